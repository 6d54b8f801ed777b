#!/bin/bash
# First GPU session: smoke, parity tests, bench sweeps, rocprof kernel trace.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -5 gpurun_out/smoke.log
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_walk.log 2>&1; tail -3 gpurun_out/bench_walk.log
for B in 8 16 24 32 128; do
  timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline > gpurun_out/bench_walk_b$B.log 2>&1; tail -1 gpurun_out/bench_walk_b$B.log
done
timeout 300 python bench.py --steps 3 --warmup 1 --xcd-map 0 --no-cpu-baseline > gpurun_out/bench_walk_noxcd.log 2>&1; tail -1 gpurun_out/bench_walk_noxcd.log
timeout 300 python bench.py --workload walk_r5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r5.log 2>&1; tail -1 gpurun_out/bench_r5.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_walk -o walk -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof_walk.log 2>&1
cd $R; find gpurun_out/prof_walk -name "*stats*" | head
