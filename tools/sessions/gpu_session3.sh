#!/bin/bash
# GPU session 3: compact (runtime-row) sweep kernel: parity, tile/batch/streams sweep, profiles.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
for T in 0 1 2 3 4 5 6 7; do for B in 24 64 128; do
  timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --tile $T --no-cpu-baseline > gpurun_out/bench_t${T}_b$B.log 2>&1; tail -1 gpurun_out/bench_t${T}_b$B.log | cut -c1-300
done; done
for T in 0 3; do
  timeout 300 python bench.py --steps 3 --warmup 1 --batch 64 --tile $T --streams 0 --no-cpu-baseline > gpurun_out/bench_t${T}_b64_nostreams.log 2>&1
  timeout 300 python bench.py --steps 3 --warmup 1 --batch 256 --tile $T --no-cpu-baseline > gpurun_out/bench_t${T}_b256.log 2>&1
  timeout 300 python bench.py --workload walk_r5 --steps 3 --warmup 1 --tile $T --no-cpu-baseline > gpurun_out/bench_r5_t$T.log 2>&1
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace -o walk -f csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace_t3 -o walk -f csv -- python $R/bench.py --steps 2 --warmup 1 --tile 3 --no-cpu-baseline > $R/gpurun_out/rocprof_trace_t3.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_tcc -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/rocprof_tcc.log 2>&1
cd $R
python tools/reduce_prof.py gpurun_out > gpurun_out/prof_summary.txt 2>&1
rm -f gpurun_out/prof_*/walk_kernel_trace.csv gpurun_out/prof_*/walk_counter_collection.csv
du -sh gpurun_out
