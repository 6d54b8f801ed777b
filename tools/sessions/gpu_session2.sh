#!/bin/bash
# GPU session 2: full parity suite, tile/batch sweep, kernel trace + PMC passes.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
for T in 0 1 2 3; do for B in 16 24 64 128; do
  timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --tile $T --no-cpu-baseline > gpurun_out/bench_t${T}_b$B.log 2>&1; tail -1 gpurun_out/bench_t${T}_b$B.log | cut -c1-400
done; done
for T in 0 2 3; do
  timeout 300 python bench.py --workload walk_r5 --steps 3 --warmup 1 --tile $T --no-cpu-baseline > gpurun_out/bench_r5_t$T.log 2>&1; tail -1 gpurun_out/bench_r5_t$T.log | cut -c1-400
done
timeout 300 python bench.py --steps 3 --warmup 1 --xcd-map 0 --no-cpu-baseline > gpurun_out/bench_noxcd.log 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace -o walk -f csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rocprof_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/rocprof_write.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_tcc -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/rocprof_tcc.log 2>&1
cd $R
# keep only the small summaries (the raw per-dispatch csv of 1000s of launches is reduced here)
python tools/reduce_prof.py gpurun_out > gpurun_out/prof_summary.txt 2>&1
tail -40 gpurun_out/prof_summary.txt
du -sh gpurun_out
