#!/bin/bash
# GPU session 4: clamped-plane mask (no zero plane) + merged single-launch sweeps.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
for M in 1 0; do for T in 0 1 3 4 6 7; do for B in 24 64 128; do
  timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --tile $T --merged $M --no-cpu-baseline > gpurun_out/bench_m${M}_t${T}_b$B.log 2>&1; tail -1 gpurun_out/bench_m${M}_t${T}_b$B.log | cut -c1-200
done; done; done
for T in 0 4; do
  timeout 300 python bench.py --steps 3 --warmup 1 --batch 256 --tile $T --no-cpu-baseline > gpurun_out/bench_m1_t${T}_b256.log 2>&1
  timeout 300 python bench.py --workload walk_r5 --steps 3 --warmup 1 --tile $T --no-cpu-baseline > gpurun_out/bench_r5_t$T.log 2>&1
done
cd /tmp
for T in 0 4; do
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_trace_t$T -o walk -f csv -- python $R/bench.py --steps 2 --warmup 1 --tile $T --no-cpu-baseline > $R/gpurun_out/rocprof_trace_t$T.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch_t$T -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --tile $T --no-cpu-baseline > $R/gpurun_out/rocprof_fetch_t$T.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/prof_tcc_t$T -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --tile $T --no-cpu-baseline > $R/gpurun_out/rocprof_tcc_t$T.log 2>&1
done
cd $R
python tools/reduce_prof.py gpurun_out > gpurun_out/prof_summary.txt 2>&1
rm -f gpurun_out/prof_*/walk_kernel_trace.csv gpurun_out/prof_*/walk_counter_collection.csv
du -sh gpurun_out
