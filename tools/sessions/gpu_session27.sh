#!/bin/bash
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s27
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --json-out $O/bench_default.json > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_trace -o walk -f csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_trace.log 2>&1
cd $R
find $O/prof_trace -name "*kernel_stats*" -exec cp {} $O/kernel_stats.csv \;
find $O -name "walk_kernel_trace.csv" -delete
cut -c1-60,150-230 $O/kernel_stats.csv | head -8
