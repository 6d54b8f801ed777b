#!/bin/bash
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s10
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-250
timeout 900 python bench.py --workload cam --steps 3 --warmup 2 --batch 8 > $O/bench_cam.log 2>&1; tail -1 $O/bench_cam.log | cut -c1-300
timeout 900 python bench.py --workload e2e --steps 3 --warmup 2 --batch 8 > $O/bench_e2e.log 2>&1; tail -1 $O/bench_e2e.log | cut -c1-300
