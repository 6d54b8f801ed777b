#!/bin/bash
# GPU session 7: staged-load pipelining + streaming-skeleton probes.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s7
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
for B in 24 64 192; do
timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline > $O/bench_b$B.log 2>&1; tail -1 $O/bench_b$B.log | cut -c1-200
for PR in 1 2 3; do
timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --probe $PR --no-cpu-baseline > $O/bench_probe${PR}_b$B.log 2>&1; tail -1 $O/bench_probe${PR}_b$B.log | cut -c1-200
done; done
timeout 300 python bench.py --steps 3 --warmup 1 --batch 192 --merged 1 --no-cpu-baseline > $O/bench_merged_b192.log 2>&1
