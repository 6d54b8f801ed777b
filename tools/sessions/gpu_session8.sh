#!/bin/bash
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s8
mkdir -p $O
for B in 64 192; do for PR in 4; do
timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --probe $PR --no-cpu-baseline > $O/bench_probe${PR}_b$B.log 2>&1; tail -1 $O/bench_probe${PR}_b$B.log | cut -c1-200
done; done
