#!/bin/bash
# GPU session 49: experiment — state stores without sc1 (line stays in the XCD's L2; same-XCD neighbours' sc1 polls hit it).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s49
mkdir -p $O
for cfg in "5 16 1 plain_store=1 poll_delay=10" "5 16 1 plain_store=1 poll_delay=5" "5 16 1 plain_store=1 poll_delay=2" "5 16 2 plain_store=1" \
           "10 4 1 plain_store=1 poll_delay=10" "10 4 1 plain_store=1 poll_delay=5" "10 4 2 plain_store=1"; do
  timeout 60 python tools/resident_profile.py $cfg 2>&1 | tail -2 | head -1 >> $O/profile.log
done
cat $O/profile.log
timeout 120 python bench.py --workload walk_r5 --no-cpu-baseline --walk-option plain_store=1 --json-out $O/bench_r5.json > $O/bench_r5.log 2>&1; tail -1 $O/bench_r5.log | cut -c1-200
timeout 120 python bench.py --no-cpu-baseline --walk-option plain_store=1 --json-out $O/bench_default.json > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
