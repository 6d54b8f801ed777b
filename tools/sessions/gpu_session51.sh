#!/bin/bash
# GPU session 51: radius-5 plain stores gated by the in-kernel per-image XCD vote.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s51
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_resident.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for cfg in "5 16 1" "5 16 1 plain_store=1 poll_delay_plain=6" "5 16 1 plain_store=1 poll_delay_plain=3" "5 16 1 plain_store=1 poll_delay_plain=1" "5 16 2 plain_store=1" "5 16 3 plain_store=1"; do
  timeout 60 python tools/resident_profile.py $cfg 2>&1 | tail -2 | head -1 >> $O/profile.log
done
cat $O/profile.log
timeout 120 python bench.py --workload walk_r5 --no-cpu-baseline --walk-option plain_store=1 --json-out $O/bench_r5.json > $O/bench_r5.log 2>&1; tail -1 $O/bench_r5.log | cut -c1-200
timeout 120 python bench.py --workload walk_r5 --no-cpu-baseline --json-out $O/bench_r5_off.json > $O/bench_r5_off.log 2>&1; tail -1 $O/bench_r5_off.log | cut -c1-200
timeout 120 python bench.py --workload ins --no-cpu-baseline --walk-option plain_store=1 --json-out $O/bench_ins.json > $O/bench_ins.log 2>&1; tail -1 $O/bench_ins.log | cut -c1-200
