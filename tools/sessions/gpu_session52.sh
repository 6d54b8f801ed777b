#!/bin/bash
# GPU session 52: radius-5 combine with both iterations' sums formed before the stores.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s52
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_resident.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in "5 16 1" "5 16 2" "5 16 3"; do
  timeout 60 python tools/resident_profile.py $cfg 2>&1 | tail -2 | head -1 >> $O/profile.log
done
cat $O/profile.log
timeout 120 python bench.py --workload walk_r5 --no-cpu-baseline --json-out $O/bench_r5.json > $O/bench_r5.log 2>&1; tail -1 $O/bench_r5.log | cut -c1-200
