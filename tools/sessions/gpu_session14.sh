#!/bin/bash
# GPU session 14: first run of the weights-stationary persistent walk (variant 2).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s14
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_resident.py -q -x > $O/pytest_resident.log 2>&1; echo "pytest rc=$?" >> $O/pytest_resident.log; tail -15 $O/pytest_resident.log
for B in 4 16 64 192; do
timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --variant 2 --no-cpu-baseline > $O/bench_v2_b$B.log 2>&1; tail -1 $O/bench_v2_b$B.log | cut -c1-260
done
timeout 300 python bench.py --workload walk_r5 --steps 3 --warmup 1 --variant 2 --no-cpu-baseline > $O/bench_v2_r5.log 2>&1; tail -1 $O/bench_v2_r5.log | cut -c1-260
timeout 300 python bench.py --workload coco --steps 2 --warmup 1 --variant 2 --no-cpu-baseline > $O/bench_v2_coco.log 2>&1; tail -1 $O/bench_v2_coco.log | cut -c1-260
timeout 300 python bench.py --workload ins --steps 3 --warmup 1 --variant 2 --no-cpu-baseline > $O/bench_v2_ins.log 2>&1; tail -1 $O/bench_v2_ins.log | cut -c1-260
