#!/bin/bash
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s36
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_resident.py -q -x > $O/pytest_resident.log 2>&1; echo "pytest rc=$?" >> $O/pytest_resident.log; tail -5 $O/pytest_resident.log
for D in 20; do
timeout 100 python tools/resident_profile.py 10 4 1 poll_delay=$D 2>&1 | grep "wg 0" >> $O/prof.log
timeout 100 python tools/resident_profile.py 10 4 2 poll_delay=$D 2>&1 | grep "wg 0" >> $O/prof.log
done
timeout 100 python tools/resident_profile.py 10 4 3 2>&1 | grep "wg 0" >> $O/prof.log
timeout 100 python tools/resident_profile.py 10 4 4 2>&1 | grep "wg 0" >> $O/prof.log
timeout 100 python tools/resident_profile.py 5 16 1 2>&1 | grep "wg 0" >> $O/prof.log
timeout 100 python tools/resident_profile.py 5 16 3 2>&1 | grep "wg 0" >> $O/prof.log
cat $O/prof.log
timeout 300 python bench.py --steps 3 --warmup 1 --variant 2 --no-cpu-baseline > $O/bench_v2.log 2>&1; tail -1 $O/bench_v2.log | cut -c1-260
timeout 300 python bench.py --workload walk_r5 --steps 3 --warmup 1 --variant 2 --no-cpu-baseline > $O/bench_v2_r5.log 2>&1; tail -1 $O/bench_v2_r5.log | cut -c1-260
timeout 300 python bench.py --workload coco --steps 2 --warmup 1 --variant 2 --no-cpu-baseline > $O/bench_v2_coco.log 2>&1; tail -1 $O/bench_v2_coco.log | cut -c1-260
timeout 300 python bench.py --workload ins --steps 3 --warmup 1 --variant 2 --no-cpu-baseline > $O/bench_v2_ins.log 2>&1; tail -1 $O/bench_v2_ins.log | cut -c1-260
