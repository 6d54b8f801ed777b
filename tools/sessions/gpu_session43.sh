#!/bin/bash
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s43
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_resident.py -q -x > $O/pytest_resident.log 2>&1; echo "pytest rc=$?" >> $O/pytest_resident.log; tail -3 $O/pytest_resident.log
for D in 8 10 12; do timeout 100 python tools/resident_profile.py 10 4 1 poll_delay=$D 2>&1 | grep "wg 0" >> $O/prof.log; done
for a in "10 4 2" "10 4 3" "5 16 1" "5 16 2" "5 16 3"; do timeout 100 python tools/resident_profile.py $a 2>&1 | grep "wg 0" >> $O/prof.log; done
cat $O/prof.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
timeout 300 python bench.py --workload walk_r5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_r5.log 2>&1; tail -1 $O/bench_r5.log | cut -c1-200
timeout 300 python bench.py --workload ins --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_ins.log 2>&1; tail -1 $O/bench_ins.log | cut -c1-200
