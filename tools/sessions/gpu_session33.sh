#!/bin/bash
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s33
mkdir -p $O
for PO in 0 30 45 60; do
for a in "10 4 1" "10 4 3"; do timeout 100 python tools/resident_profile.py $a resident_px=2 phase_offset=$PO 2>&1 | grep "wg 0" >> $O/prof.log; done
done
for a in "10 8 1" "10 8 3"; do timeout 100 python tools/resident_profile.py $a resident_px=2 phase_offset=45 2>&1 | grep "wg 0" >> $O/prof.log; done
cat $O/prof.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --resident-px 2 > $O/bench_px2.log 2>&1; tail -1 $O/bench_px2.log | cut -c1-230
timeout 300 python bench.py --workload ins --steps 3 --warmup 1 --no-cpu-baseline --resident-px 2 > $O/bench_ins_px2.log 2>&1; tail -1 $O/bench_ins_px2.log | cut -c1-230
timeout 300 python bench.py --workload coco --steps 2 --warmup 1 --no-cpu-baseline --resident-px 2 > $O/bench_coco_px2.log 2>&1; tail -1 $O/bench_coco_px2.log | cut -c1-230
