#!/bin/bash
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s32
mkdir -p $O
for PX in 4 2; do
for a in "10 4 1" "10 4 2" "10 4 3"; do timeout 100 python tools/resident_profile.py $a resident_px=$PX 2>&1 | grep "wg 0" >> $O/prof.log; done
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --resident-px $PX > $O/bench_px$PX.log 2>&1; tail -1 $O/bench_px$PX.log | cut -c1-230
timeout 300 python bench.py --workload ins --steps 3 --warmup 1 --no-cpu-baseline --resident-px $PX > $O/bench_ins_px$PX.log 2>&1; tail -1 $O/bench_ins_px$PX.log | cut -c1-230
timeout 300 python bench.py --workload coco --steps 2 --warmup 1 --no-cpu-baseline --resident-px $PX > $O/bench_coco_px$PX.log 2>&1; tail -1 $O/bench_coco_px$PX.log | cut -c1-230
done
cat $O/prof.log
