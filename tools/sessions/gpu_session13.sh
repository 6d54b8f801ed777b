#!/bin/bash
# GPU session 13: memory-hierarchy read bandwidth vs working set (MALL question), CAM bench after the
# amax fix, small-batch walk runs.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s13
mkdir -p $O
timeout 300 tools/bin/membw 8 > $O/membw_8.log 2>&1; cat $O/membw_8.log
timeout 300 tools/bin/membw 2 > $O/membw_2.log 2>&1; cat $O/membw_2.log
timeout 600 python bench.py --workload cam --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cam.log 2>&1; tail -1 $O/bench_cam.log | cut -c1-300
for B in 8 12 16 20 24; do for T in 4 6; do
timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --unique $B --tile $T --no-cpu-baseline > $O/bench_t${T}_b$B.log 2>&1; tail -1 $O/bench_t${T}_b$B.log | cut -c1-220
done; done
