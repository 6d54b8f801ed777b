#!/bin/bash
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s9
mkdir -p $O
for T in 8 9 10 11 7; do for B in 64 192; do
timeout 300 python bench.py --steps 3 --warmup 1 --batch $B --tile $T --no-cpu-baseline > $O/bench_t${T}_b$B.log 2>&1; tail -1 $O/bench_t${T}_b$B.log | cut -c1-200
done; done
