#!/bin/bash
# GPU session 5: batch-size quantisation check.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/s5
for M in 1 0; do for T in 0 7 1; do for B in 48 96 144 192 384; do
  timeout 300 python bench.py --steps 2 --warmup 1 --batch $B --tile $T --merged $M --no-cpu-baseline > gpurun_out/s5/bench_m${M}_t${T}_b$B.log 2>&1; tail -1 gpurun_out/s5/bench_m${M}_t${T}_b$B.log | cut -c1-200
done; done; done
