#!/bin/bash
# round 6 session 10: rank tables of the fp16 problems (incl. the row-fused 3x3), then the whole GPU suite, the driver's bench, and the
# rocprofv3 kernel statistics of the same command
set -u
OUT=gpurun_out/r6_s10; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
timeout 1200 python tools/gemm16_tune.py $OUT/tune > $OUT/gemm16_tune.log 2>&1; echo "tune rc=$?"; grep "^#" $OUT/tune/gemm16_tune.txt | tail -4
ROUND=6 bash tools/gpu_session.sh 10 tests_all default prof
