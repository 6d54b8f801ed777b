#!/bin/bash
set -u
OUT=gpurun_out/r6_s17; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
timeout 2400 python tools/gemm16_tune.py $OUT/tune --sizes 512x512,375x500,500x375,333x500,500x333,334x500,332x500,374x500,500x500,281x500,400x500,500x400,357x500,442x500 > $OUT/gemm16_tune.log 2>&1; echo "tune rc=$?"; grep "^#" $OUT/tune/gemm16_tune.txt | tail -4
