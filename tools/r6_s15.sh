#!/bin/bash
# round 6 session 15: channels-last find database for the six sizes of the VOC histogram that were still untuned
set -u
OUT=gpurun_out/r6_s15; mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s)
IRN_SPLIT_GEMM=0 timeout 1500 python tools/miopen_warmup.py --channels-last 1 --single 0 --sizes 500x500,281x500,400x500,500x400,357x500,442x500 --out $OUT/miopen_db > $OUT/miopen_warmup.log 2>&1
echo "warm-up rc=$? wall $(( $(date +%s) - T0 )) s"; grep -E "^cam|^irnet|nhwc_shapes|find database" $OUT/miopen_warmup.log | tail -40
ls -la $OUT/miopen_db/*
