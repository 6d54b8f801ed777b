#!/bin/bash
# round 6 session 5: the whole GPU suite with the split-precision 1x1 convolutions as the default, the 3x3 probe, the default bench
set -u
OUT=gpurun_out/r6_s5; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
timeout 300 python tools/conv3x3_split_probe.py > $OUT/conv3x3_split_probe.txt 2>&1; cat $OUT/conv3x3_split_probe.txt | grep -v amdgpu.ids
ROUND=6 bash tools/gpu_session.sh 5 tests_all default
