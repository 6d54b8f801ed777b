#!/bin/bash
set -u
OUT=gpurun_out/r6_s12; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_split_gemm.py -m gpu -q -x > $OUT/pytest_sel.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest_sel.log
ROUND=6 bash tools/gpu_session.sh 12 default
