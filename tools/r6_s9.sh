#!/bin/bash
# round 6 session 9: row-fused 3x3 (three GEMMs over 9 cin, overlapping operand rows) vs the nine-GEMM form; 128-plane threshold
set -u
OUT=gpurun_out/r6_s9; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
timeout 600 python -m pytest tests/test_gpu_split_gemm.py -m gpu -q -s > $OUT/pytest_split.log 2>&1; echo "tests rc=$?"; grep -E "3x3 \(|passed|failed|FAILED|Error" $OUT/pytest_split.log | tail -16
run() {
  tag=$1; wl=$2; st=$3; shift 3
  env "$@" timeout 400 python bench.py --workload $wl --steps $st --warmup 2 --no-legs --no-cpu-baseline --json-out $OUT/${wl}_$tag.json > $OUT/${wl}_$tag.log 2>&1
  python - $OUT/${wl}_$tag.json "$wl $tag" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print("%-40s %8.2f img/s  %8.2f ms/step  frac %s" % (sys.argv[2], r["value"], r["ms_per_step"], r["config"].get("matrix_fp32_frac")))
except Exception as e:
    print("%-40s FAILED %r" % (sys.argv[2], e))
PY
}
run nine cam 12 IRN_CONV3X3_ROW_FUSED=0
run three cam 12
run three_128 cam 12 IRN_SPLIT_MIN_PLANES_3X3=128
run three_64 cam 12 IRN_SPLIT_MIN_PLANES_3X3=64
run three_rows2048 cam 12 IRN_SPLIT_MIN_ROWS_3X3=2048
run nine e2e 12 IRN_CONV3X3_ROW_FUSED=0
run three e2e 12
run three_128 e2e 12 IRN_SPLIT_MIN_PLANES_3X3=128
