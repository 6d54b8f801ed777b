#!/bin/bash
# round 6 session 7: the nine-GEMM split 3x3 of stage 4 — tests, then cam / e2e with it off and on
set -u
OUT=gpurun_out/r6_s7; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
timeout 600 python -m pytest tests/test_gpu_split_gemm.py -m gpu -q -x -s > $OUT/pytest_split.log 2>&1; echo "split tests rc=$?"; grep -E "3x3|unit|passed|failed|Error|assert" $OUT/pytest_split.log | tail -20
timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_reproducible.py tests/test_gpu_preflight.py -m gpu -q -s > $OUT/pytest_parity.log 2>&1; echo "parity tests rc=$?"; tail -6 $OUT/pytest_parity.log
run() {
  tag=$1; wl=$2; shift 2
  env "$@" timeout 300 python bench.py --workload $wl --steps 12 --warmup 2 --no-legs --no-cpu-baseline --json-out $OUT/${wl}_$tag.json > $OUT/${wl}_$tag.log 2>&1
  python - $OUT/${wl}_$tag.json "$wl $tag" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print("%-40s %8.2f img/s  %8.2f ms/step  frac %s" % (sys.argv[2], r["value"], r["ms_per_step"], r["config"].get("matrix_fp32_frac")))
except Exception as e:
    print("%-40s FAILED %r" % (sys.argv[2], e))
PY
}
for wl in cam e2e; do
  run 3x3_off $wl IRN_SPLIT_MIN_PLANES_3X3=100000
  run 3x3_on $wl
  run 3x3_on_rows4096 $wl IRN_SPLIT_MIN_ROWS_3X3=4096
  run 3x3_on_rows16384 $wl IRN_SPLIT_MIN_ROWS_3X3=16384
done
run 3x3_on_256 cam IRN_SPLIT_MIN_PLANES_3X3=256
run steps_on steps
run steps_voc_on steps_voc
# rank table of the fp16 problems, then the same legs with it
timeout 900 python tools/gemm16_tune.py $OUT/tune > $OUT/gemm16_tune.log 2>&1; echo "tune rc=$?"; tail -4 $OUT/gemm16_tune.log
run table_on cam
run table_on e2e
