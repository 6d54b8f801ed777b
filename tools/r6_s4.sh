#!/bin/bash
# round 6 session 4: the split-precision 1x1 GEMMs — unit tests, parity suite, then cam / e2e / steps with the path off and on and
# with other thresholds (which layers take it)
set -u
OUT=gpurun_out/r6_s4; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
timeout 600 python -m pytest tests/test_gpu_split_gemm.py -m gpu -q -x -s > $OUT/pytest_split.log 2>&1; echo "split tests rc=$?"; tail -25 $OUT/pytest_split.log
timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_conv1x1.py tests/test_gpu_reproducible.py tests/test_gpu_bn_act.py -m gpu -q -s > $OUT/pytest_parity.log 2>&1; echo "parity tests rc=$?"; tail -8 $OUT/pytest_parity.log
run() {  # run <tag> <workload> env...
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
  run off $wl IRN_SPLIT_GEMM=0
  run on_default $wl IRN_SPLIT_GEMM=1
  run on_conv3_only $wl IRN_SPLIT_GEMM=1 IRN_SPLIT_MIN_INPUT=1000000000
  run on_conv3_from128 $wl IRN_SPLIT_GEMM=1 IRN_SPLIT_MIN_PLANES=128
  run on_input_from_2e19 $wl IRN_SPLIT_GEMM=1 IRN_SPLIT_MIN_INPUT=524288
  run on_input_from_2e18 $wl IRN_SPLIT_GEMM=1 IRN_SPLIT_MIN_INPUT=262144
done
run off_fast cam IRN_SPLIT_GEMM=0 IRN_DETERMINISTIC=0
run on_fast cam IRN_SPLIT_GEMM=1 IRN_DETERMINISTIC=0
