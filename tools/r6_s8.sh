#!/bin/bash
# round 6 session 8: nine-GEMM 3x3 in one C call, borders written by the split kernel (no persistent buffers): tests, legs, profile
set -u
OUT=gpurun_out/r6_s8; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
timeout 600 python -m pytest tests/test_gpu_split_gemm.py tests/test_gpu_parity_r2.py tests/test_gpu_reproducible.py tests/test_gpu_preflight.py tests/test_gpu_steps.py -m gpu -q -s > $OUT/pytest_sel.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED" $OUT/pytest_sel.log | tail -8
run() {
  tag=$1; wl=$2; st=$3; shift 3
  env "$@" timeout 400 python bench.py --workload $wl --steps $st --warmup 1 --no-legs --no-cpu-baseline --json-out $OUT/${wl}_$tag.json > $OUT/${wl}_$tag.log 2>&1
  python - $OUT/${wl}_$tag.json "$wl $tag" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print("%-40s %8.2f img/s  %8.2f ms/step  frac %s  %s" % (sys.argv[2], r["value"], r["ms_per_step"], r["config"].get("matrix_fp32_frac"), r["config"].get("pass_seconds", "")))
except Exception as e:
    print("%-40s FAILED %r" % (sys.argv[2], e))
PY
}
run on cam 12
run on e2e 12
run on steps 2 
run on steps_voc 1
run 3x3_off steps 2 IRN_SPLIT_MIN_PLANES_3X3=100000
run 3x3_off steps_voc 1 IRN_SPLIT_MIN_PLANES_3X3=100000
run all_off steps 2 IRN_SPLIT_GEMM=0
R=$PWD; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_e2e -o e2e -f csv -- python $R/bench.py --workload e2e --steps 7 --warmup 2 --no-legs --no-cpu-baseline > $R/$OUT/prof_e2e.log 2>&1
cd $R
find $OUT/prof_e2e -name "*kernel_stats*" -exec cp {} $OUT/e2e_kernel_stats.csv \;
python tools/kernel_classes.py $OUT/e2e_kernel_stats.csv 24 > $OUT/e2e_kernel_classes.txt 2>&1; cat $OUT/e2e_kernel_classes.txt
find $OUT -name "*kernel_trace.csv" -delete
