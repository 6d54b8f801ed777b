#!/bin/bash
# One gpurun call (round ${ROUND:-5}) (results land in gpurun_out/r4_sN/; copy what should be judged into profiles/).
# usage: tools/gpu_session.sh <N> [what...]
#   what: tests tests_all testsel smoke bench default ab abopt prof pmc pmcsq <script under tools/> (the round-4/5 A/B scripts are in git history)
#   env:  TESTSEL="-k expr or paths" (testsel), AB_LIBS="libirn_hip.so other.so", AB_WL="walk coco walk_r5",
#         AB_OPTS="accel=1 accel=0" (abopt: one bench run per option string; "+" joins several options of one run)
set -u
S=${1:-1}; shift || true
WHAT=${*:-tests bench prof}
OUT=gpurun_out/r${ROUND:-5}_s$S
mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=${MIOPEN_FIND_MODE:-2}
report() {   # report <json> <label>
python - "$1" "$2" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    ro = r.get("roofline") or {}
    pe = (ro.get("power_equivalent") or {}).get("frac_of_peak")
    print("%-28s %9.1f img/s  step %8.3f ms  launch %8.3f ms  frac %.4f  power-equiv %s  applied %s" % (
        sys.argv[2], r["value"], r["ms_per_step"], ro.get("avg_launch_ms", float("nan")), ro.get("frac", float("nan")),
        "%.4f" % pe if pe else "-", (ro.get("schedule") or {}).get("operator_applications")))
except Exception as e:
    print("%-28s FAILED %r" % (sys.argv[2], e))
PY
}
for w in $WHAT; do
case $w in
tests)
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -x --durations=12 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  grep -E "differ|max \||hand-offs|passed|failed|rc=|Error|error" $OUT/pytest_gpu.log | tail -40 ;;
tests_all)
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 --durations=12 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  grep -E "differ|max \||hand-offs|passed|failed|FAILED|rc=" $OUT/pytest_gpu.log | tail -60 ;;
testsel)
  timeout 900 python -m pytest ${TESTSEL:-tests/test_gpu_schedule.py} -m gpu -q --maxfail=10 --durations=8 -s > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_sel.log
  tail -60 $OUT/pytest_sel.log ;;
smoke)
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log ;;
bench)
  timeout 900 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline --json-out $OUT/bench_walk.json > $OUT/bench_walk.log 2>&1; echo "bench rc=$?"
  report $OUT/bench_walk.json walk ;;
default)
  # exactly what the driver runs
  T0=$(date +%s)
  timeout 1500 python bench.py --json-out $OUT/bench_default.json > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "default bench rc=$? wall $(( $(date +%s) - T0 )) s"
  report $OUT/bench_default.json default
  python - <<PY
import json
r = json.load(open("$OUT/bench_default.json"))
print("legs:", {k: (round(v["value"], 1) if "value" in v else v) for k, v in r.get("legs", {}).items()})
cb = r.get("cpu_baseline") or {}
print("cpu port: %s img/s on %s threads" % (cb.get("value"), cb.get("cores")), "| label parity:", r.get("label_parity"))
print("reference algorithm:", json.dumps(cb.get("reference_algorithm"))[:900])
PY
  ;;
ab)
  for lib in ${AB_LIBS:-libirn_hip.so}; do
    [ -f irn_amd/lib/$lib ] || { echo "missing $lib"; continue; }
    for wl in ${AB_WL:-walk coco walk_r5}; do
      IRN_HIP_LIB=$PWD/irn_amd/lib/$lib timeout 300 python bench.py --workload $wl --steps ${AB_STEPS:-6} --warmup 2 --no-legs --no-cpu-baseline ${AB_ARGS:-} \
         --json-out $OUT/ab_${wl}_${lib%.so}.json > $OUT/ab_${wl}_${lib%.so}.log 2>&1
      report $OUT/ab_${wl}_${lib%.so}.json "$lib $wl"
    done
  done ;;
abopt)
  for opt in ${AB_OPTS:-accel=1 accel=0}; do
    args=""; for kv in ${opt//+/ }; do args="$args --walk-option $kv"; done
    for wl in ${AB_WL:-walk}; do
      tag=${wl}_${opt//[=+]/_}
      timeout 300 python bench.py --workload $wl --steps ${AB_STEPS:-6} --warmup 2 --no-legs --no-cpu-baseline $args --json-out $OUT/opt_$tag.json > $OUT/opt_$tag.log 2>&1
      report $OUT/opt_$tag.json "$wl $opt"
    done
  done ;;
prof)
  R=$PWD; cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_trace -o walk -f csv -- python $R/bench.py --steps 4 --warmup 1 --no-legs --no-cpu-baseline ${PROF_ARGS:-} > $R/$OUT/prof_trace.log 2>&1
  cd $R
  find $OUT/prof_trace -name "*kernel_stats*" -exec cp {} $OUT/kernel_stats.csv \;
  head -12 $OUT/kernel_stats.csv
  find $OUT -name "walk_kernel_trace.csv" -delete ;;
pmc)
  R=$PWD; cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c -d $R/$OUT/prof_$c -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-legs --no-cpu-baseline > $R/$OUT/prof_$c.log 2>&1
  done
  cd $R
  python tools/reduce_prof.py $OUT > $OUT/prof_summary.txt 2>&1; grep -E "resident_k|affinity_" $OUT/prof_summary.txt | cut -c1-24,96-200 | head -12
  find $OUT -name "walk_kernel_trace.csv" -delete; find $OUT -name "walk_counter_collection.csv" -delete ;;
pmcsq)
  R=$PWD; cd /tmp; i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set -d $R/$OUT/prof_sq$i -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-legs --no-cpu-baseline ${SQ_BENCH_ARGS:-} > $R/$OUT/prof_sq$i.log 2>&1
  done
  cd $R
  python tools/reduce_prof.py $OUT > $OUT/sq_summary.txt 2>&1; grep -E "resident_kernel|affinity_" $OUT/sq_summary.txt | cut -c1-40,96-200
  find $OUT -name "walk_counter_collection.csv" -delete ;;
combineprof)
  for m in ${DIAG_MODES:-1 2}; do for cfg in "5 32 1" "5 32 2" "10 8 1" "10 8 3"; do IRN_HIP_LIB=$PWD/irn_amd/lib/libirn_hip_diag$m.so timeout 120 python tools/combine_profile.py $cfg $m 2>&1 | tail -2; done; done > $OUT/combine_profile.txt 2>&1; cat $OUT/combine_profile.txt ;;
warmvoc)
  # tuned NHWC entries for the two image sizes that dominate VOC12 (500x375, 375x500), 8 images per trunk pass
  T0=$(date +%s)
  timeout 1200 python tools/miopen_warmup.py --channels-last 1 --single 0 --sizes ${VOC_SIZES:-375x500,500x375} --out $OUT/miopen_db_voc > $OUT/miopen_warmup_voc.log 2>&1; echo "VOC-size NHWC warm-up rc=$? wall $(( $(date +%s) - T0 )) s"; grep -E "^cam|^irnet|nhwc_shapes|find database" $OUT/miopen_warmup_voc.log ;;
libtests)
  # the walk's GPU tests against another build of the library (IRN_HIP_LIB)
  for lib in ${AB_LIBS:-libirn_hip.so}; do
    IRN_HIP_LIB=$PWD/irn_amd/lib/$lib timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_schedule.py tests/test_gpu_walk.py tests/test_gpu_parity_r2.py -m gpu -q -x -k "not forward and not edge_displacement" > $OUT/pytest_${lib%.so}.log 2>&1; echo "$lib tests rc=$?"; tail -2 $OUT/pytest_${lib%.so}.log
  done ;;
ranks)
  # the N > 1 bench path on this one-GPU box: two ranks on device 0 (gloo), then what RCCL does with a shared device
  timeout 600 python -m pytest tests/test_gpu_bench_ranks.py -m gpu -q -s --durations=5 > $OUT/pytest_ranks.log 2>&1; echo "ranks pytest rc=$?"
  grep -E "two ranks|passed|failed|rc=|Error" $OUT/pytest_ranks.log | tail -8
  bash tools/rccl_shared_probe.sh $OUT ;;
*)
  # anything else: a script under tools/ taking the output directory
  if [ -f tools/$w ]; then timeout ${RAW_TIMEOUT:-600} python tools/$w $OUT > $OUT/${w%.py}.log 2>&1; echo "$w rc=$?"; tail -${RAW_TAIL:-30} $OUT/${w%.py}.log; else echo "unknown: $w"; fi ;;
esac
done
