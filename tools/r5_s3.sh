#!/bin/bash
# round 5 session 3: a find database tuned WITH MIOpen's deterministic attribute (channels-last, fused GEMMs) — speed and bits;
# the tile-local labelling of the detection stage
set -u
OUT=gpurun_out/r5_s3; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
MIOPEN_FIND_MODE=1 timeout 900 python tools/miopen_warmup.py --channels-last 1 --single 0 --sizes 512x512 --deterministic 1 --out $OUT/miopen_det > $OUT/warmup_det.log 2>&1; el "deterministic NHWC warm-up rc=$?"
grep -E "^cam|^irnet|find database|nhwc_shapes" $OUT/warmup_det.log
run() { # run <tag> <env...>
  tag=$1; shift
  for wl in cam e2e; do
    env "$@" timeout 300 python bench.py --workload $wl --steps 12 --warmup 3 --no-legs --no-cpu-baseline --json-out $OUT/bench_${wl}_$tag.json > $OUT/bench_${wl}_$tag.log 2>&1
    python - <<PY
import json
try:
    r = json.load(open("$OUT/bench_${wl}_$tag.json")); print("%-28s %-4s %8.1f images/s  %8.2f ms/step" % ("$tag", "$wl", r["value"], r["ms_per_step"]))
except Exception as e: print("$tag $wl FAILED", e)
PY
  done
}
run shipped_nondet IRN_DETERMINISTIC=0 IRN_MIOPEN_CACHE=/tmp/mc_a
run dettuned_det   IRN_DETERMINISTIC=1 IRN_MIOPEN_SEED_DIR=$PWD/$OUT/miopen_det IRN_MIOPEN_CACHE=/tmp/mc_b
run nchw_det       IRN_DETERMINISTIC=1 IRN_CHANNELS_LAST=0 IRN_MIOPEN_CACHE=/tmp/mc_c
el "bench A/B"
for p in a b; do IRN_DETERMINISTIC=1 IRN_MIOPEN_SEED_DIR=$PWD/$OUT/miopen_det IRN_MIOPEN_CACHE=/tmp/mc_d$p timeout 300 python tools/determinism_probe.py $OUT/det_tuned_$p.json --deterministic 1 --sizes 512x512 --pairs 8 --scales 1.0,0.5,1.5,2.0 > $OUT/det_tuned_$p.log 2>&1; done
grep -E "repeat|miopen db" $OUT/det_tuned_a.log; python tools/determinism_probe.py --compare $OUT/det_tuned_a.json $OUT/det_tuned_b.json; el "determinism of the det-tuned channels-last path, 8 pairs"
IRN_DETERMINISTIC=0 timeout 600 python -m pytest tests/test_gpu_labels_instance.py -m gpu -q -x > $OUT/pytest_instance.log 2>&1; el "instance tests rc=$?"; tail -5 $OUT/pytest_instance.log
IRN_DETERMINISTIC=0 timeout 300 python tools/ins_step_breakdown.py 5 64 5 > $OUT/ins_breakdown_r5.txt 2>&1; cat $OUT/ins_breakdown_r5.txt
for b in 64 128 256; do IRN_DETERMINISTIC=0 timeout 300 python bench.py --workload ins --batch $b --steps 10 --warmup 2 --no-legs --no-cpu-baseline --json-out $OUT/bench_ins_b$b.json > $OUT/bench_ins_b$b.log 2>&1; python -c "
import json; r=json.load(open('$OUT/bench_ins_b$b.json')); print('ins batch $b: %.0f images/s' % r['value'])"; done
