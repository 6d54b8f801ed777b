#!/bin/bash
# round 5 session 12: the reproducible mode as the default (filtered database for tuned shapes + MIOpen's attribute elsewhere): bits and speed
set -u
OUT=gpurun_out/r5_s12; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
for cfg in "512x512,375x500 8" "512x512,333x500,96x128 1" "333x500,281x500 8"; do set -- $cfg
  tag=$(echo $1_$2 | tr ',' '_')
  for p in a b; do IRN_MIOPEN_CACHE=/tmp/mc_$tag$p timeout 300 python tools/determinism_probe.py $OUT/det_$tag$p.json --sizes $1 --pairs $2 --scales 1.0,0.5,1.5,2.0 > $OUT/det_$tag$p.log 2>&1; done
  echo "== sizes $1, pairs $2 (default mode)"; grep -E "repeat|miopen db" $OUT/det_${tag}a.log; python tools/determinism_probe.py --compare $OUT/det_${tag}a.json $OUT/det_${tag}b.json
done > $OUT/determinism_default_mode.txt 2>&1; cat $OUT/determinism_default_mode.txt; el "determinism"
for det in 1 0; do for wl in cam e2e steps steps_voc; do
  extra=""; [ $wl = steps ] && extra="--steps 2 --warmup 1 --batch 256"; [ $wl = steps_voc ] && extra="--steps 1 --warmup 1 --batch 256"; [ -z "$extra" ] && extra="--steps 12 --warmup 3"
  IRN_DETERMINISTIC=$det timeout 600 python bench.py --workload $wl $extra --no-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readline()); t=r['config'].get('trunk') or {}; print('IRN_DETERMINISTIC=$det %-9s %7.1f images/s' % ('$wl', r['value']), t.get('layout',''), t.get('miopen_key',''), r['config'].get('cam_trunk_passes',''))"
done; done | tee $OUT/mode_ab.txt; el "mode A/B"
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log; el "gpu tests"
