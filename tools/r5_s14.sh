#!/bin/bash
# round 5 session 14: five more VOC sizes tuned (333x500, 500x333, 334x500, 332x500, 374x500): bits in the default mode, steps_voc in both modes
set -u
OUT=gpurun_out/r5_s14; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
for p in a b; do IRN_MIOPEN_CACHE=/tmp/mc_$p timeout 300 python tools/determinism_probe.py $OUT/det_$p.json --sizes 333x500,500x333,374x500,332x500 --pairs 8 --scales 1.0,0.5,1.5,2.0 > $OUT/det_$p.log 2>&1; done
grep -E "repeat|miopen db" $OUT/det_a.log; python tools/determinism_probe.py --compare $OUT/det_a.json $OUT/det_b.json
for det in 1 0; do for wl in steps_voc cam; do
  extra="--steps 12 --warmup 3"; [ $wl = steps_voc ] && extra="--steps 1 --warmup 1 --batch 256"
  IRN_DETERMINISTIC=$det timeout 600 python bench.py --workload $wl $extra --no-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readline()); t=r['config'].get('trunk') or {}; print('IRN_DETERMINISTIC=$det %-9s %7.1f images/s' % ('$wl', r['value']), t.get('layout',''), r['config'].get('cam_trunk_passes',''), r['config'].get('pass_seconds',''))"
done; done
IRN_DETERMINISTIC=1 timeout 300 python tools/cam_layout_probe.py --sizes 333x500 2>&1 | grep -v "^MIOpen\|amdgpu" | cut -c1-200
