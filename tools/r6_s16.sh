#!/bin/bash
# round 6 session 16: the six newly tuned VOC sizes — bits across processes in the default mode, steps_voc
set -u
OUT=gpurun_out/r6_s16; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
for p in a b; do IRN_MIOPEN_CACHE=/tmp/mc_$p timeout 400 python tools/determinism_probe.py $OUT/det_$p.json --sizes 500x500,281x500,400x500,500x400,357x500,442x500 --pairs 8 --scales 1.0,0.5,1.5,2.0 --repeat 2 > $OUT/det_$p.log 2>&1; done
grep -E "repeat|miopen db" $OUT/det_a.log | cut -c1-200
python - <<'PY'
import json
a=json.load(open("gpurun_out/r6_s16/det_a.json")); b=json.load(open("gpurun_out/r6_s16/det_b.json"))
for k in a:
    d=[n for (n,x),(_,y) in zip(a[k],b[k]) if x!=y]
    print(k, len(a[k]), "outputs,", len(d), "differ between the two processes", d[:2])
PY
for wl in steps_voc steps; do
timeout 600 python bench.py --workload $wl --steps 1 --warmup 1 --batch 256 --no-legs --no-cpu-baseline 2>$OUT/$wl.err | python -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('%-9s %7.1f images/s' % ('$wl', r['value']), r['config'].get('cam_trunk_passes',''), r['config'].get('pass_seconds',''))"
done
grep "trunk passes ran NCHW" $OUT/steps_voc.err | tail -1 | cut -c1-300
