#!/bin/bash
# round 5 session 2: layout policy for untuned sizes (+ MIOpen's deterministic attribute), full GPU suite with the fused GEMMs,
# default bench with the new legs
set -u
OUT=gpurun_out/r5_s2; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python tools/cam_layout_probe.py > $OUT/cam_layout_probe.txt 2>&1; el "layout probe rc=$?"; grep -v "^MIOpen" $OUT/cam_layout_probe.txt
for p in a b; do IRN_CHANNELS_LAST=0 timeout 300 python tools/determinism_probe.py $OUT/det_nchw_det1_$p.json --deterministic 1 > $OUT/det_nchw_det1_$p.log 2>&1; done
grep -E "repeat" $OUT/det_nchw_det1_a.log | head; python tools/determinism_probe.py --compare $OUT/det_nchw_det1_a.json $OUT/det_nchw_det1_b.json; el "determinism (NCHW + deterministic attribute)"
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 --durations=10 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "differ|passed|failed|FAILED|rc=|Error" $OUT/pytest_gpu.log | tail -40; el "gpu tests"
timeout 1500 python bench.py --json-out $OUT/bench_default.json > $OUT/bench_default.log 2> $OUT/bench_default.err; el "default bench rc=$?"
python - <<PY
import json
r = json.load(open("$OUT/bench_default.json"))
print("value %.1f ms/step %.3f frac %.4f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"]))
for k, v in r.get("legs", {}).items():
    print(k, {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk not in ("pass_seconds", "through")})
PY
