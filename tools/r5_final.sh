#!/bin/bash
# round 5, the committed build: what the driver runs at round end — pytest -m gpu, smoke(), python bench.py
set -u
OUT=gpurun_out/r5_final; mkdir -p $OUT
export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 1800 python -m pytest tests -x -q -m gpu --durations=10 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_gpu.log | tail -5; el "gpu tests"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1500 python bench.py --json-out $OUT/bench_default.json > $OUT/bench_default.log 2> $OUT/bench_default.err; el "default bench rc=$?"
python - <<PY
import json
r = json.load(open("$OUT/bench_default.json"))
print("value %.1f ms/step %.3f frac %.4f traffic %s (%s) cpu %.1f on %s x %s label_parity %s" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["traffic"], r["roofline"]["traffic_source"], r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["cpu_model"], r["label_parity"]["pixels_differing"]))
for k, v in r.get("legs", {}).items():
    print(k, {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk not in ("through",)})
PY
