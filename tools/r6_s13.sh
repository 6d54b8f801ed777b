#!/bin/bash
set -u
OUT=gpurun_out/r6_s13; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
for i in 1 2 3; do
  timeout 300 python bench.py --workload cam --steps 12 --warmup 1 --no-legs --no-cpu-baseline --json-out $OUT/cam_w1_$i.json > /dev/null 2>&1
  timeout 300 python bench.py --workload cam --steps 12 --warmup 2 --no-legs --no-cpu-baseline --json-out $OUT/cam_w2_$i.json > /dev/null 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r6_s13/cam_w*.json")):
    r=json.load(open(f)); print(f.split("/")[-1], round(r["value"],1), round(r["ms_per_step"],2))
PY
ROUND=6 bash tools/gpu_session.sh 13 default
