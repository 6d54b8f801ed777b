#!/bin/bash
# round 5 session 1: fused 1x1-convolution GEMMs (per-layer A/B, rank table, cam / e2e A/B), backbone determinism
# across processes, instance-stage breakdown.   usage: bash tools/r5_s1.sh   (results in gpurun_out/r5_s1)
set -u
OUT=gpurun_out/r5_s1; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 600 python tools/conv1x1_tune.py $OUT --sizes 512x512 > $OUT/tune.log 2>&1; el "tune rc=$?"; tail -4 $OUT/tune.log
for g in 0 1; do for wl in cam e2e; do
  IRN_FUSED_GEMM=$g timeout 300 python bench.py --workload $wl --steps 12 --warmup 3 --no-legs --no-cpu-baseline --json-out $OUT/bench_${wl}_gemm$g.json > $OUT/bench_${wl}_gemm$g.log 2>&1
  python - <<PY
import json
try:
    r = json.load(open("$OUT/bench_${wl}_gemm$g.json")); print("fused_gemm=$g %-4s %8.1f images/s  %8.2f ms/step" % ("$wl", r["value"], r["ms_per_step"]))
except Exception as e: print("fused_gemm=$g $wl FAILED", e)
PY
done; done; el "bench A/B done"
IRN_GEMM_TABLE=0 timeout 300 python bench.py --workload cam --steps 12 --warmup 3 --no-legs --no-cpu-baseline --json-out $OUT/bench_cam_gemm1_notable.json > $OUT/bench_cam_gemm1_notable.log 2>&1
python -c "import json; r=json.load(open('$OUT/bench_cam_gemm1_notable.json')); print('fused_gemm=1 no rank table cam %.1f images/s' % r['value'])"
# determinism: two processes each, default layout policy (auto) and both forced layouts
for cfg in "auto 1" "0 1" "1 1" "1 0"; do set -- $cfg
  for p in a b; do IRN_CHANNELS_LAST=$1 IRN_FUSED_GEMM=$2 timeout 300 python tools/determinism_probe.py $OUT/det_cl$1_g$2_$p.json > $OUT/det_cl$1_g$2_$p.log 2>&1; done
  echo "== channels_last=$1 fused_gemm=$2: in-process repeats"; grep -E "repeat|miopen db" $OUT/det_cl$1_g$2_a.log | head -12
  echo "== process a vs process b"; python tools/determinism_probe.py --compare $OUT/det_cl$1_g$2_a.json $OUT/det_cl$1_g$2_b.json
done > $OUT/determinism.txt 2>&1; el "determinism done"; cat $OUT/determinism.txt
timeout 300 python tools/ins_step_breakdown.py 5 64 5 > $OUT/ins_breakdown_r5.txt 2>&1; cat $OUT/ins_breakdown_r5.txt
timeout 600 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_bn_act.py -m gpu -q -x -s > $OUT/pytest_parity.log 2>&1; el "parity tests rc=$?"; tail -15 $OUT/pytest_parity.log
