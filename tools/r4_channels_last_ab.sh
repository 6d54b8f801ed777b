#!/bin/bash
# channels-last trunk: tune MIOpen's find database for the NHWC shapes (tools/miopen_warmup.py --channels-last 1), then the
# backbone legs in both layouts, each seeded from its own database.  usage: tools/r4_channels_last_ab.sh <outdir>
OUT=${1:-gpurun_out}
timeout 300 python -m pytest tests/test_gpu_bn_act.py -m gpu -q -k "channels_last" -s > $OUT/cl_tests.log 2>&1; echo "channels-last tests rc=$?"; grep -E "passed|failed|deviation" $OUT/cl_tests.log | tail -3
T0=$(date +%s)
timeout 1500 python tools/miopen_warmup.py --channels-last 1 --single 0 --suffix=-nhwc --out $OUT/miopen_db_nhwc > $OUT/miopen_warmup_nhwc.log 2>&1; echo "NHWC warm-up rc=$? wall $(( $(date +%s) - T0 )) s"; grep -E "^cam|^irnet|find database" $OUT/miopen_warmup_nhwc.log
run() {  # run <tag> <env...>
  tag=$1; shift
  for wl in cam e2e; do
    env "$@" timeout 600 python bench.py --workload $wl --steps 12 --warmup 2 --no-legs --no-cpu-baseline --json-out $OUT/cl_${tag}_$wl.json > $OUT/cl_${tag}_$wl.log 2>&1
    python - "$OUT/cl_${tag}_$wl.json" "$tag $wl" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print("%-22s %8.1f images/s  %8.2f ms per step" % (sys.argv[2], r["value"], r["ms_per_step"]))
except Exception as e:
    print("%-22s FAILED %r" % (sys.argv[2], e))
PY
  done
}
# the NHWC database becomes the user database of the channels-last runs
KEY=$(ls $OUT/miopen_db_nhwc | head -1)
rm -rf /tmp/mi_nhwc /tmp/mi_nchw; mkdir -p /tmp/mi_nhwc_seed
run nchw IRN_MIOPEN_CACHE=/tmp/mi_nchw
DEVDIR=/tmp/mi_nhwc/$(python -c "import sys; sys.path.insert(0,'.'); from irn_amd.step import _common; print(_common.miopen_cache_key())")/dev0
mkdir -p $DEVDIR && cp $OUT/miopen_db_nhwc/$KEY/* $DEVDIR/ 2>/dev/null
run nhwc IRN_CHANNELS_LAST=1 IRN_MIOPEN_CACHE=/tmp/mi_nhwc IRN_MIOPEN_SEED=0
run nhwc_untuned IRN_CHANNELS_LAST=1 IRN_MIOPEN_CACHE=/tmp/mi_nhwc_cold IRN_MIOPEN_SEED=0
