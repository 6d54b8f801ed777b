#!/bin/bash
# MIOpen find database: measure the backbone legs cold (empty user database, find mode 2 = the immediate-mode heuristic),
# fill the database for the bench's shape set with the FIND api (tools/miopen_warmup.py), measure again seeded from it.
# usage: tools/r4_miopen_ab.sh <outdir> [extra warm-up args]
OUT=${1:-gpurun_out}; shift || true
run() {  # run <tag> <env...> : cam and e2e legs as main workloads
  tag=$1; shift
  for wl in cam e2e; do
    env "$@" timeout 600 python bench.py --workload $wl --steps 12 --warmup 2 --no-legs --no-cpu-baseline --json-out $OUT/miopen_${tag}_$wl.json > $OUT/miopen_${tag}_$wl.log 2>&1
    python - "$OUT/miopen_${tag}_$wl.json" "$tag $wl" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print("%-22s %8.1f images/s  %8.2f ms per step" % (sys.argv[2], r["value"], r["ms_per_step"]))
except Exception as e:
    print("%-22s FAILED %r" % (sys.argv[2], e))
PY
  done
}
rm -rf /tmp/mi_cold /tmp/mi_seeded
run cold IRN_MIOPEN_CACHE=/tmp/mi_cold IRN_MIOPEN_SEED=0
T0=$(date +%s)
timeout 1200 python tools/miopen_warmup.py "$@" > $OUT/miopen_warmup.log 2>&1; echo "warm-up rc=$? wall $(( $(date +%s) - T0 )) s"; tail -6 $OUT/miopen_warmup.log
mkdir -p $OUT/miopen_db && cp -r irn_amd/data/miopen/* $OUT/miopen_db/ 2>/dev/null; du -sh $OUT/miopen_db
run seeded IRN_MIOPEN_CACHE=/tmp/mi_seeded
ls -la /tmp/mi_seeded/*/dev0 2>/dev/null | head
