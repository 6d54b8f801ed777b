#!/bin/bash
# Diagnosis of round 3's session-6 hang: what does RCCL do when two ranks share ONE device?  Runs the two-rank bench with
# the RCCL probe forced (IRN_RCCL_ALLOW_SHARED=1) under a 30 s probe deadline and NCCL_DEBUG=WARN; the run must end in the
# gloo group either way.  usage: tools/rccl_shared_probe.sh <outdir>
OUT=${1:-gpurun_out}
T0=$(date +%s)
IRN_RCCL_ALLOW_SHARED=1 IRN_RCCL_PROBE_TIMEOUT_S=30 NCCL_DEBUG=WARN timeout 300 python bench.py --gpus 2 --rank-devices 0,0 --backend auto \
   --no-legs --no-cpu-baseline --batch 16 --steps 2 --warmup 1 --launch-timeout-s 240 > $OUT/rccl_shared.out 2> $OUT/rccl_shared.err
echo "rc=$? wall $(( $(date +%s) - T0 )) s"
grep -E "RCCL|NCCL WARN|Duplicate|process_group|irn_amd.parallel" $OUT/rccl_shared.err | head -20
python - <<PY
import json
try:
    l=[x for x in open("$OUT/rccl_shared.out") if x.startswith("{")][-1]; d=json.loads(l)
    print("line:", d["n_gpus"], d["value"], d["config"]["process_group"])
except Exception as e: print("no line", e)
PY
