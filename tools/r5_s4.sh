#!/bin/bash
# round 5 session 4: full GPU suite (tile-local labelling, deterministic / tie-proof two-worker test, N > 1 bench workloads, hung RCCL probe),
# instance stage after the labelling change, host capacity on this box's cores
set -u
OUT=gpurun_out/r5_s4; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 python -m pytest tests/test_gpu_labels_instance.py -m gpu -q -x > $OUT/pytest_instance.log 2>&1; el "instance tests rc=$?"; tail -3 $OUT/pytest_instance.log
timeout 300 python tools/ins_step_breakdown.py 5 64 5 > $OUT/ins_breakdown_r5.txt 2>&1; cat $OUT/ins_breakdown_r5.txt
for b in 64 128 256; do timeout 300 python bench.py --workload ins --batch $b --steps 10 --warmup 2 --no-legs --no-cpu-baseline --json-out $OUT/bench_ins_b$b.json > $OUT/bench_ins_b$b.log 2>&1; python -c "
import json; r=json.load(open('$OUT/bench_ins_b$b.json')); print('ins batch $b: %.0f images/s' % r['value'])"; done; el "ins bench"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --durations=10 -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "differ|two ranks|two workers|hung|passed|failed|FAILED|rc=|Error" $OUT/pytest_gpu.log | tail -50; el "gpu tests"
timeout 600 python tools/host_capacity_probe.py --procs 1,8 --images 192 > $OUT/host_capacity.txt 2>&1; cat $OUT/host_capacity.txt; el "host capacity"
