#!/bin/bash
# round 5 session 5: batch-composition dependence of the instance path, host capacity (realistic PNG payload), fused-affinity prologue
# probe, MIOpen's own fused ops, instance legs at batch 128
set -u
OUT=gpurun_out/r5_s5; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 python tools/debug_ins_batch.py > $OUT/debug_ins_batch.txt 2>&1; grep -v "^MIOpen\|amdgpu.ids" $OUT/debug_ins_batch.txt; el "ins batch debug"
timeout 120 tools/bin/affinity_prologue_probe 64 > $OUT/affinity_prologue_probe.txt 2>&1; cat $OUT/affinity_prologue_probe.txt; el "prologue probe"
timeout 300 python tools/miopen_fused_probe.py > $OUT/miopen_fused_probe.txt 2>&1; grep -v "^MIOpen\|amdgpu.ids" $OUT/miopen_fused_probe.txt; el "miopen fused probe"
for wl in ins ins_r10; do for b in 64 128; do timeout 300 python bench.py --workload $wl --batch $b --steps 10 --warmup 2 --no-legs --no-cpu-baseline --json-out $OUT/bench_${wl}_b$b.json > $OUT/bench_${wl}_b$b.log 2>&1; python -c "
import json; r=json.load(open('$OUT/bench_${wl}_b$b.json')); print('$wl batch $b: %.0f images/s' % r['value'])"; done; done; el "ins bench"
timeout 300 python tools/ins_step_breakdown.py 5 128 5 > $OUT/ins_breakdown_r5.txt 2>&1; grep -v "amdgpu.ids" $OUT/ins_breakdown_r5.txt
timeout 600 python tools/host_capacity_probe.py --procs 1,8 --images 192 --demand-steps 79 > $OUT/host_capacity.txt 2>&1; cat $OUT/host_capacity.txt; el "host capacity"
