#!/bin/bash
# round 5 session 8: kernel-class profiles of cam / e2e (the profiled run now finds the shipped database), walk trace + PMC traffic
set -u
OUT=gpurun_out/${S:-r5_s8}; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
R=$PWD
for wl in cam e2e; do
  cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$wl -o $wl -f csv -- python $R/bench.py --workload $wl --steps 5 --warmup 2 --no-legs --no-cpu-baseline --json-out $R/$OUT/bench_$wl.json > $R/$OUT/prof_$wl.log 2>&1; cd $R
  find $OUT/prof_$wl -name "*kernel_stats*" -exec cp {} $OUT/${wl}_kernel_stats.csv \;
  python -c "import json; r=json.load(open('$OUT/bench_$wl.json')); print('$wl under the profiler: %.1f images/s' % r['value'], r['config']['trunk'])"
  python tools/kernel_classes.py $OUT/${wl}_kernel_stats.csv 30 > $OUT/${wl}_kernel_classes.txt 2>&1; head -22 $OUT/${wl}_kernel_classes.txt
  find $OUT -name "*_kernel_trace.csv" -delete
done; el "backbone profiles"
