#!/bin/bash
set -u
OUT=gpurun_out/r6_s14; mkdir -p $OUT
export TMPDIR=/tmp MIOPEN_FIND_MODE=2
timeout 300 python -m pytest tests/test_gpu_split_gemm.py -m gpu -q > $OUT/pytest_split.log 2>&1; echo "split tests rc=$?"; tail -2 $OUT/pytest_split.log
R=$PWD; cd /tmp
for wl in e2e cam; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$wl -o $wl -f csv -- python $R/bench.py --workload $wl --steps 7 --warmup 2 --no-legs --no-cpu-baseline > $R/$OUT/prof_$wl.log 2>&1
find $R/$OUT/prof_$wl -name "*kernel_stats*" -exec cp {} $R/$OUT/${wl}_kernel_stats.csv \;
python $R/tools/kernel_classes.py $R/$OUT/${wl}_kernel_stats.csv 30 > $R/$OUT/${wl}_kernel_classes.txt 2>&1
done
cd $R; cat $OUT/e2e_kernel_classes.txt
find $OUT -name "*kernel_trace.csv" -delete
