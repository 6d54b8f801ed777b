#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r6_s11; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD; cd /tmp
echo "== trivial torch script under rocprofv3 --kernel-trace"
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p0 -o t -f csv -- python -c "import torch; x=torch.randn(1024,1024,device='cuda'); print((x@x).sum().item())" > $OUT/p0.log 2>&1; echo "rc=$?"; ls $OUT/p0 2>/dev/null | head
echo "== import irn_amd lib only"
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p1 -o t -f csv -- python -c "import sys; sys.path.insert(0,'$R'); import torch; from irn_amd import ops; x=torch.randn(1024,1024,device='cuda'); print((x@x).sum().item())" > $OUT/p1.log 2>&1; echo "rc=$?"
echo "== bench walk under --pmc FETCH_SIZE"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/p2 -o t -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-legs --no-cpu-baseline --no-traffic > $OUT/p2.log 2>&1; echo "rc=$?"; find $OUT/p2 -name "*.csv" | head; f=$(find $OUT/p2 -name "*counter_collection.csv" | head -1); [ -n "$f" ] && (head -3 $f; grep -c resident_kernel $f)
echo "== bench walk under --pmc, HSA_... exit via os._exit hook"
IRN_BENCH_HARD_EXIT=1 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/p3 -o t -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-legs --no-cpu-baseline --no-traffic > $OUT/p3.log 2>&1; echo "rc=$?"; find $OUT/p3 -name "*.csv" | head
grep -n "SIGSEGV" $OUT/p*.log | head
find $OUT -name "*kernel_trace.csv" -delete
