#!/bin/bash
# GPU session 53: round-1 verification of the final state: full GPU tests, smoke, every bench workload, rocprof kernel
# stats and PMC traffic of the default bench command.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s53
mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --json-out $O/bench_default.json > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
timeout 200 python bench.py --workload walk_r5 --no-cpu-baseline --json-out $O/bench_r5.json > $O/bench_r5.log 2>&1; tail -1 $O/bench_r5.log | cut -c1-200
timeout 200 python bench.py --workload ins --no-cpu-baseline --json-out $O/bench_ins.json > $O/bench_ins.log 2>&1; tail -1 $O/bench_ins.log | cut -c1-200
timeout 200 python bench.py --workload coco --steps 2 --warmup 1 --no-cpu-baseline --json-out $O/bench_coco.json > $O/bench_coco.log 2>&1; tail -1 $O/bench_coco.log | cut -c1-200
timeout 200 python bench.py --workload e2e --steps 3 --warmup 1 --json-out $O/bench_e2e.json > $O/bench_e2e.log 2>&1; tail -1 $O/bench_e2e.log | cut -c1-200
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_trace -o walk -f csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$O/prof_fetch -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/rocprof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$O/prof_write -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/rocprof_write.log 2>&1
cd $R
python tools/reduce_prof.py $O > $O/prof_summary.txt 2>&1
find $O/prof_trace -name "*kernel_stats*" -exec cp {} $O/kernel_stats.csv \;
find $O -name "walk_kernel_trace.csv" -delete; find $O -name "walk_counter_collection.csv" -delete
grep -h "resident_k" $O/prof_summary.txt | cut -c1-20,100-200
