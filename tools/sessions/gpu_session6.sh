#!/bin/bash
# GPU session 6: defaults (tile 7, per-width streams, batch 192): full tests, all workloads, final profiles.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/s6
O=gpurun_out/s6
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
for T in 0 4 7; do timeout 300 python bench.py --workload walk_r5 --tile $T --no-cpu-baseline > $O/bench_r5_t$T.log 2>&1; tail -1 $O/bench_r5_t$T.log | cut -c1-200; done
timeout 600 python bench.py --workload ins --no-cpu-baseline > $O/bench_ins.log 2>&1; tail -1 $O/bench_ins.log | cut -c1-200
timeout 900 python bench.py --workload coco --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_coco.log 2>&1; tail -1 $O/bench_coco.log | cut -c1-200
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_trace -o walk -f csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$O/rocprof_trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/$O/prof_fetch -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/$O/prof_write -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/rocprof_write.log 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/$O/prof_tcc -o walk -f csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/$O/rocprof_tcc.log 2>&1
cd $R
python tools/reduce_prof.py $O > $O/prof_summary.txt 2>&1
rm -f $O/prof_*/walk_kernel_trace.csv $O/prof_*/walk_counter_collection.csv
du -sh $O
