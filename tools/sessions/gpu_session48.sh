#!/bin/bash
# GPU session 48: degree summed inside the resident kernel (no degree_kernel launch on the resident path).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s48
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_walk.py -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --json-out $O/bench_default.json > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
timeout 300 python bench.py --workload walk_r5 --no-cpu-baseline --json-out $O/bench_r5.json > $O/bench_r5.log 2>&1; tail -1 $O/bench_r5.log | cut -c1-200
timeout 300 python bench.py --workload ins --no-cpu-baseline --json-out $O/bench_ins.json > $O/bench_ins.log 2>&1; tail -1 $O/bench_ins.log | cut -c1-200
