#!/bin/bash
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s28
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_steps.py tests/test_gpu_labels_instance.py -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
timeout 600 python bench.py --workload cam --steps 3 --warmup 1 --json-out $O/bench_cam.json > $O/bench_cam.log 2>&1; tail -1 $O/bench_cam.log | cut -c1-230
timeout 900 python bench.py --workload e2e --steps 3 --warmup 1 --json-out $O/bench_e2e.json > $O/bench_e2e.log 2>&1; tail -1 $O/bench_e2e.log | cut -c1-230
