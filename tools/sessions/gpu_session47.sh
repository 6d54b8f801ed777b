#!/bin/bash
# GPU session 47: device input pipeline (irn_msf_pack): parity tests, step integration, timing, cam bench leg.
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s47
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_msf.py tests/test_gpu_steps.py -x -q -m gpu > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 200 python tools/msf_bench.py > $O/msf_bench.log 2>&1; tail -4 $O/msf_bench.log
timeout 200 python tools/msf_bench.py 375 500 >> $O/msf_bench.log 2>&1; tail -3 $O/msf_bench.log
timeout 400 python bench.py --workload cam --steps 3 --warmup 1 --json-out $O/bench_cam.json > $O/bench_cam.log 2>&1; tail -1 $O/bench_cam.log | cut -c1-300
