#!/bin/bash
# GPU session 54: last check of the round: full GPU tests on the final build, default bench (also through torch.distributed.run).
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s54
mkdir -p $O
timeout 400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --json-out $O/bench_default.json > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
