#!/bin/bash
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s11
mkdir -p $O
timeout 600 python tools/cam_tune.py fast > $O/cam_tune_fast.log 2>&1; grep "img/s" $O/cam_tune_fast.log
timeout 1200 python tools/cam_tune.py normal > $O/cam_tune_normal.log 2>&1; grep "img/s" $O/cam_tune_normal.log
