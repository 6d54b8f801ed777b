#!/bin/bash
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/s12
mkdir -p $O
timeout 900 python tools/cam_profile.py > $O/cam_profile.log 2>&1; grep "ms/img" $O/cam_profile.log
