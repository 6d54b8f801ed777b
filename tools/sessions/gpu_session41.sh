#!/bin/bash
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s41
mkdir -p $O
for D in 6 8 10 12 14 16; do
timeout 100 python tools/resident_profile.py 10 4 1 poll_delay=$D 2>&1 | grep "wg 0" >> $O/prof.log
timeout 100 python tools/resident_profile.py 5 16 1 poll_delay=$D 2>&1 | grep "wg 0" >> $O/prof.log
done
cat $O/prof.log
