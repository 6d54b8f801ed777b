#!/bin/bash
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s38
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_resident.py -q -x > $O/pytest_resident.log 2>&1; echo "pytest rc=$?" >> $O/pytest_resident.log; tail -3 $O/pytest_resident.log
for D in 20 24 28 32 36; do
timeout 100 python tools/resident_profile.py 10 4 1 poll_delay=$D 2>&1 | grep "wg 0" >> $O/prof.log
timeout 100 python tools/resident_profile.py 5 16 1 poll_delay=$D 2>&1 | grep "wg 0" >> $O/prof.log
done
cat $O/prof.log
