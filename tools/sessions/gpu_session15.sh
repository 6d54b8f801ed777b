#!/bin/bash
set -x
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/s15
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_resident.py -q -x > $O/pytest_resident.log 2>&1; echo "pytest rc=$?" >> $O/pytest_resident.log; tail -5 $O/pytest_resident.log
for D in 0 2 4 6 8 10 12 16; do for S in 0 3 6; do
timeout 100 python tools/resident_profile.py 10 4 1 poll_delay=$D poll_stagger=$S 2>&1 | grep "wg 0" >> $O/prof.log
done; done
timeout 100 python tools/resident_profile.py 10 4 3 poll_delay=4 poll_stagger=4 2>&1 | grep "wg 0" >> $O/prof.log
timeout 100 python tools/resident_profile.py 10 4 3 poll_delay=8 poll_stagger=4 2>&1 | grep "wg 0" >> $O/prof.log
timeout 100 python tools/resident_profile.py 5 16 1 poll_delay=4 poll_stagger=4 2>&1 | grep "wg 0" >> $O/prof.log
cat $O/prof.log
