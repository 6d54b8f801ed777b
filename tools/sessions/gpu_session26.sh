#!/bin/bash
set -x
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
O=$R/gpurun_out/s26
mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $O/prof_lds -o p -f csv -- python $R/tools/resident_profile.py 10 8 3 > $O/lds.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS -d $O/prof_sq2 -o p -f csv -- python $R/tools/resident_profile.py 10 8 3 > $O/sq2.log 2>&1
cd $R
python tools/reduce_prof.py gpurun_out/s26 2>&1 | grep -i "resident" | cut -c1-20,95-200
find gpurun_out/s26 -name "*counter_collection.csv" -delete
