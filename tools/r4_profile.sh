#!/bin/bash
# Per-step time stamps of the resident walk (PROF instantiation) for the classes of jobs the benches are made of.
# usage: tools/r4_profile.sh <outdir> [lib.so]
OUT=${1:-gpurun_out}; LIB=${2:-}
[ -n "$LIB" ] && export IRN_HIP_LIB=$PWD/irn_amd/lib/$LIB
for cfg in "5 32 1" "5 32 2" "5 32 3" "10 8 1" "10 8 2" "10 8 3"; do
  timeout 120 python tools/resident_profile.py $cfg 2>&1 | tail -4
done
