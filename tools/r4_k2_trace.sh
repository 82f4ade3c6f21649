#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/frave_amd/libfri_hip_tuning.so FRI_HIP_TUNING=1
timeout -k 10 200 python3 tools/trace_k2.py > $OUT/trace_k2.log 2>&1; cat $OUT/trace_k2.log
