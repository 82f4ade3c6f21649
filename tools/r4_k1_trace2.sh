#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/frave_amd/libfri_hip_tuning.so TRACE_SLOTS=40
FRI_HIP_STRIDED_SHARES=1 python3 tools/trace_timeline.py k1 1 > $OUT/trace_strided.log 2>&1
FRI_HIP_STRIDED_SHARES=1 FRI_HIP_BAND_ROWS=8 python3 tools/trace_timeline.py k1 1 > $OUT/trace_strided_b8.log 2>&1
FRI_HIP_BAND_ROWS=72 python3 tools/trace_timeline.py k1 1 > $OUT/trace_b72.log 2>&1
head -22 $OUT/trace_strided.log; tail -14 $OUT/trace_strided.log
