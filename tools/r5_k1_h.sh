#!/bin/bash
# Round 5: shares cut by cost (rim tiles weigh more) on contiguous shares - A/B and the timeline of the winner.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
C72=FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72
AB_BATCH=24 python3 tools/k1_ab_hbm.py 3 -:$C72 -:$C72,FRI_HIP_RIM_COST=50 -:$C72,FRI_HIP_RIM_COST=100 -:$C72,FRI_HIP_RIM_COST=200 -:$C72,FRI_HIP_RIM_COST=100,FRI_HIP_RANK_WEIGHTS=1.4,1.15,0.85,0.6 - -:AB_TUNE=1 > $OUT/ab_rim.log 2>&1
cat $OUT/ab_rim.log
T=frave_amd/libfri_hip_tuning.so
FRI_HIP_LIBRARY=$T TRACE_SLOTS=40 FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72 FRI_HIP_RIM_COST=100 python3 tools/trace_timeline.py k1 > $OUT/timeline_c72_rim100.log 2>&1
sed -n 1,32p $OUT/timeline_c72_rim100.log; tail -14 $OUT/timeline_c72_rim100.log
