#!/bin/bash
# Round 5: where K1's time goes now - timing-only ablations (tuning build) for the single launch and for 24 images per launch, default and tuned tiling;
# the per-workgroup timeline of both tilings with every byte from HBM.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
T=frave_amd/libfri_hip_tuning.so
L=""
for a in 0 1 2 3 4 5 6 7; do L="$L $T:FRI_HIP_K1_ABLATE=$a"; done
AB_BATCH=24 python3 tools/k1_ab_hbm.py 2 $L > $OUT/ablate_default.log 2>&1
cat $OUT/ablate_default.log
L=""
for a in 0 1 2 3 4 5 6 7; do L="$L $T:FRI_HIP_K1_ABLATE=$a,FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72"; done
AB_BATCH=24 python3 tools/k1_ab_hbm.py 2 $L > $OUT/ablate_c72.log 2>&1
cat $OUT/ablate_c72.log
FRI_HIP_LIBRARY=$T TRACE_SLOTS=40 python3 tools/trace_timeline.py k1 > $OUT/timeline_default.log 2>&1
FRI_HIP_LIBRARY=$T TRACE_SLOTS=40 FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72 python3 tools/trace_timeline.py k1 > $OUT/timeline_c72.log 2>&1
head -32 $OUT/timeline_default.log; head -32 $OUT/timeline_c72.log
