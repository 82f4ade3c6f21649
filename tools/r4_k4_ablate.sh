#!/bin/bash
# K4: timing-only ablations (tuning build) and the per-workgroup timeline
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/frave_amd/libfri_hip_tuning.so FRI_HIP_TUNING=1
for a in 0 1 2 3; do
  echo "== FRI_HIP_K4_ABLATE=$a (1: no sums, 2: no staging loads)" >> $OUT/ablate.log
  FRI_HIP_K4_ABLATE=$a K2_SLOTS=24 timeout -k 10 200 python3 tools/k2_time.py 2>&1 | grep slots >> $OUT/ablate.log
done
cat $OUT/ablate.log
K4_MODE=0 timeout -k 10 200 python3 tools/trace_k4.py > $OUT/trace0.log 2>&1; cat $OUT/trace0.log
K4_MODE=1 timeout -k 10 200 python3 tools/trace_k4.py > $OUT/trace1.log 2>&1; cat $OUT/trace1.log
