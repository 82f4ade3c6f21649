#!/bin/bash
# Round 5: prototype of a per-share feedback on measured exit times (tools/k1_share_feedback.py, tuning build).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FRI_HIP_LIBRARY=frave_amd/libfri_hip_tuning.so timeout -k 10 800 python3 tools/k1_share_feedback.py 8 2>&1 | grep -v amdgpu.ids | tee $OUT/feedback.txt
