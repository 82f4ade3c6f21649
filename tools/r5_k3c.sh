#!/bin/bash
# Round 5: what K3's time is made of - timing-only ablations of the tuning build (FRI_HIP_K3_ABLATE: 1 no global stores, 2 no LDS scatter, 4 no transform, 8 coefficient loads for the first tile only).
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in 1 2; do for A in 0 1 2 4 8 3 5 12 13 15; do
  echo -n "ablate $A: "; FRI_HIP_K3_ABLATE=$A FRI_HIP_LIBRARY=frave_amd/libfri_hip_tuning.so K2_SLOTS=12 K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/hist_blocks=default//' | grep -o "K3 *[0-9.]* us"
done; done | tee $OUT/k3_ablate.txt
