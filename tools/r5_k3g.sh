#!/bin/bash
# Round 5: K3 without its second barrier per tile (timing only, tuning build, FRI_HIP_K3_ABLATE=16): the most a double-buffered tile rectangle could save.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for A in 0 16; do
  echo -n "ablate $A: "; FRI_HIP_K3_ABLATE=$A FRI_HIP_LIBRARY=frave_amd/libfri_hip_tuning.so K2_SLOTS=12 K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | grep -o "K3 *[0-9.]* us"
  echo -n "ablate $A, bands of 16 rows: "; FRI_HIP_INV_BAND_ROWS=16 FRI_HIP_K3_ABLATE=$A FRI_HIP_LIBRARY=frave_amd/libfri_hip_tuning.so K2_SLOTS=12 K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | grep -o "K3 *[0-9.]* us"
done; done | tee $OUT/k3_barrier.txt
