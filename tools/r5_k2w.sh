#!/bin/bash
# Round 5: K2's long first tile - is it a CU that starts issuing from idle? The tuning build spins n x 8 multiply-adds per lane in the prologue while the first tile's staging
# loads are in flight (FRI_HIP_K2_ABLATE = n << 8); per-tile phases by time stamps, and the kernel's duration.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for n in 0 100 300 600 0 300; do
  echo "== warm-up iterations $n"
  FRI_HIP_K2_ABLATE=$((n << 8)) FRI_HIP_LIBRARY=frave_amd/libfri_hip_tuning.so python3 tools/trace_k2_phases.py 2>&1 | grep -v amdgpu.ids | head -4
  FRI_HIP_K2_ABLATE=$((n << 8)) FRI_HIP_LIBRARY=frave_amd/libfri_hip_tuning.so K2_SLOTS=12 K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | grep -o "K2 *[0-9.]* us"
done | tee $OUT/k2_warm.txt
