#!/bin/bash
# Round 5: K2 with one load per lane from its first output cells in the prologue (is the first tile's extra time the address translation of its first stores?); interleaved A/B.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for L in "" build_variants/libfri_hip_k2warm.so; do
  echo -n "${L:-in-tree}: "; FRI_HIP_LIBRARY=$L K2_SLOTS=12 K2_TRUSTED=1 K5=1 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ' | grep -o "chain forward.*gather *[0-9.]* us\|K2 *[0-9.]* us\|roundtrip=[A-Za-z]*" | tr '\n' ' '; echo
done; done | tee $OUT/k2_warm_pages.txt
