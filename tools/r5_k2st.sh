#!/bin/bash
# Round 5: K2 with plain instead of nontemporal output stores (K5's order loads were faster without the hint): standalone over rotating planes, and inside the chains.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for L in "" build_variants/libfri_hip_k2plainst.so; do
  echo "${L:-in-tree (nt stores)}: $(FRI_HIP_LIBRARY=$L K2_SLOTS=12 K2_TRUSTED=1 K5=1 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ' | sed 's/hist_blocks=default//; s/roundtrip.*//')"
done; done | tee $OUT/k2_plain_stores.txt
