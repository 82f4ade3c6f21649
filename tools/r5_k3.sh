#!/bin/bash
# Round 5: K3 with plain instead of nontemporal coefficient loads (K5's order loads were 10 % faster plain), interleaved A/B over rotating planes; K5 after its change.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for L in "" build_variants/libfri_hip_k3plain.so; do
  echo -n "${L:-in-tree (nt loads)}: "; FRI_HIP_LIBRARY=$L K2_SLOTS=12 K2_TRUSTED=1 K5=1 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/hist_blocks=default//'
done; done | tee $OUT/k3_plain.txt
for r in 1 2; do for L in "" build_variants/libfri_hip_k3plain.so; do
  echo -n "RGB ${L:-in-tree (nt loads)}: "; SWEEP_C=3 FRI_HIP_LIBRARY=$L K2_SLOTS=4 K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/hist_blocks=default//'
done; done | tee -a $OUT/k3_plain.txt
