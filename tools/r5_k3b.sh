#!/bin/bash
# Round 5: K3 v2 (prefetched coefficients landed in front of the tile's stores, the items' chains in one block, no run-time division, list entries without clamps,
# 24-bit multiplies in the write-out) against the previous library: parity tests, then interleaved A/B over rotating planes (plane, RGB, 16384^2).
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_tune.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for r in 1 2 3; do for L in "" build_variants/libfri_hip_k3old.so; do
  echo -n "${L:-in-tree (v2)}: "; FRI_HIP_LIBRARY=$L K2_SLOTS=12 K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/hist_blocks=default//'
done; done | tee $OUT/k3_v2.txt
for r in 1 2; do for L in "" build_variants/libfri_hip_k3old.so; do
  echo -n "RGB ${L:-in-tree (v2)}: "; SWEEP_C=3 FRI_HIP_LIBRARY=$L K2_SLOTS=4 K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/hist_blocks=default//'
done; done | tee -a $OUT/k3_v2.txt
for L in "" build_variants/libfri_hip_k3old.so; do
  echo -n "16384 ${L:-in-tree (v2)}: "; K2_SIZE=16384 FRI_HIP_LIBRARY=$L K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/hist_blocks=default//'
done | tee -a $OUT/k3_v2.txt
