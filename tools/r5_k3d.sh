#!/bin/bash
# Round 5: K3 with 512-thread workgroups (one item per wave at eight cells per tile) against 256, and smaller tiles; interleaved, rotating planes.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; shift; env "$@" K2_SLOTS=12 K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | grep -o "K3 *[0-9.]* us *\|roundtrip=[A-Za-z]*" | tr '\n' ' '; echo; }
for r in 1 2; do
  run "256 threads (in-tree)" A=0
  run "512 threads, 4 shares per CU" FRI_HIP_LIBRARY=build_variants/libfri_hip_k3t512.so
  run "512 threads, 3 shares per CU" FRI_HIP_LIBRARY=build_variants/libfri_hip_k3t512.so FRI_HIP_RANKS=3
  run "512 threads, 2 shares per CU" FRI_HIP_LIBRARY=build_variants/libfri_hip_k3t512.so FRI_HIP_RANKS=2
  run "256 threads, 4 cells per tile" FRI_HIP_CELLS_PER_TILE=4
  run "256 threads, 6 cells per tile" FRI_HIP_CELLS_PER_TILE=6
  run "256 threads, bands of 16 rows" FRI_HIP_INV_BAND_ROWS=16
  run "256 threads, bands of 48 rows" FRI_HIP_INV_BAND_ROWS=48
done | tee $OUT/k3_threads.txt
