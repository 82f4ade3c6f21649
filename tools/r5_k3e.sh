#!/bin/bash
# Round 5: K3 with larger tiles (more items per wave, fewer barriers and list loads per byte) at 4 / 3 / 2 shares per CU; interleaved, rotating planes.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; shift; env "$@" K2_SLOTS=12 K2_TRUSTED=1 K5=0 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids | tail -1 | grep -o "K3 *[0-9.]* us *\|roundtrip=[A-Za-z]*" | tr '\n' ' '; echo; }
for r in 1 2; do
  run "8 cells (in-tree)" A=0
  for c in 10 12 16; do for k in 4 3 2; do
    run "$c cells per tile, $k shares per CU" FRI_HIP_CELLS_PER_TILE=$c FRI_HIP_RANKS=$k
  done; done
done | tee $OUT/k3_tiles.txt
