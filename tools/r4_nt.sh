#!/bin/bash
set -u
cd $GRAFT_REPO_ROOT
export K1_SLOTS=32
for i in 1 2 3; do
  for C in 1 3; do
    [ $C = 3 ] && export K1_SLOTS=12 || export K1_SLOTS=32
    echo -n "C=$C in-tree: "; SWEEP_C=$C python3 tools/k1_run.py 300 | tail -1
    echo -n "C=$C ntload : "; SWEEP_C=$C FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/build_variants/libfri_hip_ntload.so python3 tools/k1_run.py 300 | tail -1
  done
done
python3 tools/k1_sweep_hbm.py --slots 32 --launches 300 --rounds 3 "" "FRI_HIP_TARGET_WGS=1280" "FRI_HIP_TARGET_WGS=1280 FRI_HIP_RANK_WEIGHTS=1.25,1.1,0.95,0.8" "FRI_HIP_RANK_WEIGHTS=1.35,1.1,0.9,0.65" "FRI_HIP_RANK_WEIGHTS=1.3,1.15,0.9,0.65" "FRI_HIP_RANK_WEIGHTS=1.25,1.1,0.9,0.75" "FRI_HIP_CELLS_PER_TILE=7"
