#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 tools/k1_sweep_hbm.py --slots 32 --launches 300 --rounds 3 "" "FRI_HIP_BAND_ROWS=8" "FRI_HIP_BAND_ROWS=16" "FRI_HIP_BAND_ROWS=60" "FRI_HIP_BAND_ROWS=72" "FRI_HIP_BAND_ROWS=76" "FRI_HIP_BAND_ROWS=80" "FRI_HIP_BAND_ROWS=88" > $OUT/sweep_c1.log 2>&1
SWEEP_C=3 python3 tools/k1_sweep_hbm.py --slots 12 --launches 200 --rounds 3 "" "FRI_HIP_BAND_ROWS=20" "FRI_HIP_BAND_ROWS=40" "FRI_HIP_BAND_ROWS=64" "FRI_HIP_BAND_ROWS=72" "FRI_HIP_BAND_ROWS=80" > $OUT/sweep_c3.log 2>&1
cat $OUT/sweep_c1.log $OUT/sweep_c3.log
