#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
SWEEP_W=2048 SWEEP_H=2048 python3 tools/k1_sweep_hbm.py --slots 128 --launches 400 --rounds 3 "" "FRI_HIP_BAND_ROWS=8" "FRI_HIP_BAND_ROWS=16" "FRI_HIP_BAND_ROWS=48" "FRI_HIP_BAND_ROWS=64" "FRI_HIP_BAND_ROWS=72" "FRI_HIP_BAND_ROWS=80" > $OUT/sweep_2048.log 2>&1
SWEEP_W=1920 SWEEP_H=1080 python3 tools/k1_sweep_hbm.py --slots 256 --launches 512 --rounds 3 "" "FRI_HIP_BAND_ROWS=8" "FRI_HIP_BAND_ROWS=16" "FRI_HIP_BAND_ROWS=48" "FRI_HIP_BAND_ROWS=64" "FRI_HIP_BAND_ROWS=72" "FRI_HIP_BAND_ROWS=80" > $OUT/sweep_1080p.log 2>&1
SWEEP_W=6000 SWEEP_H=4000 python3 tools/k1_sweep_hbm.py --slots 24 --launches 200 --rounds 3 "" "FRI_HIP_BAND_ROWS=8" "FRI_HIP_BAND_ROWS=16" "FRI_HIP_BAND_ROWS=48" "FRI_HIP_BAND_ROWS=64" "FRI_HIP_BAND_ROWS=72" "FRI_HIP_BAND_ROWS=80" > $OUT/sweep_6000.log 2>&1
SWEEP_W=8192 SWEEP_H=8192 python3 tools/k1_sweep_hbm.py --slots 8 --launches 100 --rounds 3 "" "FRI_HIP_BAND_ROWS=8" "FRI_HIP_BAND_ROWS=16" "FRI_HIP_BAND_ROWS=64" "FRI_HIP_BAND_ROWS=72" "FRI_HIP_BAND_ROWS=80" > $OUT/sweep_8192.log 2>&1
cat $OUT/sweep_2048.log $OUT/sweep_1080p.log $OUT/sweep_6000.log $OUT/sweep_8192.log
