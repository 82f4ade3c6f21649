#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
S="FRI_HIP_STRIDED_SHARES=1"
SWEEP_C=3 python3 tools/k1_sweep_hbm.py --slots 12 --launches 200 --rounds 3 "" "$S" "$S FRI_HIP_RANK_WEIGHTS=1.2,1.0,0.8,0" "$S FRI_HIP_RANK_WEIGHTS=1.1,1.0,0.9,0" "$S FRI_HIP_BAND_ROWS=24" "$S FRI_HIP_BAND_ROWS=40" "$S FRI_HIP_BAND_ROWS=64" "$S FRI_HIP_BAND_ROWS=8" > $OUT/c3.log 2>&1
SWEEP_W=1920 SWEEP_H=1080 python3 tools/k1_sweep_hbm.py --slots 256 --launches 512 --rounds 3 "" "$S" > $OUT/1080p.log 2>&1
SWEEP_W=6000 SWEEP_H=4000 python3 tools/k1_sweep_hbm.py --slots 24 --launches 200 --rounds 3 "" "$S" "$S FRI_HIP_BAND_ROWS=8" "$S FRI_HIP_BAND_ROWS=16" > $OUT/6000.log 2>&1
SWEEP_W=2048 SWEEP_H=2048 python3 tools/k1_sweep_hbm.py --slots 128 --launches 400 --rounds 3 "" "$S" > $OUT/2048.log 2>&1
SWEEP_W=8192 SWEEP_H=8192 python3 tools/k1_sweep_hbm.py --slots 8 --launches 100 --rounds 3 "" "$S" > $OUT/8192.log 2>&1
python3 tools/k1_sweep_hbm.py --slots 32 --launches 300 --rounds 3 "" "$S" "$S FRI_HIP_BAND_ROWS=16" "$S FRI_HIP_BAND_ROWS=12" "$S FRI_HIP_BAND_ROWS=20" "$S FRI_HIP_BAND_ROWS=24" "$S FRI_HIP_BAND_ROWS=16 FRI_HIP_RANK_WEIGHTS=1.2,1.07,0.93,0.8" "$S FRI_HIP_BAND_ROWS=8 FRI_HIP_RANK_WEIGHTS=1.2,1.07,0.93,0.8" > $OUT/c1.log 2>&1
cat $OUT/c3.log $OUT/1080p.log $OUT/6000.log $OUT/2048.log $OUT/8192.log $OUT/c1.log
