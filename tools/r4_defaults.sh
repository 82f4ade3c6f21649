#!/bin/bash
# Round 4: the new tiling defaults (interleaved shares, forward bands of 16 rows, the inverse kernel's own tiling) against round 3's, in the HBM-bound regime.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
OLD="FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=32 FRI_HIP_INV_SHARED=1"
OLD3="FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=16 FRI_HIP_INV_SHARED=1 FRI_HIP_RANK_WEIGHTS=1.3,1.1,0.6,0"
python3 tools/k1_sweep_hbm.py --slots 32 --launches 300 --rounds 3 "" "$OLD" "FRI_HIP_BAND_ROWS=8" "FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72" > $OUT/c1_4096.log 2>&1
SWEEP_C=3 python3 tools/k1_sweep_hbm.py --slots 12 --launches 200 --rounds 3 "" "$OLD3" > $OUT/c3_4096.log 2>&1
SWEEP_W=1920 SWEEP_H=1080 python3 tools/k1_sweep_hbm.py --slots 256 --launches 512 --rounds 3 "" "$OLD" > $OUT/1080p.log 2>&1
SWEEP_W=1920 SWEEP_H=1080 SWEEP_C=3 python3 tools/k1_sweep_hbm.py --slots 128 --launches 256 --rounds 3 "" "$OLD3" > $OUT/1080p_c3.log 2>&1
SWEEP_W=6000 SWEEP_H=4000 python3 tools/k1_sweep_hbm.py --slots 24 --launches 200 --rounds 3 "" "$OLD" > $OUT/6000.log 2>&1
SWEEP_W=2048 SWEEP_H=2048 python3 tools/k1_sweep_hbm.py --slots 128 --launches 400 --rounds 3 "" "$OLD" > $OUT/2048.log 2>&1
SWEEP_W=8192 SWEEP_H=8192 python3 tools/k1_sweep_hbm.py --slots 8 --launches 100 --rounds 3 "" "$OLD" > $OUT/8192.log 2>&1
SWEEP_W=16384 SWEEP_H=16384 python3 tools/k1_sweep_hbm.py --slots 2 --launches 30 --rounds 3 "" "$OLD" > $OUT/16384.log 2>&1
cat $OUT/c1_4096.log $OUT/c3_4096.log $OUT/1080p.log $OUT/1080p_c3.log $OUT/6000.log $OUT/2048.log $OUT/8192.log $OUT/16384.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
