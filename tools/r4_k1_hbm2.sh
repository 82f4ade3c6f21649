#!/bin/bash
# Round 4: band height / tile size of K1 in the HBM-bound regime (32 rotating slots), and the streaming floor of the memory system for K1's byte mix.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
make -s -C tools/micro stream_floor
./tools/micro/stream_floor 32 300 > $OUT/stream_floor_32.log 2>&1
./tools/micro/stream_floor 4 300 > $OUT/stream_floor_4.log 2>&1
python3 tools/k1_sweep_hbm.py --slots 32 --launches 300 --rounds 3 "" "FRI_HIP_BAND_ROWS=8" "FRI_HIP_BAND_ROWS=12" "FRI_HIP_BAND_ROWS=16" "FRI_HIP_BAND_ROWS=20" "FRI_HIP_BAND_ROWS=24" "FRI_HIP_BAND_ROWS=28" \
   "FRI_HIP_BAND_ROWS=36" "FRI_HIP_BAND_ROWS=40" "FRI_HIP_BAND_ROWS=48" "FRI_HIP_BAND_ROWS=56" "FRI_HIP_BAND_ROWS=64" "FRI_HIP_BAND_ROWS=96" > $OUT/sweep_band.log 2>&1
python3 tools/k1_sweep_hbm.py --slots 32 --launches 300 --rounds 3 "" "FRI_HIP_BAND_ROWS=48 FRI_HIP_CELLS_PER_TILE=10 FRI_HIP_TILE_BYTES=24576" "FRI_HIP_BAND_ROWS=48 FRI_HIP_CELLS_PER_TILE=12 FRI_HIP_TILE_BYTES=24576" \
   "FRI_HIP_BAND_ROWS=16 FRI_HIP_CELLS_PER_TILE=10 FRI_HIP_TILE_BYTES=24576" "FRI_HIP_BAND_ROWS=16 FRI_HIP_CELLS_PER_TILE=12 FRI_HIP_TILE_BYTES=24576" "FRI_HIP_BAND_ROWS=48 FRI_HIP_RANK_WEIGHTS=1,1,1,1" \
   "FRI_HIP_BAND_ROWS=48 FRI_HIP_RANK_WEIGHTS=1.45,1.15,0.85,0.55" "FRI_HIP_BAND_ROWS=48 FRI_HIP_RANK_WEIGHTS=1.2,1.05,0.95,0.8" "FRI_HIP_BAND_ROWS=48 FRI_HIP_CELLS_PER_TILE=6" "FRI_HIP_BAND_ROWS=48 FRI_HIP_CELLS_PER_TILE=7" > $OUT/sweep_tile.log 2>&1
SWEEP_C=3 python3 tools/k1_sweep_hbm.py --slots 12 --launches 200 --rounds 3 "" "FRI_HIP_BAND_ROWS=8" "FRI_HIP_BAND_ROWS=12" "FRI_HIP_BAND_ROWS=24" "FRI_HIP_BAND_ROWS=32" "FRI_HIP_BAND_ROWS=48" > $OUT/sweep_band_c3.log 2>&1
cat $OUT/stream_floor_32.log $OUT/stream_floor_4.log $OUT/sweep_band.log $OUT/sweep_tile.log $OUT/sweep_band_c3.log
