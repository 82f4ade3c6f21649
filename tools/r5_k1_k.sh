#!/bin/bash
# Round 5: cell-granular contiguous shares with a tile's fixed cost of 6 / 12 / 24 cells' worth in the share cut, and rank weights; interleaved A/B.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
C72=FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72
python3 tools/k1_ab_hbm.py 3 -:$C72 -:$C72,FRI_HIP_CELL_SHARES=1,FRI_HIP_TILE_COST=6 -:$C72,FRI_HIP_CELL_SHARES=1,FRI_HIP_TILE_COST=12 -:$C72,FRI_HIP_CELL_SHARES=1,FRI_HIP_TILE_COST=24 -:$C72,FRI_HIP_CELL_SHARES=1,FRI_HIP_TILE_COST=12,FRI_HIP_RANK_WEIGHTS=1.4,1.15,0.85,0.6 -:$C72,FRI_HIP_CELL_SHARES=1,FRI_HIP_TILE_COST=12,FRI_HIP_RANK_WEIGHTS=1.6,1.2,0.8,0.4 > $OUT/ab_cells2.log 2>&1
cat $OUT/ab_cells2.log
