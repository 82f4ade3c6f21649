#!/bin/bash
# Round 5: K1 with 3 / 2 resident workgroups per CU (768 / 512 shares) against 4, contiguous / 72 rows; interleaved A/B.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
C72=FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72
AB_BATCH=24 python3 tools/k1_ab_hbm.py 3 -:$C72 -:$C72,FRI_HIP_RANKS=3 -:$C72,FRI_HIP_RANKS=2 -:$C72,FRI_HIP_RANKS=3,FRI_HIP_RANK_WEIGHTS=1.2,1.0,0.8 -:$C72,FRI_HIP_TARGET_WGS=2048 2>&1 | tee $OUT/ab_ranks.log
