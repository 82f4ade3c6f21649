#!/bin/bash
# Round 5: whole tiles per share by dispatch rank (TilingParams::rank_tiles) - parity with a vector pinned, interleaved A/B of the four vectors at 4096^2 on contiguous / 72 and the
# default tiling, then the tuner with them among its phase-2 candidates.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
C72=FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72
FRI_HIP_TUNING=1 FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72 FRI_HIP_RANK_TILES=5,5,4,3 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not config5 and not short_shares" > $OUT/tests_rt.log 2>&1 || { tail -30 $OUT/tests_rt.log; exit 1; }
tail -2 $OUT/tests_rt.log
AB_BATCH=24 python3 tools/k1_ab_hbm.py 4 -:$C72 -:$C72,FRI_HIP_RANK_TILES=5,5,4,3 -:$C72,FRI_HIP_RANK_TILES=6,4,4,3 -:$C72,FRI_HIP_RANK_TILES=6,5,3,3 -:$C72,FRI_HIP_RANK_TILES=6,5,4,2 - -:FRI_HIP_RANK_TILES=5,5,4,3 -:FRI_HIP_RANK_TILES=6,4,4,3 -:AB_TUNE=1 > $OUT/ab_rt.log 2>&1
cat $OUT/ab_rt.log
