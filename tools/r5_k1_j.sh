#!/bin/bash
# Round 5: contiguous shares cut at CELL granularity (TilingParams::cell_shares) - parity with the knob pinned, then interleaved A/B against whole-tile shares.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
C72=FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72
FRI_HIP_TUNING=1 FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72 FRI_HIP_CELL_SHARES=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not config5 and not short_shares" > $OUT/tests_cells.log 2>&1 || { tail -30 $OUT/tests_cells.log; exit 1; }
tail -2 $OUT/tests_cells.log
AB_BATCH=24 python3 tools/k1_ab_hbm.py 4 -:$C72 -:$C72,FRI_HIP_CELL_SHARES=1 -:FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=48 -:FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=48,FRI_HIP_CELL_SHARES=1 -:FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=32,FRI_HIP_CELL_SHARES=1 -:$C72,FRI_HIP_CELL_SHARES=1,FRI_HIP_RANK_WEIGHTS=1.4,1.15,0.85,0.6 -:$C72,FRI_HIP_CELL_SHARES=1,FRI_HIP_RANK_WEIGHTS=1.2,1.05,0.95,0.8 - > $OUT/ab_cells.log 2>&1
cat $OUT/ab_cells.log
