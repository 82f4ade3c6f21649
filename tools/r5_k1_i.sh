#!/bin/bash
# Round 5: contiguous shares walked from share-specific starts (rotated) - parity, A/B.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
C72=FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72
FRI_HIP_TUNING=1 FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72 FRI_HIP_ROTATE_SHARES=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not config5 and not short_shares" > $OUT/tests_rot.log 2>&1 || { tail -30 $OUT/tests_rot.log; exit 1; }
tail -2 $OUT/tests_rot.log
AB_BATCH=24 python3 tools/k1_ab_hbm.py 4 -:$C72 -:$C72,FRI_HIP_ROTATE_SHARES=1 -:FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=80 -:FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=80,FRI_HIP_ROTATE_SHARES=1 - > $OUT/ab_rot.log 2>&1
cat $OUT/ab_rot.log
