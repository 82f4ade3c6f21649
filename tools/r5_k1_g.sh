#!/bin/bash
# Round 5: staggered starts by dispatch rank (experiment), then the K2 / K4 / K3 instruction-cache counters.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
C72=FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72
AB_BATCH=24 python3 tools/k1_ab_hbm.py 3 -:$C72 -:$C72,FRI_HIP_K1_STAGGER=8 -:$C72,FRI_HIP_K1_STAGGER=16 -:$C72,FRI_HIP_K1_STAGGER=32 -:$C72,FRI_HIP_K1_STAGGER=16,FRI_HIP_RANK_WEIGHTS=1.4,1.15,0.85,0.6 \
   - -:FRI_HIP_K1_STAGGER=8 -:FRI_HIP_K1_STAGGER=16 -:FRI_HIP_K1_STAGGER=32 > $OUT/ab_stagger.log 2>&1
cat $OUT/ab_stagger.log
