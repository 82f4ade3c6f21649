#!/bin/bash
# Round 5, timing only: K1 with its coefficients leaving as int16 (build_variants/libfri_hip_k1i16.so = make DEFS=-DFRI_K1_I16_EXPERIMENT) against the product - what halfword
# coefficient planes inside the chains would gain on the forward kernel (single launch, 24-image batch, RGB).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
C72=FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72
AB_BATCH=24 python3 tools/k1_ab_hbm.py 3 -:$C72 build_variants/libfri_hip_k1i16.so:$C72 - build_variants/libfri_hip_k1i16.so 2>&1 | tee $OUT/ab_i16.log
AB_C=3 python3 tools/k1_ab_hbm.py 3 - build_variants/libfri_hip_k1i16.so 2>&1 | tee -a $OUT/ab_i16.log
