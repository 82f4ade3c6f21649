#!/bin/bash
# Round 5: the row-run prefetch of K1 - parity with it switched on, then the A/B in the HBM regime (planes, RGB, 6000x4000, 2048^2, 1080p).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FRI_HIP_TUNING=1 FRI_HIP_K1_PREFETCH=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_graph.py -m gpu -x -q -k "not config5" > $OUT/tests_pf.log 2>&1 || { tail -30 $OUT/tests_pf.log; exit 1; }
tail -2 $OUT/tests_pf.log
AB_BATCH=24 python3 tools/k1_ab_hbm.py 3 - -:FRI_HIP_K1_PREFETCH=1 -:FRI_HIP_K1_PREFETCH=2 -:FRI_HIP_K1_PREFETCH=1,FRI_HIP_BAND_ROWS=8 -:FRI_HIP_K1_PREFETCH=1,FRI_HIP_BAND_ROWS=32 -:FRI_HIP_K1_PREFETCH=1,FRI_HIP_CELLS_PER_TILE=9 -:AB_TUNE=1 > $OUT/ab_c1.log 2>&1
cat $OUT/ab_c1.log
AB_C=3 python3 tools/k1_ab_hbm.py 2 - -:FRI_HIP_K1_PREFETCH=1 -:FRI_HIP_K1_PREFETCH=2 -:AB_TUNE=1 > $OUT/ab_c3.log 2>&1
cat $OUT/ab_c3.log
AB_W=6000 AB_H=4000 python3 tools/k1_ab_hbm.py 2 - -:FRI_HIP_K1_PREFETCH=1 -:AB_TUNE=1 > $OUT/ab_6000.log 2>&1
cat $OUT/ab_6000.log
AB_W=2048 AB_H=2048 python3 tools/k1_ab_hbm.py 2 - -:FRI_HIP_K1_PREFETCH=1 -:AB_TUNE=1 > $OUT/ab_2048.log 2>&1
cat $OUT/ab_2048.log
AB_W=1920 AB_H=1080 python3 tools/k1_ab_hbm.py 2 - -:FRI_HIP_K1_PREFETCH=1 -:AB_TUNE=1 > $OUT/ab_1080.log 2>&1
cat $OUT/ab_1080.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k config5 > $OUT/tests_config5.log 2>&1; tail -5 $OUT/tests_config5.log
