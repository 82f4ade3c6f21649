#!/bin/bash
# Round 5: what K1's instructions cost (valu_rate2), the K1 parity tests on the new butterflies, and the first A/B in the HBM regime:
# round 4's library against the in-tree one, untuned and tuned, planes and RGB; the two-stream launch period.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
make -s -C tools/micro valu_rate2 && ./tools/micro/valu_rate2 > $OUT/valu_rate2.txt 2>&1
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
AB_STREAMS=2 python3 tools/k1_ab_hbm.py 3 build_variants/libfri_hip_r4.so - -:AB_TUNE=1 > $OUT/ab_c1.log 2>&1
cat $OUT/ab_c1.log
AB_C=3 AB_STREAMS=2 python3 tools/k1_ab_hbm.py 3 build_variants/libfri_hip_r4.so - -:AB_TUNE=1 > $OUT/ab_c3.log 2>&1
cat $OUT/ab_c3.log
AB_W=6000 AB_H=4000 python3 tools/k1_ab_hbm.py 2 build_variants/libfri_hip_r4.so - -:AB_TUNE=1 > $OUT/ab_6000.log 2>&1
cat $OUT/ab_6000.log
AB_W=2048 AB_H=2048 python3 tools/k1_ab_hbm.py 2 build_variants/libfri_hip_r4.so - -:AB_TUNE=1 > $OUT/ab_2048.log 2>&1
cat $OUT/ab_2048.log
cat $OUT/valu_rate2.txt
