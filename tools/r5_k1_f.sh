#!/bin/bash
# Round 5: tail shares (short extra shares behind the resident round, dealt by the hardware as slots free up) - parity with them on, A/B, timeline; K5 with 3 / 4 chunks in flight.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FRI_HIP_TUNING=1 FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72 FRI_HIP_TAIL_WGS=512 FRI_HIP_TAIL_PERCENT=15 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "not config5" > $OUT/tests_tail.log 2>&1 || { tail -30 $OUT/tests_tail.log; exit 1; }
tail -2 $OUT/tests_tail.log
C72=FRI_HIP_STRIDED_SHARES=0,FRI_HIP_BAND_ROWS=72
AB_BATCH=24 python3 tools/k1_ab_hbm.py 3 - -:$C72 -:$C72,FRI_HIP_TAIL_WGS=256,FRI_HIP_TAIL_PERCENT=8 -:$C72,FRI_HIP_TAIL_WGS=512,FRI_HIP_TAIL_PERCENT=15 -:$C72,FRI_HIP_TAIL_WGS=1024,FRI_HIP_TAIL_PERCENT=25 \
   -:FRI_HIP_TAIL_WGS=512,FRI_HIP_TAIL_PERCENT=15 -:FRI_HIP_TAIL_WGS=256,FRI_HIP_TAIL_PERCENT=8 -:AB_TUNE=1 > $OUT/ab_c1.log 2>&1
cat $OUT/ab_c1.log
T=frave_amd/libfri_hip_tuning.so
FRI_HIP_LIBRARY=$T TRACE_SLOTS=40 FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72 FRI_HIP_TAIL_WGS=512 FRI_HIP_TAIL_PERCENT=15 python3 tools/trace_timeline.py k1 > $OUT/timeline_c72_tail.log 2>&1
head -34 $OUT/timeline_c72_tail.log
for L in "" build_variants/libfri_hip_k5c3.so build_variants/libfri_hip_k5c4.so "" build_variants/libfri_hip_k5c3.so build_variants/libfri_hip_k5c4.so; do
  echo "== K5 lib: ${L:-in-tree (2 chunks)}"; FRI_HIP_LIBRARY=$L K2_SLOTS=12 K2_TRUSTED=1 python3 tools/k2_time.py 2>&1 | grep -v amdgpu.ids
done > $OUT/k5_chunks.log 2>&1
cat $OUT/k5_chunks.log
