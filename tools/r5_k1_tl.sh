#!/bin/bash
# Round 5: raw per-workgroup stamps of K1 (tuning build, HBM regime, contiguous / 72 rows and the default tiling), three processes each, for an off-box look at who finishes late.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72 FRI_HIP_CELLS_PER_TILE=8 TRACE_SLOTS=40 TRACE_DUMP=$OUT/k1_c72_$r.npz FRI_HIP_LIBRARY=frave_amd/libfri_hip_tuning.so python3 tools/trace_timeline.py k1 > $OUT/k1_c72_$r.txt 2>&1
  TRACE_SLOTS=40 TRACE_DUMP=$OUT/k1_default_$r.npz FRI_HIP_LIBRARY=frave_amd/libfri_hip_tuning.so python3 tools/trace_timeline.py k1 > $OUT/k1_default_$r.txt 2>&1
done
tail -25 $OUT/k1_c72_1.txt
