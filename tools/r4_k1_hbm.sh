#!/bin/bash
# Round 4: K1's single-image launch when every byte comes from HBM (rotating over >= 24 slots): knob sweeps and the per-workgroup timeline in both regimes.
# usage: tools/r4_k1_hbm.sh <tag>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 tools/k1_sweep_hbm.py --slots 32 --launches 300 --rounds 3 "" "FRI_HIP_RANK_WEIGHTS=1,1,1,1" "FRI_HIP_RANK_WEIGHTS=1.15,1.05,0.95,0.85" "FRI_HIP_RANK_WEIGHTS=1.45,1.15,0.85,0.55" \
   "FRI_HIP_TARGET_WGS=2048 FRI_HIP_RANK_WEIGHTS=1,1,1,1" "FRI_HIP_TARGET_WGS=4096 FRI_HIP_RANK_WEIGHTS=1,1,1,1" "FRI_HIP_TARGET_WGS=1536 FRI_HIP_RANK_WEIGHTS=1,1,1,1" \
   "FRI_HIP_CELLS_PER_TILE=6" "FRI_HIP_CELLS_PER_TILE=4" "FRI_HIP_BAND_ROWS=16" "FRI_HIP_BAND_ROWS=48" "FRI_HIP_RANKS=3" > $OUT/sweep_c1.log 2>&1
echo sweep1 > $OUT/progress.txt
FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/frave_amd/libfri_hip_tuning.so TRACE_SLOTS=4 python3 tools/trace_timeline.py k1 1 > $OUT/trace_4slots.log 2>&1
FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/frave_amd/libfri_hip_tuning.so TRACE_SLOTS=40 python3 tools/trace_timeline.py k1 1 > $OUT/trace_40slots.log 2>&1
echo traces >> $OUT/progress.txt
# ablations in the HBM regime (tuning build): no staging / no stores
for ab in 0 1 4 5; do
  echo "ablate=$ab"; FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/frave_amd/libfri_hip_tuning.so FRI_HIP_K1_ABLATE=$ab K1_SLOTS=40 python3 tools/k1_run.py 300 2>&1 | grep K1
done > $OUT/ablate_40slots.log 2>&1
cat $OUT/sweep_c1.log $OUT/ablate_40slots.log
