#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# parity first: K1 / K3 with the interleaved shares against the oracle (the env knob needs the tuning opt-in)
FRI_HIP_TUNING=1 FRI_HIP_STRIDED_SHARES=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "transform or round or inverse or config" > $OUT/parity.log 2>&1 || { tail -20 $OUT/parity.log; exit 1; }
tail -2 $OUT/parity.log
S="FRI_HIP_STRIDED_SHARES=1"
python3 tools/k1_sweep_hbm.py --slots 32 --launches 300 --rounds 3 "" "$S" "$S FRI_HIP_RANK_WEIGHTS=1,1,1,1" "$S FRI_HIP_RANK_WEIGHTS=1.15,1.05,0.95,0.85" "$S FRI_HIP_BAND_ROWS=8" "$S FRI_HIP_BAND_ROWS=16" "$S FRI_HIP_BAND_ROWS=48" "$S FRI_HIP_BAND_ROWS=72" \
  "$S FRI_HIP_BAND_ROWS=8 FRI_HIP_RANK_WEIGHTS=1,1,1,1" "$S FRI_HIP_BAND_ROWS=72 FRI_HIP_RANK_WEIGHTS=1,1,1,1" "FRI_HIP_BAND_ROWS=8" "FRI_HIP_BAND_ROWS=72" > $OUT/sweep.log 2>&1
python3 tools/k1_sweep_hbm.py --slots 8 --launches 300 --rounds 3 "" "$S" "$S FRI_HIP_BAND_ROWS=8" > $OUT/sweep_8slots.log 2>&1
cat $OUT/sweep.log $OUT/sweep_8slots.log
