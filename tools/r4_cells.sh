#!/bin/bash
# K1: 9 (10) cells per tile against 8, several sizes
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { SWEEP_W=$1 SWEEP_H=$2 timeout -k 10 400 python3 tools/k1_sweep_hbm.py --slots $3 --launches $4 --rounds 3 "" "FRI_HIP_CELLS_PER_TILE=9" "FRI_HIP_CELLS_PER_TILE=10" >> $OUT/sweep.log 2>&1; }
run 4096 4096 32 300
run 1920 1080 64 300
run 2048 2048 48 300
run 6000 4000 16 200
run 8192 8192 8 100
cat $OUT/sweep.log
