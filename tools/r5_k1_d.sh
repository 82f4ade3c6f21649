#!/bin/bash
# Round 5: the tuner's second phase (XCD balance) - tuned against untuned, planes / RGB / 6000x4000 / 2048^2; then bench.py through the RCCL path on one rank.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
AB_BATCH=24 python3 tools/k1_ab_hbm.py 4 - -:AB_TUNE=1 > $OUT/ab_c1.log 2>&1
cat $OUT/ab_c1.log
AB_C=3 python3 tools/k1_ab_hbm.py 2 - -:AB_TUNE=1 > $OUT/ab_c3.log 2>&1
cat $OUT/ab_c3.log
AB_W=6000 AB_H=4000 python3 tools/k1_ab_hbm.py 2 - -:AB_TUNE=1 > $OUT/ab_6000.log 2>&1
cat $OUT/ab_6000.log
AB_W=2048 AB_H=2048 python3 tools/k1_ab_hbm.py 2 - -:AB_TUNE=1 > $OUT/ab_2048.log 2>&1
cat $OUT/ab_2048.log
FRI_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err || { echo FORCE_DIST failed; tail -5 $OUT/bench_force_dist.err; }
cat $OUT/bench_force_dist.json
