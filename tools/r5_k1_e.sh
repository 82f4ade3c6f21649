#!/bin/bash
# Round 5: the tuner with its second phase (neighbours of the winner, XCD groups, rank weights) against the untuned plan; then the bench lines (default, K = 20).
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
AB_W=1920 AB_H=1080 python3 tools/k1_ab_hbm.py 2 - -:AB_TUNE=1 > $OUT/ab_1080.log 2>&1
cat $OUT/ab_1080.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_k20.json 2> $OUT/bench_k20.err
python3 bench.py --no-extras --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 - <<PY
import json
for n in ("bench_k20","bench_default"):
    d=json.load(open("$OUT/%s.json"%n))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_us"], d["timed_region"], d["roofline_batch"]["frac"], d["two_stream_launch_period"]["us"], d["config"]["forward_tiling"].get("winner"))
PY
