#!/bin/bash
# Round 5, first call: the RCCL path of bench.py on one rank (nccl init, the warm-up barriers, the MAX all-reduce), the default line, the driver's K = 20 line.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
FRI_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err || { echo FORCE_DIST failed; tail -20 $OUT/bench_force_dist.err; exit 1; }
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_k20.json 2> $OUT/bench_k20.err || exit 1
python3 bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err || exit 1
python3 - <<PY
import json
for n in ("bench_force_dist","bench_k20","bench_default"):
    d=json.load(open("$OUT/%s.json"%n))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_us"], d["timed_region"])
PY
