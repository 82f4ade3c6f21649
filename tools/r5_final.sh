#!/bin/bash
# Round 5, what the driver does at round end: build check, smoke, the GPU suite, the bench (N = 1, the driver's K = 20 and the default).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err || { tail -20 $OUT/bench_k20.err; exit 1; }
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 - <<PY
import json
for n in ("bench_k20","bench_default"):
    lines=[l for l in open("$OUT/%s.json"%n).read().splitlines() if l.strip()]
    assert len(lines)==1, (n, len(lines))
    d=json.loads(lines[0])
    print(n, d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["kernel_us"], "traffic", d["roofline"]["traffic"], "batch", d["roofline_batch"]["frac"], d["roofline_batch"]["traffic"], "2s", d["two_stream_launch_period"]["us"], d["timed_region"], d["config"]["forward_tiling"].get("winner"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
