#!/bin/bash
# Round 5: the array-route chain with the fit on dual planes (K1 writes the caller's int32 planes and the plan's compact int16 planes; the fit and the scan read the compact ones):
# the chain / fit / parity tests, then bench.py's extras.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python3 -m pytest tests/test_encode_chain.py tests/test_gpu_fit.py tests/test_gpu_parity.py tests/test_gpu_compact.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
python3 bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("K1", d["roofline"]["kernel_us"], d["roofline"]["frac"])
for k, v in d["extras"].items():
    if isinstance(v, dict) and "us" in v: print(f"  {k:48s} {v['us']:8.2f} us  {v.get('frac')}")
PY
