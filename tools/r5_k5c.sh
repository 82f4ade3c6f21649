#!/bin/bash
# Round 5: K5 with the order in compact form (2-byte codes + per-chunk cell tables) against the flat 4-byte order (FRI_HIP_K5_FLAT_ORDER=1): parity, then kernel-trace durations.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_emit.py tests/test_encode_chain.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for v in compact flat compact2 flat2; do
  B=0; [ ${v:0:4} = flat ] && B=1
  FRI_HIP_K5_FLAT_ORDER=$B K2_SLOTS=12 K2_TRUSTED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $R/tools/k2_time.py > $OUT/trace_$v.log 2>&1
  echo "== $v"; grep "chain forward" $OUT/trace_$v.log; python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "symbol_" in r["Name"]:
            print("  ", r["Name"][27:75], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done 2>&1 | tee $OUT/k5_compact.txt
