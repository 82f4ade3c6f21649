#!/bin/bash
# Round 5: what bounds K5's gather - symbol_gather_kernel durations for synthetic orders (tools/k5_order_probe.py).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in ${K5_ORDERS:-real sorted lines16_reuse lines16_fresh}; do
  K5_ORDER=$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $R/tools/k5_order_probe.py > $OUT/trace_$v.log 2>&1
  tail -1 $OUT/trace_$v.log; python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "symbol_gather" in r["Name"]:
            print("   symbol_gather_kernel", r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done 2>&1 | tee $OUT/k5_orders.txt
