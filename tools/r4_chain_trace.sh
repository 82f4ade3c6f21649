#!/bin/bash
# per-kernel times inside the encode chains (rocprofv3 kernel trace over tools/chain_hbm.py)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_chain -- python3 $GRAFT_REPO_ROOT/tools/chain_hbm.py > $OUT/trace_chain.log 2>&1
grep CHAIN $OUT/trace_chain.log
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace_chain/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fri::" in r["Name"]:
            print(f"{r['Name'].split('fri::')[-1][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs'])/1000:8.2f}")
PY
