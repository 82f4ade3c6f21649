#!/bin/bash
export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
# Runs on the GPU box: bench.py under rocprofv3 (kernel trace + stats), then PMC passes for HBM traffic of K1.
# usage: tools/profile_bench.sh <tag>   -> gpurun_out/<tag>/...
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --extras > $OUT/bench_extras.json 2> $OUT/bench_extras.err
# the default command (single-image K1 launches only): its kernel stats are the ones roofline.kernel_us must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_traced.json 2> $OUT/trace.log
# and once more with the extras (8-image launches of the same kernel, K2, K3)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_extras -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --extras > $OUT/bench_traced_extras.json 2> $OUT/trace_extras.log
K1_SPIN_UP=0 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/k1_run.py 40 > $OUT/pmc_fetch.log 2>&1
K1_SPIN_UP=0 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/k1_run.py 40 > $OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT/pmc_fetch fwd_transform > $OUT/pmc_summary.txt
python3 tools/pmc_summary.py $OUT/pmc_write fwd_transform >> $OUT/pmc_summary.txt
python3 - <<PY
import csv, glob
rows = []
for tag, sub in (("", "trace"), ("_extras", "trace_extras")):
  rows = []
  for f in glob.glob("$OUT/" + sub + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
  with open("$OUT/kernel_stats_short" + tag + ".csv", "w") as o:
    w = csv.writer(o)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
PY
cat $OUT/bench.json; echo; cat $OUT/pmc_summary.txt; head -5 $OUT/kernel_stats_short.csv
