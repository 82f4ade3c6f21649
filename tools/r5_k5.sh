#!/bin/bash
# Round 5: what K5's gather costs - kernel-trace durations of symbol_gather_kernel for experiment builds (plain order loads, no gather, order loads without gather, one chunk).
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in intree k5m1 k5m2 k5m3 k5c1; do
  L=""; [ $v != intree ] && L=$R/build_variants/libfri_hip_$v.so
  FRI_HIP_LIBRARY=$L K2_SLOTS=12 K2_TRUSTED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $R/tools/k2_time.py > $OUT/trace_$v.log 2>&1
  echo "== $v"; python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "symbol_gather" in r["Name"] or "predict_histogram_kernel3<false, true>" in r["Name"]:
            print("  ", r["Name"][27:75], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
done 2>&1 | tee $OUT/k5_variants.txt
