#!/bin/bash
export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
# Round-2 evidence beside tools/profile_bench.sh (bench.py + K1 traffic): per-kernel durations of K1 at C = 1, C = 3 and 16384^2, of K2 / K3 /
# K4 at 4096^2, and the PMC counters (separate passes, no trace domains mixed in) of K2 and K4.
# usage: tools/profile_round2.sh <tag>   -> gpurun_out/<tag>/...
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# kernel traces (the program directly behind `--`; its knobs through the environment of this shell)
K1_SIZE=4096 SWEEP_C=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k1_c1 -- python3 $R/tools/k1_run.py 400 > $OUT/trace_k1_c1.log 2>&1
K1_SIZE=4096 SWEEP_C=3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k1_c3 -- python3 $R/tools/k1_run.py 200 > $OUT/trace_k1_c3.log 2>&1
K1_SIZE=16384 SWEEP_C=1 K1_SPIN_UP=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k1_16k -- python3 $R/tools/k1_run.py 40 > $OUT/trace_k1_16k.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k2k3k4 -- python3 $R/tools/k2_time.py > $OUT/trace_k2k3k4.log 2>&1
SWEEP_C=3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k2k3k4_c3 -- python3 $R/tools/k2_time.py > $OUT/trace_k2k3k4_c3.log 2>&1
# counters of K2 and K4 (tools/k2_time.py launches each 21 times)
pass() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/k2_time.py > $OUT/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
cd $R
python3 - <<PY
import csv, glob
out = open("$OUT/kernel_stats_round2.csv", "w")
w = csv.writer(out)
w.writerow(["run", "Name", "Calls", "AverageNs", "MinNs", "MaxNs", "StdDev"])
for run in ("trace_k1_c1", "trace_k1_c3", "trace_k1_16k", "trace_k2k3k4", "trace_k2k3k4_c3"):
    for f in glob.glob("$OUT/" + run + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fri::" in r["Name"]:
                w.writerow([run, r["Name"][:100], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
out.close()
print(open("$OUT/kernel_stats_round2.csv").read())
PY
for needle in predict_histogram_kernel3 fit_accumulate_kernel2\<0 fit_accumulate_kernel2\<1 inverse_transform; do
  echo "== $needle: mean per launch =="
  for p in sq1 sq2 tcc1 tcc2; do python3 tools/pmc_summary.py $OUT/$p "$needle"; done
done > $OUT/pmc_k2_k4_summary.txt
cat $OUT/pmc_k2_k4_summary.txt
grep -h "us/launch\|data=" $OUT/trace_*.log
