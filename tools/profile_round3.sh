#!/bin/bash
export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
# Round-3 evidence: bench.py (plain, and under rocprofv3 --kernel-trace --stats), per-kernel durations of K1 (C = 1, RGB, 16384^2), K2 / K3 / K4 / K5
# at 4096^2, and the PMC counters (separate passes, no trace domains mixed in) of K1 (traffic), K2, K4, K3 and K1 RGB.
# usage: tools/profile_round3.sh <tag>   -> gpurun_out/<tag>/...
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_traced.json 2> $OUT/trace_bench.log
echo "bench done" > $OUT/progress.txt
K1_SIZE=4096 SWEEP_C=3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k1_c3 -- python3 $R/tools/k1_run.py 200 > $OUT/trace_k1_c3.log 2>&1
K1_SIZE=16384 SWEEP_C=1 K1_SPIN_UP=200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k1_16k -- python3 $R/tools/k1_run.py 40 > $OUT/trace_k1_16k.log 2>&1
K2_TRUSTED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k2k3k4 -- python3 $R/tools/k2_time.py > $OUT/trace_k2k3k4.log 2>&1
K2_TRUSTED=1 SWEEP_C=3 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k2k3k4_c3 -- python3 $R/tools/k2_time.py > $OUT/trace_k2k3k4_c3.log 2>&1
K2_TRUSTED=1 K2_SIZE=16384 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k2k3k4_16k -- python3 $R/tools/k2_time.py > $OUT/trace_k2k3k4_16k.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_chain -- python3 $R/tools/chain_time.py > $OUT/trace_chain.log 2>&1
echo "traces done" >> $OUT/progress.txt
pass() { dir=$1; shift; script=$1; shift; K2_TRUSTED=1 K1_SPIN_UP=0 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$dir -- python3 $R/tools/$script > $OUT/$dir.log 2>&1; echo "$dir" >> $OUT/progress.txt; }
# K2 / K3 / K4 / K5 (tools/k2_time.py launches each 21 times)
pass sq1 k2_time.py SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass sq2 k2_time.py SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass tcc1 k2_time.py FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc2 k2_time.py WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# K1 plane (traffic for bench.py's roofline.traffic) and K1 RGB
pass k1_fetch "k1_run.py 40" FETCH_SIZE GRBM_GUI_ACTIVE
pass k1_write "k1_run.py 40" WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
export SWEEP_C=3
pass rgb_sq2 "k1_run.py 40" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass rgb_fetch "k1_run.py 40" FETCH_SIZE GRBM_GUI_ACTIVE
pass rgb_write "k1_run.py 40" WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
unset SWEEP_C
cd $R
python3 - <<PY
import csv, glob
out = open("$OUT/kernel_stats_round3.csv", "w")
w = csv.writer(out)
w.writerow(["run", "Name", "Calls", "AverageNs", "MinNs", "MaxNs", "StdDev"])
for run in ("trace_bench", "trace_k1_c3", "trace_k1_16k", "trace_k2k3k4", "trace_k2k3k4_c3", "trace_k2k3k4_16k", "trace_chain"):
    for f in glob.glob("$OUT/" + run + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fri::" in r["Name"]:
                w.writerow([run, r["Name"][:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
out.close()
print(open("$OUT/kernel_stats_round3.csv").read())
PY
for needle in "predict_histogram_kernel3<true, false>" "predict_histogram_kernel3<false, false>" "predict_histogram_kernel3<false, true>" "fit_accumulate_kernel2<0" "fit_accumulate_kernel2<1" inverse_transform symbol_gather symbol_stream; do
  echo "== $needle: mean per launch =="
  for p in sq1 sq2 tcc1 tcc2; do python3 tools/pmc_summary.py $OUT/$p "$needle"; done
done > $OUT/pmc_k2_k4_k3_k5_summary.txt
{ echo "== K2 standalone, predict_histogram_kernel3<true, false> (checked staging, caller vouches: no exact kernel behind it): mean per launch, 4096x4096x1 ==";
  for p in sq1 sq2 tcc1 tcc2; do python3 tools/pmc_summary.py $OUT/$p "predict_histogram_kernel3<true, false>"; done; } > $OUT/pmc_k2_summary.txt
{ echo "== K1 plane (fwd_transform_quant_kernel<1,...>) =="; python3 tools/pmc_summary.py $OUT/k1_fetch fwd_transform; python3 tools/pmc_summary.py $OUT/k1_write fwd_transform;
  echo "== K1 RGB (fwd_transform_quant_kernel<3,...>) =="; for p in rgb_sq2 rgb_fetch rgb_write; do python3 tools/pmc_summary.py $OUT/$p fwd_transform; done; } > $OUT/pmc_k1_summary.txt
cat $OUT/pmc_k2_k4_k3_k5_summary.txt $OUT/pmc_k1_summary.txt
cat $OUT/bench.json
grep -h "us/launch\|data=\|chain" $OUT/trace_*.log
