#!/bin/bash
# Round 5: the counters VERDICT r4 asked for on K2's first tile - instruction cache requests / misses - next to the instruction mix, for K2, K4, K3 (tools/k2_time.py
# launches each over 12 rotating planes). Separate --pmc passes, no trace domains mixed in.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
grep -i -o 'SQC_[A-Z_0-9]*' $OUT/counters_avail.txt | sort -u > $OUT/sqc_counters.txt
grep -i -o 'SQ_[A-Z_0-9]*IFETCH[A-Z_0-9]*\|SQ_[A-Z_0-9]*INST_LEVEL[A-Z_0-9]*\|SQ_WAIT_INST[A-Z_0-9_]*\|SQ_INST_CYCLES[A-Z_0-9_]*' $OUT/counters_avail.txt | sort -u >> $OUT/sqc_counters.txt
cat $OUT/sqc_counters.txt
pass() { dir=$1; shift; K2_TRUSTED=1 K2_SLOTS=12 K5=0 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$dir -- python3 $R/tools/k2_time.py > $OUT/$dir.log 2>&1; echo "$dir" >> $OUT/progress.txt; }
pass ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass ic2 SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass ic3 SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_REQ SQC_DCACHE_REQ SQC_DCACHE_MISSES
cd $R
for needle in "predict_histogram_kernel3<false, false>" "fit_accumulate_kernel2<0" "fit_accumulate_kernel2<1" inverse_transform; do
  echo "== $needle: mean per launch =="
  for p in ic1 ic2 ic3; do python3 tools/pmc_summary.py $OUT/$p "$needle"; done
done > $OUT/pmc_icache_summary.txt
cat $OUT/pmc_icache_summary.txt; tail -3 $OUT/ic1.log
