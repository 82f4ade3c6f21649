#!/bin/bash
export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
# usage: tools/prof_tlb.sh <outdir-under-gpurun_out>
# Address-translation counters of K1 (TCP -> UTCL1 -> UTCL2) at 4096^2 (85 MB touched per launch) and 16384^2 (1.36 GB): separate
# rocprofv3 --pmc passes, no trace domains mixed in, the program directly behind `--`. Summary per launch of fwd_transform* kernels.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() { size=$1; name=$2; shift 2; K1_SIZE=$size K1_SPIN_UP=0 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${name}_$size -- python3 $GRAFT_REPO_ROOT/tools/k1_run.py 12 > $OUT/${name}_$size.log 2>&1; }
for size in 4096 16384; do
  pass $size t1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
  pass $size t2 TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum
  pass $size t3 TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
  pass $size t4 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
  pass $size t5 GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum
done
cd $GRAFT_REPO_ROOT
for size in 4096 16384; do
  echo "== K1 ${size}x${size}x1, mean per launch =="
  for t in t1 t2 t3 t4 t5; do python3 tools/pmc_summary.py $OUT/${t}_$size fwd_transform; done
  grep -h "us/launch" $OUT/t1_$size.log
done > $OUT/summary.txt
cat $OUT/summary.txt
