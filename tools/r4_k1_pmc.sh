#!/bin/bash
# Round 4: K1's memory-side counters in the HBM-bound regime (32 rotating slots) for band heights 8 / 32 / 72; separate --pmc passes, no trace domains.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export FRI_HIP_TUNING=1 K1_SLOTS=32 K1_SPIN_UP=0
for band in 8 32 72; do
  export FRI_HIP_BAND_ROWS=$band
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/b${band}_fetch -- python3 $R/tools/k1_run.py 64 > $OUT/b${band}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/b${band}_write -- python3 $R/tools/k1_run.py 64 > $OUT/b${band}_write.log 2>&1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum --output-format csv -d $OUT/b${band}_ea -- python3 $R/tools/k1_run.py 64 > $OUT/b${band}_ea.log 2>&1
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_sum --output-format csv -d $OUT/b${band}_eaw -- python3 $R/tools/k1_run.py 64 > $OUT/b${band}_eaw.log 2>&1
  echo "== band $band =="
  for p in fetch write ea eaw; do python3 $R/tools/pmc_summary.py $OUT/b${band}_$p fwd_transform; done
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
