#!/bin/bash
export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
# usage: tools/prof_pmc.sh <outdir-under-gpurun_out> <python script + args...>
# Collects PMC counters in separate passes (no trace domains mixed in), CSV output.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
SCRIPT="$*"
cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift; K1_SPIN_UP=0 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/$SCRIPT > $OUT/$name.log 2>&1; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
echo done
