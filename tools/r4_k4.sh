#!/bin/bash
# K4 after a change: the fit's tests, then K2/K3/K4 timings over 24 rotating planes (HBM regime), the chain, the timeline
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fit.py tests/test_encode_chain.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/fit_tests.log 2>&1 || { tail -40 $OUT/fit_tests.log; exit 1; }
tail -2 $OUT/fit_tests.log
K2_SLOTS=24 timeout -k 10 300 python3 tools/k2_time.py > $OUT/k2_time.log 2>&1
cat $OUT/k2_time.log
timeout -k 10 300 python3 tools/chain_hbm.py > $OUT/chain.log 2>&1; tail -6 $OUT/chain.log
export FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/frave_amd/libfri_hip_tuning.so FRI_HIP_TUNING=1
K4_MODE=0 timeout -k 10 200 python3 tools/trace_k4.py > $OUT/trace0.log 2>&1; tail -4 $OUT/trace0.log
K4_MODE=1 timeout -k 10 200 python3 tools/trace_k4.py > $OUT/trace1.log 2>&1; tail -4 $OUT/trace1.log
# LDS bank conflicts of the two fit kernels (its own pass: --pmc without trace domains)
cd /tmp && export TMPDIR=/tmp
unset FRI_HIP_LIBRARY
K2_TRUSTED=1 K2_SLOTS=12 K5=0 timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/tools/k2_time.py > $OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT && for n in "fit_accumulate_kernel2<0" "fit_accumulate_kernel2<1"; do echo "== $n"; python3 tools/pmc_summary.py $OUT/pmc "$n"; done
