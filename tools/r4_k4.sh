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
