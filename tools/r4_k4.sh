#!/bin/bash
# K4 after a change: the fit's tests, then K2/K3/K4 timings over 24 rotating planes (HBM regime)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fit.py tests/test_encode_chain.py -m gpu -x -q > $OUT/fit_tests.log 2>&1 || { tail -40 $OUT/fit_tests.log; exit 1; }
tail -2 $OUT/fit_tests.log
K2_SLOTS=24 timeout -k 10 300 python3 tools/k2_time.py > $OUT/k2_time.log 2>&1
cat $OUT/k2_time.log
