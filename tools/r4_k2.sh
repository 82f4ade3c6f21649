#!/bin/bash
# K2 after a change: its parity tests (and every test that runs a chain), timings over 24 rotating planes, the chains, the timeline
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_encode_chain.py tests/test_gpu_fuzz.py tests/test_emit.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
K2_SLOTS=24 K5=0 timeout -k 10 300 python3 tools/k2_time.py 2>&1 | grep slots
K2_TRUSTED=1 K2_SLOTS=24 K5=0 timeout -k 10 300 python3 tools/k2_time.py 2>&1 | grep slots
timeout -k 10 300 python3 tools/chain_hbm.py 2>&1 | grep CHAIN
timeout -k 10 300 python3 tools/chain_hbm.py 2>&1 | grep CHAIN
export FRI_HIP_LIBRARY=$GRAFT_REPO_ROOT/frave_amd/libfri_hip_tuning.so FRI_HIP_TUNING=1
timeout -k 10 200 python3 tools/trace_k2.py 2>&1 | tail -16
