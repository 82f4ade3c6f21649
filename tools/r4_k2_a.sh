#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_encode_chain.py tests/test_gpu_fuzz.py tests/test_emit.py tests/test_config4.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
K2_TRUSTED=1 python3 tools/k2_time.py > $OUT/k2_time.log 2>&1
python3 tools/chain_time.py > $OUT/chain.log 2>&1
cat $OUT/k2_time.log $OUT/chain.log
tools/r4_k1_w72.sh $1_w72
