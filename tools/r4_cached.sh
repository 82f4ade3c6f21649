#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "nt stores     "; python3 tools/chain_hbm.py 2>&1 | grep CHAIN
  echo -n "cached stores "; FRI_HIP_K1_CACHED_STORES=1 python3 tools/chain_hbm.py 2>&1 | grep CHAIN
done
