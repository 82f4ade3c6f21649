#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 tools/k1_sweep_hbm.py --slots 32 --launches 300 --rounds 4 "" "FRI_HIP_K1_CACHED_STORES=1"
python3 tools/k1_sweep_hbm.py --slots 8 --launches 300 --rounds 4 "" "FRI_HIP_K1_CACHED_STORES=1"
SWEEP_C=3 python3 tools/k1_sweep_hbm.py --slots 12 --launches 200 --rounds 3 "" "FRI_HIP_K1_CACHED_STORES=1"
SWEEP_W=1920 SWEEP_H=1080 python3 tools/k1_sweep_hbm.py --slots 256 --launches 512 --rounds 3 "" "FRI_HIP_K1_CACHED_STORES=1"
SWEEP_W=8192 SWEEP_H=8192 python3 tools/k1_sweep_hbm.py --slots 8 --launches 100 --rounds 3 "" "FRI_HIP_K1_CACHED_STORES=1"
