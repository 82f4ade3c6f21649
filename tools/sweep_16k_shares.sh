#!/bin/bash
# K1 on large single images: shares per launch (FRI_HIP_TARGET_WGS, equal rank weights) against the default of one share per resident workgroup.
# usage (GPU box): bash tools/sweep_16k_shares.sh [size ...]
for size in ${@:-8192 16384}; do
  echo "== ${size}x${size}: default plan"; K1_SIZE=$size K1_SPIN_UP=100 timeout -k 10 120 python tools/k1_run.py 40 || exit 1
  for wgs in 2048 3072 4096 6144 8192 12288; do
    echo "target_wgs=$wgs"; K1_SIZE=$size K1_SPIN_UP=100 FRI_HIP_TUNING=1 FRI_HIP_TARGET_WGS=$wgs FRI_HIP_RANK_WEIGHTS=1,1,1,1 timeout -k 10 120 python tools/k1_run.py 40 || exit 1
  done
done
