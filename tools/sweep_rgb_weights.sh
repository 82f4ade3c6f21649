#!/bin/bash
# K1 RGB at 4096^2: share sizes by dispatch rank (three resident workgroups per CU). A/B inside one call; first and last line = the default.
run() { SWEEP_C=3 FRI_HIP_TUNING=1 FRI_HIP_RANK_WEIGHTS=$1 timeout -k 10 120 python tools/k1_run.py 200 | sed "s/^/weights $1: /"; }
SWEEP_C=3 timeout -k 10 120 python tools/k1_run.py 200 | sed "s/^/default: /"
for w in 1.3,1.1,0.6,0 1.2,1.0,0.8,0 1.3,1.0,0.7,0 1.25,1.05,0.7,0 1.1,1.0,0.9,0 1,1,1,0 1.4,1.1,0.5,0 1.2,1.1,0.7,0 1.15,1.05,0.8,0; do run $w || exit 1; done
SWEEP_C=3 timeout -k 10 120 python tools/k1_run.py 200 | sed "s/^/default: /"
