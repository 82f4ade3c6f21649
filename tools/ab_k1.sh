#!/bin/bash
export FRI_HIP_TUNING=1  # the library reads its tuning knobs from the environment only with this opt-in
# A/B of two builds of libfri_hip.so on one GPU box, interleaved so that clock / box effects hit both alike.
# usage: tools/ab_k1.sh <other-library.so> [rounds]     (the in-tree build is "new")
OTHER=$1; N=${2:-4}
for C in 1 3; do
  for i in $(seq $N); do
    echo -n "C=$C new: "; SWEEP_C=$C python tools/k1_run.py 200 | tail -1
    echo -n "C=$C old: "; SWEEP_C=$C FRI_HIP_LIBRARY=$OTHER python tools/k1_run.py 200 | tail -1
  done
done
