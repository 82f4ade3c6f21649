#!/bin/bash
# K1 RGB at 4096^2: tile parameters (cells per tile, band rows) against the default plan; first and last line = the default.
run() { echo "== $*"; env SWEEP_C=3 FRI_HIP_TUNING=1 "$@" timeout -k 10 120 python tools/k1_run.py 200; }
SWEEP_C=3 timeout -k 10 120 python tools/k1_run.py 200
run FRI_HIP_CELLS_PER_TILE=4
run FRI_HIP_CELLS_PER_TILE=4 FRI_HIP_BAND_ROWS=32
run FRI_HIP_CELLS_PER_TILE=3 FRI_HIP_BAND_ROWS=32
run FRI_HIP_CELLS_PER_TILE=6
run FRI_HIP_BAND_ROWS=24
run FRI_HIP_BAND_ROWS=8
SWEEP_C=3 timeout -k 10 120 python tools/k1_run.py 200
