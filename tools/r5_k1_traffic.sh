#!/bin/bash
# Round 5: K1's HBM traffic (separate --pmc passes, no trace domains) on each tiling the tuner ends up choosing at 4096^2 - bench.py reports the figure of the tiling its plan measured.
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; for c in FETCH_SIZE WRITE_SIZE; do env "$@" K1_SLOTS=32 K1_SPIN_UP=0 rocprofv3 --pmc $c --output-format csv -d $OUT/${name}_$c -- python3 $R/tools/k1_run.py 64 > $OUT/${name}_$c.log 2>&1; done; echo $name >> $OUT/progress.txt; }
run interleaved_band16_cells8 FRI_HIP_STRIDED_SHARES=1 FRI_HIP_BAND_ROWS=16 FRI_HIP_CELLS_PER_TILE=8
run interleaved_band16_cells9 FRI_HIP_STRIDED_SHARES=1 FRI_HIP_BAND_ROWS=16 FRI_HIP_CELLS_PER_TILE=9
run contiguous_band72_cells8 FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=72 FRI_HIP_CELLS_PER_TILE=8
run contiguous_band80_cells8 FRI_HIP_STRIDED_SHARES=0 FRI_HIP_BAND_ROWS=80 FRI_HIP_CELLS_PER_TILE=8
run interleaved_band8_cells8 FRI_HIP_STRIDED_SHARES=1 FRI_HIP_BAND_ROWS=8 FRI_HIP_CELLS_PER_TILE=8
cd $R
for n in interleaved_band16_cells8 interleaved_band16_cells9 contiguous_band72_cells8 contiguous_band80_cells8 interleaved_band8_cells8; do
  echo "== $n"; for c in FETCH_SIZE WRITE_SIZE; do python3 tools/pmc_summary.py $OUT/${n}_$c fwd_transform; done
done > $OUT/traffic_by_tiling.txt
cat $OUT/traffic_by_tiling.txt
