#!/bin/bash
# Round 5: K1's HBM traffic (separate --pmc passes, no trace domains) on every tiling the tuner can end up with at 4096^2 (its phase-1 candidates and their phase-2 neighbours) -
# bench.py reports the figure of the tiling its plan measured. usage: r5_k1_traffic.sh <tag> [more]   (more: the neighbours too); tools/k1_traffic_table.py merges the result
# into profiles/r05_k1_traffic_by_tiling.json
set -u
export FRI_HIP_TUNING=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
LIST="interleaved_band16_cells8 interleaved_band16_cells9 contiguous_band72_cells8 contiguous_band80_cells8 interleaved_band8_cells8"
[ "${2:-}" = more ] && LIST="contiguous_band64_cells8 contiguous_band88_cells8 contiguous_band72_cells9 contiguous_band72_cells7 contiguous_band80_cells9 contiguous_band64_cells9 contiguous_band48_cells8 contiguous_band96_cells8 contiguous_band8_cells8 interleaved_band16_cells7 interleaved_band24_cells8 interleaved_band32_cells8 interleaved_band8_cells9 interleaved_band16_cells10 interleaved_band24_cells9 contiguous_band56_cells8"
for name in $LIST; do
  s=1; [ ${name%%_*} = contiguous ] && s=0
  b=${name#*_band}; b=${b%%_*}; c=${name##*_cells}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    FRI_HIP_STRIDED_SHARES=$s FRI_HIP_BAND_ROWS=$b FRI_HIP_CELLS_PER_TILE=$c K1_SLOTS=32 K1_SPIN_UP=0 rocprofv3 --pmc $ctr --output-format csv -d $OUT/${name}_$ctr -- python3 $R/tools/k1_run.py 64 > $OUT/${name}_$ctr.log 2>&1
  done
  echo $name >> $OUT/progress.txt
done
cd $R
for n in $LIST; do
  echo "== $n"; for c in FETCH_SIZE WRITE_SIZE; do python3 tools/pmc_summary.py $OUT/${n}_$c fwd_transform; done
done > $OUT/traffic_by_tiling.txt
cat $OUT/traffic_by_tiling.txt
