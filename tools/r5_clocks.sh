#!/bin/bash
# Round 5: can an ordinary user see the clocks / power while K1 runs? rocm-smi next to a sustained K1 loop and next to an idle device.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
{ echo "== idle"; timeout 20 rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -v "^$" | head -40; } > $OUT/clocks.txt
K1_SIZE=4096 K1_SLOTS=24 K1_SPIN_UP=250000 python3 tools/k1_run.py 100 > $OUT/k1_run.log 2>&1 &
PID=$!
sleep 2.5
for i in 1 2 3; do { echo "== busy sample $i"; timeout 20 rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|mclk\|fclk\|socclk\|power\|err" | head -12; } >> $OUT/clocks.txt; sleep 0.5; done
wait $PID
cat $OUT/k1_run.log | tail -1 >> $OUT/clocks.txt
cat $OUT/clocks.txt
