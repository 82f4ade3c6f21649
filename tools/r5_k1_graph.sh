#!/bin/bash
# Round 5: K1 replayed from a HIP graph against the native launch loop: events, then kernel durations from a trace of the same script.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd $R
python3 tools/k1_graph_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/graph.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/k1_graph_probe.py > $OUT/run.log 2>&1
python3 - <<PY | tee -a $OUT/graph.txt
import csv, glob
import numpy as np
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fwd_transform_quant_kernel<1, false, true, 4, true, true, false>" in r["Kernel_Name"]]
st = np.array([int(r["Start_Timestamp"]) for r in rows]); en = np.array([int(r["End_Timestamp"]) for r in rows])
o = np.argsort(st); st, en = st[o], en[o]
dur = (en - st) / 1000.0
gap = np.r_[0, (st[1:] - en[:-1]) / 1000.0]
cut = np.r_[0, np.flatnonzero(gap[1:] > 8.0) + 1, len(st)]
print("groups of back-to-back launches (gaps > 8 us split): size, mean kernel duration us, mean of the last half")
for a, b in zip(cut[:-1], cut[1:]):
    if b - a >= 90: print(f"  {b - a:5d}  {dur[a:b].mean():6.2f}  {dur[a + (b - a) // 2:b].mean():6.2f}")
PY
