#!/bin/bash
# Round 5: K1's kernel durations by position behind a drain (tools/k1_after_drain.py under rocprofv3 --kernel-trace).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/k1_after_drain.py > $OUT/run.log 2>&1
python3 - <<PY | tee $OUT/by_position.txt
import csv, glob
import numpy as np
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fwd_transform_quant_kernel<1, false, true, 4, true, true, false>" in r["Kernel_Name"]]
st = np.array([int(r["Start_Timestamp"]) for r in rows]); en = np.array([int(r["End_Timestamp"]) for r in rows])
o = np.argsort(st); st, en = st[o], en[o]
gap = st[1:] - en[:-1]
cut = np.r_[0, np.flatnonzero(gap > 20000) + 1, len(st)]
groups = [(a, b) for a, b in zip(cut[:-1], cut[1:])]
g60 = [(a, b) for a, b in groups if b - a == 60]
g400 = [(a, b) for a, b in groups if b - a == 400]
print(len(g60), "groups of 60 behind a synchronise,", len(g400), "groups of 400")
D = np.array([(en[a:b] - st[a:b]) / 1000.0 for a, b in g60])
P = np.array([np.r_[np.nan, (st[a + 1:b] - st[a:b - 1]) / 1000.0] for a, b in g60])
print("position: mean kernel duration us | mean start-to-start period us")
for i in list(range(0, 12)) + list(range(12, 60, 6)):
    print(f"  {i:3d}: {D[:, i].mean():6.2f} | {np.nanmean(P[:, i]):6.2f}")
print("groups of 400: mean duration of launches 0-19 / 20-59 / 200-399:", " / ".join(f"{np.mean([((en[a:b] - st[a:b]) / 1000.0)[lo:hi].mean() for a, b in g400]):.2f}" for lo, hi in ((0, 20), (20, 60), (200, 400))))
PY
