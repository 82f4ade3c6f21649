#!/bin/bash
# Round 5: K5 with the planes of a launch sharing the order loads - parity (emit / chain tests, the fuzz), then kernel-trace durations against one plane per workgroup.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_emit.py tests/test_encode_chain.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for v in shared per_plane shared2 per_plane2; do
  B=0; [ ${v:0:3} = per ] && B=1
  FRI_HIP_TUNING=1 FRI_HIP_K5_PER_PLANE=$B rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $R/tools/k5_planes_probe.py > $OUT/trace_$v.log 2>&1
  echo "== $v"; grep "image(s)" $OUT/trace_$v.log; python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace_$v/**/*kernel_trace.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "symbol_gather" in r["Kernel_Name"]]
    by={}
    for r in rows: by.setdefault((r["Kernel_Name"][27:60], r["Grid_Size_Y"]), []).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000.0)
    for k,v in by.items(): print("  ", k, len(v), "launches, mean %.1f us, min %.1f" % (sum(v)/len(v), min(v)))
PY
done 2>&1 | tee $OUT/k5_planes.txt
