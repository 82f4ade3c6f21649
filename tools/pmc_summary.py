"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch for kernels matching a substring."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)

import csv
import glob
import sys
from collections import defaultdict

root, needle = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if needle in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f}")
