"""Which single coefficients the two value-pass kernels count differently: sets one node to 1 at a time (tiny image). GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FRI_HIP_TUNING"] = "1"
import numpy as np

import frave_amd

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 80)
ctx = frave_amd.Context(0)
os.environ["FRI_HIP_K4_VALUE3"] = "0"
P0 = frave_amd.Plan(ctx, w, h, 1)
os.environ["FRI_HIP_K4_VALUE3"] = "1"
P1 = frave_amd.Plan(ctx, w, h, 1)
F = P0.num_cells
valid = P0.valid_mask().reshape(F, 16)
print(f"{w}x{h}: {F} cells")
co = np.zeros((1, F, 512), np.int32)
for c in range(F):
    for hp in range(512):
        if not (valid[c, hp >> 5] >> (hp & 31)) & 1:
            co[0, c, hp] = np.iinfo(np.int32).min
n_bad = 0
for c in range(F):
    for hp in range(512):
        if co[0, c, hp] != 0:
            continue
        co[0, c, hp] = 1
        g0, g1 = P0.fit_value_sums(co, 0), P1.fit_value_sums(co, 0)
        co[0, c, hp] = 0
        if not np.array_equal(g0, g1):
            n_bad += 1
            if n_bad <= 25:
                d = [(g, k, int(g0[g][k][k]), int(g1[g][k][k])) for g in range(3) for k in range(7) if g0[g][k][k] != g1[g][k][k]]
                print(f"cell {c} heap {hp} (level {int(np.log2(hp + 1)) if hp else 0}): (group, column, old, new) {d}")
print(n_bad, "nodes counted differently")
