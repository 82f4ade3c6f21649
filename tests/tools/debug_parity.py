"""Print where K1 output differs from the oracle for a given shape. GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import frave_amd
from oracle import fri_oracle as O
from tests.common import gen_image

w, h, c = (int(v) for v in sys.argv[1:4])
ctx = frave_amd.Context(0)
P = frave_amd.Plan(ctx, w, h, c)
img = gen_image("noise", w, h, c, w + h)
W = O.Wavelet(img, h, w, c)
want = W.coefficients()
for rep in range(3):
    got = P.transform_quant(img)
    bad = got != want
    print(f"rep {rep}: mismatches {int(bad.sum())} of {bad.size}")
    if bad.any():
        ch, cell, idx = np.nonzero(bad)
        print("  channels", np.bincount(ch, minlength=c).tolist())
        cells = np.unique(cell)
        print("  cells", len(cells), cells[:20].tolist())
        print("  idx sample", idx[:20].tolist())
        k = cells[0]
        print("  cell", k, "center", P.centers()[k].tolist(), "got", got[ch[0], k, :8].tolist(), "want", want[ch[0], k, :8].tolist())
