"""Random-shape parity sweep of K1 / K2 / K3 / K4 against the CPU oracle (GPU box). Complements tests/test_gpu_parity.py, whose
shapes are fixed. usage: fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import frave_amd
from oracle import fri_oracle as O
from tests.common import gen_image, random_params

def run(n_cases, seed, ctx=None):
    """n_cases random (shape, content, channel count, quantiser, parameters) cases; returns the number of mismatching ones."""
    rng = np.random.default_rng(seed)
    ctx = ctx or frave_amd.Context(0)
    bad = 0
    t0 = time.time()
    for case in range(n_cases):
        kind = ["noise", "smooth", "const"][int(rng.integers(0, 3))]
        if rng.random() < 0.3:  # thin / tiny images
            w, h = int(rng.integers(1, 60)), int(rng.integers(1, 900))
            if rng.random() < 0.5:
                w, h = h, w
        else:
            w, h = int(rng.integers(46, 1400)), int(rng.integers(46, 900))
        c = 1 if rng.random() < 0.5 else 3
        img = gen_image(kind, w, h, c, int(rng.integers(0, 1 << 30)))
        try:
            P = frave_amd.Plan(ctx, w, h, c)
        except frave_amd.api.FriHipError as e:
            print(f"case {case}: {w}x{h}x{c}: plan error {e}")
            continue
        W = O.Wavelet(img, h, w, c)
        q = np.ones(32, np.int32)
        if rng.random() < 0.3:
            q[:10] = rng.integers(1, 9, 10)
        co = P.transform_quant(img, q)
        W.quantize(q)
        ok = np.array_equal(co, W.coefficients())
        msgs = [] if ok else ["K1"]
        ch = int(rng.integers(0, c))
        vp, wp = random_params(int(rng.integers(0, 1000)))
        b, p, hist, oob = P.predict_histogram(co, ch, vp, wp)
        wb, wpred, whist, woob = W.predict(ch, vp, wp)
        if not (np.array_equal(b, wb) and np.array_equal(p, wpred) and np.array_equal(hist, whist) and oob == woob):
            msgs.append("K2")
        back = P.inverse_transform(co)  # the oracle's inverse takes the coefficients as they are (identity dequantiser)
        if not np.array_equal(back, W.to_raster()):
            msgs.append("K3")
        # nine images in ONE launch: the merged batch shares of the plan
        d_px = torch.from_numpy(np.tile(img.reshape(1, -1), (9, 1))).cuda()
        d_co = torch.empty((9, P.coef_count), dtype=torch.int32, device="cuda")
        P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), qmatrix=q, n_images=9, pixel_stride=P.pixel_bytes, coef_stride=P.coef_count)
        torch.cuda.synchronize()
        if not bool((d_co.cpu().numpy().reshape(9, -1) == co.reshape(1, -1)).all()):
            msgs.append("K1-batch")
        if msgs:
            bad += 1
            print(f"case {case}: {w}x{h}x{c} {kind} q={q[:10].tolist()}: MISMATCH in {msgs}", flush=True)
        P.close()
        W.close()
    print(f"{n_cases} cases, {bad} mismatching, {time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
