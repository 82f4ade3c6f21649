#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/fri_oracle.c).

The reference ships no golden vectors for this path and cannot be built in this image (Rust), so these fixtures are
outputs of the oracle -- itself pinned to the known answers of SURVEY.md section 8c by tests/test_oracle_kat.py -- on the
seeded inputs of tests/common.py. They are data (inputs are regenerated from the seed; expected outputs are stored).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import fri_oracle as O  # noqa: E402
from tests.common import KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, gen_image, kat_image, random_params  # noqa: E402

CASES = {
    # name: (image factory, w, h, c, qmatrix or None, params)
    "kat_64x48_rgb": (lambda: kat_image(64, 48), 64, 48, 3, None, "kat"),
    "noise_100x37_rgb_q": (lambda: gen_image("noise", 100, 37, 3, 9), 100, 37, 3, [1, 2, 3, 5, 7, -3, 16, 255, 256, 1000], "random7"),
    "smooth_200x120_luma": (lambda: gen_image("smooth", 200, 120, 1, 4), 200, 120, 1, None, "kat"),
}


def build(name):
    make, w, h, c, q, params = CASES[name]
    img = make()
    W = O.Wavelet(img, h, w, c)
    raw = W.coefficients()
    qm = np.ones(32, np.int32)
    if q is not None:
        qm[: len(q)] = q
    assert W.quantize(qm) == 0
    out = {"width": w, "height": h, "channels": c, "qmatrix": qm, "centers": W.centers(), "coefs_raw": raw, "coefs": W.coefficients()}
    for ch in range(c):
        vp, wp = (KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS) if params == "kat" else random_params(7 + ch)
        b, p, hist, oob = W.predict(ch, vp, wp)
        out.update({f"value_params_{ch}": vp, f"width_params_{ch}": wp, f"bucket_{ch}": b, f"prediction_{ch}": p, f"hist_{ch}": hist, f"oob_{ch}": np.uint64(oob)})
    out["decoded"] = W.to_raster()  # quantization::decode is applied by the ABI, the oracle's inverse takes the coefficients as they are
    return out


# .frv files of the host emit path (oracle/emit_oracle.py: literal scan_level walk, finalize_context, Python rans64, serialize).
# Byte-stream parity with the reference's `rans` crate is unpinned: these pin OUR stream against regressions.
EMIT_CASES = {"emit_mixed_129x65_luma": (129, 65, 1, 7), "emit_mixed_96x257_rgb": (96, 257, 3, 7)}


def build_frv(name):
    from oracle import emit_oracle
    from tests.test_emit import _arrays

    w, h, c, seed = EMIT_CASES[name]
    W, coefs, bucket, pred, hist, vp, wp = _arrays(w, h, c, seed)
    return emit_oracle.encode_image(W, coefs, bucket, pred, hist, vp, wp)


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **build(name))
        print("wrote", name)
    for name in EMIT_CASES:
        with open(os.path.join(HERE, name + ".frv"), "wb") as f:
            f.write(build_frv(name))
        print("wrote", name)
