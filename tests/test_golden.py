"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them (guards the oracle against silent edits). GPU: the HIP path reproduces them
through the C ABI without running the oracle."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _input_image(name):
    from tests.golden.make_golden import CASES

    return CASES[name][0]()


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(oracle, path):
    from tests.golden.make_golden import build

    g = np.load(path)
    now = build(os.path.basename(path)[:-4])
    assert sorted(now) == sorted(g.files)
    for k in g.files:
        assert np.array_equal(np.asarray(now[k]), g[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_reproduces_golden(path):
    import frave_amd

    g = np.load(path)
    name = os.path.basename(path)[:-4]
    w, h, c = int(g["width"]), int(g["height"]), int(g["channels"])
    ctx = frave_amd.Context(0)
    P = frave_amd.Plan(ctx, w, h, c)
    img = _input_image(name)
    assert np.array_equal(P.centers(), g["centers"])
    assert np.array_equal(P.transform_quant(img), g["coefs_raw"])
    co = P.transform_quant(img, g["qmatrix"])
    assert np.array_equal(co, g["coefs"])
    for ch in range(c):
        b, p, hist, oob = P.predict_histogram(co, ch, g[f"value_params_{ch}"], g[f"width_params_{ch}"])
        assert np.array_equal(b, g[f"bucket_{ch}"]) and np.array_equal(p, g[f"prediction_{ch}"])
        assert np.array_equal(hist, g[f"hist_{ch}"]) and oob == int(g[f"oob_{ch}"])
    # inverse of the stored (quantised) coefficients with an all-ones matrix = the oracle's extract_values on them
    assert np.array_equal(P.inverse_transform(co), g["decoded"])
