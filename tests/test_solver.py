"""The 6 x 6 solves behind the fit (fri_hip_solve6, fri_hip_fit_value_params / _width_params; the counterpart of lstsq in
ContextModeler::optimize_value_prediction / optimize_width_prediction, context_modeling.rs:144-202), host only: the Cholesky route for
safely positive definite systems and the minimum-norm eigen-decomposition route for the rest, against numpy's lstsq on the explicit
design matrix."""
import ctypes

import numpy as np
import pytest

import frave_amd as fa


def solve6(m, y):
    m = np.ascontiguousarray(m, np.float64)
    y = np.ascontiguousarray(y, np.float64)
    x = np.zeros(6)
    fa.load_library().fri_hip_solve6(m.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), x.ctypes.data_as(ctypes.c_void_p))
    return x


def design(rng, n, kind):
    a = rng.integers(-200, 200, (n, 6)).astype(np.float64)
    if kind == "dependent":  # rank 5: an exact linear dependence
        a[:, 3] = 2 * a[:, 2]
    elif kind == "zero column":  # a feature that is zero everywhere (flat regions)
        a[:, 1] = 0
    elif kind == "nearly dependent":  # positive definite on paper, below the Cholesky route's pivot threshold in practice
        a[:, 4] = a[:, 0] + 1e-5 * rng.standard_normal(n)
    elif kind == "all zero":
        a[:] = 0
    return a


@pytest.mark.parametrize("kind", ["full rank", "dependent", "zero column", "nearly dependent", "all zero"])
def test_solve6_is_the_least_squares_solution(kind):
    rng = np.random.default_rng(abs(hash(kind)) % 1000)
    for trial in range(40):
        n = int(rng.integers(8, 3000))
        a = design(rng, n, kind)
        b = rng.integers(-200, 200, n).astype(np.float64)
        x = solve6(a.T @ a, a.T @ b)
        ref = np.linalg.lstsq(a, b, rcond=1e-7)[0]
        r, r_ref = np.linalg.norm(a @ x - b), np.linalg.norm(a @ ref - b)
        assert r <= r_ref * (1 + 1e-9) + 1e-9  # no worse a fit than the SVD's
        if kind != "nearly dependent":  # there the two may split the nearly dependent pair differently; the fit is what counts
            assert np.allclose(x, ref, rtol=1e-6, atol=1e-7)
        if kind in ("dependent", "zero column", "all zero"):  # the minimum-norm solution: no component along the null space
            assert np.linalg.norm(x) <= np.linalg.norm(ref) * (1 + 1e-6) + 1e-9


def test_fit_params_use_the_triangles_and_the_all_zero_rows():
    rng = np.random.default_rng(5)
    n = 4000
    a = rng.integers(-255, 256, (3, n, 6)).astype(np.int64)
    v = rng.integers(-255, 256, (3, n)).astype(np.int64)
    u = np.concatenate([a, v[:, :, None]], axis=2)
    gram = np.einsum("gni,gnj->gij", u, u)
    vp = fa.fit_value_params(np.stack([gram[g][np.triu_indices(7)] for g in range(3)]))
    for g in range(3):
        ref = np.linalg.lstsq(a[g].astype(np.float64), v[g].astype(np.float64), rcond=None)[0]
        assert np.allclose(vp[g], ref.astype(np.float32), rtol=1e-5, atol=1e-7)
    # width fit: features w with a constant 1, residuals r >= 0; `rows` counts the reference's all-zero rows too (feature 1, residual 0)
    w = np.abs(rng.integers(-300, 300, (3, n, 6))).astype(np.int64)
    w[:, :, 0] = 1
    r = np.abs(rng.standard_normal((3, n))) * 10
    wtw = np.einsum("gni,gnj->gij", w, w)
    wtr = np.einsum("gni,gn->gi", w.astype(np.float64), r)
    extra = np.array([100, 0, 2500], np.uint64)
    wp = fa.fit_width_params(np.stack([wtw[g][np.triu_indices(6)] for g in range(3)]), wtr, extra + n)
    for g in range(3):
        w_all = np.concatenate([w[g].astype(np.float64), np.tile([1.0, 0, 0, 0, 0, 0], (int(extra[g]), 1))])
        r_all = np.concatenate([r[g], np.zeros(int(extra[g]))])
        ref = np.linalg.lstsq(w_all, r_all, rcond=None)[0]
        assert np.allclose(wp[g], ref.astype(np.float32), rtol=1e-5, atol=1e-6)
