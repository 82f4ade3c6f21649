"""Fit accumulators against sums computed on the CPU from the oracle's restatement of
ContextModeler::get_neighbour_values (context_modeling.rs:25-77). Integer sums are bit-exact; the f64 residual sums are
compared with a tolerance (summation order is not fixed on the device)."""
import numpy as np
import pytest

from tests.common import KAT_VALUE_PARAMS, gen_image, random_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import frave_amd

    c = frave_amd.Context(0)
    yield c
    c.close()


def _groups():
    p = np.arange(512)
    level = np.floor(np.log2(np.maximum(p, 1))).astype(int)
    g = np.where(level == 8, 0, np.where(level == 7, 1, 2))
    return g, p >= 2


def _cpu_sums(oracle, W, ch, value_params):
    co = W.coefficients()[ch].astype(np.int64)  # [F][512]
    some = co != oracle.NONE
    nv = W.neighbour_values(ch).astype(np.int64)  # [F][512][6]
    g, fit_row = _groups()
    use = some & fit_row[None, :]
    gram = np.zeros((3, 7, 7), np.int64)
    wtw = np.zeros((3, 6, 6), np.int64)
    wtr = np.zeros((3, 6), np.float64)
    vp = np.asarray(value_params, np.float32)
    for grp in range(3):
        m = use & (g == grp)[None, :]
        v = nv[m]  # [n][6]
        val = co[m]
        u = np.concatenate([v, val[:, None]], axis=1)
        gram[grp] = u.T @ u
        # f32 prediction, left to right, one rounding per op (prediction.rs:199-204 / nalgebra gemv)
        vf = v.astype(np.float32)
        pf = vf[:, 0] * vp[grp, 0]
        for k in range(1, 6):
            pf = (pf + vf[:, k] * vp[grp, k]).astype(np.float32)
        res = np.abs(val.astype(np.float32) - pf).astype(np.float32)
        w = np.stack([np.ones(len(v), np.int64), np.abs(v[:, 0] - v[:, 3]), np.abs(v[:, 1] - v[:, 2]), np.abs(v[:, 4] - v[:, 5]), np.abs(v[:, 1] - v[:, 5]),
                      np.abs(v[:, 2] - v[:, 4])], axis=1)
        wtw[grp] = w.T @ w
        wtr[grp] = (w.astype(np.float64) * res.astype(np.float64)[:, None]).sum(0)
    return gram, wtw, wtr


@pytest.mark.parametrize("shape", [(10, 10, 3), (100, 37, 3), (300, 200, 1), (512, 512, 3)])
@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_fit_sums_match_cpu(ctx, oracle, shape, kind):
    import frave_amd

    w, h, c = shape
    img = gen_image(kind, w, h, c, 21)
    P = frave_amd.Plan(ctx, w, h, c)
    W = oracle.Wavelet(img, h, w, c)
    co = P.transform_quant(img)
    for ch in range(c):
        vp, _ = random_params(3 + ch, scale=0.2)
        want_gram, want_wtw, want_wtr = _cpu_sums(oracle, W, ch, vp)
        gram = P.fit_value_sums(co, ch)
        assert np.array_equal(gram, want_gram)
        wtw, wtr, rows = P.fit_width_sums(co, ch, vp)
        assert np.array_equal(wtw, want_wtw)
        # W^T r: f32 partial sums over the 16 nodes a lane has in a tile (the reference's whole fit is f32), fixed point with 20 fraction bits from there on
        # (each of the <= 512 partial sums per tile truncated by < 2^-20: k4_fit.hip, fit_f32_to_fixed)
        assert np.allclose(wtr, want_wtr, rtol=1e-6, atol=1e-3 + P.num_cells * 32 * 2.0 ** -20)
        assert rows.tolist() == [P.num_cells * 256, P.num_cells * 128, P.num_cells * 128]


@pytest.mark.parametrize("shape", [(700, 500, 3), (2048, 1536, 1)])
def test_the_fit_is_reproducible_bit_for_bit(ctx, shape):
    """W^T r leaves the kernel through integer (fixed-point) adds, which commute: the sums, the fitted parameters, the buckets and hence the encoder's bytes do not
    depend on the order in which workgroups finish (rounds 1-3 added doubles in arrival order: two runs could round to different f32 width parameters)."""
    import frave_amd

    w, h, c = shape
    img = gen_image("noise", w, h, c, 5)
    img[:, : w // 2] = gen_image("smooth", w // 2, h, c, 6)
    P = frave_amd.Plan(ctx, w, h, c)
    co = P.transform_quant(img)
    vp = KAT_VALUE_PARAMS
    first = [P.fit_width_sums(co, ch, vp)[1].copy() for ch in range(c)]
    for rep in range(8):
        for ch in range(c):
            assert np.array_equal(P.fit_width_sums(co, ch, vp)[1].view(np.uint64), first[ch].view(np.uint64)), f"W^T r differs between runs (rep {rep}, channel {ch})"
    a = P.encode_image(img, fit=True)
    for rep in range(4):
        b = frave_amd.Plan(ctx, w, h, c).encode_image(img, fit=True)  # (another plan: other accumulators, other arrival orders)
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))  # value / width parameters
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])  # buckets, predictions, histograms


def test_fitted_parameters_reduce_the_residual(ctx, oracle):
    """Solve the two 6 x 6 systems on the host and use the result: it must be the least-squares optimum (no worse than
    numpy's lstsq on the explicit design matrix), and K2 must run with it."""
    import frave_amd
    from frave_amd.api import solve_normal_equations

    w, h, c = 640, 360, 1
    img = gen_image("smooth", w, h, c, 8)
    P = frave_amd.Plan(ctx, w, h, c)
    W = oracle.Wavelet(img, h, w, c)
    co = P.transform_quant(img)
    gram = P.fit_value_sums(co, 0)
    vp = np.stack([solve_normal_equations(gram[g][:6, :6], gram[g][:6, 6]) for g in range(3)]).astype(np.float32)
    # reference route: explicit design matrix + lstsq (float64)
    cov = W.coefficients()[0].astype(np.float64)
    some = cov != oracle.NONE
    nv = W.neighbour_values(0).astype(np.float64)
    g, fit_row = _groups()
    for grp in range(3):
        m = some & (fit_row & (g == grp))[None, :]
        A, b = nv[m], cov[m]
        x_ref = np.linalg.lstsq(A, b, rcond=None)[0]
        r_ref = np.linalg.norm(A @ x_ref - b)
        r_got = np.linalg.norm(A @ vp[grp].astype(np.float64) - b)
        assert r_got <= r_ref * (1 + 1e-5)
        assert np.allclose(vp[grp], x_ref, rtol=1e-3, atol=1e-4)
    wtw, wtr, rows = P.fit_width_sums(co, 0, vp)
    wp = []
    for grp in range(3):
        H = wtw[grp].astype(np.float64)
        H[0, 0] += float(rows[grp]) - float(wtw[grp][0, 0])  # the reference's all-zero rows: feature 1, residual 0
        wp.append(solve_normal_equations(H, wtr[grp]))
    wp = np.stack(wp).astype(np.float32)
    assert np.isfinite(wp).all()
    b_, p_, hist, oob = P.predict_histogram(co, 0, vp, wp)
    W.quantize(np.ones(32, np.int32))
    wb, wpred, whist, woob = W.predict(0, vp, wp)
    assert np.array_equal(hist, whist) and np.array_equal(b_, wb) and np.array_equal(p_, wpred) and oob == woob


def test_fit_reports_coefficients_outside_its_range(ctx, oracle):
    """The fit kernels' 32-bit partial sums need |coefficient| <= 256 (everything the forward transform produces). A larger Some value is
    an error (FRI_HIP_ERR_OUT_OF_RANGE), not silently overflowed sums; None entries and in-range arrays pass; the plan stays usable."""
    import frave_amd as fa

    w, h, c = 200, 150, 1
    img = gen_image("noise", w, h, c, 4)
    P = fa.Plan(ctx, w, h, c)
    co = P.transform_quant(img)
    good = P.fit_value_sums(co, 0)
    bad = co.copy()
    some = np.flatnonzero(bad.reshape(-1) != fa.NONE)
    bad.reshape(-1)[some[1234]] = 300
    with pytest.raises(fa.FriHipError) as e:
        P.fit_value_sums(bad, 0)
    assert e.value.code == -7
    with pytest.raises(fa.FriHipError) as e:
        P.fit_width_sums(bad, 0, KAT_VALUE_PARAMS)
    assert e.value.code == -7
    with pytest.raises(fa.FriHipError) as e:
        P.predict_image(bad, fit=True)
    assert e.value.code == -7
    assert np.array_equal(P.fit_value_sums(co, 0), good)  # and the accumulators were left clean


def test_width_sums_report_value_parameters_from_nowhere(ctx):
    """W^T r travels as 64-bit fixed point; a partial sum of 2^24 or more (value parameters that are huge, infinite or NaN) does not fit and used to
    wrap or turn negative without a word. It is clamped and counted with the out-of-range values: the host form reports FRI_HIP_ERR_OUT_OF_RANGE, sane
    parameters right afterwards give the sums they always gave."""
    import frave_amd as fa

    w, h, c = 320, 200, 1
    P = fa.Plan(ctx, w, h, c)
    co = P.transform_quant(gen_image("noise", w, h, c, 6))
    good = P.fit_width_sums(co, 0, KAT_VALUE_PARAMS)
    for bad_value in (1e30, float("inf"), float("nan"), -3e12):
        vp = np.array(KAT_VALUE_PARAMS, np.float32).copy()
        vp[:, 2] = bad_value
        with pytest.raises(fa.FriHipError) as e:
            P.fit_width_sums(co, 0, vp)
        assert e.value.code == -7, bad_value
    again = P.fit_width_sums(co, 0, KAT_VALUE_PARAMS)
    assert all(np.array_equal(a, b) for a, b in zip(again, good))


def test_device_solves_return_the_bits_of_the_host_solves(ctx):
    """The 6 x 6 solves of the asynchronous chain run on the device (fit_solve_kernel) from the source the host functions are built from
    (csrc/solve6.hpp): for the same sums the same parameters, bit for bit - on the LDL^T route (textured data), on the eigen-decomposition
    route (rank-deficient and nearly dependent systems) and for all-zero sums."""
    import torch

    import frave_amd as fa

    P = fa.Plan(ctx, 300, 200, 1)
    rng = np.random.default_rng(11)
    iu7, iu6 = np.triu_indices(7), np.triu_indices(6)
    kinds = ["full", "full", "dependent", "zero column", "nearly dependent", "all zero", "full", "tiny"]
    n = 3 * len(kinds)
    gram, wtw, wtr = np.zeros((n, 3, 28), np.int64), np.zeros((n, 3, 21), np.int64), np.zeros((n, 3, 6), np.float64)
    for k in range(n):
        for g in range(3):
            kind = kinds[(k + g) % len(kinds)]
            rows = 4 if kind == "tiny" else int(rng.integers(50, 4000))
            a = rng.integers(-255, 256, (rows, 7)).astype(np.int64)
            if kind == "dependent":
                a[:, 3] = 2 * a[:, 2] // 2
                a[:, 2] = a[:, 3]
            elif kind == "zero column":
                a[:, 1] = 0
            elif kind == "nearly dependent":
                a[:, 4] = a[:, 0] + (rng.random(rows) < 0.001)
            elif kind == "all zero":
                a[:] = 0
            gram[k, g] = (a.T @ a)[iu7]
            w = np.abs(a[:, :6])
            w[:, 0] = 0 if kind == "all zero" else 1
            r = np.abs(rng.standard_normal(rows)) * (0 if kind == "all zero" else 7.5)
            wtw[k, g] = (w.T @ w)[iu6]
            wtr[k, g] = (w.astype(np.float64) * r[:, None]).sum(0)
    d_params = torch.zeros((n, 2, 3, 6), dtype=torch.float32, device="cuda")
    d_gram, d_wtw, d_wtr = torch.from_numpy(gram).cuda(), torch.from_numpy(wtw).cuda(), torch.from_numpy(wtr).cuda()
    P.fit_value_params_batch_dev(n, d_gram.data_ptr(), d_params.data_ptr())
    P.fit_width_params_batch_dev(n, d_wtw.data_ptr(), d_wtr.data_ptr(), d_params.data_ptr())
    torch.cuda.synchronize()
    got = d_params.cpu().numpy()
    rows = np.array([P.num_cells * 256, P.num_cells * 128, P.num_cells * 128], np.uint64)
    for k in range(n):
        assert np.array_equal(got[k, 0].view(np.uint32), fa.fit_value_params(gram[k]).view(np.uint32)), k
        assert np.array_equal(got[k, 1].view(np.uint32), fa.fit_width_params(wtw[k], wtr[k], rows).view(np.uint32)), k
    P.close()


def test_fit_at_4096_against_the_oracle(ctx, oracle):
    """BASELINE config 2's size: the fit sums of a 4096 x 4096 plane against sums built with numpy from the oracle's get_neighbour_values
    (17 M rows), then the whole chain with the fit (fri_hip_encode_image: K1 -> sums -> device solves -> sums -> device solves -> K2): the
    value parameters are the host solve of the oracle's exact sums bit for bit, the width parameters agree to rounding, and the scan's outputs
    are the oracle's predictor run with the chain's parameters."""
    import frave_amd as fa

    w = h = 4096
    img = gen_image("noise", w, h, 1, 33)
    img[:, : w // 2] = gen_image("smooth", w // 2, h, 1, 34)
    P = fa.Plan(ctx, w, h, 1)
    W = oracle.Wavelet(img, h, w, 1)
    co, vp, wp, b, p, hist, oob = P.encode_image(img, fit=True)
    assert np.array_equal(co, W.coefficients())
    iu7, iu6 = np.triu_indices(7), np.triu_indices(6)
    want_gram, want_wtw, want_wtr = _cpu_sums(oracle, W, 0, vp[0])
    assert np.array_equal(P.fit_value_sums(co, 0), want_gram)
    assert np.array_equal(vp[0].view(np.uint32), fa.fit_value_params(np.stack([want_gram[g][iu7] for g in range(3)])).view(np.uint32))
    wtw, wtr, rows = P.fit_width_sums(co, 0, vp[0])
    assert np.array_equal(wtw, want_wtw)
    assert np.allclose(wtr, want_wtr, rtol=1e-6, atol=1e-3)
    assert np.allclose(wp[0], fa.fit_width_params(np.stack([want_wtw[g][iu6] for g in range(3)]), want_wtr, rows), rtol=1e-4, atol=1e-6)
    W.quantize(np.ones(32, np.int32))
    wb, wpred, whist, woob = W.predict(0, vp[0], wp[0])
    assert np.array_equal(b[0], wb) and np.array_equal(p[0], wpred) and np.array_equal(hist[0], whist) and int(oob[0]) == woob
    W.close()
    P.close()
