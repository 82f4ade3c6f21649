"""The device-resident encode chain (fri_hip_encode_image: K1 -> fit sums -> solves -> K2 for all channels, coefficients never leaving
device memory) and the batched K2 / K4 / K3 entry points, against the CPU oracle. Reference: FRIEncoder::encode's stage chain
(encoder.rs:19-48), prediction::encode (stages/prediction.rs:224-323), the channel loop at :231."""
import numpy as np
import pytest

from tests.common import KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, gen_image, random_params

pytestmark = pytest.mark.gpu
ONES = np.ones(32, np.int32)


@pytest.fixture(scope="module")
def ctx():
    import frave_amd as fa

    c = fa.Context(0)
    yield c
    c.close()


def mixed_image(w, h, c, seed):
    img = gen_image("noise", w, h, c, seed)
    img[:, : w // 2] = gen_image("smooth", w // 2, h, c, seed + 1)  # half smooth, half noise: every ANS context is populated
    return img


def oracle_outputs(oracle, img, w, h, c, vp, wp, q=ONES):
    W = oracle.Wavelet(img, h, w, c)
    W.quantize(q)
    co = W.coefficients()
    per = [W.predict(ch, vp[ch], wp[ch]) for ch in range(c)]
    W.close()
    return co, per


@pytest.mark.parametrize("shape", [(512, 512, 3), (1920, 1080, 1), (4096, 4096, 3), (100, 37, 3)])
def test_encode_image_with_given_parameters(ctx, oracle, shape):
    import frave_amd as fa

    w, h, c = shape
    img = mixed_image(w, h, c, 7)
    P = fa.Plan(ctx, w, h, c)
    vp = np.stack([random_params(11 + ch)[0] for ch in range(c)])
    wp = np.stack([random_params(11 + ch)[1] for ch in range(c)])
    q = ONES.copy()
    if w == 512:
        q[:10] = [1, 2, 3, 1, 2, 1, 4, 1, 2, 3]
    co, vp2, wp2, b, p, hist, oob = P.encode_image(img, q, fit=False, value_params=vp, width_params=wp)
    assert np.array_equal(vp2, vp) and np.array_equal(wp2, wp)  # inputs, untouched
    want_co, per = oracle_outputs(oracle, img, w, h, c, vp, wp, q)
    assert np.array_equal(co, want_co)
    for ch in range(c):
        wb, wpred, whist, woob = per[ch]
        assert np.array_equal(b[ch], wb) and np.array_equal(p[ch], wpred) and np.array_equal(hist[ch], whist) and int(oob[ch]) == woob
    P.close()


@pytest.mark.parametrize("shape", [(512, 512, 3), (777, 333, 1)])
def test_encode_image_fits_like_the_stage_by_stage_path(ctx, oracle, shape):
    """fit = 1: the parameters equal the ones the single-stage entry points give (same exact integer sums, same solver), the
    fit itself is checked against numpy's lstsq in tests/test_gpu_fit.py; outputs = the oracle's predictor run with those parameters."""
    import frave_amd as fa

    w, h, c = shape
    img = mixed_image(w, h, c, 21)
    P = fa.Plan(ctx, w, h, c)
    co, vp, wp, b, p, hist, oob = P.encode_image(img, fit=True)
    # stage by stage through the host-pointer entry points
    co2 = P.transform_quant(img)
    assert np.array_equal(co, co2)
    iu7, iu6 = np.triu_indices(7), np.triu_indices(6)
    for ch in range(c):
        gram = P.fit_value_sums(co2, ch)
        vp_ref = fa.fit_value_params(np.stack([gram[g][iu7] for g in range(3)]))
        assert np.array_equal(vp[ch], vp_ref)
        wtw, wtr, rows = P.fit_width_sums(co2, ch, vp_ref)
        wp_ref = fa.fit_width_params(np.stack([wtw[g][iu6] for g in range(3)]), wtr, rows)
        # the f64 sums of w * r are accumulated in an order that is not fixed: the last bits of the width parameters may differ between launches
        assert np.allclose(wp[ch], wp_ref, rtol=1e-4, atol=1e-6)
    want_co, per = oracle_outputs(oracle, img, w, h, c, vp, wp)
    for ch in range(c):
        wb, wpred, whist, woob = per[ch]
        assert np.array_equal(b[ch], wb) and np.array_equal(p[ch], wpred) and np.array_equal(hist[ch], whist) and int(oob[ch]) == woob
        assert int(hist[ch].sum()) + int(oob[ch]) == P.num_some
    P.close()


def test_predict_image_on_foreign_coefficients(ctx, oracle):
    """prediction::encode alone, on coefficients the forward kernel did not produce (the exact int32 kernel takes over)."""
    import frave_amd as fa

    w, h, c = 300, 260, 3
    img = gen_image("noise", w, h, c, 3)
    P = fa.Plan(ctx, w, h, c)
    W = oracle.Wavelet(img, h, w, c)
    co = P.transform_quant(img)
    valid = co != oracle.NONE
    rng = np.random.default_rng(9)
    rnd = rng.integers(-3000, 3000, co.shape, dtype=np.int32)
    rnd[~valid] = oracle.NONE
    vp = np.stack([random_params(3 + ch)[0] for ch in range(c)])
    wp = np.stack([random_params(3 + ch)[1] for ch in range(c)])
    W.set_coefficients(rnd)
    _, _, b, p, hist, oob = P.predict_image(rnd, fit=False, value_params=vp, width_params=wp)
    for ch in range(c):
        wb, wpred, whist, woob = W.predict(ch, vp[ch], wp[ch])
        assert np.array_equal(b[ch], wb) and np.array_equal(p[ch], wpred) and np.array_equal(hist[ch], whist) and int(oob[ch]) == woob
    W.close()
    P.close()


def test_config3_batched_chain_over_1080p_frames(ctx, oracle):
    """BASELINE config 3: 256 x 1920x1080 frames. K1 in one launch, the fit sums of all 256 planes in one launch each, K2 for all planes in
    one launch, K3 for all frames in one launch. Sampled frames against the oracle; every plane against the single-plane entry points
    through checksums; every frame through the lossless round trip."""
    import torch

    import frave_amd as fa

    w, h, c, n = 1920, 1080, 1, 256
    P = fa.Plan(ctx, w, h, c)
    F, plane = P.num_cells, P.num_cells * 512
    gen = torch.Generator(device="cuda").manual_seed(5)
    d_px = torch.randint(0, 256, (n, P.pixel_bytes), dtype=torch.uint8, device="cuda", generator=gen)
    d_px[:, : P.pixel_bytes // 2] >>= 3  # a darker, smoother upper half
    d_co = torch.empty((n, plane), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=n, pixel_stride=P.pixel_bytes, coef_stride=plane)
    # fit: value sums of all planes, solves on the host, width sums of all planes, solves
    d_gram = torch.empty((n, 3, 28), dtype=torch.int64, device="cuda")
    P.fit_value_sums_batch_dev(n, d_co.data_ptr(), plane, d_gram.data_ptr(), stream=s)
    gram = d_gram.cpu().numpy()
    params = np.zeros((n, 2, 3, 6), np.float32)
    for k in range(n):
        params[k, 0] = fa.fit_value_params(gram[k])
    d_params = torch.from_numpy(params).cuda()
    d_wtw = torch.empty((n, 3, 21), dtype=torch.int64, device="cuda")
    d_wtr = torch.empty((n, 3, 6), dtype=torch.float64, device="cuda")
    P.fit_width_sums_batch_dev(n, d_co.data_ptr(), plane, d_params.data_ptr(), d_wtw.data_ptr(), d_wtr.data_ptr(), stream=s)
    wtw, wtr = d_wtw.cpu().numpy(), d_wtr.cpu().numpy()
    rows = np.array([F * 256, F * 128, F * 128], np.uint64)
    for k in range(n):
        params[k, 1] = fa.fit_width_params(wtw[k], wtr[k], rows)
    d_params = torch.from_numpy(params).cuda()
    # K2 for all planes
    d_b = torch.empty((n, plane), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, plane), dtype=torch.int32, device="cuda")
    d_h = torch.empty((n, 10, 1024), dtype=torch.int32, device="cuda")
    d_o = torch.empty(n, dtype=torch.int64, device="cuda")
    P.predict_histogram_batch_dev(n, d_co.data_ptr(), plane, d_params.data_ptr(), d_b.data_ptr(), d_p.data_ptr(), plane, d_h.data_ptr(), d_o.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert bool((d_h.sum(dim=(1, 2)) + d_o == P.num_some).all())
    # every plane: the single-plane entry points give the same sums and the same outputs
    d_g1 = torch.empty((3, 28), dtype=torch.int64, device="cuda")
    d_b1 = torch.empty(plane, dtype=torch.uint8, device="cuda")
    d_p1 = torch.empty(plane, dtype=torch.int32, device="cuda")
    d_h1 = torch.empty((10, 1024), dtype=torch.int32, device="cuda")
    d_o1 = torch.empty(1, dtype=torch.int64, device="cuda")
    for k in range(0, n, 5):
        P.fit_value_sums_dev(d_co[k].data_ptr(), 0, d_g1.data_ptr(), stream=s)
        assert torch.equal(d_g1, d_gram[k]), k
        P.predict_histogram_dev(d_co[k].data_ptr(), 0, params[k, 0], params[k, 1], d_b1.data_ptr(), d_p1.data_ptr(), d_h1.data_ptr(), d_o1.data_ptr(), stream=s)
        assert torch.equal(d_b1, d_b[k]) and torch.equal(d_p1, d_p[k]) and torch.equal(d_h1, d_h[k]) and int(d_o1) == int(d_o[k]), k
    # sampled frames against the oracle
    for k in (0, 101, 255):
        img = d_px[k].cpu().numpy()
        W = oracle.Wavelet(img, h, w, c)
        assert np.array_equal(d_co[k].cpu().numpy().reshape(1, F, 512), W.coefficients())
        W.quantize(ONES)
        wb, wpred, whist, woob = W.predict(0, params[k, 0], params[k, 1])
        assert np.array_equal(d_b[k].cpu().numpy().reshape(F, 512), wb) and np.array_equal(d_p[k].cpu().numpy().reshape(F, 512), wpred)
        assert np.array_equal(d_h[k].cpu().numpy().astype(np.uint32), whist) and int(d_o[k]) == woob
        W.close()
    # K3 for all frames in one launch: lossless
    d_back = torch.empty((n, P.pixel_bytes), dtype=torch.uint8, device="cuda")
    P.inverse_transform_batch_dev(n, d_co.data_ptr(), plane, d_back.data_ptr(), P.pixel_bytes, stream=s)
    torch.cuda.synchronize()
    assert torch.equal(d_back, d_px)
    # The same batch as ONE asynchronous call (fri_hip_encode_image_batch_dev: K1 -> value sums -> device solves -> width sums -> device solves ->
    # K2, nothing but enqueues): coefficients and value parameters are the stage-by-stage ones bit for bit (exact integer sums, one solver source),
    # the width parameters agree to the rounding of their f64 sums, and sampled frames are the oracle's predictor run with the batch's own parameters.
    d_co2 = torch.empty_like(d_co)
    d_par2 = torch.zeros((n, 2, 3, 6), dtype=torch.float32, device="cuda")
    d_b2, d_p2, d_h2, d_o2 = torch.empty_like(d_b), torch.empty_like(d_p), torch.empty_like(d_h), torch.empty_like(d_o)
    d_rng = torch.ones(n, dtype=torch.int64, device="cuda")
    P.encode_image_batch_dev(n, d_px.data_ptr(), P.pixel_bytes, d_par2.data_ptr(), d_co2.data_ptr(), plane, d_b2.data_ptr(), d_p2.data_ptr(), plane, d_h2.data_ptr(),
                             d_o2.data_ptr(), fit=True, d_fit_out_of_range=d_rng.data_ptr(), stream=s)
    torch.cuda.synchronize()
    par2 = d_par2.cpu().numpy()
    assert torch.equal(d_co2, d_co) and int(d_rng.abs().sum()) == 0
    assert np.array_equal(par2[:, 0].view(np.uint32), params[:, 0].view(np.uint32))
    assert np.allclose(par2[:, 1], params[:, 1], rtol=1e-4, atol=1e-6)
    assert bool((d_h2.sum(dim=(1, 2)) + d_o2 == P.num_some).all())
    for k in (3, 200):
        W = oracle.Wavelet(d_px[k].cpu().numpy(), h, w, c)
        W.quantize(ONES)
        wb, wpred, whist, woob = W.predict(0, par2[k, 0], par2[k, 1])
        assert np.array_equal(d_b2[k].cpu().numpy().reshape(F, 512), wb) and np.array_equal(d_p2[k].cpu().numpy().reshape(F, 512), wpred)
        assert np.array_equal(d_h2[k].cpu().numpy().astype(np.uint32), whist) and int(d_o2[k]) == woob
        W.close()
    # parameters given (fit = 0): the batch call is K1 + the batched scan
    P.encode_image_batch_dev(n, d_px.data_ptr(), P.pixel_bytes, d_params.data_ptr(), d_co2.data_ptr(), plane, d_b2.data_ptr(), d_p2.data_ptr(), plane, d_h2.data_ptr(),
                             d_o2.data_ptr(), fit=False, stream=s)
    torch.cuda.synchronize()
    assert torch.equal(d_b2, d_b) and torch.equal(d_p2, d_p) and torch.equal(d_h2, d_h) and torch.equal(d_o2, d_o)
    P.close()


def test_encode_image_batch_dev_rgb_images_back_to_back(ctx, oracle):
    """Three RGB images in one asynchronous chain (nine planes, evenly spaced because the images lie back to back); a layout that is not
    evenly spaced is refused."""
    import torch

    import frave_amd as fa

    w, h, c, n = 640, 360, 3, 3
    P = fa.Plan(ctx, w, h, c)
    F, plane = P.num_cells, P.num_cells * 512
    imgs = [mixed_image(w, h, c, 40 + k) for k in range(n)]
    d_px = torch.from_numpy(np.stack([im.reshape(-1) for im in imgs])).cuda()
    d_co = torch.empty((n, c, plane), dtype=torch.int32, device="cuda")
    d_par = torch.zeros((n, c, 2, 3, 6), dtype=torch.float32, device="cuda")
    d_b = torch.empty((n, c, plane), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((n, c, plane), dtype=torch.int32, device="cuda")
    d_h = torch.empty((n, c, 10, 1024), dtype=torch.int32, device="cuda")
    d_o = torch.empty((n, c), dtype=torch.int64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    P.encode_image_batch_dev(n, d_px.data_ptr(), P.pixel_bytes, d_par.data_ptr(), d_co.data_ptr(), c * plane, d_b.data_ptr(), d_p.data_ptr(), c * plane, d_h.data_ptr(),
                             d_o.data_ptr(), fit=True, stream=s)
    torch.cuda.synchronize()
    par = d_par.cpu().numpy()
    for k in range(n):
        co1, vp1, wp1, b1, p1, hist1, oob1 = P.encode_image(imgs[k], fit=True)  # the one-image call
        assert np.array_equal(d_co[k].cpu().numpy().reshape(c, F, 512), co1)
        assert np.array_equal(par[k, :, 0].view(np.uint32), vp1.view(np.uint32)) and np.allclose(par[k, :, 1], wp1, rtol=1e-4, atol=1e-6)
        W = oracle.Wavelet(imgs[k], h, w, c)
        W.quantize(ONES)
        for ch in range(c):
            wb, wpred, whist, woob = W.predict(ch, par[k, ch, 0], par[k, ch, 1])
            assert np.array_equal(d_b[k, ch].cpu().numpy().reshape(F, 512), wb) and np.array_equal(d_p[k, ch].cpu().numpy().reshape(F, 512), wpred)
            assert np.array_equal(d_h[k, ch].cpu().numpy().astype(np.uint32), whist) and int(d_o[k, ch]) == woob
        W.close()
    with pytest.raises(fa.FriHipError) as e:
        P.encode_image_batch_dev(n, d_px.data_ptr(), P.pixel_bytes, d_par.data_ptr(), d_co.data_ptr(), c * plane + 512, d_b.data_ptr(), d_p.data_ptr(), c * plane, d_h.data_ptr(),
                                 d_o.data_ptr(), fit=True, stream=s)
    assert e.value.code == -1
    P.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_batched_entry_points_on_random_small_shapes(ctx, oracle, seed):
    """The batch launches (planes / images on the second grid dimension, a plane on an eighth of the machine) against the
    single-plane entry points and the oracle, on shapes from one tile to a few dozen, with padded strides and 1..9 planes."""
    import torch

    import frave_amd as fa

    rng = np.random.default_rng(seed)
    s = torch.cuda.current_stream().cuda_stream
    for case in range(6):
        w, h = (int(rng.integers(1, 50)), int(rng.integers(1, 700))) if case == 0 else (int(rng.integers(40, 900)), int(rng.integers(40, 600)))
        n = int(rng.integers(1, 10))
        P = fa.Plan(ctx, w, h, 1)
        F, plane = P.num_cells, P.num_cells * 512
        pad = 512 * int(rng.integers(0, 3))  # strides larger than a plane
        imgs = [gen_image(["noise", "smooth", "const"][int(rng.integers(0, 3))], w, h, 1, int(rng.integers(0, 1 << 20))) for _ in range(n)]
        d_px = torch.from_numpy(np.stack([im.reshape(-1) for im in imgs])).cuda()
        d_co = torch.full((n, plane + pad), 12345, dtype=torch.int32, device="cuda")
        P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=n, pixel_stride=P.pixel_bytes, coef_stride=plane + pad)
        params = np.stack([np.stack(random_params(int(rng.integers(0, 1000)))) for _ in range(n)]).astype(np.float32)  # [n][2][3][6]
        d_params = torch.from_numpy(params).cuda()
        d_b = torch.zeros((n, plane + pad), dtype=torch.uint8, device="cuda")
        d_p = torch.zeros((n, plane + pad), dtype=torch.int32, device="cuda")
        d_h = torch.empty((n, 10, 1024), dtype=torch.int32, device="cuda")
        d_o = torch.empty(n, dtype=torch.int64, device="cuda")
        P.predict_histogram_batch_dev(n, d_co.data_ptr(), plane + pad, d_params.data_ptr(), d_b.data_ptr(), d_p.data_ptr(), plane + pad, d_h.data_ptr(), d_o.data_ptr(), stream=s)
        d_g = torch.empty((n, 3, 28), dtype=torch.int64, device="cuda")
        P.fit_value_sums_batch_dev(n, d_co.data_ptr(), plane + pad, d_g.data_ptr(), stream=s)
        d_w = torch.empty((n, 3, 21), dtype=torch.int64, device="cuda")
        d_r = torch.empty((n, 3, 6), dtype=torch.float64, device="cuda")
        P.fit_width_sums_batch_dev(n, d_co.data_ptr(), plane + pad, d_params.data_ptr(), d_w.data_ptr(), d_r.data_ptr(), stream=s)
        d_back = torch.zeros((n, P.pixel_bytes + 16), dtype=torch.uint8, device="cuda")
        P.inverse_transform_batch_dev(n, d_co.data_ptr(), plane + pad, d_back.data_ptr(), P.pixel_bytes + 16, stream=s)
        torch.cuda.synchronize()
        assert bool((d_co[:, plane:] == 12345).all()) and bool((d_back[:, P.pixel_bytes :] == 0).all())  # nothing written between the planes
        d_g1 = torch.empty((3, 28), dtype=torch.int64, device="cuda")
        d_w1 = torch.empty((3, 21), dtype=torch.int64, device="cuda")
        d_r1 = torch.empty((3, 6), dtype=torch.float64, device="cuda")
        for k in range(n):
            W = oracle.Wavelet(imgs[k], h, w, 1)
            assert np.array_equal(d_co[k, :plane].cpu().numpy().reshape(1, F, 512), W.coefficients()), (w, h, n, k)
            W.quantize(ONES)
            wb, wpred, whist, woob = W.predict(0, params[k, 0], params[k, 1])
            want_back = W.to_raster()  # (images thinner than a cell are not covered by the reference's lattice: compare with ITS way back)
            W.close()
            assert np.array_equal(d_b[k, :plane].cpu().numpy().reshape(F, 512), wb) and np.array_equal(d_p[k, :plane].cpu().numpy().reshape(F, 512), wpred), (w, h, n, k)
            assert np.array_equal(d_h[k].cpu().numpy().astype(np.uint32), whist) and int(d_o[k]) == woob, (w, h, n, k)
            P.fit_value_sums_dev(d_co[k].data_ptr(), 0, d_g1.data_ptr(), stream=s)
            P.fit_width_sums_dev(d_co[k].data_ptr(), 0, params[k, 0], d_w1.data_ptr(), d_r1.data_ptr(), stream=s)
            assert torch.equal(d_g1, d_g[k]) and torch.equal(d_w1, d_w[k]), (w, h, n, k)
            assert torch.allclose(d_r1, d_r[k], rtol=1e-9, atol=1e-6), (w, h, n, k)
            assert np.array_equal(d_back[k, : P.pixel_bytes].cpu().numpy(), want_back.reshape(-1)), (w, h, n, k)
        P.close()
