"""Host emit path: frave_amd/host/emit.cpp through libfri_emit.so against oracle/emit_oracle.py,
which restates the reference step by step (literal scan_level walk, finalize_context, rans64, serialize). CPU only, except the
last test, which feeds the emitter with the arrays the kernels produce."""
import os
import struct

import numpy as np
import pytest

from oracle import emit_oracle, fri_oracle  # noqa: E402
from tests.common import KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, gen_image, random_params  # noqa: E402

import frave_amd.emit as emit  # noqa: E402

SIZES = [(10, 10), (64, 48), (100, 37), (33, 17), (300, 200), (200, 300), (129, 65), (5, 5), (1, 300), (512, 512)]


@pytest.mark.parametrize("size", SIZES)
def test_symbol_order_is_the_reference_walk(size):
    w, h = size
    W = fri_oracle.Wavelet(gen_image("noise", w, h, 1, 1), h, w, 1)
    centers = W.centers()
    lit = [fri_oracle.literal(i) for i in range(11)]
    for level in range(9):
        got = emit.symbol_order(centers, level)
        assert len(got) == W.num_cells << level
        # position of (cell, heap): centre + sum over the path bits (Fractal::new, wavelet_transform.rs:42-69)
        i = got[:, 1].astype(np.int64) - (1 << level)
        pos = centers[got[:, 0]].astype(np.int64)
        for j in range(level):
            bit = (i >> j) & 1
            pos[:, 0] += bit * lit[9 - level + j][0]
            pos[:, 1] += bit * lit[9 - level + j][1]
        ref = W.sorted_level(level).astype(np.int64)
        if len(ref) != len(got):
            # the reference's own walk loses nodes here and its assertion (wavelet_transform.rs:701) fires: only ever seen for
            # images thinner than a cell, which libfri cannot encode anyway (thin-image limitation)
            assert min(w, h) < 46, (size, level, len(ref), len(got))
            continue
        assert np.array_equal(pos, ref), (size, level)


def _laplace_counts(rng, bucket, n, outliers):
    width = emit_oracle.WIDTHS[bucket]
    k = np.rint(rng.laplace(0.0, width, n)).astype(np.int64)
    sym = np.where(k >= 0, 2 * k, -2 * k - 1)
    counts = np.bincount(sym[sym < 1024], minlength=1024).astype(np.uint32)
    for s in outliers:
        counts[s] += 1
    return counts


@pytest.mark.parametrize("bucket", range(10))
@pytest.mark.parametrize("n", [40, 3000, 200000])
def test_finalize_context_matches_restatement(bucket, n):
    rng = np.random.default_rng(100 * bucket + n)
    counts = _laplace_counts(rng, bucket, n, outliers=rng.integers(300, 1023, 5))
    f, cdf, off, bits = emit.finalize_context(counts, bucket)
    c = emit_oracle.Context()
    c.freqs = [int(v) for v in counts]
    c.max_freq_bits = emit_oracle.trailing_zeros(emit_oracle.prev_power_two(int(counts.sum())))
    c.finalize(bucket)
    assert f.tolist() == c.freqs and cdf.tolist() == c.cdf and off.tolist() == c.off and bits == c.max_freq_bits
    # what the coder needs: every observed symbol keeps a slot, the slots tile [0, 2^bits)
    assert all(f[s] > 0 for s in np.flatnonzero(counts)) and int(f.sum()) == 1 << bits and np.array_equal(np.cumsum(f) - f, cdf)


def test_empty_context_is_an_error_like_the_reference_panic():
    with pytest.raises(emit.EmitError, match="empty context"):
        emit.finalize_context(np.zeros(1024, np.uint32), 3)
    with pytest.raises(ZeroDivisionError):
        c = emit_oracle.Context()
        c.max_freq_bits = 64
        c.finalize(3)


def _mixed_image(w, h, c, seed):
    """left half smooth, right half noise: small and large prediction widths, so that all ten contexts get symbols"""
    a = gen_image("smooth", w, h, c, seed).reshape(h, w, c)
    b = gen_image("noise", w, h, c, seed + 1).reshape(h, w, c)
    return np.where((np.arange(w) < w // 2)[None, :, None], a, b).astype(np.uint8).reshape(-1)


def _arrays(w, h, c, seed, kind="mixed"):
    img = _mixed_image(w, h, c, seed) if kind == "mixed" else gen_image(kind, w, h, c, seed)
    W = fri_oracle.Wavelet(img, h, w, c)
    coefs = W.coefficients()
    vps, wps, bs, ps, hs = [], [], [], [], []
    for ch in range(c):
        vp, wp = KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS
        b, p, hist, oob = W.predict(ch, vp, wp)
        assert oob == 0
        vps.append(np.asarray(vp, np.float32).reshape(3, 6)), wps.append(np.asarray(wp, np.float32).reshape(3, 6))
        bs.append(b), ps.append(p), hs.append(hist)
    return W, coefs, np.stack(bs), np.stack(ps), np.stack(hs), np.stack(vps), np.stack(wps)


@pytest.mark.parametrize("shape", [(129, 65, 1), (300, 200, 1), (160, 120, 3), (96, 257, 3)])
def test_frv_bytes_match_restatement_and_decode(shape):
    w, h, c = shape
    W, coefs, bucket, pred, hist, vp, wp = _arrays(w, h, c, 7)
    assert (hist.sum(axis=2) > 0).all(), "the test image must populate every context"
    want = emit_oracle.encode_image(W, coefs, bucket, pred, hist, vp, wp)
    got = emit.encode_image(w, h, W.centers(), coefs, bucket, pred, hist, vp, wp)
    assert got == want
    assert got[:4] == b"frif" and struct.unpack("<II", got[4:12]) == (h, w) and got[-2:] == b"\xff\xdf"
    emit.check_image(got, W.centers(), coefs, bucket, pred)  # parse -> rebuild the models -> decode all symbols
    for ch in range(c):  # and the streams carry exactly the symbols in reference order
        sym, bk = emit.channel_symbols(W.centers(), coefs[ch], bucket[ch], pred[ch])
        ref = emit_oracle.stream_symbols(W, ch, coefs[ch], bucket[ch], pred[ch])
        assert list(zip(sym.tolist(), bk.tolist())) == ref


@pytest.mark.parametrize("name", ["emit_mixed_129x65_luma", "emit_mixed_96x257_rgb"])
def test_frv_golden_files(name):
    """committed .frv fixtures (tests/golden/make_golden.py): the product and the restatement both still produce them"""
    from tests.golden.make_golden import EMIT_CASES, build_frv

    w, h, c, seed = EMIT_CASES[name]
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".frv"), "rb") as f:
        golden = f.read()
    W, coefs, bucket, pred, hist, vp, wp = _arrays(w, h, c, seed)
    assert emit.encode_image(w, h, W.centers(), coefs, bucket, pred, hist, vp, wp) == golden
    assert build_frv(name) == golden


def test_image_with_an_empty_context_is_reported_like_the_reference_panic():
    w, h, c = 64, 48, 1  # too few symbols for ten contexts
    W, coefs, bucket, pred, hist, vp, wp = _arrays(w, h, c, 7, kind="noise")
    assert (hist.sum(axis=2) == 0).any()
    with pytest.raises(ZeroDivisionError):
        emit_oracle.encode_image(W, coefs, bucket, pred, hist, vp, wp)
    with pytest.raises(emit.EmitError, match="empty context"):
        emit.encode_image(w, h, W.centers(), coefs, bucket, pred, hist, vp, wp)


def test_corrupted_stream_is_detected():
    w, h, c = 129, 65, 1
    W, coefs, bucket, pred, hist, vp, wp = _arrays(w, h, c, 3)
    frv = bytearray(emit.encode_image(w, h, W.centers(), coefs, bucket, pred, hist, vp, wp))
    frv[len(frv) // 2] ^= 0x55
    with pytest.raises(emit.EmitError):
        emit.check_image(bytes(frv), W.centers(), coefs, bucket, pred)


@pytest.mark.parametrize("shape", [(129, 65, 1), (300, 200, 1), (160, 120, 3), (96, 257, 3), (512, 512, 1)])
def test_full_decode_recovers_the_coefficients(shape):
    """entropy_coding::decode (:352-443): the decoder knows only the container. It recomputes every symbol's context from the
    coefficients decoded before it, so the planes come back iff the stream order is causal for the 6-neighbour context and the
    host predictor computes bit for bit what the encoder side (here: the oracle's predict) computed."""
    w, h, c = shape
    W, coefs, bucket, pred, hist, vp, wp = _arrays(w, h, c, 21)
    frv = emit.encode_image(w, h, W.centers(), coefs, bucket, pred, hist, vp, wp)
    dw, dh, dc, centers, got = emit.decode_image(frv)
    assert (dw, dh, dc) == (w, h, c) and np.array_equal(centers, W.centers())
    assert np.array_equal(got.reshape(-1), np.asarray(coefs).reshape(-1))
    W.set_coefficients(got)  # and the oracle's inverse transform turns them back into the pixels
    assert np.array_equal(np.asarray(W.to_raster()).reshape(-1), _mixed_image(w, h, c, 21))


@pytest.mark.parametrize("shape", [(129, 65, 1), (96, 120, 3)])
def test_literal_decoder_of_the_restatement_reads_the_product_stream(shape):
    """oracle/emit_oracle.decode_image walks the reference's decoder literally (hash-map lattice, get_lf / get_hf_context_bucket
    per symbol on the coefficients decoded so far); it must read the product's .frv back to the planes the stream was made from,
    and agree with the product's own decoder."""
    w, h, c = shape
    W, coefs, bucket, pred, hist, vp, wp = _arrays(w, h, c, 31)
    frv = emit.encode_image(w, h, W.centers(), coefs, bucket, pred, hist, vp, wp)
    ow, oh, oc, got = emit_oracle.decode_image(frv)
    assert (ow, oh, oc) == (w, h, c)
    assert np.array_equal(got.reshape(-1), np.asarray(coefs).reshape(-1))
    assert np.array_equal(emit.decode_image(frv)[4].reshape(-1), got.reshape(-1))


def test_full_decode_with_fitted_looking_parameters():
    """non-dyadic f32 parameters: the rounding order of the predictor matters now (prediction.rs:190-206)"""
    w, h, c = 200, 150, 3
    img = _mixed_image(w, h, c, 5)
    W = fri_oracle.Wavelet(img, h, w, c)
    coefs = W.coefficients()
    rng = np.random.default_rng(5)
    bs, ps, hs, vps, wps = [], [], [], [], []
    for ch in range(c):
        vp = (np.asarray(KAT_VALUE_PARAMS, np.float32).reshape(3, 6) * rng.uniform(0.8, 1.2, (3, 6))).astype(np.float32)
        wp = (np.asarray(KAT_WIDTH_PARAMS, np.float32).reshape(3, 6) * rng.uniform(0.8, 1.2, (3, 6))).astype(np.float32)
        b, p, hist, oob = W.predict(ch, vp, wp)
        assert oob == 0
        bs.append(b), ps.append(p), hs.append(hist), vps.append(vp), wps.append(wp)
    frv = emit.encode_image(w, h, W.centers(), coefs, np.stack(bs), np.stack(ps), np.stack(hs), np.stack(vps), np.stack(wps))
    _, _, _, _, got = emit.decode_image(frv)
    assert np.array_equal(got.reshape(-1), np.asarray(coefs).reshape(-1))


@pytest.mark.parametrize("name", ["emit_mixed_129x65_luma", "emit_mixed_96x257_rgb"])
def test_golden_frv_files_decode(name):
    from tests.golden.make_golden import EMIT_CASES

    w, h, c, seed = EMIT_CASES[name]
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".frv"), "rb") as f:
        golden = f.read()
    W, coefs, *_ = _arrays(w, h, c, seed)
    dw, dh, dc, _, got = emit.decode_image(golden)
    assert (dw, dh, dc) == (w, h, c) and np.array_equal(got.reshape(-1), np.asarray(coefs).reshape(-1))


@pytest.mark.parametrize("n,seed", [(1, 1), (1000, 2), (65535, 3), (65536, 4), (65537, 5), (300000, 6), (3000001, 7)])
def test_context_parallel_rans_coder_equals_the_one_loop_coder(n, seed):
    """Large channels are coded context by context on threads and stitched (emit.cpp encode_symbols); the stream must be byte for byte
    what the reference's single loop over the symbols (entropy_coding.rs:332-347) gives - also around the span size of the stitching pass."""
    emit.rans_selfcheck(n, seed)


def test_full_decode_rejects_damaged_files():
    w, h, c = 129, 65, 1
    W, coefs, bucket, pred, hist, vp, wp = _arrays(w, h, c, 3)
    frv = emit.encode_image(w, h, W.centers(), coefs, bucket, pred, hist, vp, wp)
    with pytest.raises(emit.EmitError, match="Invalid signature"):
        emit.decode_image(b"frig" + frv[4:])
    with pytest.raises(emit.EmitError):
        emit.decode_image(frv[: len(frv) // 2])
    with pytest.raises(emit.EmitError):  # a stream cut short inside the DAT segment, lengths patched to match
        at = frv.index(b"\xff\xb4")
        n = struct.unpack("<Q", frv[at + 2 : at + 10])[0]
        emit.decode_image(frv[: at + 2] + struct.pack("<Q", 40) + frv[at + 10 : at + 50] + frv[at + 10 + n :])
    damaged = bytearray(frv)
    damaged[frv.index(b"\xff\xb4") + 200] ^= 0x55  # inside the rANS words: decodes to other coefficients or fails, never crashes
    try:
        _, _, _, _, got = emit.decode_image(bytes(damaged))
        assert not np.array_equal(got.reshape(-1), np.asarray(coefs).reshape(-1))
    except emit.EmitError:
        pass


def _numpy_streams(order, coefs, bucket, pred):
    """What K5 computes, in numpy: per channel, bucket << 10 | pack_signed(coef - prediction) of the nodes in `order`."""
    c = coefs.shape[0]
    out = []
    for ch in range(c):
        d = (coefs[ch].reshape(-1)[order].astype(np.int64) - pred[ch].reshape(-1)[order].astype(np.int64)).astype(np.int32)
        sym = ((d.astype(np.uint32) << 1) ^ (d >> 31).astype(np.uint32)) & 1023
        out.append((bucket[ch].reshape(-1)[order].astype(np.uint32) << 10 | sym).astype(np.uint16))
    return np.stack(out)


@pytest.mark.parametrize("name", ["emit_mixed_129x65_luma", "emit_mixed_96x257_rgb"])
def test_stream_route_reproduces_the_golden_files(name):
    """The symbol-stream route (stream order without the None nodes + 2-byte symbols + fri_emit_encode_image_from_streams) makes the committed
    .frv files byte for byte; the stream order is the reference's (oracle walk), None nodes taken out."""
    import frave_amd
    from tests.golden.make_golden import EMIT_CASES

    w, h, c, seed = EMIT_CASES[name]
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".frv"), "rb") as f:
        golden = f.read()
    W, coefs, bucket, pred, hist, vp, wp = _arrays(w, h, c, seed)
    P = frave_amd.Plan(None, w, h, c)  # host-only plan: geometry getters
    assert np.array_equal(P.centers(), W.centers())
    order = emit.stream_order(P.centers(), P.valid_mask())
    assert len(order) == P.num_some and len(np.unique(order)) == len(order)
    some = coefs[0].reshape(-1) != fri_oracle.NONE
    assert some[order].all() and some.sum() == len(order)
    for ch in range(c):  # the sequence the reference's loop feeds to its coder
        ref = emit_oracle.stream_symbols(W, ch, coefs[ch], bucket[ch], pred[ch])
        st = _numpy_streams(order, coefs, bucket, pred)[ch]
        assert list(zip((st & 1023).tolist(), (st >> 10).tolist())) == ref
    assert emit.encode_image_from_streams(w, h, _numpy_streams(order, coefs, bucket, pred), hist, vp, wp) == golden


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(129, 65, 1), (700, 90, 1), (100, 37, 3), (512, 512, 3), (640, 360, 1)])
def test_device_symbol_stream_is_the_references_walk(shape):
    """The device route to the emitter's input (fri_hip_encode_image_symbols: K1 -> K2 halfwords -> K5) against the reference's own loop, restated literally
    (oracle/emit_oracle.stream_symbols: entropy_coding::encode, entropy_coding.rs:285-336, walking sort_lattice / scan_level, wavelet_transform.rs:505-705) over
    the ORACLE's coefficients, buckets and predictions: symbol for symbol, bucket for bucket, in stream order. (test_symbol_stream_on_the_device checks the
    device's stream against a gather of the device's own arrays; the order against the literal walk is otherwise a CPU-only test.)"""
    import frave_amd

    w, h, c = shape
    img = _mixed_image(w, h, c, 21)
    W = fri_oracle.Wavelet(img, h, w, c)
    coefs = W.coefficients()
    ctx = frave_amd.Context(0)
    P = frave_amd.Plan(ctx, w, h, c)
    P.set_stream_order()
    vp = np.tile(np.asarray(KAT_VALUE_PARAMS, np.float32).reshape(1, 3, 6), (c, 1, 1))
    wp = np.tile(np.asarray(KAT_WIDTH_PARAMS, np.float32).reshape(1, 3, 6), (c, 1, 1))
    sym, _, _, hist, oob = P.encode_image_symbols(img, fit=False, value_params=vp, width_params=wp)
    assert not oob.any()
    for ch in range(c):
        b, p, want_hist, want_oob = W.predict(ch, KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS)
        assert want_oob == 0 and np.array_equal(hist[ch], want_hist)
        ref = emit_oracle.stream_symbols(W, ch, coefs[ch], b, p)  # [(symbol, bucket)] as the reference's loop feeds its coder
        assert len(ref) == P.num_some
        got = sym[ch]
        assert list(zip((got & 1023).tolist(), (got >> 10).tolist())) == ref
    P.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(129, 65, 1), (640, 360, 3), (4096, 4096, 1)])
def test_symbol_stream_on_the_device(shape):
    """K5 (fri_hip_symbol_stream_batch_dev): the stream equals the gather of the device's own arrays in stream order, and the .frv made from it is the
    one the array route (fri_emit_encode_image: gather on the host) makes."""
    import torch

    import frave_amd

    w, h, c = shape
    img = _mixed_image(w, h, c, 13)
    ctx = frave_amd.Context(0)
    P = frave_amd.Plan(ctx, w, h, c)
    order = P.set_stream_order()
    F, plane, n = P.num_cells, P.num_cells * 512, P.num_some
    assert len(order) == n
    d_px = torch.from_numpy(img.reshape(-1)).cuda()
    d_co = torch.empty((c, plane), dtype=torch.int32, device="cuda")
    d_b = torch.empty((c, plane), dtype=torch.uint8, device="cuda")
    d_p = torch.empty((c, plane), dtype=torch.int32, device="cuda")
    d_h = torch.empty((c, 10, 1024), dtype=torch.int32, device="cuda")
    d_o = torch.empty(c, dtype=torch.int64, device="cuda")
    d_par = torch.zeros((c, 2, 3, 6), dtype=torch.float32, device="cuda")
    d_st = torch.full((c, n + 8), 0xFFFF, dtype=torch.uint16, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    P.encode_image_batch_dev(1, d_px.data_ptr(), P.pixel_bytes, d_par.data_ptr(), d_co.data_ptr(), c * plane, d_b.data_ptr(), d_p.data_ptr(), c * plane, d_h.data_ptr(), d_o.data_ptr(),
                             fit=True, stream=s)
    P.symbol_stream_batch_dev(c, d_co.data_ptr(), plane, d_b.data_ptr(), d_p.data_ptr(), plane, d_st.data_ptr(), n + 8, stream=s)
    torch.cuda.synchronize()
    assert int(d_o.abs().sum()) == 0
    st = d_st.cpu().numpy()
    assert (st[:, n:] == 0xFFFF).all()  # nothing behind a plane's symbols
    co, b, p = d_co.cpu().numpy().reshape(c, F, 512), d_b.cpu().numpy().reshape(c, F, 512), d_p.cpu().numpy().reshape(c, F, 512)
    assert np.array_equal(st[:, :n], _numpy_streams(order, co, b, p))
    par = d_par.cpu().numpy()
    hist = d_h.cpu().numpy().astype(np.uint32)
    from_streams = emit.encode_image_from_streams(w, h, np.ascontiguousarray(st[:, :n]), hist, par[:, 0], par[:, 1])
    from_arrays = emit.encode_image(w, h, P.centers(), co, b, p, hist, par[:, 0], par[:, 1])
    assert from_streams == from_arrays
    if w <= 640:
        dw, dh, dc, centers, coefs = emit.decode_image(from_streams)
        assert np.array_equal(coefs, co)
    # The halfword route (fri_hip_encode_symbols_batch_dev: the scan writes bucket << 10 | symbol per node, one 2-byte gather per symbol): two images in one
    # chain, the first the image above - same streams, histograms and parameters as the array route; the node words are the counters the nodes bumped.
    n_img = 2
    img2 = np.stack([img, _mixed_image(w, h, c, 14)])
    d_px2 = torch.from_numpy(img2.reshape(-1)).cuda()
    d_co2 = torch.empty((n_img, c, plane), dtype=torch.int32, device="cuda")
    d_w2 = torch.full((n_img, c, plane), 0xEEEE, dtype=torch.uint16, device="cuda")
    d_st2 = torch.full((n_img * c * n + 8,), 0xFFFF, dtype=torch.uint16, device="cuda")
    d_h2 = torch.empty((n_img, c, 10, 1024), dtype=torch.int32, device="cuda")
    d_o2 = torch.empty((n_img, c), dtype=torch.int64, device="cuda")
    d_par2 = torch.zeros((n_img, c, 2, 3, 6), dtype=torch.float32, device="cuda")
    d_r2 = torch.zeros((n_img, c), dtype=torch.int64, device="cuda")
    P.encode_symbols_batch_dev(n_img, d_px2.data_ptr(), P.pixel_bytes, None, True, d_par2.data_ptr(), d_co2.data_ptr(), c * plane, d_w2.data_ptr(), c * plane, d_st2.data_ptr(),
                               c * n, d_h2.data_ptr(), d_o2.data_ptr(), d_r2.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert int(d_o2.abs().sum()) == 0 and int(d_r2.abs().sum()) == 0
    st2 = d_st2.cpu().numpy()
    assert (st2[n_img * c * n:] == 0xFFFF).all()
    st2 = st2[: n_img * c * n].reshape(n_img, c, n)
    assert np.array_equal(st2[0], st[:, :n]) and np.array_equal(d_h2[0].cpu().numpy(), d_h.cpu().numpy()) and np.array_equal(d_par2[0].cpu().numpy(), par)
    assert np.array_equal(d_co2[0].cpu().numpy().reshape(c, F, 512), co)
    words = d_w2.cpu().numpy().reshape(n_img, c, -1)
    for k in range(n_img):
        for ch in range(c):
            assert np.array_equal(words[k, ch][order], st2[k, ch])  # the stream is the gather of the node words ...
            assert np.array_equal(np.bincount(st2[k, ch], minlength=10240), d_h2[k, ch].cpu().numpy().reshape(-1))  # ... and the histogram counts exactly them
    # second image against the array route run on it alone
    P.encode_image_batch_dev(1, d_px2[P.pixel_bytes:].data_ptr(), P.pixel_bytes, d_par.data_ptr(), d_co.data_ptr(), c * plane, d_b.data_ptr(), d_p.data_ptr(), c * plane, d_h.data_ptr(),
                             d_o.data_ptr(), fit=True, stream=s)
    torch.cuda.synchronize()
    co, b, p = d_co.cpu().numpy().reshape(c, F, 512), d_b.cpu().numpy().reshape(c, F, 512), d_p.cpu().numpy().reshape(c, F, 512)
    assert np.array_equal(st2[1], _numpy_streams(order, co, b, p))
    P.close()


@pytest.mark.gpu
def test_symbol_stream_entry_points_reject_what_they_cannot_do():
    """No stream order uploaded, a stream order that is no permutation of the Some nodes, strides that overlap: error codes, and the plan stays usable."""
    import torch

    import frave_amd

    w, h, c = 160, 96, 3
    ctx = frave_amd.Context(0)
    P = frave_amd.Plan(ctx, w, h, c)
    plane, n = P.num_cells * 512, P.num_some
    img = _mixed_image(w, h, c, 3)
    with pytest.raises(frave_amd.api.FriHipError):  # no order yet
        P.encode_image_symbols(img)
    order = emit.stream_order(P.centers(), P.valid_mask())
    bad = order.copy()
    bad[0] = bad[1]  # a node twice, another never
    with pytest.raises(frave_amd.api.FriHipError):
        P.set_stream_order(bad)
    with pytest.raises(frave_amd.api.FriHipError):
        P.set_stream_order(order[:-1])
    P.set_stream_order(order)
    d_px = torch.from_numpy(np.stack([img, img]).reshape(-1)).cuda()
    d_co = torch.empty((2, c, plane), dtype=torch.int32, device="cuda")
    d_w = torch.empty((2, c, plane), dtype=torch.uint16, device="cuda")
    d_st = torch.empty((2, c, n), dtype=torch.uint16, device="cuda")
    d_h = torch.empty((2, c, 10, 1024), dtype=torch.int32, device="cuda")
    d_o = torch.empty((2, c), dtype=torch.int64, device="cuda")
    d_par = torch.zeros((2, c, 2, 3, 6), dtype=torch.float32, device="cuda")
    args = lambda **k: dict(dict(n_images=2, pixel_stride=P.pixel_bytes, coef_stride=c * plane, word_stride=c * plane, symbol_stride=c * n), **k)
    def call(**k):
        a = args(**k)
        P.encode_symbols_batch_dev(a["n_images"], d_px.data_ptr(), a["pixel_stride"], None, True, d_par.data_ptr(), d_co.data_ptr(), a["coef_stride"], d_w.data_ptr(), a["word_stride"],
                                   d_st.data_ptr(), a["symbol_stride"], d_h.data_ptr(), d_o.data_ptr())
    for wrong in (dict(n_images=0), dict(symbol_stride=c * n - 1), dict(word_stride=c * plane - 512), dict(pixel_stride=P.pixel_bytes - 1), dict(coef_stride=c * plane + 512)):
        with pytest.raises(frave_amd.api.FriHipError):
            call(**wrong)
    call()  # and the plan still works
    torch.cuda.synchronize()
    sym, vp, wp, hist, oob = P.encode_image_symbols(img)
    assert np.array_equal(d_st[0].cpu().numpy(), sym) and np.array_equal(d_st[1].cpu().numpy(), sym) and int(oob.sum()) == 0
    P.close()


@pytest.mark.gpu
def test_emit_from_device_arrays():
    import frave_amd

    w, h, c = 512, 384, 3
    img = _mixed_image(w, h, c, 9)
    ctx = frave_amd.Context(0)
    P = frave_amd.Plan(ctx, w, h, c)
    co = P.transform_quant(img)
    bs, ps, hs, vps, wps = [], [], [], [], []
    for ch in range(c):
        vp, wp = KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS
        b, p, hist, oob = P.predict_histogram(co, ch, vp, wp)
        assert oob == 0
        bs.append(b), ps.append(p), hs.append(hist)
        vps.append(np.asarray(vp, np.float32).reshape(3, 6)), wps.append(np.asarray(wp, np.float32).reshape(3, 6))
    frv = emit.encode_image(w, h, P.centers(), co, np.stack(bs), np.stack(ps), np.stack(hs), np.stack(vps), np.stack(wps))
    emit.check_image(frv, P.centers(), co, np.stack(bs), np.stack(ps))
    assert 0 < len(frv) < w * h * c * 2


@pytest.mark.gpu
def test_driver_encodes_pnm_files(tmp_path):
    """fri_driver encode-file: PGM / PPM in, .frv out (device stages + host emit + stream self-check in one process)"""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    driver = os.path.join(root, "frave_amd", "host", "fri_driver")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "frave_amd", "host")])  # relinks if anything it is made of changed
    for c, magic in ((1, b"P5"), (3, b"P6")):
        w, h = 320, 200
        img = _mixed_image(w, h, c, 11)
        src, dst = tmp_path / f"in{c}.pnm", tmp_path / f"out{c}.frv"
        src.write_bytes(magic + b"\n# a comment\n%d %d\n255\n" % (w, h) + img.tobytes())
        out = subprocess.run([driver, "encode-file", str(src), str(dst)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert "lossless" in out.stdout
        frv = dst.read_bytes()
        assert frv[:4] == b"frif" and struct.unpack("<II", frv[4:12]) == (h, w) and frv[-2:] == b"\xff\xdf"
        # and back: decode-file = container -> host entropy decoding -> device dequantisation + inverse transform -> PNM
        back = tmp_path / f"back{c}.pnm"
        out = subprocess.run([driver, "decode-file", str(dst), str(back)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert back.read_bytes() == magic + b"\n%d %d\n255\n" % (w, h) + img.tobytes()
        if c == 3:  # 24-bit BMP in and out (rows bottom-up, BGR, padded to 4 bytes - 322 * 3 = 966 -> 968)
            bw, bh = 322, 150
            bimg = _mixed_image(bw, bh, 3, 12)
            stride = (bw * 3 + 3) & ~3
            rows = np.zeros((bh, stride), np.uint8)
            rows[:, : bw * 3] = bimg.reshape(bh, bw, 3)[::-1, :, ::-1].reshape(bh, bw * 3)
            head = b"BM" + struct.pack("<IHHI", 54 + stride * bh, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, bw, bh, 1, 24, 0, stride * bh, 0, 0, 0, 0)
            bmp, frv2, bmp2, ppm2 = tmp_path / "in.bmp", tmp_path / "bmp.frv", tmp_path / "back.bmp", tmp_path / "back_bmp.ppm"
            bmp.write_bytes(head + rows.tobytes())
            out = subprocess.run([driver, "encode-file", str(bmp), str(frv2)], capture_output=True, text=True, timeout=120)
            assert out.returncode == 0, out.stderr
            out = subprocess.run([driver, "decode-file", str(frv2), str(bmp2)], capture_output=True, text=True, timeout=120)
            assert out.returncode == 0, out.stderr
            assert bmp2.read_bytes() == bmp.read_bytes()
            out = subprocess.run([driver, "decode-file", str(frv2), str(ppm2)], capture_output=True, text=True, timeout=120)
            assert out.returncode == 0 and ppm2.read_bytes() == b"P6\n%d %d\n255\n" % (bw, bh) + bimg.tobytes()
        # the same stream through the Python binding: decoded planes are the ones the device produced
        import frave_amd

        dw, dh, dc, centers, coefs = emit.decode_image(frv)
        P = frave_amd.Plan(frave_amd.Context(0), w, h, c)
        assert (dw, dh, dc) == (w, h, c) and np.array_equal(centers, P.centers())
        assert np.array_equal(coefs.reshape(-1), np.asarray(P.transform_quant(img)).reshape(-1))
        assert np.array_equal(np.asarray(P.inverse_transform(coefs)).reshape(-1), img)
