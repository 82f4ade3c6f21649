"""The symbol-stream chains' compact coefficient planes (round 5): when the caller does not ask for the coefficients (fri_hip_encode_symbols_batch_dev with
d_coefs = NULL; fri_hip_encode_image_symbols always) they travel between the forward kernel, the fit and the scan as int16 with None as 0. Everything the caller
does get - streams, histograms, out-of-alphabet counts, fitted parameters, range counts, the node words of Some nodes - must be the same bits as with int32 planes."""
import numpy as np
import pytest

from tests.common import gen_image, random_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import frave_amd as fa

    c = fa.Context(0)
    yield c
    c.close()


def _run(P, torch, imgs, fit, params, compact, q=None, direct=False):
    n_img, c = len(imgs), P.channels
    plane, n = P.num_cells * 512, P.num_some
    d_px = torch.from_numpy(np.stack([im.reshape(-1) for im in imgs])).cuda()
    d_co = None if compact else torch.full((n_img, c, plane), 7, dtype=torch.int32, device="cuda")
    d_w = torch.full((n_img, c, plane), 0xEEEE, dtype=torch.uint16, device="cuda")
    d_st = torch.full((n_img * c * n + 8,), 0xFFFF, dtype=torch.uint16, device="cuda")
    d_h = torch.full((n_img, c, 10, 1024), -1, dtype=torch.int32, device="cuda")
    d_o = torch.full((n_img, c), -1, dtype=torch.int64, device="cuda")
    d_r = torch.full((n_img, c), -1, dtype=torch.int64, device="cuda")
    d_par = torch.from_numpy(np.broadcast_to(params, (n_img, c, 2, 3, 6)).astype(np.float32).copy()).cuda()
    s = torch.cuda.current_stream().cuda_stream
    P.encode_symbols_batch_dev(n_img, d_px.data_ptr(), P.pixel_bytes, q, fit, d_par.data_ptr(), 0 if compact else d_co.data_ptr(), c * plane, 0 if direct else d_w.data_ptr(), c * plane,
                               d_st.data_ptr(), c * n, d_h.data_ptr(), d_o.data_ptr(), d_r.data_ptr() if fit else None, stream=s)
    torch.cuda.synchronize()
    st = d_st.cpu().numpy()
    assert (st[n_img * c * n:] == 0xFFFF).all()
    return (st[: n_img * c * n].reshape(n_img, c, n), d_h.cpu().numpy(), d_o.cpu().numpy(), d_par.cpu().numpy(), d_r.cpu().numpy() if fit else None,
            d_w.cpu().numpy().reshape(n_img, c, plane), None if compact else d_co.cpu().numpy())


@pytest.mark.parametrize("shape", [(46, 46, 1), (129, 65, 1), (640, 360, 3), (1000, 777, 1), (1920, 1080, 3), (4096, 4096, 1)])
def test_compact_planes_change_nothing_the_caller_sees(ctx, shape):
    import torch

    import frave_amd as fa

    w, h, c = shape
    P = fa.Plan(ctx, w, h, c)
    order = P.set_stream_order()
    vp, wp = random_params(11)
    params = np.stack([np.asarray(vp, np.float32).reshape(3, 6), np.asarray(wp, np.float32).reshape(3, 6)])
    kinds = ["noise", "smooth"] if w * h < 4e6 else ["noise"]
    for n_img in (1, 2) if w * h < 4e6 else (1,):
        imgs = [gen_image(kinds[k % len(kinds)], w, h, c, 70 + k) for k in range(n_img)]
        for fit in (False, True):
            for q in (None, np.array([3, 2, 2, 1, 1, 1, 1, 1, 1, 1] + [1] * 22, np.int32)):
                ref = _run(P, torch, imgs, fit, params, compact=False, q=q)
                got = _run(P, torch, imgs, fit, params, compact=True, q=q)
                assert np.array_equal(got[0], ref[0]), "streams"
                assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), "histograms / out-of-alphabet counts"
                assert np.array_equal(got[3].view(np.uint32), ref[3].view(np.uint32)), "parameters"
                if fit:
                    assert np.array_equal(got[4], ref[4]) and not ref[4].any()
                for k in range(n_img):
                    for ch in range(c):
                        assert np.array_equal(got[5][k, ch][order], ref[5][k, ch][order])  # the node words of the Some nodes
                # ... and without the node words either (d_node_words = NULL): the scan writes the streams itself, no gather kernel
                dr = _run(P, torch, imgs, fit, params, compact=True, q=q, direct=True)
                assert np.array_equal(dr[0], ref[0]), "streams written by the scan"
                assert np.array_equal(dr[1], ref[1]) and np.array_equal(dr[2], ref[2]) and np.array_equal(dr[3].view(np.uint32), ref[3].view(np.uint32))
                assert (dr[5] == 0xEEEE).all()  # nobody touched the node-word buffer
    P.close()


def test_compact_planes_grow_with_the_batch(ctx):
    """The plan's compact planes are sized by the largest call so far; a larger batch behind a smaller one finds room."""
    import torch

    import frave_amd as fa

    w, h, c = 320, 200, 3
    P = fa.Plan(ctx, w, h, c)
    P.set_stream_order()
    vp, wp = random_params(5)
    params = np.stack([np.asarray(vp, np.float32).reshape(3, 6), np.asarray(wp, np.float32).reshape(3, 6)])
    for n_img in (1, 5, 2, 9):
        imgs = [gen_image("noise", w, h, c, 90 + k) for k in range(n_img)]
        ref = _run(P, torch, imgs, True, params, compact=False)
        got = _run(P, torch, imgs, True, params, compact=True)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[3].view(np.uint32), ref[3].view(np.uint32))
    P.close()


def test_compact_chains_on_two_streams_take_turns(ctx):
    """The planes belong to one chain at a time: a chain on another stream waits for the previous one's event. Eight chains alternating over two streams without a host
    synchronise in between, each with its own outputs, give what eight ordered chains give."""
    import torch

    import frave_amd as fa

    w, h, c = 1000, 700, 1
    P = fa.Plan(ctx, w, h, c)
    P.set_stream_order()
    plane, n = P.num_cells * 512, P.num_some
    vp, wp = random_params(3)
    params = np.stack([np.asarray(vp, np.float32).reshape(3, 6), np.asarray(wp, np.float32).reshape(3, 6)])
    imgs = [gen_image("noise" if k % 2 else "smooth", w, h, c, 200 + k) for k in range(8)]
    want = [_run(P, torch, [im], True, params, compact=False)[0] for im in imgs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = []
    for k, im in enumerate(imgs):
        d_px = torch.from_numpy(im.reshape(-1).copy()).cuda()
        bufs.append(dict(px=d_px, w=torch.empty(plane, dtype=torch.uint16, device="cuda"), st=torch.full((n + 8,), 0xFFFF, dtype=torch.uint16, device="cuda"),
                         h=torch.empty((10, 1024), dtype=torch.int32, device="cuda"), o=torch.empty(1, dtype=torch.int64, device="cuda"), r=torch.empty(1, dtype=torch.int64, device="cuda"),
                         par=torch.from_numpy(params.reshape(-1).copy()).cuda()))
    torch.cuda.synchronize()
    for k, b in enumerate(bufs):
        s = streams[k % 2]
        P.encode_symbols_batch_dev(1, b["px"].data_ptr(), P.pixel_bytes, None, True, b["par"].data_ptr(), 0, plane, b["w"].data_ptr(), plane, b["st"].data_ptr(), n,
                                   b["h"].data_ptr(), b["o"].data_ptr(), b["r"].data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
    for k, b in enumerate(bufs):
        assert np.array_equal(b["st"].cpu().numpy()[:n], want[k][0, 0]), k
    P.close()
