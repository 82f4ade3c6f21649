"""fri_hip_plan_tune_forward (round 5): the plan measures candidate cuts of the cell lattice into tiles / shares and keeps the fastest. Any partition gives the same
coefficients (Fractal::extract_coefficients is per-cell independent, wavelet_transform.rs:179-225): the tuned plan against the oracle, the remembered winner, the
multi-stream launch loop, and the calls that must refuse."""
import numpy as np
import pytest

from tests.common import gen_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import frave_amd as fa

    c = fa.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("shape", [(1920, 1080, 1), (1024, 768, 3), (333, 777, 1)])
def test_tuned_plan_gives_the_oracles_coefficients(ctx, oracle, shape):
    import frave_amd as fa

    w, h, c = shape
    img = gen_image("noise", w, h, c, 31)
    img[: h // 3] = gen_image("smooth", w, h // 3, c, 32)
    want = oracle.Wavelet(img, h, w, c).coefficients()
    P = fa.Plan(ctx, w, h, c)
    before = P.tiling()
    rep = P.tune_forward(24)
    assert rep["tuned"] is True and rep["winner"] in rep["candidates_us"] and len(rep["candidates_us"]) >= 4
    assert all(v > 0 for v in rep["candidates_us"].values())
    assert np.array_equal(P.transform_quant(img), want)  # single launch on the winner
    q = np.ones(32, np.int32)
    q[:10] = [1, 2, 3, 1, 2, 1, 4, 1, 2, 3]
    W = oracle.Wavelet(img, h, w, c)
    W.quantize(q)
    assert np.array_equal(P.transform_quant(img, q), W.coefficients())
    flipped = img[::-1].copy()
    batch = P.transform_quant_batch([img] * 4 + [flipped] + [img] * 4, None)  # nine images in one launch: the merged batch shares of the winner
    assert np.array_equal(batch[0], want) and np.array_equal(batch[8], want)
    assert np.array_equal(batch[4], oracle.Wavelet(flipped, h, w, c).coefficients())
    assert np.array_equal(P.inverse_transform(want), img.reshape(-1))  # the inverse kernel walks its own tiling: untouched
    # a second plan of the shape starts from the remembered winner without measuring
    P2 = fa.Plan(ctx, w, h, c)
    assert P2.tiling() == P.tiling()
    assert np.array_equal(P2.transform_quant(img), want)
    # tuning again is allowed and changes nothing observable
    rep2 = P.tune_forward(16)
    assert rep2["tuned"] is True
    assert np.array_equal(P.transform_quant(img), want)
    print(shape, before, "->", P.tiling(), rep["winner"])
    P.close(), P2.close()


def test_every_candidate_tiling_gives_the_same_coefficients(ctx, oracle):
    """The candidates the tuner chooses from, pinned one by one through the tuning knobs: all of them are the oracle's transform."""
    import os

    import frave_amd as fa

    w, h, c = 1280, 720, 1
    img = gen_image("noise", w, h, c, 5)
    want = oracle.Wavelet(img, h, w, c).coefficients()
    keys = ("FRI_HIP_TUNING", "FRI_HIP_STRIDED_SHARES", "FRI_HIP_BAND_ROWS", "FRI_HIP_CELLS_PER_TILE", "FRI_HIP_RANK_WEIGHTS")
    saved = {k: os.environ.get(k) for k in keys}
    try:
        os.environ["FRI_HIP_TUNING"] = "1"
        for spec in ({"FRI_HIP_STRIDED_SHARES": "0", "FRI_HIP_BAND_ROWS": "72"}, {"FRI_HIP_STRIDED_SHARES": "0", "FRI_HIP_BAND_ROWS": "80"},
                     {"FRI_HIP_STRIDED_SHARES": "1", "FRI_HIP_BAND_ROWS": "16", "FRI_HIP_CELLS_PER_TILE": "9"}, {"FRI_HIP_STRIDED_SHARES": "0", "FRI_HIP_BAND_ROWS": "8"},
                     {"FRI_HIP_STRIDED_SHARES": "1", "FRI_HIP_BAND_ROWS": "24", "FRI_HIP_RANK_WEIGHTS": "1.4,1.15,0.85,0.6"}):
            for k in keys[1:]:
                os.environ.pop(k, None)
            os.environ.update(spec)
            P = fa.Plan(ctx, w, h, c)
            assert np.array_equal(P.transform_quant(img), want), spec
            assert P.tune_forward()["tuned"] is False  # a pinned tiling is left alone
            P.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_multi_stream_launch_loop_and_refusals(ctx):
    import torch

    import frave_amd as fa

    w, h, c = 1024, 1024, 1
    P = fa.Plan(ctx, w, h, c)
    slots = 6
    d_px = torch.randint(0, 256, (slots, P.pixel_bytes), dtype=torch.uint8, device="cuda")
    d_co = torch.zeros((slots, P.coef_count), dtype=torch.int32, device="cuda")
    ref = torch.empty(P.coef_count, dtype=torch.int32, device="cuda")
    one = P.time_transform_quant_dev(slots, d_px.data_ptr(), P.pixel_bytes, d_co.data_ptr(), P.coef_count, 30)
    d_co.zero_()
    two = P.time_transform_quant_streams_dev(slots, d_px.data_ptr(), P.pixel_bytes, d_co.data_ptr(), P.coef_count, 30, 2)
    assert one > 0 and two > 0
    for k in range(slots):  # every slot was transformed by the two-stream loop exactly as a plain launch does it
        P.transform_quant_dev(d_px[k].data_ptr(), ref.data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(d_co[k], ref)
    with pytest.raises(fa.FriHipError):
        P.time_transform_quant_streams_dev(slots, d_px.data_ptr(), P.pixel_bytes, d_co.data_ptr(), P.coef_count, 30, 9)
    host_only = fa.Plan(None, w, h, c)
    with pytest.raises(fa.FriHipError) as e:
        host_only.tune_forward()
    assert e.value.code == -3  # FRI_HIP_ERR_NO_DEVICE
    P.close()
