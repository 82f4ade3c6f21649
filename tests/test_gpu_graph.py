"""HIP-graph capture: the forward kernel alone can be captured and replayed (its launch carries no host-side state); every entry point that
launches the scan (predict + histogram) refuses a capturing stream - its histogram hand-over numbers launches on the host, a replay would reuse a number."""
import ctypes as C

import numpy as np
import pytest

from tests.common import KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, gen_image

pytestmark = pytest.mark.gpu
RELAXED = 2  # hipStreamCaptureModeRelaxed


@pytest.fixture(scope="module")
def ctx():
    import frave_amd as fa

    c = fa.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def hip():
    return C.CDLL("libamdhip64.so")


def test_forward_kernel_replays_from_a_graph(ctx, oracle, hip):
    import torch

    import frave_amd as fa

    w, h, c = 640, 360, 1
    P = fa.Plan(ctx, w, h, c)
    imgs = [gen_image("noise", w, h, c, 40 + k) for k in range(3)]
    d_px = torch.from_numpy(imgs[0].reshape(-1).copy()).cuda()
    d_co = torch.empty(P.coef_count, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)
    torch.cuda.synchronize()
    assert hip.hipStreamBeginCapture(sp, RELAXED) == 0
    P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s.cuda_stream)
    graph, ex = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamEndCapture(sp, C.byref(graph)) == 0 and graph.value
    assert hip.hipGraphInstantiate(C.byref(ex), graph, None, None, 0) == 0
    for img in imgs:  # every replay transforms what the pixel buffer holds NOW
        d_px.copy_(torch.from_numpy(img.reshape(-1).copy()))
        d_co.fill_(7)
        torch.cuda.synchronize()
        assert hip.hipGraphLaunch(ex, sp) == 0
        s.synchronize()
        assert np.array_equal(d_co.cpu().numpy().reshape(c, -1, 512), oracle.Wavelet(img, h, w, c).coefficients())
    hip.hipGraphExecDestroy(ex)
    hip.hipGraphDestroy(graph)
    P.close()


def test_the_scan_refuses_a_capturing_stream(ctx, hip):
    import torch

    import frave_amd as fa

    w, h, c = 320, 200, 1
    P = fa.Plan(ctx, w, h, c)
    F = P.num_cells
    d_px = torch.from_numpy(gen_image("smooth", w, h, c, 3).reshape(-1).copy()).cuda()
    d_co = torch.empty(P.coef_count, dtype=torch.int32, device="cuda")
    d_b = torch.empty(F * 512, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(F * 512, dtype=torch.int32, device="cuda")
    d_h = torch.empty(10 * 1024, dtype=torch.int32, device="cuda")
    d_o = torch.empty(1, dtype=torch.int64, device="cuda")
    s = torch.cuda.Stream()
    sp = C.c_void_p(s.cuda_stream)
    # once outside any capture: scratch and accumulators exist, the result is the reference for what follows
    P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s.cuda_stream)
    P.predict_histogram_dev(d_co.data_ptr(), 0, KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    want = d_h.clone()
    assert int(want.sum()) + int(d_o) == P.num_some
    assert hip.hipStreamBeginCapture(sp, RELAXED) == 0
    try:
        with pytest.raises(fa.FriHipError) as e:
            P.predict_histogram_dev(d_co.data_ptr(), 0, KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=s.cuda_stream)
        assert e.value.code == -1 and "graph" in str(e.value)  # FRI_HIP_ERR_INVALID_ARGUMENT
    finally:
        graph = C.c_void_p()
        hip.hipStreamEndCapture(sp, C.byref(graph))
        if graph.value:
            hip.hipGraphDestroy(graph)
    # the plan is unharmed: the next ordinary launch counts exactly what the first did
    d_h.fill_(99)
    P.predict_histogram_dev(d_co.data_ptr(), 0, KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    assert torch.equal(d_h, want)
    P.close()
