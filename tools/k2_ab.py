"""K2 figures for an A/B of two builds (tools/ab_lib.py-style, one process per library): prints `K2AB promised generic chain chain_minus_k1 words_chain`
in microseconds (HIP events around 30 back-to-back calls each; 4096x4096x1, one buffer set: the plane comes from the Infinity Cache in all of them)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
F = plan.num_cells
d_px = torch.randint(0, 256, (plan.pixel_bytes,), dtype=torch.uint8, device="cuda")
d_co = torch.empty(plan.coef_count, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
wp = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (3, 1))
d_b = torch.empty(F * 512, dtype=torch.uint8, device="cuda")
d_p = torch.empty(F * 512, dtype=torch.int32, device="cuda")
d_h = torch.empty(10 * 1024, dtype=torch.int32, device="cuda")
d_o = torch.empty(1, dtype=torch.int64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(1500):
    plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s)


def timed(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e3


k1 = timed(lambda: plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s))
generic = timed(lambda: plan.predict_histogram_dev(d_co.data_ptr(), 0, vp, wp, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=s))
assert int(d_h.sum()) + int(d_o) == plan.num_some
plan.assume_forward_coefficients(True)
promised = timed(lambda: plan.predict_histogram_dev(d_co.data_ptr(), 0, vp, wp, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=s))
assert int(d_h.sum()) + int(d_o) == plan.num_some
plan.assume_forward_coefficients(False)
d_par = torch.from_numpy(np.stack([vp, wp]).reshape(-1)).cuda()
chain = timed(lambda: plan.encode_image_batch_dev(1, d_px.data_ptr(), plan.pixel_bytes, d_par.data_ptr(), d_co.data_ptr(), F * 512, d_b.data_ptr(), d_p.data_ptr(), F * 512,
                                                  d_h.data_ptr(), d_o.data_ptr(), fit=False, stream=s))
assert int(d_h.sum()) + int(d_o) == plan.num_some
print(f"K2AB {promised:.2f} {generic:.2f} {chain:.2f} {chain - k1:.2f} {k1:.2f}")
