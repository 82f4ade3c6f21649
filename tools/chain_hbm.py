"""The device-resident encode chain at 4096x4096x1 when nothing is left in the caches from the image before: rotating over SLOTS (default 24) pixel /
coefficient slots like bench.py. Prints K1 alone, K1 -> K2 (given parameters), K1 -> fit -> K2 and K1 -> K2 (halfwords) -> K5 in microseconds per image
(HIP events around one pass over the slots). For A/B runs of plan-level knobs: set them in the environment (FRI_HIP_TUNING=1 is set here)."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

slots = int(os.environ.get("SLOTS", "24"))
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
F, plane = plan.num_cells, plan.num_cells * 512
d_px = torch.randint(0, 256, (slots, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((slots, plan.coef_count), dtype=torch.int32, device="cuda")
d_b = torch.empty(plane, dtype=torch.uint8, device="cuda")
d_p = torch.empty(plane, dtype=torch.int32, device="cuda")
d_h = torch.empty(10 * 1024, dtype=torch.int32, device="cuda")
d_o = torch.empty(1, dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
wp = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (3, 1))
d_par = torch.from_numpy(np.stack([vp, wp]).reshape(-1)).cuda()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
px = lambda k: d_px[k].data_ptr()
co = lambda k: d_co[k].data_ptr()
plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 3000, stream=s)


def timed(fn, passes=2):
    fn(slots - 1)
    torch.cuda.synchronize()
    ev0.record()
    for i in range(passes * slots):
        fn(i % slots)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / (passes * slots) * 1e3


k1 = timed(lambda k: plan.transform_quant_dev(px(k), co(k), stream=s))
given = timed(lambda k: plan.encode_image_batch_dev(1, px(k), plan.pixel_bytes, d_par.data_ptr(), co(k), plane, d_b.data_ptr(), d_p.data_ptr(), plane, d_h.data_ptr(), d_o.data_ptr(),
                                                    fit=False, stream=s))
assert int(d_h.sum()) + int(d_o) == plan.num_some
fit = timed(lambda k: plan.encode_image_batch_dev(1, px(k), plan.pixel_bytes, d_par.data_ptr(), co(k), plane, d_b.data_ptr(), d_p.data_ptr(), plane, d_h.data_ptr(), d_o.data_ptr(),
                                                  fit=True, stream=s))
assert int(d_h.sum()) + int(d_o) == plan.num_some
plan.set_stream_order()
n = plan.num_some
d_w = torch.empty(plane, dtype=torch.uint16, device="cuda")
d_s = torch.empty(n, dtype=torch.uint16, device="cuda")
d_par.copy_(torch.from_numpy(np.stack([vp, wp]).reshape(-1)))
sym = timed(lambda k: plan.encode_symbols_batch_dev(1, px(k), plan.pixel_bytes, None, False, d_par.data_ptr(), co(k), plane, d_w.data_ptr(), plane, d_s.data_ptr(), n, d_h.data_ptr(),
                                                    d_o.data_ptr(), stream=s))
print(f"CHAIN slots={slots}: K1 {k1:.2f}  K1->K2 {given:.2f} (K2 {given - k1:.2f})  K1->fit->K2 {fit:.2f} (fit {fit - given:.2f})  K1->K2w->K5 {sym:.2f} us per image")
