"""One measurement pass for tools/ab_lib.py: prints `AB k1 k2guard chain k2 k4v k4w k3 chain_fit` in microseconds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

SIZE = int(os.environ.get("AB_SIZE", "4096"))
C = int(os.environ.get("AB_C", "1"))
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, SIZE, SIZE, C)
F = plan.num_cells
d_px = torch.randint(0, 256, (plan.pixel_bytes,), dtype=torch.uint8, device="cuda")
d_co = torch.empty(plan.coef_count, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
wp = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (3, 1))
d_b = torch.empty(C * F * 512, dtype=torch.uint8, device="cuda")
d_p = torch.empty(C * F * 512, dtype=torch.int32, device="cuda")
d_h = torch.empty(C * 10 * 1024, dtype=torch.int32, device="cuda")
d_o = torch.empty(C, dtype=torch.int64, device="cuda")
d_back = torch.empty(plan.pixel_bytes, dtype=torch.uint8, device="cuda")
d_gi = torch.empty(3 * 28, dtype=torch.int64, device="cuda")
d_gd = torch.empty(18, dtype=torch.float64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(300):  # spin-up: steady clocks
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


vp3, wp3 = np.tile(vp, (C, 1, 1)).copy(), np.tile(wp, (C, 1, 1)).copy()
k1 = timed(lambda: plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s))
k2g = timed(lambda: plan.predict_histogram_dev(d_co.data_ptr(), 0, vp, wp, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=s))
chain = timed(lambda: plan.encode_image_dev(d_px.data_ptr(), d_co.data_ptr(), d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), vp3, wp3, fit=False, stream=s))
k4v = timed(lambda: plan.fit_value_sums_dev(d_co.data_ptr(), 0, d_gi.data_ptr(), stream=s))
k4w = timed(lambda: plan.fit_width_sums_dev(d_co.data_ptr(), 0, vp, d_gi.data_ptr(), d_gd.data_ptr(), stream=s))
k3 = timed(lambda: plan.inverse_transform_dev(d_co.data_ptr(), d_back.data_ptr(), stream=s))
vpf, wpf = np.zeros((C, 3, 6), np.float32), np.zeros((C, 3, 6), np.float32)
fitc = timed(lambda: plan.encode_image_dev(d_px.data_ptr(), d_co.data_ptr(), d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), vpf, wpf, fit=True, stream=s), reps=15)
assert int(d_h.sum()) == plan.num_some * C
print(f"AB {k1:.2f} {k2g:.2f} {chain:.2f} {chain - k1:.2f} {k4v:.2f} {k4w:.2f} {k3:.2f} {fitc:.2f}")
