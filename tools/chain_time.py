"""Time the device-resident encode chain at 4096x4096xC with and without the fit, through the host-parameter entry point (fri_hip_encode_image_dev: wall
clock) and the asynchronous one (fri_hip_encode_image_batch_dev: HIP events). GPU only. Under `rocprofv3 --kernel-trace` its trace feeds tools/chain_gaps.py."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

C = int(os.environ.get("SWEEP_C", "1"))
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, C)
F = plan.num_cells
d_px = torch.randint(0, 256, (plan.pixel_bytes,), dtype=torch.uint8, device="cuda")
d_co = torch.empty(plan.coef_count, dtype=torch.int32, device="cuda")
d_b = torch.empty(C * F * 512, dtype=torch.uint8, device="cuda")
d_p = torch.empty(C * F * 512, dtype=torch.int32, device="cuda")
d_h = torch.empty(C * 10 * 1024, dtype=torch.int32, device="cuda")
d_o = torch.empty(C, dtype=torch.int64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (C, 3, 1))
wp = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (C, 3, 1))
plan.time_transform_quant_dev(1, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 2000, stream=s)  # clocks up
for fit in (False, True):
    vpf, wpf = vp.copy(), wp.copy()
    call = lambda: plan.encode_image_dev(d_px.data_ptr(), d_co.data_ptr(), d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), vpf, wpf, fit=fit, stream=s)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 50
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    print(f"4096x4096x{C} encode chain, fit={fit}: {us:8.1f} us per image (wall clock, {reps} images back to back)")

# the asynchronous form (fri_hip_encode_image_batch_dev: parameters stay in device memory, nothing but enqueues), timed with events on the stream
d_par = torch.zeros((C, 2, 3, 6), dtype=torch.float32, device="cuda")
d_par[:, 0] = torch.from_numpy(vp).cuda()
d_par[:, 1] = torch.from_numpy(wp).cuda()
d_r = torch.zeros(C, dtype=torch.int64, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for fit in (False, True):
    call = lambda: plan.encode_image_batch_dev(1, d_px.data_ptr(), plan.pixel_bytes, d_par.data_ptr(), d_co.data_ptr(), C * F * 512, d_b.data_ptr(), d_p.data_ptr(), C * F * 512,
                                               d_h.data_ptr(), d_o.data_ptr(), fit=fit, d_fit_out_of_range=d_r.data_ptr(), stream=s)
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(50):
        call()
    ev1.record()
    torch.cuda.synchronize()
    print(f"4096x4096x{C} asynchronous chain, fit={fit}: {ev0.elapsed_time(ev1) / 50 * 1e3:8.1f} us per image (events, 50 images back to back)")
