"""K1 over a batch of 1920x1080 frames in one launch (BASELINE config 3). GPU only. usage: k1_batch_frames.py [C] [n_frames]"""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 1920, 1080, C)
d_px = torch.randint(0, 256, (n, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((n, plan.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2):
    plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=n, pixel_stride=plan.pixel_bytes, coef_stride=plan.coef_count)
torch.cuda.synchronize()
ev0.record()
reps = 10
for _ in range(reps):
    plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=n, pixel_stride=plan.pixel_bytes, coef_stride=plan.coef_count)
ev1.record()
torch.cuda.synchronize()
us = ev0.elapsed_time(ev1) / reps * 1e3
alg = (plan.pixel_bytes + plan.coef_count * 4) * n
print(f"{n} x 1920x1080x{C}: {us:9.1f} us per launch, {us / n:6.2f} us per frame, {1920 * 1080 * n / us:10.1f} Mpix/s, {alg / us / 1e3:7.1f} GB/s algorithmic ({alg / us / 1e3 / 80:.1f} % of 8 TB/s)  tiling {plan.tiling()}")
