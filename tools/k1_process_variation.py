"""K1's launch period over successive bursts of 20 launches inside one process (after the spin-up): how much of the run-to-run spread of a
short bench run is the process (its buffers' placement) and how much the moment. GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
slots = 8
d_px = torch.randint(0, 256, (slots, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((slots, plan.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 4000, stream=s)
out = []
for rep in range(12):
    torch.cuda.synchronize()
    out.append(plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 20, stream=s))
print("bursts of 20:", " ".join(f"{u:.2f}" for u in out), f"| 400: {plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 400, stream=s):.2f}")
