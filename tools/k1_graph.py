"""K1 at 4096^2: plain back-to-back launches vs the same launches replayed from a HIP graph (rotating slots). GPU only."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, size, size, 1)
slots = 8
d_px = torch.randint(0, 256, (slots, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((slots, plan.coef_count), dtype=torch.int32, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def launch_all(stream):
    for k in range(slots):
        plan.transform_quant_dev(d_px[k].data_ptr(), d_co[k].data_ptr(), stream=stream)


s = torch.cuda.current_stream().cuda_stream
plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 3000, stream=s)  # spin-up
for _ in range(5):
    launch_all(s)
torch.cuda.synchronize()
ev0.record()
for _ in range(50):
    launch_all(s)
ev1.record()
torch.cuda.synchronize()
print(f"{size}^2 plain launches : {ev0.elapsed_time(ev1) / (50 * slots) * 1e3:.2f} us per launch")
print(f"{size}^2 native loop    : {plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 400, stream=s):.2f} us per launch")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    launch_all(torch.cuda.current_stream().cuda_stream)
g.replay()
torch.cuda.synchronize()
ev0.record()
for _ in range(50):
    g.replay()
ev1.record()
torch.cuda.synchronize()
print(f"{size}^2 graph of {slots}     : {ev0.elapsed_time(ev1) / (50 * slots) * 1e3:.2f} us per launch")
