"""Per-image K1 time when n images go in one launch (grid.y = n): separates steady-state throughput from ramp/tail."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

C = int(os.environ.get("SWEEP_C", "1"))
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, C)
print(plan.tiling())
nmax = 32
d_px = torch.randint(0, 256, (nmax, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((nmax, plan.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
alg = plan.pixel_bytes + plan.coef_count * 4
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]:
    reps = max(4, 64 // n)
    for _ in range(2):
        plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=n, pixel_stride=plan.pixel_bytes, coef_stride=plan.coef_count)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=n, pixel_stride=plan.pixel_bytes, coef_stride=plan.coef_count)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / reps
    print(f"n_images={n:3d}: {us:9.2f} us/launch  {us / n:8.2f} us/image  {alg * n / us / 1e3:8.1f} GB/s", flush=True)
