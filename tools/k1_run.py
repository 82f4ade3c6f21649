"""Minimal K1 driver for rocprofv3: N event-timed launches over rotating slots."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
C = int(os.environ.get("SWEEP_C", "1"))
ctx = frave_amd.Context(0)
SIZE = int(os.environ.get("K1_SIZE", "4096"))
plan = frave_amd.Plan(ctx, SIZE, SIZE, C)
slots = int(os.environ.get("K1_SLOTS", "8" if SIZE <= 4096 else "1"))
d_px = torch.randint(0, 256, (slots, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((slots, plan.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
if os.environ.get("K1_TUNE") == "1":  # the plan measures its forward tiling like bench.py does (without it: fri_hip_plan_create's default, or what the FRI_HIP_* knobs pin)
    print("tune:", plan.tune_forward())
nb = int(os.environ.get("K1_BATCH", "0"))
if nb > 1:  # the batch form: nb distinct images per launch (grid.y = nb), n launches
    nb = min(nb, slots)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    go = lambda: plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=nb, pixel_stride=plan.pixel_bytes, coef_stride=plan.coef_count)
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n):
        go()
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / n
    print(f"K1 batch {SIZE}x{SIZE}x{C}: {us:.2f} us/launch = {us / nb:.2f} us/image over {n} launches of {nb} distinct images, {plan.tiling()['n_wg']} shares")
    sys.exit(0)
spin = int(os.environ.get("K1_SPIN_UP", "3000"))  # untimed: a fresh GPU needs tens of ms of work to reach steady clocks (bench.py does the same)
if spin:
    plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, spin, stream=s)
us = plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, n, stream=s)
print(f"K1 {SIZE}x{SIZE}x{C}: {us:.2f} us/launch over {n} launches, {slots} rotating slot(s), {plan.tiling()['n_wg']} shares")
