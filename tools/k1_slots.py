"""Is K1's single-image launch period an HBM figure? The bench loop (`fri_hip_time_transform_quant_dev`) over 8, 16, 24, 40 and 64 rotating
slots, interleaved rounds inside one process on one box. 8 slots = 134 MB of pixels (fits the 256 MiB Infinity Cache), 24 = 403 MB, 40 = 671 MB;
with the coefficients a slot is 85 MB. If the period does not grow with the slot count, the pixels of the 8-slot loop do not come from the cache.

usage: python3 tools/k1_slots.py [launches per measurement = 400] [rounds = 5]        (env SWEEP_C=3 for RGB)"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
C = int(os.environ.get("SWEEP_C", "1"))
counts = [int(x) for x in os.environ.get("K1_SLOT_COUNTS", "8,16,24,40,64").split(",")]
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, C)
top = max(counts)
d_px = torch.randint(0, 256, (top, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((top, plan.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
alg = plan.pixel_bytes + plan.coef_count * 4
plan.time_transform_quant_dev(8, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 4000, stream=s)  # spin-up, as bench.py
res = {k: [] for k in counts}
for r in range(rounds):
    for k in counts:
        plan.time_transform_quant_dev(k, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 2 * k, stream=s)  # this rotation's own warm-up
        res[k].append(plan.time_transform_quant_dev(k, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, n, stream=s))
for k in counts:
    med = statistics.median(res[k])
    print(f"K1 4096x4096x{C} slots={k:3d} (pixels {k * plan.pixel_bytes / 1e6:7.1f} MB, with coefficients {k * alg / 1e6:7.1f} MB): median {med:.3f} us/launch "
          f"= {alg / med / 1e3 / 8000:.4f} of 8 TB/s; rounds {' '.join(f'{x:.2f}' for x in res[k])}", flush=True)
