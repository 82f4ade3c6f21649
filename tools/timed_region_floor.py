"""What bench.py's timed region costs beyond its kernels: wall time of (synchronize, K launches through the library's native loop,
synchronize) for several K, against the HIP-event time of the same launches. wall(K) = a + b K: a is the fixed cost a short run
(the driver's --steps 20) pays. GPU only."""
import os
import sys
import time

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
for K in (1, 2, 5, 10, 20, 50, 100, 400):
    best = None
    for rep in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        us = plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, K, stream=s)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        cand = ((t2 - t0) * 1e6, (t1 - t0) * 1e6, us * K)
        best = cand if best is None or cand[0] < best[0] else best
    print(f"K={K:4d}: wall {best[0]:8.1f} us (call {best[1]:8.1f} us), events {best[2]:8.1f} us, wall - events {best[0] - best[2]:6.1f} us, wall/K {best[0] / K:7.2f} us")
