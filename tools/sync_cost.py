"""What the synchronize behind bench.py's timed region costs once the GPU is already idle (the native loop has polled its end event), and whether a stream
query / stream synchronize in front of it makes the device-wide synchronize cheaper. GPU only."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
slots = 8
d_px = torch.randint(0, 256, (slots, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((slots, plan.coef_count), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream()
s = st.cuda_stream
plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 2000, stream=s)
torch.cuda.synchronize()
for K in (5, 20, 100):
    for mode in ("device sync", "stream query, device sync", "stream sync, device sync", "stream sync only"):
        ts = []
        for rep in range(15):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, K, stream=s)
            t1 = time.perf_counter()
            if mode.startswith("stream query"):
                st.query()
            if mode.startswith("stream sync"):
                st.synchronize()
            if mode.endswith("device sync"):
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            ts.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6))
        ts = np.array(ts)
        print(f"K={K:3d} {mode:28s}: call {np.median(ts[:, 0]):7.1f} us, after the call {np.median(ts[:, 1]):6.1f} us, together {np.median(ts.sum(1)):7.1f} us")
