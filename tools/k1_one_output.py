"""K1 sustained, 24 rotating pixel slots, coefficients written (a) to 24 rotating planes (1.6 GB), (b) always to the SAME plane (68 MB, a quarter of the 256 MiB Infinity Cache):
does the cache absorb the kernel's nontemporal stores? HIP events through the native loop, median of 5 x 400 launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import frave_amd

ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
print("tune:", plan.tune_forward().get("winner"))
SLOTS = 24
d_px = torch.randint(0, 256, (SLOTS, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((SLOTS, plan.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for name, cs in (("24 rotating coefficient planes", plan.coef_count), ("one coefficient plane", 0), ("24 rotating coefficient planes", plan.coef_count), ("one coefficient plane", 0)):
    run = lambda n: plan.time_transform_quant_dev(SLOTS, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), cs, n, stream=s)
    run(2000)
    print(f"{name:32s}: {np.median([run(400) for _ in range(5)]):6.2f} us per launch", flush=True)
