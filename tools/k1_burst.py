"""K1 at 4096^2, one image per launch from HBM: microseconds per launch of a BURST of n back-to-back launches behind an idle gap, against the burst's length and the gap's
(round 5: the driver's 20 steps measure 16.9-17.6 us, the default 400 steps 17.9-18.6 - what is the time scale?). HIP events through the library's native loop."""
import os, sys, time
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
run = lambda n: plan.time_transform_quant_dev(SLOTS, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, n, stream=s)
run(4000)
print("burst length (gap 5 ms): us per launch, median of 12 bursts")
for n in (5, 10, 20, 50, 100, 200, 400, 1000, 4000):
    v = []
    for _ in range(12):
        torch.cuda.synchronize()
        time.sleep(0.005)
        v.append(run(n))
    print(f"  n = {n:5d}: {np.median(v):6.2f}   (min {min(v):6.2f} max {max(v):6.2f})", flush=True)
print("gap in front of a burst of 20: us per launch, median of 12 bursts")
for gap in (0.0, 0.0002, 0.001, 0.005, 0.02, 0.1):
    v = []
    for _ in range(12):
        run(400)  # busy right up to the gap
        torch.cuda.synchronize()
        if gap:
            time.sleep(gap)
        v.append(run(20))
    print(f"  gap {gap * 1e3:6.1f} ms: {np.median(v):6.2f}   (min {min(v):6.2f} max {max(v):6.2f})", flush=True)
print("back to back, no gap: 20-launch pieces of one long run")
torch.cuda.synchronize()
run(2000)
v = [run(20) for _ in range(40)]
print(f"  {np.median(v):6.2f}   (min {min(v):6.2f} max {max(v):6.2f})")
