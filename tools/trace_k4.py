"""Per-workgroup timeline of the fit kernels (K4) at 4096x4096 (tuning build: make -C frave_amd/csrc tuning; FRI_HIP_LIBRARY). GPU only.
K4_MODE=0 value sums, 1 width sums."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")
import sys

os.environ["FRI_HIP_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

mode = int(os.environ.get("K4_MODE", "0"))
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
s = torch.cuda.current_stream().cuda_stream
d_px = torch.randint(0, 256, (plan.pixel_bytes,), dtype=torch.uint8, device="cuda")
d_co = torch.empty(plan.coef_count, dtype=torch.int32, device="cuda")
plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s)
vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
d_g = torch.empty(3 * 28, dtype=torch.int64, device="cuda")
d_w = torch.empty(18, dtype=torch.float64, device="cuda")
for _ in range(3):
    if mode == 0:
        plan.fit_value_sums_dev(d_co.data_ptr(), 0, d_g.data_ptr(), stream=s)
    else:
        plan.fit_width_sums_dev(d_co.data_ptr(), 0, vp, d_g.data_ptr(), d_w.data_ptr(), stream=s)
torch.cuda.synchronize()
n_wg = int(os.environ.get("FRI_HIP_HIST_BLOCKS", "512"))
tr = plan.read_trace().astype(np.int64)[:n_wg]
t0 = tr[:, 0].min()
us = lambda a: (a - t0) / 100.0
pc = lambda a: " ".join(f"{np.percentile(a, q):7.2f}" for q in (0, 10, 50, 90, 100))
print(f"mode {mode}: {len(tr)} workgroups")
print("                      min     p10     p50     p90     max   [us since first entry]")
print("entry              ", pc(us(tr[:, 0])))
print("prologue done      ", pc(us(tr[:, 1])))
prev = tr[:, 1]
for i in range(11):
    m = tr[:, 2 + i] > tr[:, 1]
    if not m.any():
        break
    print(f"tile {i} done ({m.sum():4d})", pc(us(tr[m, 2 + i])), "  duration", pc((tr[m, 2 + i] - prev[m]) / 100.0))
    prev = np.where(m, tr[:, 2 + i], prev)
print("loop done          ", pc(us(tr[:, 13])))
print("ticket drawn       ", pc(us(tr[:, 15])), "  merge", pc((tr[:, 15] - tr[:, 13]) / 100.0))
