"""Per-workgroup timeline of the pipelined K2 at 4096x4096 (instrumented build: make -C frave_amd/csrc trace). GPU only."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

os.environ["FRI_HIP_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

ctx = frave_amd.Context(0)
ctx_cus = int(os.environ.get("TRACE_ROWS", os.environ.get("FRI_HIP_PRED_BLOCKS", "256")))  # rows of the trace = workgroups of the launch
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
F = plan.num_cells
s = torch.cuda.current_stream().cuda_stream
d_px = torch.randint(0, 256, (plan.pixel_bytes,), dtype=torch.uint8, device="cuda")
d_co = torch.empty(plan.coef_count, dtype=torch.int32, device="cuda")
plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s)
vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
wp = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (3, 1))
d_b = torch.empty(F * 512, dtype=torch.uint8, device="cuda")
d_p = torch.empty(F * 512, dtype=torch.int32, device="cuda")
d_h = torch.empty(10 * 1024, dtype=torch.int32, device="cuda")
d_o = torch.empty(1, dtype=torch.int64, device="cuda")
for _ in range(3):
    plan.predict_histogram_dev(d_co.data_ptr(), 0, vp, wp, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=s)
torch.cuda.synchronize()
tr = plan.read_trace().astype(np.int64)
tr = tr[: min(len(tr), ctx_cus)]  # rows beyond the K2 grid still hold the forward kernel's stamps
t0 = tr[:, 0].min()
us = lambda a: (a - t0) / 100.0
pc = lambda a: " ".join(f"{np.percentile(a, q):7.2f}" for q in (0, 10, 50, 90, 100))
print(f"{len(tr)} workgroups")
print("                      min     p10     p50     p90     max   [us since first entry]")
print("entry              ", pc(us(tr[:, 0])))
if (tr[:, 11] > 0).all():  # (one-off stamps of a hacked tuning build: slot lists in LDS / the first tile's loads have landed)
    print("ring in LDS        ", pc(us(tr[:, 11])))
    print("tile 0 data landed ", pc(us(tr[:, 12])))
print("prologue done      ", pc(us(tr[:, 1])))
prev = tr[:, 1]
for i in range(11):
    m = (tr[:, 2 + i] > tr[:, 1]) & (tr[:, 2 + i] <= tr[:, 13])  # (a slot this launch did not stamp holds an older kernel's)
    if not m.any():
        break
    print(f"tile {i} done ({m.sum():4d})", pc(us(tr[m, 2 + i])), "  duration", pc((tr[m, 2 + i] - prev[m]) / 100.0))
    prev = np.where(m, tr[:, 2 + i], prev)
print("loop done          ", pc(us(tr[:, 13])))
print("exit               ", pc(us(tr[:, 15])), "  merge", pc((tr[:, 15] - tr[:, 13]) / 100.0))
# which workgroups take the long tiles?
dur = np.zeros((len(tr), 11))
prev = tr[:, 1]
for i in range(11):
    m = (tr[:, 2 + i] > tr[:, 1]) & (tr[:, 2 + i] <= tr[:, 13])
    dur[m, i] = (tr[m, 2 + i] - prev[m]) / 100.0
    prev = np.where(m, tr[:, 2 + i], prev)
slow = np.argwhere(dur > 7.5)
xcc = (tr[:, 14] >> 32) & 0xF
hw = tr[:, 14] & 0xFFFFFFFF
print(f"{len(slow)} tile iterations longer than 7.5 us:")
for w, i in slow[:40]:
    print(f"  workgroup {w:3d} tile {i}  {dur[w, i]:5.2f} us  ends at {us(tr[w, 2 + i]):6.2f}  xcc {xcc[w]} se {(hw[w] >> 13) & 7} cu {(hw[w] >> 8) & 15}")
