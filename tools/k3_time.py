"""Time K3 (inverse transform) at 4096x4096, optionally under the timing-only ablation flags. GPU only.
usage: python tools/k3_time.py [C] [ablate,ablate,...]"""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
flags = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
ctx = frave_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
SLOTS = 4  # > Infinity Cache
for ab in flags:
    os.environ["FRI_HIP_K3_ABLATE"] = str(ab)  # read at plan creation
    plan = frave_amd.Plan(ctx, 4096, 4096, C)
    d_px = torch.randint(0, 256, (SLOTS, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
    d_co = torch.empty((SLOTS, plan.coef_count), dtype=torch.int32, device="cuda")
    d_back = torch.empty((SLOTS, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
    for k in range(SLOTS):
        plan.transform_quant_dev(d_px[k].data_ptr(), d_co[k].data_ptr(), stream=s)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    for i in range(SLOTS):
        plan.inverse_transform_dev(d_co[i].data_ptr(), d_back[i].data_ptr(), stream=s)
    torch.cuda.synchronize()
    ok = bool(torch.equal(d_back, d_px)) if ab == 0 else None
    ev0.record()
    for i in range(reps):
        plan.inverse_transform_dev(d_co[i % SLOTS].data_ptr(), d_back[i % SLOTS].data_ptr(), stream=s)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) / reps * 1e3
    alg = plan.pixel_bytes + plan.coef_count * 4
    print(f"C={C} ablate={ab:2d}  K3 {us:8.2f} us  {alg / us / 1e6:6.2f} TB/s algorithmic  tiling={plan.tiling()}  roundtrip={ok}", flush=True)
    del plan
