"""One K1 measurement pass in the HBM-bound regime (launch loop over enough rotating slots that neither pixels nor coefficients come from the
256 MiB Infinity Cache) for tools/k1_ab_hbm.py. Prints `AB <us per launch> [<two-stream period>]`.
env: AB_W, AB_H (4096), AB_C (1), AB_SLOTS (24 planes / 12 RGB), AB_TUNE=1 (fri_hip_plan_tune_forward first, report on stderr), AB_STREAMS=2 (also the n-stream period)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

W, H, C = int(os.environ.get("AB_W", "4096")), int(os.environ.get("AB_H", "4096")), int(os.environ.get("AB_C", "1"))
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, W, H, C)
slots = int(os.environ.get("AB_SLOTS", "0")) or max(2, min(64, (420 << 20) // plan.pixel_bytes + 1))
if os.environ.get("AB_TUNE") == "1" and hasattr(plan, "tune_forward"):
    print("tune:", plan.tune_forward(), file=sys.stderr)
d_px = torch.randint(0, 256, (slots, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((slots, plan.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
run = lambda k: plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, k, stream=s)
n = int(os.environ.get("AB_LAUNCHES", "300"))
run(max(200, int(60000 / max(1.0, run(20)))))  # ~60 ms of spin-up
one = statistics.median(run(n) for _ in range(5))
out = f"AB {one:.3f}"
ns = int(os.environ.get("AB_STREAMS", "0"))
if ns > 1:
    try:
        two = statistics.median(plan.time_transform_quant_streams_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, n, ns) for _ in range(5))
    except AttributeError:  # a library from before round 5
        two = float("nan")
    out += f" {two:.3f}"
nb = min(slots, int(os.environ.get("AB_BATCH", "0")))
if nb > 1:  # many distinct images per launch: the steady state without one launch's ramp and tail
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def batch():
        plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=nb, pixel_stride=plan.pixel_bytes, coef_stride=plan.coef_count)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(4):
            plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=nb, pixel_stride=plan.pixel_bytes, coef_stride=plan.coef_count)
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) * 1e3 / 4 / nb

    out += f" {statistics.median(batch() for _ in range(3)):.3f}"
print(out)
