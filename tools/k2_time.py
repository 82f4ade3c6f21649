"""Time K2 (predict + histogram), K3 (inverse) and K4 (fit sums) at K2_SIZE x K2_SIZE (default 4096). GPU only."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

C = int(os.environ.get("SWEEP_C", "1"))
kind = os.environ.get("K2_DATA", "noise")
ctx = frave_amd.Context(0)
SIZE = int(os.environ.get("K2_SIZE", "4096"))
plan = frave_amd.Plan(ctx, SIZE, SIZE, C)
F = plan.num_cells
if kind == "noise":
    d_px = torch.randint(0, 256, (plan.pixel_bytes,), dtype=torch.uint8, device="cuda")
else:  # smooth: hot histogram bins
    y, x = torch.meshgrid(torch.arange(SIZE, device="cuda"), torch.arange(SIZE, device="cuda"), indexing="ij")
    d_px = (((x + 2 * y) >> 3) + torch.randint(0, 8, (SIZE, SIZE), device="cuda")).to(torch.uint8).reshape(-1).repeat_interleave(C)
# K2_SLOTS > 1: the kernels rotate over that many coefficient planes (the same image transformed into each), so that their input comes from HBM and not from
# the 256 MiB Infinity Cache (one 68 MB plane re-read by every launch stays in it: K2 42 instead of 47 us)
SLOTS = int(os.environ.get("K2_SLOTS", "1"))
d_cos = torch.empty((SLOTS, plan.coef_count), dtype=torch.int32, device="cuda")
d_co = d_cos[0]
s = torch.cuda.current_stream().cuda_stream
for k in range(SLOTS):
    plan.transform_quant_dev(d_px.data_ptr(), d_cos[k].data_ptr(), stream=s)
_rot = [0]


def co():
    _rot[0] = (_rot[0] + 1) % SLOTS
    return d_cos[_rot[0]].data_ptr()
vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
wp = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (3, 1))
d_b = torch.empty(F * 512, dtype=torch.uint8, device="cuda")
d_p = torch.empty(F * 512, dtype=torch.int32, device="cuda")
d_h = torch.empty(10 * 1024, dtype=torch.int32, device="cuda")
d_o = torch.empty(1, dtype=torch.int64, device="cuda")
d_back = torch.empty(plan.pixel_bytes, dtype=torch.uint8, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e3


if os.environ.get("K2_TRUSTED") == "1":  # the coefficients are K1's: no exact-int32 guard launch behind K2 (fri_hip_plan_assume_forward_coefficients)
    plan.assume_forward_coefficients(True)
k2 = timed(lambda: plan.predict_histogram_dev(co(), 0, vp, wp, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=s))
k3 = timed(lambda: plan.inverse_transform_dev(co(), d_back.data_ptr(), stream=s))
d_gi = torch.empty(3 * 28, dtype=torch.int64, device="cuda")
d_gd = torch.empty(18, dtype=torch.float64, device="cuda")
k4a = timed(lambda: plan.fit_value_sums_dev(co(), 0, d_gi.data_ptr(), stream=s))
k4b = timed(lambda: plan.fit_width_sums_dev(co(), 0, vp, d_gi.data_ptr(), d_gd.data_ptr(), stream=s))
k5 = float("nan")
if os.environ.get("K5", "1") == "1" and C == 1:  # the symbol stream kernel (needs the stream order: ~0.3 s of host time at 4096^2)
    plan.set_stream_order()
    d_st = torch.empty(plan.num_some, dtype=torch.uint16, device="cuda")
    k5 = timed(lambda: plan.symbol_stream_batch_dev(1, d_co.data_ptr(), F * 512, d_b.data_ptr(), d_p.data_ptr(), F * 512, d_st.data_ptr(), plan.num_some, stream=s))
    # the halfword route: forward -> scan (2 B per node out) -> gather (2 B per symbol in), next to the array route's forward -> scan
    d_par = torch.from_numpy(np.concatenate([np.asarray(vp, np.float32).reshape(-1), np.asarray(wp, np.float32).reshape(-1)])).cuda()
    d_w = torch.empty(F * 512, dtype=torch.uint16, device="cuda")
    ca = timed(lambda: plan.encode_image_batch_dev(1, d_px.data_ptr(), plan.pixel_bytes, d_par.data_ptr(), d_co.data_ptr(), F * 512, d_b.data_ptr(), d_p.data_ptr(), F * 512,
                                                   d_h.data_ptr(), d_o.data_ptr(), fit=False, stream=s))
    cw = timed(lambda: plan.encode_symbols_batch_dev(1, d_px.data_ptr(), plan.pixel_bytes, None, False, d_par.data_ptr(), d_co.data_ptr(), F * 512, d_w.data_ptr(), F * 512,
                                                     d_st.data_ptr(), plan.num_some, d_h.data_ptr(), d_o.data_ptr(), stream=s))
    print(f"chain forward -> scan (arrays) {ca:7.2f} us; forward -> scan (halfwords) -> gather {cw:7.2f} us")
ok = bool(torch.equal(d_back, d_px))
print(f"slots={SLOTS} data={kind} hist_blocks={os.environ.get('FRI_HIP_HIST_BLOCKS', 'default')}  K2 {k2:8.2f} us  K3 {k3:8.2f} us  fit_value {k4a:7.2f} us  fit_width {k4b:7.2f} us  K5 {k5:7.2f} us  roundtrip={ok}  hist_total={int(d_h.sum())} (expect {plan.num_some})")
