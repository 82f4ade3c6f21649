#!/usr/bin/env python3
"""Runs the five BASELINE.json configs on one MI355X and prints a report: kernel-only Mpixels/s and
algorithmic GB/s per kernel, fraction of the 8 TB/s HBM roofline, CPU-oracle Mpixels/s where it is run, parity verdicts.
    python tests/tools/report_configs.py [--quick]
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import frave_amd
from oracle import fri_oracle as O
from tests.common import KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, gen_image

QUICK = "--quick" in sys.argv
PEAK = 8000.0
ctx = frave_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e3  # us


def line(name, us, pixels, nbytes):
    gbs = nbytes / us / 1e3
    print(f"    {name:34s} {us:10.2f} us  {pixels / us:12.1f} Mpix/s  {gbs:8.1f} GB/s algorithmic  {gbs / PEAK * 100:5.1f} % of 8 TB/s")


def kernels(w, h, c, slots, label):
    P = frave_amd.Plan(ctx, w, h, c)
    F = P.num_cells
    try:  # like bench.py: the plan measures its forward tiling (later plans of the shape in this process start from the winner)
        tuned = P.tune_forward()
    except frave_amd.api.FriHipError as e:
        tuned = {"tuned": False, "why": str(e)}
    print(f"  {label}: {w}x{h}x{c}, F={F} cells, forward tiling {tuned.get('winner') or ('default: ' + str(tuned.get('why', 'not tunable')))}, {P.tiling()}")
    d_px = torch.randint(0, 256, (slots, P.pixel_bytes), dtype=torch.uint8, device="cuda")
    d_co = torch.empty((slots, P.coef_count), dtype=torch.int32, device="cuda")
    k1 = P.time_transform_quant_dev(slots, d_px.data_ptr(), P.pixel_bytes, d_co.data_ptr(), P.coef_count, 2000 if w * h < 1e8 else 40, stream=s)  # untimed: tens of ms of work, like bench.py's spin-up
    k1 = P.time_transform_quant_dev(slots, d_px.data_ptr(), P.pixel_bytes, d_co.data_ptr(), P.coef_count, 400 if w * h < 1e8 else 40, stream=s)
    alg1 = P.pixel_bytes + P.coef_count * 4
    line("K1 transform+quant (all channels)", k1, w * h, alg1)
    d_b = torch.empty(F * 512, dtype=torch.uint8, device="cuda")
    d_p = torch.empty(F * 512, dtype=torch.int32, device="cuda")
    d_h = torch.empty(10 * 1024, dtype=torch.int32, device="cuda")
    d_o = torch.empty(1, dtype=torch.int64, device="cuda")
    d_back = torch.empty(P.pixel_bytes, dtype=torch.uint8, device="cuda")
    vp, wp = np.asarray(KAT_VALUE_PARAMS, np.float32).reshape(3, 6), np.asarray(KAT_WIDTH_PARAMS, np.float32).reshape(3, 6)
    co0, pb, pp, ph, po = d_co[0].data_ptr(), d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr()  # keep the host side of a call short: the loop is timed by events
    # every kernel rotates over the slots (round 4): its input comes from HBM, not from the 256 MiB Infinity Cache (one re-read plane would stay in it)
    rot = [0]

    def co():  # the next slot's coefficients
        rot[0] = (rot[0] + 1) % slots
        return co0 + rot[0] * P.coef_count * 4

    def px():  # the pixels of the slot co() will hand out next (evaluated in front of it in an argument list)
        return d_px[(rot[0] + 1) % slots].data_ptr()

    k2 = timed(lambda: P.predict_histogram_dev(co(), 0, vp, wp, pb, pp, ph, po, stream=s), 30)
    line("K2 predict+histogram (per channel)", k2, w * h, F * 512 * 9 + 40960)
    # the same with the caller's promise that the planes are the forward kernel's output (fri_hip_plan_assume_forward_coefficients: still checked, no exact
    # kernel behind it) - what a host that keeps the coefficients between calls would use; the chains below need no promise
    P.assume_forward_coefficients(True)
    k2p = timed(lambda: P.predict_histogram_dev(co(), 0, vp, wp, pb, pp, ph, po, stream=s), 30)
    P.assume_forward_coefficients(False)
    line("K2, forward coefficients promised", k2p, w * h, F * 512 * 9 + 40960)
    pk = d_back.data_ptr()
    k3 = timed(lambda: P.inverse_transform_dev(co(), pk, stream=s), 30)
    ok = bool(torch.equal(d_back, d_px[rot[0]]))
    line("K3 inverse (all channels)", k3, w * h, alg1)
    d_g = torch.empty(3 * 28, dtype=torch.int64, device="cuda")
    d_w = torch.empty(18, dtype=torch.float64, device="cuda")
    pg, pw = d_g.data_ptr(), d_w.data_ptr()
    k4v = timed(lambda: P.fit_value_sums_dev(co(), 0, pg, stream=s), 30)
    k4w = timed(lambda: P.fit_width_sums_dev(co(), 0, vp, pg, pw, stream=s), 30)
    line("K4 fit value sums (per channel)", k4v, w * h, F * 512 * 4)
    line("K4 fit width sums (per channel)", k4w, w * h, F * 512 * 4)
    # the device part of FRIEncoder::encode in one call, all channels, coefficients staying in HBM
    d_ba = torch.empty(c * F * 512, dtype=torch.uint8, device="cuda")
    d_pa = torch.empty(c * F * 512, dtype=torch.int32, device="cuda")
    d_ha = torch.empty(c * 10 * 1024, dtype=torch.int32, device="cuda")
    d_oa = torch.empty(c, dtype=torch.int64, device="cuda")
    vpc, wpc = np.stack([vp] * c).astype(np.float32), np.stack([wp] * c).astype(np.float32)
    px0 = d_px[0].data_ptr()
    enc = timed(lambda: P.encode_image_dev(px(), co(), d_ba.data_ptr(), d_pa.data_ptr(), d_ha.data_ptr(), d_oa.data_ptr(), vpc, wpc, fit=False, stream=s), 20)
    line("encode chain K1 -> K2 (params given)", enc, w * h, alg1 + c * (F * 512 * 9 + 40960))
    vpf, wpf = np.zeros((c, 3, 6), np.float32), np.zeros((c, 3, 6), np.float32)
    encf = timed(lambda: P.encode_image_dev(px(), co(), d_ba.data_ptr(), d_pa.data_ptr(), d_ha.data_ptr(), d_oa.data_ptr(), vpf, wpf, fit=True, stream=s), 20)
    line("encode chain with the fit (device solves)", encf, w * h, alg1 + c * (F * 512 * 17 + 40960))
    tot = int(d_h.sum()) + int(d_o.item())
    print(f"    lossless K3(K1(x)) == x: {ok};  histogram total {tot} == Some coefficients {P.num_some}: {tot == P.num_some}")
    return P, d_px, d_co


print("== config 1: 512x512 plumbing (CPU oracle round trip; GPU parity) ==")
g = gen_image("smooth", 512, 512, 1, 1)
t0 = time.perf_counter()
W1 = O.Wavelet(g, 512, 512, 1)
W3 = O.Wavelet(np.repeat(g, 3, axis=2), 512, 512, 3)
dt = time.perf_counter() - t0
print(f"  oracle: luma round trip lossless {np.array_equal(W1.to_raster(), g.reshape(-1))}; RGB(R=G=B) channel 0 == luma {np.array_equal(W1.coefficients()[0], W3.coefficients()[0])}; "
      f"F={W1.num_cells} (BFS {W1.num_bfs_cells}); {dt:.2f} s for both")
P = frave_amd.Plan(ctx, 512, 512, 1)
print(f"  GPU == oracle, bit for bit: {np.array_equal(P.transform_quant(g), W1.coefficients())}")

print("== config 2: single 4096x4096 on 1 MI355X ==")
for c in (1, 3):
    P, d_px, d_co = kernels(4096, 4096, c, 24 if c == 1 else 8, f"C={c}")
    if not QUICK:
        img = d_px[0].cpu().numpy()
        t0 = time.perf_counter()
        W = O.Wavelet(img, 4096, 4096, c)
        dt = time.perf_counter() - t0
        P.transform_quant_dev(d_px[0].data_ptr(), d_co[0].data_ptr(), stream=s)
        torch.cuda.synchronize()
        same = np.array_equal(d_co[0].cpu().numpy().reshape(c, P.num_cells, 512), W.coefficients())
        print(f"    CPU oracle transform: {dt:.1f} s = {4096 * 4096 / dt / 1e6:.2f} Mpix/s (1 thread);  GPU == oracle bit for bit: {same}")
        W.close()
    del d_px, d_co

print("== config 3: 256 x 1920x1080 on 1 MI355X ==")
P = frave_amd.Plan(ctx, 1920, 1080, 1)
n = 256
d_px = torch.randint(0, 256, (n, P.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((n, P.coef_count), dtype=torch.int32, device="cuda")
us = timed(lambda: P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=n, pixel_stride=P.pixel_bytes, coef_stride=P.coef_count), 5)
line(f"K1, one launch over {n} frames", us, n * 1920 * 1080, n * (P.pixel_bytes + P.coef_count * 4))
# the whole device chain over the batch: fit sums of all planes in one launch each (the 6 x 6 solves of 256 planes on the host in between), K2 for all planes in one launch
F3, plane3 = P.num_cells, P.num_cells * 512
d_gram = torch.empty((n, 3, 28), dtype=torch.int64, device="cuda")
d_wtw = torch.empty((n, 3, 21), dtype=torch.int64, device="cuda")
d_wtr = torch.empty((n, 3, 6), dtype=torch.float64, device="cuda")
d_par = torch.from_numpy(np.tile(np.stack([KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS]).astype(np.float32), (n, 1, 1, 1))).cuda()
d_b3 = torch.empty((n, plane3), dtype=torch.uint8, device="cuda")
d_p3 = torch.empty((n, plane3), dtype=torch.int32, device="cuda")
d_h3 = torch.empty((n, 10, 1024), dtype=torch.int32, device="cuda")
d_o3 = torch.empty(n, dtype=torch.int64, device="cuda")
k1b = us
k4vb = timed(lambda: P.fit_value_sums_batch_dev(n, d_co.data_ptr(), plane3, d_gram.data_ptr(), stream=s), 5)
k4wb = timed(lambda: P.fit_width_sums_batch_dev(n, d_co.data_ptr(), plane3, d_par.data_ptr(), d_wtw.data_ptr(), d_wtr.data_ptr(), stream=s), 5)
k2b = timed(lambda: P.predict_histogram_batch_dev(n, d_co.data_ptr(), plane3, d_par.data_ptr(), d_b3.data_ptr(), d_p3.data_ptr(), plane3, d_h3.data_ptr(), d_o3.data_ptr(), stream=s), 5)
line(f"K4 value sums, one launch over {n} planes", k4vb, n * 1920 * 1080, n * plane3 * 4)
line(f"K4 width sums, one launch over {n} planes", k4wb, n * 1920 * 1080, n * plane3 * 4)
line(f"K2, one launch over {n} planes", k2b, n * 1920 * 1080, n * (plane3 * 9 + 40960))
print(f"    device chain per frame: K1 {k1b / n:.2f} + K4 {k4vb / n:.2f} + {k4wb / n:.2f} + K2 {k2b / n:.2f} = {(k1b + k4vb + k4wb + k2b) / n:.2f} us/frame ({n * 1920 * 1080 / (k1b + k4vb + k4wb + k2b):.0f} Mpix/s), "
      f"without the fit {(k1b + k2b) / n:.2f} us/frame")
del d_b3, d_p3, d_px, d_co
drv = os.path.join(ROOT, "frave_amd", "host", "fri_driver")
if os.path.exists(drv):
    print("   ", subprocess.run([drv, "batch", "1920", "1080", "1", "256"], capture_output=True, text=True).stdout.strip())
    # pixels -> .frv bytes: device chains to the emitter's input and host rANS emits pipelined (libfri::encode_batch_bytes)
    print("   ", subprocess.run([drv, "batch-frv", "1920", "1080", "1", "256", "--emitters", "12"], capture_output=True, text=True).stdout.strip())

print("== config 4: 1024 x 4096x4096 over 8 GPUs -> this GPU's share is 128 images (image i -> rank i mod 8, no collective) ==")
P = frave_amd.Plan(ctx, 4096, 4096, 1)
n = 32
d_px = torch.randint(0, 256, (n, P.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((n, P.coef_count), dtype=torch.int32, device="cuda")


def share():
    for _ in range(4):  # 4 launches x 32 images = 128
        P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=n, pixel_stride=P.pixel_bytes, coef_stride=P.coef_count)


us = timed(share, 3)
line("K1, 128 images as 4 launches of 32", us, 128 * 4096 * 4096, 128 * (P.pixel_bytes + P.coef_count * 4))
del d_px, d_co

print("== config 5: single 16384x16384, histogram on the device ==")
kernels(16384, 16384, 1, 2, "C=1")
