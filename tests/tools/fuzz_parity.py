"""Random-shape parity sweep of K1 / K2 / K3 / K4 against the CPU oracle (GPU box). Complements tests/test_gpu_parity.py, whose
shapes are fixed. usage: fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import frave_amd
from oracle import fri_oracle as O
from tests.common import gen_image, random_params

def run(n_cases, seed, ctx=None):
    """n_cases random (shape, content, channel count, quantiser, parameters) cases; returns the number of mismatching ones."""
    rng = np.random.default_rng(seed)
    ctx = ctx or frave_amd.Context(0)
    bad = chains = 0
    t0 = time.time()
    for case in range(n_cases):
        kind = ["noise", "smooth", "const"][int(rng.integers(0, 3))]
        if rng.random() < 0.3:  # thin / tiny images
            w, h = int(rng.integers(1, 60)), int(rng.integers(1, 900))
            if rng.random() < 0.5:
                w, h = h, w
        elif os.environ.get("FUZZ_LARGE") == "1":  # mid-size and large planes (round 5): other share counts, band layouts and list sizes than the small shapes reach
            w, h = int(rng.integers(1400, 6000)), int(rng.integers(900, 4500))
        else:
            w, h = int(rng.integers(46, 1400)), int(rng.integers(46, 900))
        c = 1 if rng.random() < 0.5 else 3
        img = gen_image(kind, w, h, c, int(rng.integers(0, 1 << 30)))
        try:
            P = frave_amd.Plan(ctx, w, h, c)
        except frave_amd.api.FriHipError as e:
            print(f"case {case}: {w}x{h}x{c}: plan error {e}")
            continue
        if os.environ.get("FUZZ_TUNE") == "1" and case % 2 == 0:  # every other plan measures its forward tiling first (round 5): any winner must give the oracle's coefficients
            P.tune_forward(8)
        W = O.Wavelet(img, h, w, c)
        q = np.ones(32, np.int32)
        if rng.random() < 0.3:
            q[:10] = rng.integers(1, 9, 10)
        co = P.transform_quant(img, q)
        W.quantize(q)
        ok = np.array_equal(co, W.coefficients())
        msgs = [] if ok else ["K1"]
        ch = int(rng.integers(0, c))
        vp, wp = random_params(int(rng.integers(0, 1000)))
        b, p, hist, oob = P.predict_histogram(co, ch, vp, wp)
        wb, wpred, whist, woob = W.predict(ch, vp, wp)
        if not (np.array_equal(b, wb) and np.array_equal(p, wpred) and np.array_equal(hist, whist) and oob == woob):
            msgs.append("K2")
        back = P.inverse_transform(co)  # the oracle's inverse takes the coefficients as they are (identity dequantiser)
        if not np.array_equal(back, W.to_raster()):
            msgs.append("K3")
        # nine images in ONE launch: the merged batch shares of the plan
        d_px = torch.from_numpy(np.tile(img.reshape(1, -1), (9, 1))).cuda()
        d_co = torch.empty((9, P.coef_count), dtype=torch.int32, device="cuda")
        P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), qmatrix=q, n_images=9, pixel_stride=P.pixel_bytes, coef_stride=P.coef_count)
        torch.cuda.synchronize()
        if not bool((d_co.cpu().numpy().reshape(9, -1) == co.reshape(1, -1)).all()):
            msgs.append("K1-batch")
        # the chain with the fit (device-side solves) and the symbol stream route on the same image: the fitted parameters equal the host solves of the
        # stage-by-stage sums bit for bit, the scan with them equals the oracle's, the stream is the gather of the array route's outputs
        if P.num_some > 0 and not (kind == "const" and rng.random() < 0.5):
            try:
                co2, vpf, wpf, b2, p2, h2, oob2 = P.encode_image(img, q, fit=True)
                chains += 1
                if not np.array_equal(co2, co):
                    msgs.append("chain-K1")
                ok_fit = True
                for cc in range(c):
                    gram = P.fit_value_sums(co, cc)
                    iu = np.triu_indices(7)
                    vh = frave_amd.api.fit_value_params(np.stack([gram[g][iu] for g in range(3)]))
                    wtw, wtr, rows = P.fit_width_sums(co, cc, vh)
                    iu6 = np.triu_indices(6)
                    wh = frave_amd.api.fit_width_params(np.stack([wtw[g][iu6] for g in range(3)]), wtr, rows)
                    ok_fit = ok_fit and np.array_equal(vh.view(np.uint32), vpf[cc].view(np.uint32)) and np.array_equal(wh.view(np.uint32), wpf[cc].view(np.uint32))
                if not ok_fit:
                    msgs.append("fit(device solve != host solve)")
                if np.isfinite(vpf).all() and np.isfinite(wpf).all():
                    wb2, wp2, wh2, wo2 = W.predict(ch, vpf[ch].reshape(-1), wpf[ch].reshape(-1))
                    if not (np.array_equal(b2[ch], wb2) and np.array_equal(p2[ch], wp2) and np.array_equal(h2[ch], wh2) and int(oob2[ch]) == wo2):
                        msgs.append("chain-K2")
                order = P.set_stream_order()
                sym, vps, wps, hs, oobs = P.encode_image_symbols(img, q, fit=True)
                if not (np.array_equal(vps.view(np.uint32), vpf.view(np.uint32)) and np.array_equal(wps.view(np.uint32), wpf.view(np.uint32)) and np.array_equal(hs, h2)):
                    msgs.append("symbols-route(params/hist)")
                if int(oob2.sum()) == 0:
                    for cc in range(c):
                        d = (co[cc].reshape(-1)[order].astype(np.int64) - p2[cc].reshape(-1)[order].astype(np.int64)).astype(np.int32)
                        ref = (b2[cc].reshape(-1)[order].astype(np.uint32) << 10 | (((d.astype(np.uint32) << 1) ^ (d >> 31).astype(np.uint32)) & 1023)).astype(np.uint16)
                        if not np.array_equal(sym[cc], ref):
                            msgs.append(f"K5 ch{cc}")
            except frave_amd.api.FriHipError as e:
                msgs.append(f"chain error {e}")
        if msgs:
            bad += 1
            print(f"case {case}: {w}x{h}x{c} {kind} q={q[:10].tolist()}: MISMATCH in {msgs}", flush=True)
        if (case + 1) % (10 if os.environ.get("FUZZ_LARGE") == "1" else 250) == 0:  # a sign of life for long runs (a silent command is taken to be hung after a few minutes)
            print(f"... {case + 1} cases, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
        P.close()
        W.close()
    print(f"{n_cases} cases ({chains} with the fitted chain and the symbol stream route), {bad} mismatching, {time.time() - t0:.0f} s")
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
