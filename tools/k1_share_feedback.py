"""Prototype (round 5): can measured per-share exit times re-cut K1's shares? Tuning build (trace stamps). Each iteration: a plan with the current whole-tile counts per share
(FRI_HIP_SHARE_TILES_FILE), HIP-event time of the kernel over rotating HBM-resident slots, per-share exits (median over several traced launches), then tiles move from the
latest shares to the earliest ones. usage: FRI_HIP_LIBRARY=frave_amd/libfri_hip_tuning.so python3 tools/k1_share_feedback.py [iterations]"""
import os, sys, tempfile
os.environ["FRI_HIP_TUNING"] = "1"
os.environ["FRI_HIP_TRACE"] = "1"
os.environ.setdefault("FRI_HIP_STRIDED_SHARES", "0")
os.environ.setdefault("FRI_HIP_BAND_ROWS", "72")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import frave_amd

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = frave_amd.Context(0)
SLOTS = 32
d_px = d_co = None
path = os.path.join(tempfile.gettempdir(), "share_tiles.txt")
counts = None
s = torch.cuda.current_stream().cuda_stream
best = None
for it in range(iters):
    if counts is not None:
        np.savetxt(path, counts, fmt="%d")
        os.environ["FRI_HIP_SHARE_TILES_FILE"] = path
    plan = frave_amd.Plan(ctx, 4096, 4096, 1)
    if d_px is None:
        d_px = torch.randint(0, 256, (SLOTS, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
        d_co = torch.empty((SLOTS, plan.coef_count), dtype=torch.int32, device="cuda")
    _, _, wg = plan.tile_table()
    nt = np.diff(wg)
    if counts is not None:
        assert np.array_equal(nt, counts), "the plan did not take the counts"
    plan.time_transform_quant_dev(SLOTS, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 2000, stream=s)
    us = np.median([plan.time_transform_quant_dev(SLOTS, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 200, stream=s) for _ in range(5)])
    exits, pros = [], []
    for k in range(12):  # traced launches over different slots: the per-share exit relative to the first entry
        plan.transform_quant_dev(d_px[(5 * k) % SLOTS].data_ptr(), d_co[(5 * k) % SLOTS].data_ptr(), stream=s)
        torch.cuda.synchronize()
        tr = plan.read_trace().astype(np.int64)[: len(nt)]
        t0 = tr[:, 0].min()
        exits.append((tr[:, 15] - t0) / 100.0)
        pros.append((tr[:, 1] - t0) / 100.0)
    ex, pro = np.median(exits, axis=0), np.median(pros, axis=0)
    span = np.median([e.max() for e in exits])
    print(f"iteration {it}: {us:6.2f} us per launch (events, 5 x 200 launches); traced: last exit {span:5.2f} us, median-per-share exits p10/50/90/100 "
          f"{np.percentile(ex, 10):5.2f} {np.percentile(ex, 50):5.2f} {np.percentile(ex, 90):5.2f} {ex.max():5.2f}; tiles per share {np.bincount(nt)[1:].tolist()}", flush=True)
    if best is None or us < best[0]:
        best = (us, it)
    # move tiles: a share's per-tile time tau = (exit - prologue) / tiles; greedy - take a tile from the share with the latest predicted exit, give it to the one whose
    # predicted exit after receiving it is the earliest, while that lowers the maximum; at most 8 % of the shares change per iteration (the measurement is noisy)
    tau = (ex - pro) / nt
    pred = ex.copy()
    new = nt.copy()
    for _ in range(len(nt) // 12):
        src = int(np.argmax(np.where(new > 1, pred, -1)))
        cand = pred + tau
        dst = int(np.argmin(cand))
        if cand[dst] >= pred[src] - 0.3 or src == dst:
            break
        new[src] -= 1
        pred[src] -= tau[src]
        new[dst] += 1
        pred[dst] += tau[dst]
    counts = new
    plan.close()
print(f"best: iteration {best[1]} at {best[0]:.2f} us")
