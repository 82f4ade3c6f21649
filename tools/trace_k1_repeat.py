"""Are K1's slow tile iterations the same ones from launch to launch? (instrumented build) GPU only."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

os.environ["FRI_HIP_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
s = torch.cuda.current_stream().cuda_stream
SLOTS = 4
d_px = torch.randint(0, 256, (SLOTS, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((SLOTS, plan.coef_count), dtype=torch.int32, device="cuda")
tiles, cells, wg_tiles = plan.tile_table()
n_tiles = np.diff(wg_tiles)
runs = []
for rep in range(6):
    for k in range(SLOTS):
        plan.transform_quant_dev(d_px[k].data_ptr(), d_co[k].data_ptr(), stream=s)
    torch.cuda.synchronize()
    tr = plan.read_trace().astype(np.int64)
    dur = np.zeros((len(tr), 8))
    prev = tr[:, 1]
    for i in range(8):
        m = n_tiles > i
        dur[m, i] = (tr[m, 2 + i] - prev[m]) / 100.0
        prev = np.where(m, tr[:, 2 + i], prev)
    runs.append(dur)
    span = (tr[:, 15].max() - tr[:, 0].min()) / 100.0
    print(f"run {rep}: span {span:.2f} us, iterations > 5 us: {(dur > 5).sum()}, > 4 us: {(dur > 4).sum()}")
slow = [set(map(tuple, np.argwhere(d > 5))) for d in runs]
common = set.intersection(*slow[1:])
print(f"slow in all of runs 1-5: {len(common)}; union {len(set.union(*slow[1:]))}")
cnt = {}
for sset in slow[1:]:
    for e in sset:
        cnt[e] = cnt.get(e, 0) + 1
rep_hist = np.bincount(list(cnt.values()), minlength=6)
print("how often is an iteration slow over 5 runs (1x..5x):", rep_hist[1:].tolist())
mean = np.mean(runs[1:], axis=0)
wg_mean = mean.sum(axis=1)
order = np.argsort(wg_mean)[::-1][:12]
for w in order:
    t0 = wg_tiles[w]
    ys = tiles[t0:t0 + n_tiles[w], 1]
    print(f"  share {w:4d}: mean tile-time sum {wg_mean[w]:6.2f} us, tiles {n_tiles[w]}, rows y={ys.min()}..{ys.max()}, cells {tiles[t0:t0 + n_tiles[w], 5].sum()}")
# per dispatch rank (block b = idx * 8 + x  <->  share x * q + idx for n_wg % 8 == 0)
n = len(wg_mean)
q = n >> 3
share = np.arange(n)
idx = share % q
rank = np.minimum(idx * 8 // (n // 4), 3) if n % 8 == 0 else np.zeros(n, int)
rank = np.minimum((idx * 8 + share // q) // (n // 4), 3)
cells_per_share = np.array([tiles[wg_tiles[w]:wg_tiles[w + 1], 5].sum() for w in range(n)])
for r in range(4):
    m = rank == r
    print(f"rank {r}: {m.sum():4d} shares, cells {cells_per_share[m].mean():5.1f}, tiles {n_tiles[m].mean():4.2f}, tile-time sum mean {wg_mean[m].mean():6.2f} p90 {np.percentile(wg_mean[m], 90):6.2f} max {wg_mean[m].max():6.2f}")
interior = np.zeros(len(cells), bool)
print("iterations slow in all runs (share, tile index, cells in tile, tile x_lo, y_lo, width):")
for (w, i) in sorted(common)[:20]:
    t = tiles[wg_tiles[w] + i]
    print(f"  share {w:4d} rank {rank[w]} tile {i}: {t[5]} cells at x={t[0]} y={t[1]} w={t[2]} rows={t[3]}  mean {mean[w, i]:.2f} us")
# edge tiles (touching the image border) vs inner tiles, per rank
W = H = 4096
edge_t = (tiles[:, 0] == 0) | (tiles[:, 1] == 0) | (tiles[:, 0] + tiles[:, 2] >= W) | (tiles[:, 1] + tiles[:, 3] >= H)
for r in range(4):
    e, ne = [], []
    for w in np.flatnonzero(rank == r):
        for i in range(n_tiles[w]):
            (e if edge_t[wg_tiles[w] + i] else ne).append(mean[w, i] if i < mean.shape[1] else np.nan)
    print(f"rank {r}: edge tiles {len(e):4d} mean {np.nanmean(e):5.2f} us   inner tiles {len(ne):5d} mean {np.nanmean(ne):5.2f} us")
