"""Per-workgroup timeline of K1 (forward) or K3 (inverse) at 4096x4096 from the diagnostic trace. GPU only.
usage: FRI_HIP_TRACE=1 python tools/trace_timeline.py [k1|k3] [C]"""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

os.environ["FRI_HIP_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import frave_amd

which = sys.argv[1] if len(sys.argv) > 1 else "k1"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, C)
s = torch.cuda.current_stream().cuda_stream
SLOTS = int(os.environ.get("TRACE_SLOTS", "4"))  # 4: the pixels stay in the Infinity Cache; 40: everything comes from HBM
d_px = torch.randint(0, 256, (SLOTS, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((SLOTS, plan.coef_count), dtype=torch.int32, device="cuda")
d_back = torch.empty((SLOTS, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
for k in range(SLOTS):
    plan.transform_quant_dev(d_px[k].data_ptr(), d_co[k].data_ptr(), stream=s)
for rep in range(3 if SLOTS <= 8 else 2):  # the last launch (cold slot) is the one read back
    for k in range(SLOTS):
        if which == "k1":
            plan.transform_quant_dev(d_px[k].data_ptr(), d_co[k].data_ptr(), stream=s)
        else:
            plan.inverse_transform_dev(d_co[k].data_ptr(), d_back[k].data_ptr(), stream=s)
torch.cuda.synchronize()
tr = plan.read_trace().astype(np.int64)
_, _, wg_tiles = plan.tile_table()
if os.environ.get("TRACE_DUMP"):  # the raw stamps (slot 14 = HW_ID | XCC_ID << 32: which CU a workgroup ran on) for an analysis off the box
    np.savez(os.environ["TRACE_DUMP"], trace=tr, wg_tiles=np.asarray(wg_tiles))
n_tiles = np.diff(wg_tiles)
t0 = tr[:, 0].min()
us = lambda a: (a - t0) / 100.0
entry, pro, end = us(tr[:, 0]), us(tr[:, 1]), us(tr[:, 15])
print(f"{which} C={C}: {len(tr)} workgroups, tiles per share {n_tiles.min()}..{n_tiles.max()} (mean {n_tiles.mean():.2f})")
pc = lambda a: " ".join(f"{np.percentile(a, q):7.2f}" for q in (0, 10, 50, 90, 100))
print("                      min     p10     p50     p90     max   [us since first entry]")
print("entry              ", pc(entry))
print("prologue done      ", pc(pro))
for i in range(int(n_tiles.max())):
    m = n_tiles > i
    print(f"tile {i} done ({m.sum():4d})", pc(us(tr[m, 2 + i])))
print("exit               ", pc(end))
print("prologue duration  ", pc(pro - entry))
dur = []
for i in range(int(n_tiles.max())):
    m = n_tiles > i
    prev = tr[m, 1] if i == 0 else tr[m, 1 + i]
    dur.append((tr[m, 2 + i] - prev) / 100.0)
    print(f"tile {i} duration    ", pc(dur[-1]))
print("lifetime           ", pc(end - entry))
xcc = (tr[:, 14] >> 32) & 0xF
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"xcc {x}: {m.sum():4d} wgs  entry {entry[m].min():6.2f}..{entry[m].max():6.2f}  exit {end[m].min():6.2f}..{end[m].max():6.2f}")

# ---- where does the spread come from? ----------------------------------------------------------
life = end - entry
for n in sorted(set(n_tiles)):
    m = n_tiles == n
    print(f"shares with {n} tiles: {m.sum():4d}  lifetime {pc(life[m])}")
hw = tr[:, 14] & 0xFFFFFFFF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 0x7
key = (xcc * 8 + se) * 32 + sh * 16 + cu
cu_mean = {}
for k in np.unique(key):
    m = key == k
    cu_mean[int(k)] = (m.sum(), life[m].mean(), life[m].min(), life[m].max())
cnt = np.array([v[0] for v in cu_mean.values()])
means = np.array([v[1] for v in cu_mean.values()])
print(f"{len(cu_mean)} distinct (xcc, se, sh, cu); workgroups per CU min/mean/max {cnt.min()}/{cnt.mean():.2f}/{cnt.max()}")
print("per-CU mean lifetime   ", pc(means))
within = np.array([v[3] - v[2] for v in cu_mean.values()])
print("within-CU max-min      ", pc(within))
print(f"variance: total {life.var():.2f}, between CUs {np.average((means - life.mean()) ** 2, weights=cnt):.2f}")
for x in range(8):
    for s_ in range(8):
        m = (xcc == x) & (se == s_)
        if m.any():
            print(f"xcc {x} se {s_}: {m.sum():3d} wgs on {len(np.unique(key[m])):2d} CUs  lifetime mean {life[m].mean():6.2f} max {life[m].max():6.2f}")
order = np.argsort(end)
print("last 12 shares to exit (share id, tiles, entry, exit, xcc/se/cu):")
for i in order[-12:]:
    print(f"  share {i:4d} tiles {n_tiles[i]} entry {entry[i]:5.2f} exit {end[i]:6.2f}  {xcc[i]}/{se[i]}/{sh[i]}/{cu[i]}")
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"trace_{which}_c{C}.npy"), tr)
