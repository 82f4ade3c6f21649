"""K1 tuning knobs in the HBM-bound regime: the single-image launch loop over enough rotating slots that neither pixels nor coefficients
can come from the 256 MiB Infinity Cache (tools/k1_slots.py: 8 slots 16.5 us, 24+ slots 20.3 us). Every plan-level knob set is a string of
NAME=VALUE pairs; the sets are measured in interleaved rounds inside one process (the boxes of the pool differ by more than most effects).

usage: python3 tools/k1_sweep_hbm.py [--slots 32] [--launches 300] [--rounds 3] "" "FRI_HIP_RANK_WEIGHTS=1,1,1,1" "FRI_HIP_TARGET_WGS=2048 FRI_HIP_RANK_WEIGHTS=1,1,1,1" ...
(env SWEEP_C=3 for RGB)"""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

args = sys.argv[1:]
opts = {"--slots": 32, "--launches": 300, "--rounds": 3}
sets = []
i = 0
while i < len(args):
    if args[i] in opts:
        opts[args[i]] = int(args[i + 1])
        i += 2
    else:
        sets.append(args[i])
        i += 1
if not sets:
    sets = [""]
C = int(os.environ.get("SWEEP_C", "1"))
SW, SH = int(os.environ.get("SWEEP_W", "4096")), int(os.environ.get("SWEEP_H", "4096"))
slots, n, rounds = opts["--slots"], opts["--launches"], opts["--rounds"]
ctx = frave_amd.Context(0)
KNOBS = ("FRI_HIP_K1_CACHED_STORES", "FRI_HIP_INV_STRIDED_SHARES", "FRI_HIP_INV_SHARED", "FRI_HIP_INV_BAND_ROWS", "FRI_HIP_STRIDED_SHARES", "FRI_HIP_BAND_ROWS", "FRI_HIP_CELLS_PER_TILE", "FRI_HIP_CELLS_PER_WG", "FRI_HIP_TILE_BYTES", "FRI_HIP_TARGET_WGS", "FRI_HIP_RANKS", "FRI_HIP_RANK_WEIGHTS")


def make_plan(spec):
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in spec.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
    return frave_amd.Plan(ctx, SW, SH, C)


plans = []
for spec in sets:
    try:
        plans.append((spec, make_plan(spec)))
    except Exception as e:  # a knob set the kernel's LDS / register budget refuses
        print(f"[{spec}] refused: {e}", flush=True)
base = plans[0][1]
d_px = torch.randint(0, 256, (slots, base.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((slots, base.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
alg = base.pixel_bytes + base.coef_count * 4
run = lambda p, k: p.time_transform_quant_dev(slots, d_px.data_ptr(), p.pixel_bytes, d_co.data_ptr(), p.coef_count, k, stream=s)
run(base, 3000)
res = {spec: [] for spec, _ in plans}
for r in range(rounds):
    for spec, p in plans:
        run(p, slots)
        res[spec].append(run(p, n))
# K3 walks the same tiles, and a launch over many images uses the merged shares: the same knob sets, so that a new default does not cost them
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, reps):
    fn(0)
    torch.cuda.synchronize()
    ev0.record()
    for i in range(reps):
        fn(i)
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e3


extra = {}
if os.environ.get("SWEEP_EXTRAS", "1") != "0":
    d_back = torch.empty((slots, base.pixel_bytes), dtype=torch.uint8, device="cuda")
    for spec, p in plans:
        k3 = [timed(lambda i: p.inverse_transform_dev(d_co[i % slots].data_ptr(), d_back[i % slots].data_ptr(), stream=s), 2 * slots) for _ in range(2)]
        nb = min(slots, 32)
        kb = [timed(lambda i: p.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=nb, pixel_stride=p.pixel_bytes, coef_stride=p.coef_count), 4) / nb for _ in range(2)]
        extra[spec] = f"K3 {min(k3):.2f} us; K1 batch of {nb} distinct images {min(kb):.2f} us/image"
for spec, p in plans:
    med = statistics.median(res[spec])
    t = p.tiling()
    print(f"{SW}x{SH}x{C} [{spec or 'default'}] n_wg={t['n_wg']} tiles={t.get('n_tiles')} cells_per_tile={t.get('cells_per_tile')} band_rows={t.get('band_rows')}: median {med:.2f} us = {alg / med / 1e3 / 8000:.4f}; "
          f"rounds {' '.join(f'{x:.2f}' for x in res[spec])}; {extra.get(spec, '')}", flush=True)
