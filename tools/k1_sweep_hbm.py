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
slots, n, rounds = opts["--slots"], opts["--launches"], opts["--rounds"]
ctx = frave_amd.Context(0)
KNOBS = ("FRI_HIP_BAND_ROWS", "FRI_HIP_CELLS_PER_TILE", "FRI_HIP_CELLS_PER_WG", "FRI_HIP_TILE_BYTES", "FRI_HIP_TARGET_WGS", "FRI_HIP_RANKS", "FRI_HIP_RANK_WEIGHTS")


def make_plan(spec):
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in spec.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
    return frave_amd.Plan(ctx, 4096, 4096, C)


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
for spec, p in plans:
    med = statistics.median(res[spec])
    t = p.tiling()
    print(f"[{spec or 'default'}] n_wg={t['n_wg']} tiles={t.get('n_tiles')} cells_per_tile={t.get('cells_per_tile')} band_rows={t.get('band_rows')}: median {med:.2f} us = {alg / med / 1e3 / 8000:.4f}; "
          f"rounds {' '.join(f'{x:.2f}' for x in res[spec])}", flush=True)
