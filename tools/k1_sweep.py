"""Sweep K1 tiling parameters (plan-level env knobs) and print the event-timed kernel time. GPU only."""
import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import frave_amd

W = H = int(os.environ.get("SWEEP_SIZE", "4096"))
C = int(os.environ.get("SWEEP_C", "1"))
combos = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "," in a] or [(32, 9, 1024)]
ctx = frave_amd.Context(0)
slots = 8 if W <= 4096 else 2
for band, cpt, cpwg in combos:
    os.environ["FRI_HIP_BAND_ROWS"] = str(band)
    os.environ["FRI_HIP_CELLS_PER_TILE"] = str(cpt)
    os.environ["FRI_HIP_TARGET_WGS"] = str(cpwg)
    plan = frave_amd.Plan(ctx, W, H, C)
    d_px = torch.randint(0, 256, (slots, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
    d_co = torch.empty((slots, plan.coef_count), dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 20, stream=s)
    us = plan.time_transform_quant_dev(slots, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 200, stream=s)
    alg = plan.pixel_bytes + plan.coef_count * 4
    print(f"ablate={os.environ.get('FRI_HIP_K1_ABLATE','0')} tiles={plan.num_cells} band_rows={band:3d} cells_per_tile={cpt:3d} target_wgs={cpwg:4d} {plan.tiling()}  {us:8.2f} us  {alg / us / 1e3:8.1f} GB/s", flush=True)
    del d_px, d_co
    plan.close()
