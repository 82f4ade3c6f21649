"""Debug aid: batched K2 (planes on grid.y) against the single-plane entry point on the same coefficients; prints where they differ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import frave_amd as fa
from tests.common import random_params

w, h, n = int(os.environ.get("DBG_W", "1920")), int(os.environ.get("DBG_H", "1080")), int(os.environ.get("DBG_N", "24"))
ctx = fa.Context(0)
P = fa.Plan(ctx, w, h, 1)
F, plane = P.num_cells, P.num_cells * 512
gen = torch.Generator(device="cuda").manual_seed(5)
d_px = torch.randint(0, 256, (n, P.pixel_bytes), dtype=torch.uint8, device="cuda", generator=gen)
d_px[:, : P.pixel_bytes // 2] >>= 3
d_co = torch.empty((n, plane), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
P.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s, n_images=n, pixel_stride=P.pixel_bytes, coef_stride=plane)
params = np.stack([np.stack(random_params(k)) for k in range(n)]).astype(np.float32)
d_params = torch.from_numpy(params).cuda()
d_b = torch.zeros((n, plane), dtype=torch.uint8, device="cuda")
d_p = torch.zeros((n, plane), dtype=torch.int32, device="cuda")
d_h = torch.zeros((n, 10, 1024), dtype=torch.int32, device="cuda")
d_o = torch.zeros(n, dtype=torch.int64, device="cuda")
for rep in range(3):
    P.predict_histogram_batch_dev(n, d_co.data_ptr(), plane, d_params.data_ptr(), d_b.data_ptr(), d_p.data_ptr(), plane, d_h.data_ptr(), d_o.data_ptr(), stream=s)
    torch.cuda.synchronize()
    tot = (d_h.sum(dim=(1, 2)) + d_o).cpu().numpy()
    print("rep", rep, "planes with a wrong total:", np.flatnonzero(tot != P.num_some).tolist(), (tot - P.num_some)[tot != P.num_some].tolist())
d_b1 = torch.zeros(plane, dtype=torch.uint8, device="cuda")
d_p1 = torch.zeros(plane, dtype=torch.int32, device="cuda")
d_h1 = torch.zeros((10, 1024), dtype=torch.int32, device="cuda")
d_o1 = torch.zeros(1, dtype=torch.int64, device="cuda")
for k in range(n):
    P.predict_histogram_dev(d_co[k].data_ptr(), 0, params[k, 0], params[k, 1], d_b1.data_ptr(), d_p1.data_ptr(), d_h1.data_ptr(), d_o1.data_ptr(), stream=s)
    torch.cuda.synchronize()
    bad_b = (d_b1 != d_b[k]).nonzero().flatten().cpu().numpy()
    bad_p = (d_p1 != d_p[k]).nonzero().flatten().cpu().numpy()
    bad_h = int((d_h1 != d_h[k]).sum())
    if len(bad_b) or len(bad_p) or bad_h or int(d_o1) != int(d_o[k]):
        print(f"plane {k}: bucket diffs {len(bad_b)} prediction diffs {len(bad_p)} hist bins {bad_h} oob {int(d_o1)} vs {int(d_o[k])} single total {int(d_h1.sum()) + int(d_o1)}")
        for idx in bad_p[:12]:
            print("   cell", idx // 512, "heap", idx % 512, "single", int(d_p1[idx]), "batch", int(d_p[k, idx]))
print("done")
# the asynchronous chain, repeatedly
d_co2 = torch.empty_like(d_co)
d_par2 = torch.zeros((n, 2, 3, 6), dtype=torch.float32, device="cuda")
for rep in range(6):
    d_h.zero_(); d_o.zero_()
    P.encode_image_batch_dev(n, d_px.data_ptr(), P.pixel_bytes, d_par2.data_ptr(), d_co2.data_ptr(), plane, d_b.data_ptr(), d_p.data_ptr(), plane, d_h.data_ptr(), d_o.data_ptr(), fit=True, stream=s)
    torch.cuda.synchronize()
    tot = (d_h.sum(dim=(1, 2)) + d_o).cpu().numpy()
    bad = np.flatnonzero(tot != P.num_some)
    print("chain rep", rep, "wrong planes:", bad.tolist()[:20], (tot - P.num_some)[bad].tolist()[:20], "oob", d_o[bad].cpu().numpy().tolist()[:20] if len(bad) else "")
    par = d_par2.cpu().numpy()
    if not np.isfinite(par).all():
        print("   non-finite parameters in planes", np.flatnonzero(~np.isfinite(par).all(axis=(1, 2, 3))).tolist()[:20])
