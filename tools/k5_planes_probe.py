"""K5 over several planes per launch (round 5): the symbol-stream chain for one RGB image and for a batch of eight planes, run under rocprofv3 --kernel-trace; the
symbol_gather*_kernel rows are the result (FRI_HIP_TUNING=1 FRI_HIP_K5_PER_PLANE=1: one plane per workgroup, as through round 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import frave_amd

ctx = frave_amd.Context(0)
s = torch.cuda.current_stream().cuda_stream
for (c, n_img) in ((3, 1), (1, 8)):
    P = frave_amd.Plan(ctx, 4096, 4096, c)
    order = P.set_stream_order()
    n, plane = P.num_some, P.num_cells * 512
    slots = 3
    d_px = torch.randint(0, 256, (slots, n_img, P.pixel_bytes), dtype=torch.uint8, device="cuda")
    d_co = torch.empty((slots, n_img, c, plane), dtype=torch.int32, device="cuda")
    d_w = torch.empty((slots, n_img, c, plane), dtype=torch.uint16, device="cuda")
    d_st = torch.full((slots, n_img * c * n + 8), 0xFFFF, dtype=torch.uint16, device="cuda")
    d_h = torch.empty((slots, n_img, c, 10, 1024), dtype=torch.int32, device="cuda")
    d_o = torch.empty((slots, n_img, c), dtype=torch.int64, device="cuda")
    d_par = torch.zeros((slots, n_img, c, 2, 3, 6), dtype=torch.float32, device="cuda")
    d_par[..., 0] = 1.0
    for it in range(9):
        k = it % slots
        P.encode_symbols_batch_dev(n_img, d_px[k].data_ptr(), P.pixel_bytes, None, False, d_par[k].data_ptr(), d_co[k].data_ptr(), c * plane, d_w[k].data_ptr(), c * plane,
                                   d_st[k].data_ptr(), c * n, d_h[k].data_ptr(), d_o[k].data_ptr(), stream=s)
    torch.cuda.synchronize()
    st = d_st[0].cpu().numpy()
    w = d_w[0].cpu().numpy().reshape(n_img * c, -1)
    ok = bool((st[n_img * c * n:] == 0xFFFF).all()) and all(np.array_equal(w[p][order], st[p * n:(p + 1) * n]) for p in range(n_img * c))
    print(f"{n_img} image(s) x {c} channel(s): every plane's stream = words[order], nothing behind the last: {ok}")
    P.close()
