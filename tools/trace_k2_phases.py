import os, sys
os.environ["FRI_HIP_TUNING"]="1"; os.environ["FRI_HIP_TRACE"]="1"
sys.path.insert(0, "/root/repo")
import numpy as np, torch, frave_amd
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
F = plan.num_cells
s = torch.cuda.current_stream().cuda_stream
d_px = torch.randint(0, 256, (plan.pixel_bytes,), dtype=torch.uint8, device="cuda")
d_co = torch.empty(plan.coef_count, dtype=torch.int32, device="cuda")
plan.transform_quant_dev(d_px.data_ptr(), d_co.data_ptr(), stream=s)
vp = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
wp = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (3, 1))
d_b = torch.empty(F * 512, dtype=torch.uint8, device="cuda"); d_p = torch.empty(F * 512, dtype=torch.int32, device="cuda")
d_h = torch.empty(10 * 1024, dtype=torch.int32, device="cuda"); d_o = torch.empty(1, dtype=torch.int64, device="cuda")
plan.assume_forward_coefficients(True)
for _ in range(3):
    plan.predict_histogram_dev(d_co.data_ptr(), 0, vp, wp, d_b.data_ptr(), d_p.data_ptr(), d_h.data_ptr(), d_o.data_ptr(), stream=s)
torch.cuda.synchronize()
tr = plan.read_trace().astype(np.int64)[:256]
pc = lambda a: " ".join(f"{np.percentile(a, q):6.2f}" for q in (10, 50, 90))
prev = tr[:, 1]
for it in range(4):
    c, m, b = tr[:, 2 + 3 * it], tr[:, 3 + 3 * it], tr[:, 4 + 3 * it]
    print(f"tile {it}: compute {pc((c - prev) / 100.0)} | commit (wait for loads + LDS writes) {pc((m - c) / 100.0)} | barrier wait {pc((b - m) / 100.0)}   [us, wave 0, p10 p50 p90]")
    prev = b
