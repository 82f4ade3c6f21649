"""K1 launch by launch behind a drain (run under rocprofv3 --kernel-trace): ten times 400 busy launches, a synchronise, then 60 launches; tools/r5_k1_drain.sh prints the mean
kernel duration by position behind the synchronise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frave_amd

ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
plan.tune_forward()
SLOTS = 24
d_px = torch.randint(0, 256, (SLOTS, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((SLOTS, plan.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
run = lambda n: plan.time_transform_quant_dev(SLOTS, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, n, stream=s)
run(4000)
for _ in range(10):
    run(400)
    torch.cuda.synchronize()
    run(60)
    torch.cuda.synchronize()
