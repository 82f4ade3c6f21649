import os
os.environ.setdefault("FRI_HIP_TUNING", "1")  # opt in to the library's tuning knobs (ablations / trace need `make -C frave_amd/csrc tuning` + FRI_HIP_LIBRARY)
import torch, time
x = torch.zeros(64, device="cuda")
s = torch.cuda.current_stream()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in (1000, 5000):
    for _ in range(100): x.add_(1)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(n): x.add_(1)
    ev1.record(); torch.cuda.synchronize()
    print(f"tiny elementwise kernel, {n} back-to-back launches: {ev0.elapsed_time(ev1)/n*1e3:.2f} us per launch")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(200): x.add_(1)
g.replay(); torch.cuda.synchronize()
ev0.record(); 
for _ in range(10): g.replay()
ev1.record(); torch.cuda.synchronize()
print(f"same in a graph of 200: {ev0.elapsed_time(ev1)/2000*1e3:.2f} us per kernel")
