"""K1 launched from a HIP graph (round 5): 96 single-image launches over rotating HBM-resident slots captured once, the graph replayed back to back; run under rocprofv3
--kernel-trace next to the library's native launch loop (the same 96 x 10 launches), tools/r5_k1_graph.sh prints both sets of kernel durations."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import frave_amd

hip = C.CDLL("libamdhip64.so")
ctx = frave_amd.Context(0)
plan = frave_amd.Plan(ctx, 4096, 4096, 1)
print("tune:", plan.tune_forward().get("winner"))
SLOTS = 24
d_px = torch.randint(0, 256, (SLOTS, plan.pixel_bytes), dtype=torch.uint8, device="cuda")
d_co = torch.empty((SLOTS, plan.coef_count), dtype=torch.int32, device="cuda")
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
N = 96
plan.time_transform_quant_dev(SLOTS, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 4000, stream=s.cuda_stream)
torch.cuda.synchronize()
assert hip.hipStreamBeginCapture(sp, 2) == 0
for k in range(N):
    plan.transform_quant_dev(d_px[k % SLOTS].data_ptr(), d_co[k % SLOTS].data_ptr(), stream=s.cuda_stream)
graph, ex = C.c_void_p(), C.c_void_p()
assert hip.hipStreamEndCapture(sp, C.byref(graph)) == 0 and graph.value
assert hip.hipGraphInstantiate(C.byref(ex), graph, None, None, 0) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    # native loop first, then the graph, each 10 x 96 launches between two events
    with torch.cuda.stream(s):
        e0.record(s)
    us_native = plan.time_transform_quant_dev(SLOTS, d_px.data_ptr(), plan.pixel_bytes, d_co.data_ptr(), plan.coef_count, 10 * N, stream=s.cuda_stream)
    s.synchronize()
    e0.record(s)
    for _ in range(10):
        assert hip.hipGraphLaunch(ex, sp) == 0
    e1.record(s)
    s.synchronize()
    print(f"round {rep}: native loop {us_native:6.2f} us per launch; graph replays {e0.elapsed_time(e1) * 1e3 / (10 * N):6.2f} us per launch (events around 10 replays of {N} launches)", flush=True)
hip.hipGraphExecDestroy(ex)
hip.hipGraphDestroy(graph)
