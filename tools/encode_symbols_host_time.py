"""fri_hip_encode_image_symbols (host buffers in and out: the call the pixels-to-.frv pipeline makes per image) timed per call; FRI_HIP_LIBRARY selects the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import frave_amd

ctx = frave_amd.Context(0)
for c in (1, 3):
    P = frave_amd.Plan(ctx, 4096, 4096, c)
    P.set_stream_order()
    img = np.random.default_rng(3).integers(0, 256, P.pixel_bytes, dtype=np.uint8)
    for fit in (False, True):
        P.encode_image_symbols(img, fit=fit)
        t = []
        for _ in range(8):
            t0 = time.perf_counter()
            P.encode_image_symbols(img, fit=fit)
            t.append(time.perf_counter() - t0)
        print(f"4096x4096x{c} fit={fit}: {np.median(t) * 1e3:7.2f} ms per call (min {min(t) * 1e3:7.2f})", flush=True)
    P.close()
