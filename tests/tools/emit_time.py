"""Host emit (symbol order, ANS models, rANS, container) timed on the CPU alone: the inputs come from the CPU oracle instead of the device
kernels (same arrays), so this runs without a GPU. First call of a geometry builds and caches the symbol order; the following ones are
what every further image of that size costs.   python tests/tools/emit_time.py [width height channels]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from frave_amd import emit
from oracle import fri_oracle
from tests.common import KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, gen_image

w, h, c = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (4096, 4096, 1)
fri_oracle.build()
img = gen_image("noise", w, h, c, 5)
img[:, : w // 2] = gen_image("smooth", w // 2, h, c, 6)
t0 = time.perf_counter()
W = fri_oracle.Wavelet(img, h, w, c)
W.quantize(np.ones(32, np.int32))
co = W.coefficients()
vp = np.stack([KAT_VALUE_PARAMS] * c)
wp = np.stack([KAT_WIDTH_PARAMS] * c)
per = [W.predict(ch, vp[ch], wp[ch]) for ch in range(c)]
centers = W.centers()
W.close()
b = np.stack([p[0] for p in per])
pr = np.stack([p[1] for p in per])
hist = np.stack([p[2] for p in per])
print(f"oracle inputs for {w}x{h}x{c}: {time.perf_counter() - t0:.1f} s, {len(centers)} cells")
for k in range(2):
    t0 = time.perf_counter()
    frv = emit.encode_image(w, h, centers, co, b, pr, hist, vp, wp)
    print(f"encode_image call {k}: {time.perf_counter() - t0:.3f} s, {len(frv)} bytes")
# The symbol stream route (K5 on the device): what is left for the host is contexts + rANS + container. The stream is built here with numpy from the
# same arrays (on the device that is fri_hip_symbol_stream_batch_dev); the stream order is geometry and computed once per plan.
import frave_amd

P = frave_amd.Plan(None, w, h, c)  # host-only plan: the Some/None masks
t0 = time.perf_counter()
order = emit.stream_order(centers, P.valid_mask())
print(f"stream_order (once per plan): {time.perf_counter() - t0:.3f} s, {len(order)} symbols per channel")
streams = []
for ch in range(c):
    d = (co[ch].reshape(-1)[order].astype(np.int64) - pr[ch].reshape(-1)[order].astype(np.int64)).astype(np.int32)
    sym = ((d.astype(np.uint32) << 1) ^ (d >> 31).astype(np.uint32)) & 1023
    streams.append((b[ch].reshape(-1)[order].astype(np.uint32) << 10 | sym).astype(np.uint16))
streams = np.stack(streams)
for k in range(2):
    t0 = time.perf_counter()
    frv2 = emit.encode_image_from_streams(w, h, streams, hist, vp, wp)
    print(f"encode_image_from_streams call {k}: {time.perf_counter() - t0:.3f} s, same bytes: {frv2 == frv}; device -> host per plane: {2 * len(order) / 1e6:.1f} MB instead of {9 * co[0].size / 1e6:.1f} MB")
for k in range(2):
    t0 = time.perf_counter()
    out = emit.decode_image(frv)
    print(f"decode_image call {k}: {time.perf_counter() - t0:.3f} s")
