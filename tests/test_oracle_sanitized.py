"""Runs the oracle's known-answer workload under AddressSanitizer + UBSan (CPU build only; GPU ASan is unavailable on this
pool). A separate process loads the instrumented library with libasan preloaded."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes, sys, numpy as np
sys.path.insert(0, %(root)r)
from oracle import fri_oracle as O
O._SO = %(so)r
from tests.common import KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, kat_image, gen_image
for (w, h, c) in [(10, 10, 3), (64, 48, 3), (1, 300, 1), (130, 75, 1)]:
    img = kat_image(w, h, c) if c == 3 else gen_image('noise', w, h, c, 1)
    W = O.Wavelet(img, h, w, c)
    W.quantize(np.ones(32, np.int32))
    for ch in range(c):
        W.predict(ch, KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS)
    W.neighbour_values(0)
    if w > 1:  # a 1-pixel-wide image is not covered by the reference's BFS lattice (DESIGN.md section 8), hence not lossless there
        assert np.array_equal(W.to_raster(), img.reshape(-1))
    for lvl in range(9):
        n = len(W.sorted_level(lvl))  # the reference's own assertion (wavelet_transform.rs:701) holds for the ordinary shapes
        assert c == 1 or n == W.num_cells << lvl
    W.close()
print('sanitized run ok')
"""


def test_oracle_under_asan_ubsan():
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    so = os.path.join(ROOT, "oracle", "libfri_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "so": so}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "sanitized run ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
