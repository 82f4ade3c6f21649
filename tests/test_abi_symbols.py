"""The C ABI libraries load on a machine without a GPU and export every symbol include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="fri_hip.h", prefix="fri_hip_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_every_header_under_include_is_covered():
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["fri_emit.h", "fri_hip.h"]


def test_emit_library_exports_every_declared_symbol():
    import frave_amd.emit as emit

    lib = emit.load_library()
    names = declared_symbols("fri_emit.h", "fri_emit_")
    assert len(names) == 9
    for name in names:
        assert hasattr(lib, name), name
        assert ctypes.cast(getattr(lib, name), ctypes.c_void_p).value


def test_header_and_binding_agree():
    import frave_amd.api as api

    assert declared_symbols() == sorted(api.SYMBOLS)


def test_library_exports_every_declared_symbol():
    import frave_amd

    lib = frave_amd.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), name
        assert ctypes.cast(getattr(lib, name), ctypes.c_void_p).value


def test_strerror_and_version():
    import frave_amd

    lib = frave_amd.load_library()
    assert b"gfx950" in lib.fri_hip_version()
    assert lib.fri_hip_strerror(0) == b"ok"
    for code in range(-6, 0):
        assert lib.fri_hip_strerror(code) not in (b"", b"unknown error")
    assert lib.fri_hip_strerror(-99) == b"unknown error"


def test_no_cpu_fallback():
    """Without a GPU the context cannot be created and a host-only plan refuses to compute: nothing silently runs on the CPU."""
    import numpy as np
    import torch

    import frave_amd

    if torch.cuda.is_available():
        pytest.skip("GPU present: this check is about the CPU-only container")
    with pytest.raises(frave_amd.FriHipError) as e:
        frave_amd.Context(0)
    assert e.value.code == -3
    plan = frave_amd.Plan(None, 64, 48, 3)
    with pytest.raises(frave_amd.FriHipError) as e:
        plan.transform_quant(np.zeros((48, 64, 3), np.uint8))
    assert e.value.code == -3
    with pytest.raises(frave_amd.FriHipError) as e:
        plan.inverse_transform(np.zeros((3, plan.num_cells, 512), np.int32))
    assert e.value.code == -3


def test_plan_argument_errors():
    import frave_amd

    for w, h, c in [(0, 10, 3), (10, 0, 1), (10, 10, 2), (10, 10, 4)]:
        with pytest.raises(frave_amd.FriHipError) as e:
            frave_amd.Plan(None, w, h, c)
        assert e.value.code == -1
    with pytest.raises(frave_amd.FriHipError):  # larger than the reference's u32 pixel index (images.rs:94)
        frave_amd.Plan(None, 65536, 65536, 3)


def test_rust_extern_blocks_are_generated_from_the_headers():
    """integration/hip_sys.rs and emit_sys.rs (the `extern "C"` blocks of the Rust side; unverifiable here: no Rust toolchain) are generated from
    include/*.h by tools/gen_hip_sys.py: the committed files are current and declare every symbol the headers declare."""
    import subprocess
    import sys

    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_hip_sys.py"), "--check"]).returncode == 0
    for header, prefix, rs in (("fri_hip.h", "fri_hip_", "hip_sys.rs"), ("fri_emit.h", "fri_emit_", "emit_sys.rs")):
        text = open(os.path.join(ROOT, "integration", rs)).read()
        assert sorted(re.findall(r"pub fn (" + prefix + r"[a-z0-9_]+)\(", text)) == declared_symbols(header, prefix)
