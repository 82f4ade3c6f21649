"""frave_amd -- MI355X (gfx950) implementation of libfri's transform / quantisation /
prediction-histogram hot path behind a C ABI (include/fri_hip.h).

The product is frave_amd/libfri_hip.so (hand-written HIP, see frave_amd/csrc). This Python package is
a thin ctypes binding used by the tests and bench.py; it never computes anything itself and has no CPU
fallback: without the built library it raises, without a gfx950 GPU Context() raises.
"""
from .api import (  # noqa: F401
    NONE,
    Context,
    FriHipError,
    Multi,
    Plan,
    build_library,
    fit_value_params,
    fit_width_params,
    library_path,
    load_library,
    shard_images,
)
