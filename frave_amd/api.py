"""ctypes binding of include/fri_hip.h. Plumbing only -- every compute call goes to libfri_hip.so."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRI_HIP_LIBRARY: path of an alternative build of the same library (kernel experiments); default = the in-tree build
_SO = os.environ.get("FRI_HIP_LIBRARY") or os.path.join(_HERE, "libfri_hip.so")
NONE = -(2 ** 31)

# every symbol include/fri_hip.h declares (tests/test_abi_symbols.py checks the header against this list)
SYMBOLS = [
    "fri_hip_strerror", "fri_hip_version", "fri_hip_ctx_create", "fri_hip_ctx_destroy", "fri_hip_backend",
    "fri_hip_last_hip_error", "fri_hip_plan_create", "fri_hip_plan_destroy", "fri_hip_plan_num_cells",
    "fri_hip_plan_num_bfs_cells", "fri_hip_plan_num_interior_cells", "fri_hip_plan_coef_count", "fri_hip_plan_pixel_bytes",
    "fri_hip_plan_centers", "fri_hip_plan_valid_mask", "fri_hip_plan_num_some", "fri_hip_plan_neighbour_cells",
    "fri_hip_plan_neighbour_table", "fri_hip_plan_tiling", "fri_hip_plan_tile_table", "fri_hip_transform_quant", "fri_hip_transform_quant_dev",
    "fri_hip_transform_quant_batch_dev", "fri_hip_transform_quant_batch", "fri_hip_predict_histogram",
    "fri_hip_predict_histogram_dev", "fri_hip_fit_value_sums", "fri_hip_fit_value_sums_dev", "fri_hip_fit_width_sums",
    "fri_hip_fit_width_sums_dev", "fri_hip_inverse_transform", "fri_hip_inverse_transform_dev",
    "fri_hip_time_transform_quant_dev", "fri_hip_plan_read_trace", "fri_hip_plan_inverse_lists",
    "fri_hip_shard_size", "fri_hip_shard_image", "fri_hip_multi_create", "fri_hip_multi_destroy", "fri_hip_multi_num_devices",
    "fri_hip_multi_plan", "fri_hip_multi_transform_quant", "fri_hip_predict_histogram_batch_dev", "fri_hip_fit_value_sums_batch_dev",
    "fri_hip_fit_width_sums_batch_dev", "fri_hip_solve6", "fri_hip_fit_value_params", "fri_hip_fit_width_params", "fri_hip_encode_image",
    "fri_hip_encode_image_dev", "fri_hip_inverse_transform_batch_dev", "fri_hip_predict_image", "fri_hip_predict_image_dev",
    "fri_hip_fit_params_batch_dev", "fri_hip_encode_image_batch_dev", "fri_hip_fit_value_params_batch_dev", "fri_hip_fit_width_params_batch_dev",
    "fri_hip_plan_assume_forward_coefficients", "fri_hip_encode_image_batch", "fri_hip_multi_encode_image",
    "fri_hip_plan_set_stream_order", "fri_hip_symbol_stream_batch_dev", "fri_hip_encode_image_symbols", "fri_hip_encode_symbols_batch_dev",
    "fri_hip_plan_set_dequantiser", "fri_hip_plan_tune_forward", "fri_hip_time_transform_quant_streams_dev",
]


class FriHipError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = load_library().fri_hip_strerror(code).decode()
        super().__init__(f"{where}: {msg} ({code}){': ' + detail if detail else ''}")


def library_path():
    return _SO


def build_library(force=False):
    """Compile frave_amd/libfri_hip.so for gfx950 with hipcc (works without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    if force and os.path.exists(_SO):
        os.remove(_SO)
    subprocess.check_call(["make", "-s", "-C", src_dir])
    return _SO


_lib = None


def load_library():
    """Load libfri_hip.so. Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise FileNotFoundError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                                "frave_amd has no CPU fallback.")
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's). If this
    # library were loaded first the dynamic linker would bind torch to /opt/rocm's copy later and torch then finds no
    # GPU; importing torch first makes both share torch's copy. Without torch (C++ callers) /opt/rocm's is used.
    if os.environ.get("FRI_HIP_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(_SO)
    if os.environ.get("FRI_HIP_LIBRARY"):
        # an alternative build selected for an A/B measurement (tools/ab_lib.py) may predate the newest entry points: give those a stub that raises
        # when called, so that the older build still loads. The in-tree library gets no such leniency (tests/test_abi_symbols.py).
        def _missing(name):
            def stub(*a):
                raise AttributeError(f"{_SO} does not export {name}")
            return stub
        for name in SYMBOLS:
            try:
                getattr(L, name)
            except AttributeError:
                setattr(L, name, _missing(name))
    vp, u32, i32, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_size_t
    L.fri_hip_strerror.restype = C.c_char_p
    L.fri_hip_strerror.argtypes = [i32]
    L.fri_hip_version.restype = C.c_char_p
    L.fri_hip_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.fri_hip_ctx_destroy.argtypes = [vp]
    L.fri_hip_backend.restype = C.c_char_p
    L.fri_hip_backend.argtypes = [vp]
    L.fri_hip_last_hip_error.restype = C.c_char_p
    L.fri_hip_last_hip_error.argtypes = [vp]
    L.fri_hip_plan_create.argtypes = [vp, u32, u32, u32, C.POINTER(vp)]
    L.fri_hip_plan_destroy.argtypes = [vp]
    for n in ("num_cells", "num_bfs_cells", "num_interior_cells"):
        f = getattr(L, "fri_hip_plan_" + n)
        f.restype, f.argtypes = u32, [vp]
    for n in ("coef_count", "pixel_bytes"):
        f = getattr(L, "fri_hip_plan_" + n)
        f.restype, f.argtypes = sz, [vp]
    L.fri_hip_plan_num_some.restype = C.c_uint64
    L.fri_hip_plan_num_some.argtypes = [vp]
    for n in ("centers", "valid_mask", "neighbour_cells", "neighbour_table", "tiling"):
        getattr(L, "fri_hip_plan_" + n).argtypes = [vp, vp]
    L.fri_hip_plan_tile_table.argtypes = [vp, vp, vp, vp]
    L.fri_hip_transform_quant.argtypes = [vp, vp, vp, vp]
    L.fri_hip_transform_quant_dev.argtypes = [vp, vp, vp, vp, vp]
    L.fri_hip_transform_quant_batch_dev.argtypes = [vp, u32, vp, sz, vp, vp, sz, vp]
    L.fri_hip_transform_quant_batch.argtypes = [vp, u32, vp, vp, vp]
    L.fri_hip_predict_histogram.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp]
    L.fri_hip_predict_histogram_dev.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp, vp, vp]
    L.fri_hip_fit_value_sums.argtypes = [vp, vp, u32, vp]
    L.fri_hip_fit_value_sums_dev.argtypes = [vp, vp, u32, vp, vp]
    L.fri_hip_fit_width_sums.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    L.fri_hip_fit_width_sums_dev.argtypes = [vp, vp, u32, vp, vp, vp, vp]
    L.fri_hip_inverse_transform.argtypes = [vp, vp, vp, vp]
    L.fri_hip_inverse_transform_dev.argtypes = [vp, vp, vp, vp, vp]
    L.fri_hip_plan_read_trace.argtypes = [vp, vp]
    L.fri_hip_plan_inverse_lists.argtypes = [vp, vp]
    L.fri_hip_time_transform_quant_dev.argtypes = [vp, u32, vp, sz, vp, vp, sz, u32, vp, C.POINTER(C.c_double)]
    L.fri_hip_time_transform_quant_streams_dev.argtypes = [vp, u32, vp, sz, vp, vp, sz, u32, u32, C.POINTER(C.c_double)]
    L.fri_hip_plan_tune_forward.argtypes = [vp, u32, C.c_char_p, sz]
    L.fri_hip_shard_size.restype, L.fri_hip_shard_size.argtypes = u32, [u32, u32, u32]
    L.fri_hip_shard_image.restype, L.fri_hip_shard_image.argtypes = u32, [u32, u32, u32]
    L.fri_hip_multi_create.argtypes = [vp, u32, u32, u32, u32, C.POINTER(vp)]
    L.fri_hip_multi_destroy.argtypes = [vp]
    L.fri_hip_multi_num_devices.restype, L.fri_hip_multi_num_devices.argtypes = u32, [vp]
    L.fri_hip_multi_plan.restype, L.fri_hip_multi_plan.argtypes = vp, [vp, u32]
    L.fri_hip_multi_transform_quant.argtypes = [vp, u32, vp, vp, vp]
    L.fri_hip_predict_histogram_batch_dev.argtypes = [vp, u32, vp, sz, vp, vp, vp, sz, vp, vp, vp]
    L.fri_hip_fit_value_sums_batch_dev.argtypes = [vp, u32, vp, sz, vp, vp]
    L.fri_hip_fit_width_sums_batch_dev.argtypes = [vp, u32, vp, sz, vp, vp, vp, vp]
    L.fri_hip_solve6.restype, L.fri_hip_solve6.argtypes = None, [vp, vp, vp]
    L.fri_hip_fit_value_params.restype, L.fri_hip_fit_value_params.argtypes = None, [vp, vp]
    L.fri_hip_fit_width_params.restype, L.fri_hip_fit_width_params.argtypes = None, [vp, vp, vp, vp]
    L.fri_hip_encode_image.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.fri_hip_encode_image_dev.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.fri_hip_inverse_transform_batch_dev.argtypes = [vp, u32, vp, sz, vp, vp, sz, vp]
    L.fri_hip_predict_image.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.fri_hip_predict_image_dev.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.fri_hip_fit_params_batch_dev.argtypes = [vp, u32, vp, sz, vp, vp, vp]
    L.fri_hip_plan_assume_forward_coefficients.argtypes = [vp, i32]
    L.fri_hip_plan_set_dequantiser.argtypes = [vp, i32]
    L.fri_hip_plan_set_stream_order.argtypes = [vp, vp, C.c_uint64]
    L.fri_hip_symbol_stream_batch_dev.argtypes = [vp, u32, vp, sz, vp, vp, sz, vp, sz, vp]
    L.fri_hip_encode_image_symbols.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.fri_hip_encode_symbols_batch_dev.argtypes = [vp, u32, vp, sz, vp, i32, vp, vp, sz, vp, sz, vp, sz, vp, vp, vp, vp]
    L.fri_hip_encode_image_batch.argtypes = [vp, u32, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.fri_hip_multi_encode_image.argtypes = [vp, u32, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    L.fri_hip_fit_value_params_batch_dev.argtypes = [vp, u32, vp, vp, vp]
    L.fri_hip_fit_width_params_batch_dev.argtypes = [vp, u32, vp, vp, vp, vp]
    L.fri_hip_encode_image_batch_dev.argtypes = [vp, u32, vp, sz, vp, i32, vp, vp, sz, vp, vp, sz, vp, vp, vp, vp]
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _q(q):
    q = np.ones(32, np.int32) if q is None else np.ascontiguousarray(q, np.int32)
    assert q.size == 32
    return q


def _untriangle(tri, n):
    """[g][n(n+1)/2] upper triangle (row major) -> [g][n][n] symmetric."""
    out = np.zeros((tri.shape[0], n, n), tri.dtype)
    iu = np.triu_indices(n)
    for g in range(tri.shape[0]):
        out[g][iu] = tri[g]
        out[g] = out[g] + out[g].T - np.diag(np.diag(out[g]))
    return out


def solve_normal_equations(ata, atb, eps=1e-12):
    """Minimum-norm least-squares solution from normal equations (what lstsq's SVD returns, up to rounding):
    x = pinv(A^T A) A^T b via the symmetric eigen-decomposition; eigenvalues <= eps * max are treated as zero."""
    ata = np.asarray(ata, np.float64)
    atb = np.asarray(atb, np.float64)
    lam, vec = np.linalg.eigh(ata)
    keep = lam > eps * max(lam.max(), 0.0)
    inv = np.where(keep, 1.0 / np.where(keep, lam, 1.0), 0.0)
    return (vec * inv) @ (vec.T @ atb)


def _check(rc, where, ctx=None):
    if rc != 0:
        detail = ""
        if ctx is not None and ctx._h:
            detail = load_library().fri_hip_last_hip_error(ctx._h).decode()
        raise FriHipError(rc, where, detail)


def shard_images(n_images, shard, n_shards):
    """Global indices of the images of `shard` (fri_hip_shard_size / fri_hip_shard_image: image i -> shard i mod n_shards)."""
    L = load_library()
    return [L.fri_hip_shard_image(k, shard, n_shards) for k in range(L.fri_hip_shard_size(n_images, shard, n_shards))]


def fit_value_params(gram_tri):
    """fri_hip_fit_value_params: gram_tri int64 [3][28] (upper triangles) -> float32 [3][6]."""
    g = np.ascontiguousarray(gram_tri, np.int64).reshape(3, 28)
    out = np.empty((3, 6), np.float32)
    load_library().fri_hip_fit_value_params(_p(g), _p(out))
    return out


def fit_width_params(wtw_tri, wtr, rows):
    """fri_hip_fit_width_params: wtw_tri int64 [3][21], wtr float64 [3][6], rows uint64 [3] -> float32 [3][6]."""
    w = np.ascontiguousarray(wtw_tri, np.int64).reshape(3, 21)
    r = np.ascontiguousarray(wtr, np.float64).reshape(3, 6)
    n = np.ascontiguousarray(rows, np.uint64).reshape(3)
    out = np.empty((3, 6), np.float32)
    load_library().fri_hip_fit_width_params(_p(w), _p(r), _p(n), _p(out))
    return out


def _encode_batch(fn, handle, where, ctx, images, channels, num_cells, qmatrix, fit, params, want_bucket, want_prediction):
    """Shared marshalling of fri_hip_encode_image_batch / fri_hip_multi_encode_image: lists of per-image arrays."""
    imgs = [np.ascontiguousarray(i, np.uint8).reshape(-1) for i in images]
    n, c, f = len(imgs), channels, num_cells
    if params is None:
        par = [np.zeros((c, 2, 3, 6), np.float32) for _ in imgs]
    else:
        par = [np.ascontiguousarray(p, np.float32).reshape(c, 2, 3, 6).copy() for p in params]
    coefs = [np.empty((c, f, 512), np.int32) for _ in imgs]
    bucket = [np.empty((c, f, 512), np.uint8) for _ in imgs] if want_bucket else None
    pred = [np.empty((c, f, 512), np.int32) for _ in imgs] if want_prediction else None
    hist = [np.empty((c, 10, 1024), np.uint32) for _ in imgs]
    oob = [np.zeros(c, np.uint64) for _ in imgs]
    arr = lambda xs: (C.c_void_p * n)(*[x.ctypes.data for x in xs]) if xs is not None else None
    q = _q(qmatrix)
    _check(fn(handle, n, arr(imgs), _p(q), 1 if fit else 0, arr(par), arr(coefs), arr(bucket), arr(pred), arr(hist), arr(oob)), where, ctx)
    return coefs, par, bucket, pred, hist, oob


class Multi:
    """fri_hip_multi: one process driving several GPUs, one ctx + plan per device, image i on devices[i mod len(devices)]."""

    def __init__(self, devices, width, height, channels):
        self._h = None
        h = C.c_void_p()
        dev = (C.c_int * len(devices))(*devices)
        _check(load_library().fri_hip_multi_create(dev, len(devices), width, height, channels, C.byref(h)), "fri_hip_multi_create")
        self._h = h
        self.channels = channels
        L = load_library()
        self.num_devices = L.fri_hip_multi_num_devices(h)
        self.num_cells = L.fri_hip_plan_num_cells(L.fri_hip_multi_plan(h, 0))

    def transform_quant(self, images, qmatrix=None):
        imgs = [np.ascontiguousarray(i, np.uint8).reshape(-1) for i in images]
        outs = [np.empty((self.channels, self.num_cells, 512), np.int32) for _ in imgs]
        n = len(imgs)
        pin = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        pout = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        q = _q(qmatrix)
        _check(load_library().fri_hip_multi_transform_quant(self._h, n, pin, _p(q), pout), "fri_hip_multi_transform_quant")
        return outs

    def encode_image(self, images, qmatrix=None, fit=True, params=None, want_bucket=True, want_prediction=True):
        """fri_hip_multi_encode_image: per image (coefs, params [C][2][3][6], bucket, prediction, hist, oob), image i on device i mod N."""
        return _encode_batch(load_library().fri_hip_multi_encode_image, self._h, "fri_hip_multi_encode_image", None, images, self.channels, self.num_cells, qmatrix, fit, params,
                             want_bucket, want_prediction)

    def close(self):
        if self._h:
            load_library().fri_hip_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """fri_hip_ctx: one per (host thread, GPU). Raises FriHipError(NO_DEVICE) without a gfx950 GPU."""

    def __init__(self, device=0):
        self._h = None
        h = C.c_void_p()
        _check(load_library().fri_hip_ctx_create(device, C.byref(h)), "fri_hip_ctx_create")
        self._h = h
        self.device = device

    @property
    def backend(self):
        return load_library().fri_hip_backend(self._h).decode()

    def close(self):
        if self._h:
            load_library().fri_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """fri_hip_plan: geometry of one (width, height, channels). ctx=None gives a host-only plan (getters only)."""

    def __init__(self, ctx, width, height, channels):
        self._h = None
        self.ctx = ctx
        self.width, self.height, self.channels = width, height, channels
        h = C.c_void_p()
        _check(load_library().fri_hip_plan_create(ctx._h if ctx else None, width, height, channels, C.byref(h)), "fri_hip_plan_create", ctx)
        self._h = h
        L = load_library()
        self.num_cells = L.fri_hip_plan_num_cells(h)
        self.num_bfs_cells = L.fri_hip_plan_num_bfs_cells(h)
        self.num_interior_cells = L.fri_hip_plan_num_interior_cells(h)
        self.coef_count = L.fri_hip_plan_coef_count(h)
        self.pixel_bytes = L.fri_hip_plan_pixel_bytes(h)
        self.num_some = L.fri_hip_plan_num_some(h)

    def close(self):
        if self._h:
            load_library().fri_hip_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream_order(self, order=None):
        """fri_hip_plan_set_stream_order; order = None builds it with the host emitter library (frave_amd.emit.stream_order). Returns the order."""
        if order is None:
            from . import emit

            order = emit.stream_order(self.centers(), self.valid_mask())
        order = np.ascontiguousarray(order, np.uint32)
        _check(load_library().fri_hip_plan_set_stream_order(self._h, _p(order), order.size), "fri_hip_plan_set_stream_order", self.ctx)
        return order

    def symbol_stream_batch_dev(self, n_planes, d_coefs, coef_stride, d_bucket, d_prediction, out_stride, d_symbols, symbol_stride, stream=0):
        _check(load_library().fri_hip_symbol_stream_batch_dev(self._h, n_planes, d_coefs, coef_stride, d_bucket, d_prediction, out_stride, d_symbols, symbol_stride, stream),
               "fri_hip_symbol_stream_batch_dev", self.ctx)

    def encode_symbols_batch_dev(self, n_images, d_pixels, pixel_stride, qmatrix, fit, d_params, d_coefs, coef_stride, d_node_words, word_stride, d_symbols, symbol_stride,
                                 d_hist, d_oob, d_fit_range=0, stream=0):
        """fri_hip_encode_symbols_batch_dev: forward -> [fit] -> scan (halfword form) -> gather into stream order; everything in device memory, asynchronous.
        d_coefs = 0 / None: the coefficients stay inside the chain as the plan's compact int16 planes (same streams, histograms and parameters; faster);
        d_node_words = 0 / None as well: the scan writes the streams itself, no node words, no gather kernel (faster still)."""
        q = _q(qmatrix)
        _check(load_library().fri_hip_encode_symbols_batch_dev(self._h, n_images, d_pixels, pixel_stride, _p(q), 1 if fit else 0, d_params, d_coefs, coef_stride, d_node_words,
                                                               word_stride, d_symbols, symbol_stride, d_hist, d_oob, d_fit_range, stream), "fri_hip_encode_symbols_batch_dev", self.ctx)

    def set_dequantiser(self, multiply):
        """fri_hip_plan_set_dequantiser: False = the reference's dividing quantization::decode (default), True = coefficient x qmatrix[layer]."""
        _check(load_library().fri_hip_plan_set_dequantiser(self._h, 1 if multiply else 0), "fri_hip_plan_set_dequantiser")

    def assume_forward_coefficients(self, on=True):
        """fri_hip_plan_assume_forward_coefficients: the predict entry points then skip the exact-kernel guard launch."""
        _check(load_library().fri_hip_plan_assume_forward_coefficients(self._h, 1 if on else 0), "fri_hip_plan_assume_forward_coefficients")

    # ---- getters -------------------------------------------------------------------------------
    def centers(self):
        out = np.empty((self.num_cells, 2), np.int32)
        _check(load_library().fri_hip_plan_centers(self._h, _p(out)), "fri_hip_plan_centers")
        return out

    def valid_mask(self):
        out = np.empty((self.num_cells, 16), np.uint32)
        _check(load_library().fri_hip_plan_valid_mask(self._h, _p(out)), "fri_hip_plan_valid_mask")
        return out

    def valid_bits(self):
        """bool [F][512] expansion of valid_mask()."""
        m = self.valid_mask()
        return ((m[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).astype(bool).reshape(self.num_cells, 512)

    def neighbour_cells(self):
        out = np.empty((self.num_cells, 8), np.int32)
        _check(load_library().fri_hip_plan_neighbour_cells(self._h, _p(out)), "fri_hip_plan_neighbour_cells")
        return out

    def tiling(self):
        out = np.empty(8, np.int32)
        _check(load_library().fri_hip_plan_tiling(self._h, _p(out)), "fri_hip_plan_tiling")
        return dict(zip(("n_wg", "n_tiles", "lds_pitch", "lds_rows", "max_tile_cells", "band_rows", "cells_per_tile", "cells_per_wg"), (int(v) for v in out)))

    def inverse_lists(self):
        out = np.zeros(5, np.uint64)
        _check(load_library().fri_hip_plan_inverse_lists(self._h, _p(out)), "fri_hip_plan_inverse_lists")
        return dict(zip(("built", "quads", "dwords", "part_bytes", "rect_bytes"), (int(v) for v in out)))

    def read_trace(self):
        """[n_wg, 16] uint64 time stamps (100 MHz ticks) of the last forward/inverse launch; needs FRI_HIP_TRACE=1 at plan creation."""
        out = np.zeros((self.tiling()["n_wg"], 16), np.uint64)
        _check(load_library().fri_hip_plan_read_trace(self._h, _p(out)), "fri_hip_plan_read_trace", self.ctx)
        return out

    def tile_table(self):
        t = self.tiling()
        tiles = np.empty((t["n_tiles"], 6), np.int32)
        cells = np.empty(self.num_cells, np.int32)
        wg = np.empty(t["n_wg"] + 1, np.int32)
        _check(load_library().fri_hip_plan_tile_table(self._h, _p(tiles), _p(cells), _p(wg)), "fri_hip_plan_tile_table")
        return tiles, cells, wg

    def neighbour_table(self):
        out = np.empty((512, 6), np.uint16)
        _check(load_library().fri_hip_plan_neighbour_table(self._h, _p(out)), "fri_hip_plan_neighbour_table")
        return out

    # ---- host-pointer entry points -------------------------------------------------------------
    def transform_quant(self, pixels, qmatrix=None):
        px = np.ascontiguousarray(pixels, np.uint8).reshape(-1)
        assert px.size == self.pixel_bytes
        out = np.empty((self.channels, self.num_cells, 512), np.int32)
        q = _q(qmatrix)
        _check(load_library().fri_hip_transform_quant(self._h, _p(px), _p(q), _p(out)), "fri_hip_transform_quant", self.ctx)
        return out

    def transform_quant_batch(self, images, qmatrix=None):
        imgs = [np.ascontiguousarray(i, np.uint8).reshape(-1) for i in images]
        outs = [np.empty((self.channels, self.num_cells, 512), np.int32) for _ in imgs]
        n = len(imgs)
        pin = (C.c_void_p * n)(*[i.ctypes.data for i in imgs])
        pout = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        q = _q(qmatrix)
        _check(load_library().fri_hip_transform_quant_batch(self._h, n, pin, _p(q), pout), "fri_hip_transform_quant_batch", self.ctx)
        return outs

    def encode_image_batch(self, images, qmatrix=None, fit=True, params=None, want_bucket=True, want_prediction=True):
        """fri_hip_encode_image_batch: the per-image encode loop for host buffers on this plan's device (lists per image, as Multi.encode_image)."""
        return _encode_batch(load_library().fri_hip_encode_image_batch, self._h, "fri_hip_encode_image_batch", self.ctx, images, self.channels, self.num_cells, qmatrix, fit, params,
                             want_bucket, want_prediction)

    def predict_histogram(self, coefs, channel, value_params, width_params, want_bucket=True, want_prediction=True):
        """(bucket, prediction, hist, n_out_of_alphabet); an output that is not wanted is passed as NULL and returned as None."""
        co = np.ascontiguousarray(coefs, np.int32)
        assert co.size == self.coef_count
        vp = np.ascontiguousarray(value_params, np.float32).reshape(3, 6)
        wp = np.ascontiguousarray(width_params, np.float32).reshape(3, 6)
        bucket = np.empty((self.num_cells, 512), np.uint8) if want_bucket else None
        pred = np.empty((self.num_cells, 512), np.int32) if want_prediction else None
        hist = np.empty((10, 1024), np.uint32)
        oob = C.c_uint64(0)
        _check(load_library().fri_hip_predict_histogram(self._h, _p(co), channel, _p(vp), _p(wp), _p(bucket) if want_bucket else None,
                                                        _p(pred) if want_prediction else None, _p(hist), C.addressof(oob)),
               "fri_hip_predict_histogram", self.ctx)
        return bucket, pred, hist, oob.value

    def fit_value_sums(self, coefs, channel):
        """gram[3][7][7] (full symmetric int64) of u = [v0..v5, value] per layer group."""
        co = np.ascontiguousarray(coefs, np.int32)
        assert co.size == self.coef_count
        tri = np.empty((3, 28), np.int64)
        _check(load_library().fri_hip_fit_value_sums(self._h, _p(co), channel, _p(tri)), "fri_hip_fit_value_sums", self.ctx)
        return _untriangle(tri, 7)

    def fit_width_sums(self, coefs, channel, value_params):
        """(wtw[3][6][6] int64 over the Some rows, wtr[3][6] float64, rows[3])."""
        co = np.ascontiguousarray(coefs, np.int32)
        assert co.size == self.coef_count
        vp = np.ascontiguousarray(value_params, np.float32).reshape(3, 6)
        tri = np.empty((3, 21), np.int64)
        wtr = np.empty((3, 6), np.float64)
        rows = np.empty(3, np.uint64)
        _check(load_library().fri_hip_fit_width_sums(self._h, _p(co), channel, _p(vp), _p(tri), _p(wtr), _p(rows)), "fri_hip_fit_width_sums", self.ctx)
        return _untriangle(tri, 6), wtr, rows

    def inverse_transform(self, coefs, qmatrix=None):
        co = np.ascontiguousarray(coefs, np.int32)
        assert co.size == self.coef_count
        out = np.empty(self.pixel_bytes, np.uint8)
        q = _q(qmatrix)
        _check(load_library().fri_hip_inverse_transform(self._h, _p(co), _p(q), _p(out)), "fri_hip_inverse_transform", self.ctx)
        return out

    # ---- device-pointer entry points (pointers are ints, e.g. torch.Tensor.data_ptr()) ---------
    def transform_quant_dev(self, d_pixels, d_coefs, qmatrix=None, stream=0, n_images=1, pixel_stride=0, coef_stride=0):
        q = _q(qmatrix)
        _check(load_library().fri_hip_transform_quant_batch_dev(self._h, n_images, d_pixels, pixel_stride, _p(q), d_coefs, coef_stride, stream),
               "fri_hip_transform_quant_batch_dev", self.ctx)

    def predict_histogram_dev(self, d_coefs, channel, value_params, width_params, d_bucket, d_prediction, d_hist, d_oob, stream=0):
        vp = np.ascontiguousarray(value_params, np.float32).reshape(3, 6)
        wp = np.ascontiguousarray(width_params, np.float32).reshape(3, 6)
        _check(load_library().fri_hip_predict_histogram_dev(self._h, d_coefs, channel, _p(vp), _p(wp), d_bucket, d_prediction, d_hist, d_oob, stream),
               "fri_hip_predict_histogram_dev", self.ctx)

    def fit_value_sums_dev(self, d_coefs, channel, d_gram, stream=0):
        _check(load_library().fri_hip_fit_value_sums_dev(self._h, d_coefs, channel, d_gram, stream), "fri_hip_fit_value_sums_dev", self.ctx)

    def fit_width_sums_dev(self, d_coefs, channel, value_params, d_wtw, d_wtr, stream=0):
        vp = np.ascontiguousarray(value_params, np.float32).reshape(3, 6)
        _check(load_library().fri_hip_fit_width_sums_dev(self._h, d_coefs, channel, _p(vp), d_wtw, d_wtr, stream), "fri_hip_fit_width_sums_dev", self.ctx)

    def encode_image(self, pixels, qmatrix=None, fit=True, value_params=None, width_params=None, want_bucket=True, want_prediction=True):
        """fri_hip_encode_image: (coefs [C][F][512], value_params [C][3][6], width_params [C][3][6], bucket, prediction, hist [C][10][1024], oob [C])."""
        px = np.ascontiguousarray(pixels, np.uint8).reshape(-1)
        assert px.size == self.pixel_bytes
        c, f = self.channels, self.num_cells
        vp = np.zeros((c, 3, 6), np.float32) if value_params is None else np.ascontiguousarray(value_params, np.float32).reshape(c, 3, 6).copy()
        wp = np.zeros((c, 3, 6), np.float32) if width_params is None else np.ascontiguousarray(width_params, np.float32).reshape(c, 3, 6).copy()
        coefs = np.empty((c, f, 512), np.int32)
        bucket = np.empty((c, f, 512), np.uint8) if want_bucket else None
        pred = np.empty((c, f, 512), np.int32) if want_prediction else None
        hist = np.empty((c, 10, 1024), np.uint32)
        oob = np.zeros(c, np.uint64)
        q = _q(qmatrix)
        _check(load_library().fri_hip_encode_image(self._h, _p(px), _p(q), 1 if fit else 0, _p(vp), _p(wp), _p(coefs), _p(bucket) if want_bucket else None,
                                                   _p(pred) if want_prediction else None, _p(hist), _p(oob)), "fri_hip_encode_image", self.ctx)
        return coefs, vp, wp, bucket, pred, hist, oob

    def encode_image_symbols(self, pixels, qmatrix=None, fit=True, value_params=None, width_params=None):
        """fri_hip_encode_image_symbols: (symbols uint16 [C][num_some], value_params, width_params, hist [C][10][1024], oob [C]); needs set_stream_order()."""
        px = np.ascontiguousarray(pixels, np.uint8).reshape(-1)
        assert px.size == self.pixel_bytes
        c = self.channels
        vp = np.zeros((c, 3, 6), np.float32) if value_params is None else np.ascontiguousarray(value_params, np.float32).reshape(c, 3, 6).copy()
        wp = np.zeros((c, 3, 6), np.float32) if width_params is None else np.ascontiguousarray(width_params, np.float32).reshape(c, 3, 6).copy()
        sym = np.empty((c, self.num_some), np.uint16)
        hist = np.empty((c, 10, 1024), np.uint32)
        oob = np.zeros(c, np.uint64)
        q = _q(qmatrix)
        _check(load_library().fri_hip_encode_image_symbols(self._h, _p(px), _p(q), 1 if fit else 0, _p(vp), _p(wp), _p(sym), _p(hist), _p(oob)), "fri_hip_encode_image_symbols", self.ctx)
        return sym, vp, wp, hist, oob

    def predict_image(self, coefs, fit=True, value_params=None, width_params=None):
        """fri_hip_predict_image: (value_params, width_params, bucket [C][F][512], prediction, hist [C][10][1024], oob [C])."""
        co = np.ascontiguousarray(coefs, np.int32)
        assert co.size == self.coef_count
        c, f = self.channels, self.num_cells
        vp = np.zeros((c, 3, 6), np.float32) if value_params is None else np.ascontiguousarray(value_params, np.float32).reshape(c, 3, 6).copy()
        wp = np.zeros((c, 3, 6), np.float32) if width_params is None else np.ascontiguousarray(width_params, np.float32).reshape(c, 3, 6).copy()
        bucket, pred = np.empty((c, f, 512), np.uint8), np.empty((c, f, 512), np.int32)
        hist, oob = np.empty((c, 10, 1024), np.uint32), np.zeros(c, np.uint64)
        _check(load_library().fri_hip_predict_image(self._h, _p(co), 1 if fit else 0, _p(vp), _p(wp), _p(bucket), _p(pred), _p(hist), _p(oob)), "fri_hip_predict_image", self.ctx)
        return vp, wp, bucket, pred, hist, oob

    def encode_image_dev(self, d_pixels, d_coefs, d_bucket, d_prediction, d_hist, d_oob, value_params, width_params, fit=False, qmatrix=None, stream=0):
        """fri_hip_encode_image_dev; value_params / width_params are float32 [C][3][6] numpy arrays (in, or out when fit)."""
        q = _q(qmatrix)
        assert value_params.dtype == np.float32 and width_params.dtype == np.float32 and value_params.size == width_params.size == self.channels * 18
        _check(load_library().fri_hip_encode_image_dev(self._h, d_pixels, _p(q), 1 if fit else 0, _p(value_params), _p(width_params), d_coefs, d_bucket, d_prediction,
                                                       d_hist, d_oob, stream), "fri_hip_encode_image_dev", self.ctx)

    def predict_histogram_batch_dev(self, n_planes, d_coefs, coef_stride, d_params, d_bucket, d_prediction, out_stride, d_hist, d_oob, stream=0):
        _check(load_library().fri_hip_predict_histogram_batch_dev(self._h, n_planes, d_coefs, coef_stride, d_params, d_bucket, d_prediction, out_stride, d_hist, d_oob,
                                                                  stream), "fri_hip_predict_histogram_batch_dev", self.ctx)

    def fit_value_sums_batch_dev(self, n_planes, d_coefs, coef_stride, d_gram, stream=0):
        _check(load_library().fri_hip_fit_value_sums_batch_dev(self._h, n_planes, d_coefs, coef_stride, d_gram, stream), "fri_hip_fit_value_sums_batch_dev", self.ctx)

    def fit_width_sums_batch_dev(self, n_planes, d_coefs, coef_stride, d_params, d_wtw, d_wtr, stream=0):
        _check(load_library().fri_hip_fit_width_sums_batch_dev(self._h, n_planes, d_coefs, coef_stride, d_params, d_wtw, d_wtr, stream),
               "fri_hip_fit_width_sums_batch_dev", self.ctx)

    def fit_value_params_batch_dev(self, n_planes, d_gram, d_params, stream=0):
        _check(load_library().fri_hip_fit_value_params_batch_dev(self._h, n_planes, d_gram, d_params, stream), "fri_hip_fit_value_params_batch_dev", self.ctx)

    def fit_width_params_batch_dev(self, n_planes, d_wtw, d_wtr, d_params, stream=0):
        _check(load_library().fri_hip_fit_width_params_batch_dev(self._h, n_planes, d_wtw, d_wtr, d_params, stream), "fri_hip_fit_width_params_batch_dev", self.ctx)

    def fit_params_batch_dev(self, n_planes, d_coefs, coef_stride, d_params, d_fit_out_of_range=None, stream=0):
        """fri_hip_fit_params_batch_dev: the whole fit (sums + 6 x 6 solves) of n_planes planes on the device, parameters into d_params."""
        _check(load_library().fri_hip_fit_params_batch_dev(self._h, n_planes, d_coefs, coef_stride, d_params, d_fit_out_of_range, stream),
               "fri_hip_fit_params_batch_dev", self.ctx)

    def encode_image_batch_dev(self, n_images, d_pixels, pixel_stride, d_params, d_coefs, coef_stride, d_bucket, d_prediction, out_stride, d_hist, d_oob,
                               fit=True, d_fit_out_of_range=None, qmatrix=None, stream=0):
        """fri_hip_encode_image_batch_dev: K1 -> (fit) -> K2 for n_images images, all in device memory, nothing but enqueues."""
        q = _q(qmatrix)
        _check(load_library().fri_hip_encode_image_batch_dev(self._h, n_images, d_pixels, pixel_stride, _p(q), 1 if fit else 0, d_params, d_coefs, coef_stride,
                                                             d_bucket, d_prediction, out_stride, d_hist, d_oob, d_fit_out_of_range, stream),
               "fri_hip_encode_image_batch_dev", self.ctx)

    def inverse_transform_batch_dev(self, n_images, d_coefs, coef_stride, d_pixels, pixel_stride, qmatrix=None, stream=0):
        q = _q(qmatrix)
        _check(load_library().fri_hip_inverse_transform_batch_dev(self._h, n_images, d_coefs, coef_stride, _p(q), d_pixels, pixel_stride, stream),
               "fri_hip_inverse_transform_batch_dev", self.ctx)

    def inverse_transform_dev(self, d_coefs, d_pixels, qmatrix=None, stream=0):
        q = _q(qmatrix)
        _check(load_library().fri_hip_inverse_transform_dev(self._h, d_coefs, _p(q), d_pixels, stream), "fri_hip_inverse_transform_dev", self.ctx)

    def tune_forward(self, launches=0):
        """fri_hip_plan_tune_forward: measure candidate tilings of the forward kernel on this device and keep the fastest. Returns the report (dict)."""
        import json

        buf = C.create_string_buffer(16384)
        _check(load_library().fri_hip_plan_tune_forward(self._h, launches, buf, len(buf)), "fri_hip_plan_tune_forward", self.ctx)
        try:
            return json.loads(buf.value.decode() or "{}")
        except ValueError:  # a report cut short by the buffer
            return {"raw": buf.value.decode(errors="replace")}

    def time_transform_quant_streams_dev(self, n_images, d_pixels, pixel_stride, d_coefs, coef_stride, iters, n_streams, qmatrix=None):
        q = _q(qmatrix)
        us = C.c_double(0)
        _check(load_library().fri_hip_time_transform_quant_streams_dev(self._h, n_images, d_pixels, pixel_stride, _p(q), d_coefs, coef_stride, iters, n_streams,
                                                                       C.byref(us)), "fri_hip_time_transform_quant_streams_dev", self.ctx)
        return us.value

    def time_transform_quant_dev(self, n_images, d_pixels, pixel_stride, d_coefs, coef_stride, iters, qmatrix=None, stream=0):
        q = _q(qmatrix)
        us = C.c_double(0)
        _check(load_library().fri_hip_time_transform_quant_dev(self._h, n_images, d_pixels, pixel_stride, _p(q), d_coefs, coef_stride, iters, stream,
                                                               C.byref(us)), "fri_hip_time_transform_quant_dev", self.ctx)
        return us.value
