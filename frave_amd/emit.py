"""ctypes binding of frave_amd/libfri_emit.so: the host side of the encode path behind the kernels (symbol order, ANS model,
rANS streams, `frif` container; frave_amd/host/emit.hpp). No GPU involved."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfri_emit.so")
_lib = None


class EmitError(RuntimeError):
    pass


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):  # host-only C++ (g++): build on first use
            import subprocess

            subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "host"), _SO])
        L = C.CDLL(_SO)
        vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
        L.fri_emit_symbol_order.argtypes = [vp, u32, u32, vp]
        L.fri_emit_finalize_context.argtypes = [vp, u32, vp, vp, vp, vp, C.c_char_p, sz]
        L.fri_emit_channel_symbols.argtypes = [vp, u32, vp, vp, vp, vp, vp, vp]
        L.fri_emit_encode_image.argtypes = [u32, u32, u32, vp, u32, vp, vp, vp, vp, vp, vp, vp, sz, vp, C.c_char_p, sz]
        L.fri_emit_check_image.argtypes = [vp, sz, u32, vp, u32, vp, vp, vp, C.c_char_p, sz]
        L.fri_emit_decode_image.argtypes = [vp, sz, vp, vp, sz, vp, C.c_char_p, sz]
        L.fri_emit_rans_selfcheck.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p, sz]
        L.fri_emit_stream_order.argtypes = [vp, u32, vp, vp, vp]
        L.fri_emit_encode_image_from_streams.argtypes = [u32, u32, u32, vp, C.c_uint64, vp, vp, vp, vp, sz, vp, C.c_char_p, sz]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def symbol_order(centers, level):
    """[(cell, heap index)] of `level` in stream order; centers = plan.centers() ([F][2] int32)."""
    c = np.ascontiguousarray(centers, np.int32)
    out = np.empty(len(c) << level, np.uint32)
    if load_library().fri_emit_symbol_order(_p(c), len(c), level, _p(out)) != 0:
        raise EmitError("fri_emit_symbol_order")
    return np.stack([out >> 9, out & 511], axis=1)


def finalize_context(counts, bucket):
    """(freqs, cdf, off_distribution_values, max_freq_bits) of the ANS model built from one context's measured counts."""
    f = np.ascontiguousarray(counts, np.uint32).copy()
    cdf = np.empty(1024, np.uint32)
    off = np.empty(1024, np.uint16)
    n_off, bits = C.c_uint32(0), C.c_uint32(0)
    err = C.create_string_buffer(256)
    rc = load_library().fri_emit_finalize_context(_p(f), bucket, _p(cdf), _p(off), C.addressof(n_off), C.addressof(bits), err, 256)
    if rc != 0:
        raise EmitError(err.value.decode() or f"fri_emit_finalize_context: {rc}")
    return f, cdf, off[: n_off.value].copy(), bits.value


def channel_symbols(centers, coefs, bucket, prediction):
    c = np.ascontiguousarray(centers, np.int32)
    co, b, p = np.ascontiguousarray(coefs, np.int32), np.ascontiguousarray(bucket, np.uint8), np.ascontiguousarray(prediction, np.int32)
    sym = np.empty(len(c) * 512, np.uint16)
    bk = np.empty(len(c) * 512, np.uint8)
    n = C.c_uint64(0)
    if load_library().fri_emit_channel_symbols(_p(c), len(c), _p(co), _p(b), _p(p), _p(sym), _p(bk), C.addressof(n)) != 0:
        raise EmitError("fri_emit_channel_symbols")
    return sym[: n.value].copy(), bk[: n.value].copy()


def encode_image(width, height, centers, coefs, bucket, prediction, hist, value_params, width_params):
    """.frv bytes. coefs/bucket/prediction [C][F][512], hist [C][10][1024], params [C][3][6]."""
    c = np.ascontiguousarray(centers, np.int32)
    co, b, p = np.ascontiguousarray(coefs, np.int32), np.ascontiguousarray(bucket, np.uint8), np.ascontiguousarray(prediction, np.int32)
    h = np.ascontiguousarray(hist, np.uint32)
    vp, wp = np.ascontiguousarray(value_params, np.float32), np.ascontiguousarray(width_params, np.float32)
    channels = co.size // (len(c) * 512)
    assert co.size == channels * len(c) * 512 and b.size == co.size and p.size == co.size and h.size == channels * 10240 and vp.size == channels * 18 and wp.size == channels * 18
    n = C.c_size_t(0)
    err = C.create_string_buffer(256)
    L = load_library()
    # one call in the common case: a symbol costs at most max_freq_bits (< 32) bits, so 4 bytes per coefficient + the container
    # overhead always suffice; the library reports the needed size (-3) if they should not
    out = np.empty(co.size * 4 + channels * (10 * 2070 + 256) + 64, np.uint8)
    rc = L.fri_emit_encode_image(width, height, channels, _p(c), len(c), _p(co), _p(b), _p(p), _p(h), _p(vp), _p(wp), _p(out), out.size, C.addressof(n), err, 256)
    if rc == -3:
        out = np.empty(n.value, np.uint8)
        rc = L.fri_emit_encode_image(width, height, channels, _p(c), len(c), _p(co), _p(b), _p(p), _p(h), _p(vp), _p(wp), _p(out), out.size, C.addressof(n), err, 256)
    if rc != 0:
        raise EmitError(err.value.decode() or f"fri_emit_encode_image: {rc}")
    out = out[: n.value]
    return out.tobytes()


def stream_order(centers, valid_mask):
    """uint32 [num_some]: cell << 9 | heap index of every symbol of a channel, in stream order (None nodes taken out)."""
    c = np.ascontiguousarray(centers, np.int32)
    m = np.ascontiguousarray(valid_mask, np.uint32)
    out = np.empty(len(c) * 512, np.uint32)
    n = C.c_uint64(0)
    if load_library().fri_emit_stream_order(_p(c), len(c), _p(m), _p(out), C.addressof(n)) != 0:
        raise EmitError("fri_emit_stream_order")
    return out[: n.value].copy()


def encode_image_from_streams(width, height, streams, hist, value_params, width_params):
    """.frv bytes from the device's symbol streams: streams uint16 [C][n_symbols] (bucket << 10 | symbol), hist [C][10][1024], params [C][3][6]."""
    st = np.ascontiguousarray(streams, np.uint16)
    h = np.ascontiguousarray(hist, np.uint32)
    channels = h.size // 10240
    n_symbols = st.size // channels
    vp, wp = np.ascontiguousarray(value_params, np.float32), np.ascontiguousarray(width_params, np.float32)
    assert st.size == channels * n_symbols and vp.size == channels * 18 and wp.size == channels * 18
    n = C.c_size_t(0)
    err = C.create_string_buffer(256)
    out = np.empty(st.size * 4 + channels * (10 * 2070 + 256) + 64, np.uint8)
    rc = load_library().fri_emit_encode_image_from_streams(width, height, channels, _p(st), n_symbols, _p(h), _p(vp), _p(wp), _p(out), out.size, C.addressof(n), err, 256)
    if rc != 0:
        raise EmitError(err.value.decode() or f"fri_emit_encode_image_from_streams: {rc}")
    return out[: n.value].tobytes()


def check_image(frv, centers, coefs, bucket, prediction):
    """Entropy-layer self-check: parse, rebuild the models from the container, decode every symbol, compare. Raises on mismatch."""
    c = np.ascontiguousarray(centers, np.int32)
    co, b, p = np.ascontiguousarray(coefs, np.int32), np.ascontiguousarray(bucket, np.uint8), np.ascontiguousarray(prediction, np.int32)
    data = np.frombuffer(frv, np.uint8)
    channels = co.size // (len(c) * 512)
    err = C.create_string_buffer(256)
    rc = load_library().fri_emit_check_image(_p(data), data.size, channels, _p(c), len(c), _p(co), _p(b), _p(p), err, 256)
    if rc != 0:
        raise EmitError(err.value.decode() or f"fri_emit_check_image: {rc}")


def rans_selfcheck(n_symbols, seed=1):
    """The context-parallel rANS coder against the plain one-loop coder on n_symbols pseudo-random symbols. Raises on a difference."""
    err = C.create_string_buffer(256)
    rc = load_library().fri_emit_rans_selfcheck(n_symbols, seed, err, 256)
    if rc != 0:
        raise EmitError(err.value.decode() or f"fri_emit_rans_selfcheck: {rc}")


def decode_image(frv):
    """A .frv back to coefficient planes (serialize::decode + entropy_coding::decode of the reference, host only).
    Returns (width, height, channels, centers [F][2] int32, coefs [channels][F][512] int32 with None = INT32_MIN)."""
    data = np.frombuffer(frv, np.uint8)
    info = np.zeros(4, np.uint32)
    err = C.create_string_buffer(256)
    L = load_library()
    rc = L.fri_emit_decode_image(_p(data), data.size, _p(info), None, 0, None, err, 256)
    if rc != -3:
        raise EmitError(err.value.decode() or f"fri_emit_decode_image: {rc}")
    w, h, c, f = (int(x) for x in info)
    coefs = np.empty((c, f, 512), np.int32)
    centers = np.empty((f, 2), np.int32)
    rc = L.fri_emit_decode_image(_p(data), data.size, _p(info), _p(coefs), coefs.size, _p(centers), err, 256)
    if rc != 0:
        raise EmitError(err.value.decode() or f"fri_emit_decode_image: {rc}")
    return w, h, c, centers, coefs
