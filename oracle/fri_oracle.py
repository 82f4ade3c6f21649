"""ctypes loader for the CPU oracle (oracle/fri_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. Nothing under frave_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfri_oracle.so")
NONE = -(2 ** 31)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("fri_oracle.c", "fri_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()  # no-op unless the library is missing or older than its sources
        L = C.CDLL(_SO)
        vp = C.c_void_p
        L.fri_oracle_from_raster.restype = vp
        L.fri_oracle_from_raster.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
        L.fri_oracle_free.argtypes = [vp]
        L.fri_oracle_from_raster_cells.restype = vp
        L.fri_oracle_from_raster_cells.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32]
        L.fri_oracle_cell.restype = C.c_int
        L.fri_oracle_cell.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, vp]
        for n in ("num_cells", "num_bfs_cells", "channels"):
            f = getattr(L, "fri_oracle_" + n)
            f.restype = C.c_uint32
            f.argtypes = [vp]
        L.fri_oracle_centers.argtypes = [vp, vp]
        L.fri_oracle_coefficients.argtypes = [vp, vp]
        L.fri_oracle_set_coefficients.argtypes = [vp, vp]
        L.fri_oracle_quantize.argtypes = [vp, vp]
        L.fri_oracle_quantize.restype = C.c_int
        L.fri_oracle_predict.argtypes = [vp, C.c_uint32, vp, vp, vp, vp]
        L.fri_oracle_predict.restype = C.c_int
        L.fri_oracle_predictors.argtypes = [vp, C.c_uint32, vp, vp]
        L.fri_oracle_neighbour_values.argtypes = [vp, C.c_uint32, vp]
        L.fri_oracle_context_at.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp]
        L.fri_oracle_context_at.restype = C.c_int
        L.fri_oracle_set_coefficient.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32]
        L.fri_oracle_set_coefficient.restype = C.c_int
        L.fri_oracle_to_raster.argtypes = [vp, vp]
        L.fri_oracle_sorted_level.argtypes = [vp, C.c_uint32, vp]
        L.fri_oracle_sorted_level.restype = C.c_int64
        L.fri_oracle_pair.argtypes = [C.c_int, C.c_int32, C.c_int, C.c_int32, vp, vp]
        L.fri_oracle_pair.restype = C.c_int
        L.fri_oracle_nearby_vectors.argtypes = [C.c_uint32, vp]
        L.fri_oracle_literal.argtypes = [C.c_uint32, vp]
        L.fri_oracle_assign_bucket.argtypes = [C.c_float]
        L.fri_oracle_assign_bucket.restype = C.c_uint32
        L.fri_oracle_pack_signed.argtypes = [C.c_int32]
        L.fri_oracle_pack_signed.restype = C.c_uint32
        L.fri_oracle_unpack_signed.argtypes = [C.c_uint32]
        L.fri_oracle_unpack_signed.restype = C.c_int32
        L.fri_oracle_quant_layer.argtypes = [C.c_uint32]
        L.fri_oracle_quant_layer.restype = C.c_uint32
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Wavelet:
    """WaveletImage restatement (stages/wavelet_transform.rs:384-432)."""

    def __init__(self, pixels, height, width, channels, centers=None):
        """centers (optional, [n][2] = (re, im)): only these cells instead of the BFS lattice (fri_oracle_from_raster_cells: sampled checks of huge images)"""
        px = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
        assert px.size == height * width * channels
        self.h, self.w, self.c = height, width, channels
        if centers is None:
            self._h = lib().fri_oracle_from_raster(_p(px), height, width, channels)
        else:
            cs = np.ascontiguousarray(centers, np.int32).reshape(-1, 2)
            self._h = lib().fri_oracle_from_raster_cells(_p(px), height, width, channels, _p(cs), len(cs))
        if not self._h:
            raise ValueError("fri_oracle_from_raster failed")

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.fri_oracle_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    @property
    def num_cells(self):
        return lib().fri_oracle_num_cells(self._h)

    @property
    def num_bfs_cells(self):
        return lib().fri_oracle_num_bfs_cells(self._h)

    def centers(self):
        out = np.empty((self.num_cells, 2), np.int32)
        lib().fri_oracle_centers(self._h, _p(out))
        return out

    def coefficients(self):
        out = np.empty((self.c, self.num_cells, 512), np.int32)
        lib().fri_oracle_coefficients(self._h, _p(out))
        return out

    def set_coefficients(self, coefs):
        a = np.ascontiguousarray(coefs, np.int32)
        assert a.shape == (self.c, self.num_cells, 512)
        lib().fri_oracle_set_coefficients(self._h, _p(a))

    def quantize(self, q):
        q = np.ascontiguousarray(q, np.int32)
        assert q.size == 32
        return lib().fri_oracle_quantize(self._h, _p(q))

    def predict(self, channel, value_params, width_params):
        vp = np.ascontiguousarray(value_params, np.float32).reshape(3, 6)
        wp = np.ascontiguousarray(width_params, np.float32).reshape(3, 6)
        hist = np.zeros((10, 1024), np.uint32)
        oob = C.c_uint64(0)
        rc = lib().fri_oracle_predict(self._h, channel, _p(vp), _p(wp), _p(hist), C.addressof(oob))
        assert rc == 0
        bucket = np.empty((self.num_cells, 512), np.uint8)
        pred = np.empty((self.num_cells, 512), np.int32)
        lib().fri_oracle_predictors(self._h, channel, _p(bucket), _p(pred))
        return bucket, pred, hist, oob.value

    def context_at(self, channel, cell, heap, vp, wp):
        """(bucket, prediction) of one node from the coefficients as they are now, or None for a None node; vp / wp: float32 [3][6]"""
        b, p = C.c_uint32(0), C.c_int32(0)
        rc = lib().fri_oracle_context_at(self._h, channel, cell, heap, _p(vp), _p(wp), C.addressof(b), C.addressof(p))
        assert rc in (0, -1)
        return None if rc else (b.value, p.value)

    def set_coefficient(self, channel, cell, heap, value):
        assert lib().fri_oracle_set_coefficient(self._h, channel, cell, heap, value) == 0

    def neighbour_values(self, channel):
        out = np.empty((self.num_cells, 512, 6), np.int32)
        lib().fri_oracle_neighbour_values(self._h, channel, _p(out))
        return out

    def to_raster(self):
        out = np.empty(self.h * self.w * self.c, np.uint8)
        lib().fri_oracle_to_raster(self._h, _p(out))
        return out

    def sorted_level(self, level):
        n = lib().fri_oracle_sorted_level(self._h, level, None)
        if n < 0:
            raise ValueError("bad level")
        out = np.empty((n, 2), np.int32)
        lib().fri_oracle_sorted_level(self._h, level, _p(out))
        return out


def cell_coefficients(pixels, height, width, channels, centers):
    """fri_oracle_cell for each centre ([n][2] = (re, im)): (coefficients [n][channels][512], retained [n] bool)"""
    px = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
    assert px.size == height * width * channels
    cs = np.ascontiguousarray(centers, np.int32).reshape(-1, 2)
    out = np.empty((len(cs), channels, 512), np.int32)
    kept = np.empty(len(cs), bool)
    f = lib().fri_oracle_cell
    for i, (re, im) in enumerate(cs):
        rc = f(_p(px), height, width, channels, int(re), int(im), out[i].ctypes.data)
        assert rc >= 0
        kept[i] = rc == 1
    return out, kept


def pair(l, r):
    """(Option l, Option r) -> (d, s) or None; wavelet_transform.rs:211-218."""
    d, s = C.c_int32(0), C.c_int32(0)
    some = lib().fri_oracle_pair(l is not None, l or 0, r is not None, r or 0, C.addressof(d), C.addressof(s))
    return (d.value, s.value) if some else None


def nearby_vectors(depth):
    out = np.empty((6, 2), np.int32)
    lib().fri_oracle_nearby_vectors(depth, _p(out))
    return [tuple(int(v) for v in row) for row in out]


def literal(i):
    out = np.empty(2, np.int32)
    lib().fri_oracle_literal(i, _p(out))
    return int(out[0]), int(out[1])


def fnv1a64(data: bytes) -> int:
    h = 0xCBF29CE484222325
    for b in data:
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def fnv1a64_np(arr) -> int:
    """FNV-1a-64 over the little-endian bytes of arr (vectorised per chunk would reorder; keep scalar but fast enough via bytes)."""
    return fnv1a64(np.ascontiguousarray(arr).tobytes())
