"""CPU restatement of libfri's host-side emit path -- TEST INFRASTRUCTURE ONLY (tests/ may import it, frave_amd/ must not).

Follows the reference step by step (citations relative to /root/reference/crates/libfri/src):
  symbol order      = the oracle's literal scan_level walk (oracle/fri_oracle.c, wavelet_transform.rs:505-705)
  stream order      = stages/entropy_coding.rs:285-329
  ANS model         = prediction.rs:302-305, entropy_coding.rs:82-159 (AnsContext::finalize_context)
  rANS              = entropy_coding.rs:332-336 over ryg_rans' published rans64 coder (the `rans` crate the reference
                      links is not part of the tree: PARITY UNPINNED for the byte stream, see frave_amd/host/emit.hpp)
  container         = stages/serialize.rs:49-117
  decoder           = stages/serialize.rs:119-268 + stages/entropy_coding.rs:205-264, :352-443 (decode_image)
Pure Python / numpy loops: small images only. f32 `exp` goes through the platform libm (ctypes), as it does for the
reference and for the product.
"""
import ctypes
import ctypes.util
import struct

import numpy as np

from . import fri_oracle

NONE = fri_oracle.NONE
CONTEXTS, ALPHABET = 10, 1024
WIDTHS = [2.5, 4.5, 6.3, 8.5, 12.7, 16.0, 20.0, 24.0, 28.0, 36.0]
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.expf.restype = ctypes.c_float
_libm.expf.argtypes = [ctypes.c_float]
f32 = np.float32


def pack_signed(k):
    return 2 * k if k >= 0 else -2 * k - 1


def unpack_signed(k):
    return k // 2 if k % 2 == 0 else -((k + 1) // 2)


def prev_power_two(x):
    n = x
    for s in (1, 2, 4, 8, 16):
        n |= n >> s
    return n ^ (n >> 1)


def trailing_zeros(x):
    return 64 if x == 0 else (x & -x).bit_length() - 1


def heap_positions(center, level):
    """image position of every node of `level` of the cell at `center` (Fractal::new, wavelet_transform.rs:42-69)."""
    lit = [fri_oracle.literal(i) for i in range(11)]
    pos = {1: tuple(center)}
    for lv in range(level):
        for p in range(1 << lv, 1 << (lv + 1)):
            x, y = pos[p]
            pos[2 * p] = (x, y)
            pos[2 * p + 1] = (x + lit[9 - lv - 1][0], y + lit[9 - lv - 1][1])
    return {pos[p]: p for p in range(1 << level, 1 << (level + 1))}


def stream_symbols(W, channel, coefs, bucket, prediction):
    """(symbol, bucket) in stream order; coefs/bucket/prediction are this channel's [F][512] planes."""
    centers = [tuple(int(v) for v in c) for c in W.centers()]
    cell_of = {c: i for i, c in enumerate(centers)}
    out = []
    lvl0 = [tuple(int(v) for v in p) for p in W.sorted_level(0)]
    for heap in (0, 1):
        for p in lvl0:
            k = cell_of[p]
            if coefs[k][heap] != NONE:
                out.append((pack_signed(int(coefs[k][heap]) - int(prediction[k][heap])), int(bucket[k][heap])))
    for level in range(1, 9):
        where = {}
        for k, c in enumerate(centers):
            for pos, heap in heap_positions(c, level).items():
                where[pos] = (k, heap)
        for p in W.sorted_level(level):
            k, heap = where[(int(p[0]), int(p[1]))]
            if coefs[k][heap] != NONE:
                out.append((pack_signed(int(coefs[k][heap]) - int(prediction[k][heap])), int(bucket[k][heap])))
    return out


class Context:
    def __init__(self):
        self.freqs = [0] * ALPHABET
        self.cdf = [0] * ALPHABET
        self.off = []
        self.max_freq_bits = 0

    def finalize(self, bucket):
        if self.max_freq_bits < 8:
            self.max_freq_bits = 8
        width = f32(WIDTHS[bucket])
        scale = f32(1 << (self.max_freq_bits & 31))
        for j in range(ALPHABET):
            x = f32(unpack_signed(j))
            e = f32(_libm.expf(ctypes.c_float(float(-(abs(x - f32(0.0))) / width))))
            lap = f32(e / f32(f32(2.0) * width))
            v = f32(lap * scale)
            lv = 0 if not v > 0 else min(int(v), 0xFFFFFFFF)
            if lv == 0 and self.freqs[j] == 0 and j in self.off:
                self.freqs[j] = 1
            elif self.freqs[j] != 0 and lv == 0:
                self.freqs[j] = 1
                self.off.append(j)
            else:
                self.freqs[j] = lv
        target = 1 << (self.max_freq_bits & 31)
        cum, acc = [], 0
        for f in self.freqs:
            cum.append(acc)
            acc += f
        cur_total = cum[-1] + self.freqs[-1]
        if cur_total == 0:
            raise ZeroDivisionError("empty context")
        for i in range(1, ALPHABET):
            cum[i] = (target * cum[i]) // cur_total
        for i in range(ALPHABET - 1):
            if self.freqs[i] != 0 and cum[i + 1] == cum[i]:
                best_freq, best = 1 << 32, None
                for j in range(ALPHABET - 1):
                    f = cum[j + 1] - cum[j]
                    if 1 < f < best_freq:
                        best_freq, best = f, j
                if best is None:
                    continue
                if best < i:
                    for j in range(best + 1, i + 1):
                        cum[j] -= 1
                else:
                    for j in range(i + 1, best + 1):
                        cum[j] += 1
        for i in range(ALPHABET - 1):
            self.freqs[i] = cum[i + 1] - cum[i]
        self.freqs[-1] = (cum[-1] - target) & 0xFFFFFFFF
        self.cdf = cum
        self.max_freq_bits = trailing_zeros(prev_power_two(sum(self.freqs) & 0xFFFFFFFF))


L = 1 << 31


def rans_encode(symbols, contexts):
    """symbols: [(symbol, bucket)] in stream order -> bytes. Ten interleaved rans64 states, one backwards word stream."""
    x = [L] * CONTEXTS
    rev = []
    for sym, b in reversed(symbols):
        c = contexts[b]
        start, freq, bits = c.cdf[sym], c.freqs[sym], c.max_freq_bits
        x_max = ((L >> bits) << 32) * freq
        if x[b] >= x_max:
            rev.append(x[b] & 0xFFFFFFFF)
            x[b] >>= 32
        x[b] = ((x[b] // freq) << bits) + (x[b] % freq) + start
    for s in range(CONTEXTS):
        rev.append(x[s] >> 32)
        rev.append(x[s] & 0xFFFFFFFF)
    return b"".join(struct.pack("<I", w) for w in reversed(rev))


def rans_decode(data, buckets, contexts):
    words = list(struct.unpack("<%dI" % (len(data) // 4), data))
    pos = 0
    x = []
    for _ in range(CONTEXTS):
        x.append(words[pos] | words[pos + 1] << 32)
        pos += 2
    out = []
    for b in buckets:
        c = contexts[b]
        s = CONTEXTS - b - 1
        bits = c.max_freq_bits
        v = x[s] & ((1 << bits) - 1)
        sym = max(i for i in range(ALPHABET) if c.cdf[i] <= v and c.freqs[i] > 0)
        x[s] = c.freqs[sym] * (x[s] >> bits) + v - c.cdf[sym]
        if x[s] < L:
            x[s] = (x[s] << 32) | words[pos]
            pos += 1
        out.append(sym)
    return out


def encode_image(W, coefs, bucket, prediction, hist, value_params, width_params):
    """All arrays per channel: coefs/bucket/prediction [C][F][512], hist [C][10][1024], params [C][3][6] -> .frv bytes."""
    C = W.c
    out = bytearray(b"frif")
    out += struct.pack("<III", W.h, W.w, ((1 if C == 1 else 2) << 30) | (1 << 28))
    for ch in range(C):
        contexts = []
        for b in range(CONTEXTS):
            c = Context()
            c.freqs = [int(v) for v in hist[ch][b]]
            c.max_freq_bits = trailing_zeros(prev_power_two(sum(c.freqs) & 0xFFFFFFFF))
            c.finalize(b)
            contexts.append(c)
        data = rans_encode(stream_symbols(W, ch, coefs[ch], bucket[ch], prediction[ch]), contexts)
        out += b"\xff\xbb"
        out += np.asarray(value_params[ch], "<f4").tobytes() + np.asarray(width_params[ch], "<f4").tobytes()
        for c in contexts:
            out += b"\xff\xb2" + struct.pack("<IQ", c.max_freq_bits, len(c.off)) + b"".join(struct.pack("<H", v) for v in c.off)
        out += b"\xff\xb4" + struct.pack("<Q", len(data)) + data + b"\xff\xb8"
    out += b"\xff\xdf"
    return bytes(out)


def stream_nodes(W):
    """(cell, heap) of every node in stream order, None nodes included (entropy_coding.rs:369-443 walks sorted_lattice)."""
    centers = [tuple(int(v) for v in c) for c in W.centers()]
    cell_of = {c: i for i, c in enumerate(centers)}
    lvl0 = [cell_of[tuple(int(v) for v in p)] for p in W.sorted_level(0)]
    out = [(k, 0) for k in lvl0] + [(k, 1) for k in lvl0]
    for level in range(1, 9):
        where = {}
        for k, c in enumerate(centers):
            for pos, heap in heap_positions(c, level).items():
                where[pos] = (k, heap)
        out += [where[(int(p[0]), int(p[1]))] for p in W.sorted_level(level)]
    return out


def decode_image(frv):
    """serialize::decode + entropy_coding::decode, literally: a Wavelet of the all-zero image (WaveletImage::from_metadata), then
    symbol by symbol the context from the coefficients decoded so far (fri_oracle_context_at = get_lf / get_hf_context_bucket on
    the oracle's hash maps), the symbol from rANS state CONTEXT_AMOUNT - bucket - 1, and the coefficient written back.
    Returns (width, height, channels, coefs [C][F][512])."""
    assert frv[:4] == b"frif", "Invalid signature for FRIF image."
    h, w, meta = struct.unpack("<III", frv[4:16])
    C = 1 if (meta >> 30) & 3 == 1 else 3
    W = fri_oracle.Wavelet(np.zeros(h * w * C, np.uint8), h, w, C)
    nodes = stream_nodes(W)
    o, ch = 16, 0
    vp = wp = None
    contexts, data = [], b""
    while True:
        seg = frv[o : o + 2]
        o += 2
        if seg == b"\xff\xbb":
            prm = np.frombuffer(frv[o : o + 144], "<f4").astype(np.float32)
            vp, wp = np.ascontiguousarray(prm[:18].reshape(3, 6)), np.ascontiguousarray(prm[18:].reshape(3, 6))
            o += 144
        elif seg == b"\xff\xb2":
            c = Context()
            c.max_freq_bits, n = struct.unpack("<IQ", frv[o : o + 12])
            o += 12
            c.off = list(struct.unpack("<%dH" % n, frv[o : o + 2 * n]))
            o += 2 * n
            c.finalize(len(contexts))  # serialize.rs:232: rebuilt from the two fields
            contexts.append(c)
        elif seg == b"\xff\xb4":
            (n,) = struct.unpack("<Q", frv[o : o + 8])
            data = frv[o + 8 : o + 8 + n]
            o += 8 + n
        elif seg == b"\xff\xb8":
            words = list(struct.unpack("<%dI" % (len(data) // 4), data))
            x = [words[2 * i] | words[2 * i + 1] << 32 for i in range(CONTEXTS)]
            pos = 2 * CONTEXTS
            for cell, heap in nodes:
                ctx = W.context_at(ch, cell, heap, vp, wp)
                if ctx is None:  # entropy_coding.rs:421-425
                    continue
                b, pred = ctx
                c, s = contexts[b], CONTEXTS - b - 1  # :239
                bits = c.max_freq_bits
                v = x[s] & ((1 << bits) - 1)
                sym = max(i for i in range(ALPHABET) if c.cdf[i] <= v and c.freqs[i] > 0)  # :243-256
                x[s] = c.freqs[sym] * (x[s] >> bits) + v - c.cdf[sym]
                if x[s] < L:
                    x[s] = (x[s] << 32) | words[pos]
                    pos += 1
                value = unpack_signed(sym) + pred  # :263
                W.set_coefficient(ch, cell, heap, int(np.int32(np.uint32(value & 0xFFFFFFFF))))
            ch += 1
            contexts, data = [], b""
        elif seg == b"\xff\xdf":
            assert ch == C
            return w, h, C, W.coefficients()
        else:
            raise ValueError("Malformed image bytes")
