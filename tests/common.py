"""Shared synthetic inputs for the parity tests."""
import numpy as np

# Exact dyadic HF parameters from SURVEY.md section 8c (f32 evaluation order cannot matter).
KAT_VALUE_PARAMS = np.tile(np.array([0.25, 0.25, 0.25, 0.125, 0.0625, 0.0625], np.float32), (3, 1))
KAT_WIDTH_PARAMS = np.tile(np.array([1.0, 0.5, 0.25, 0.25, 0.125, 0.125], np.float32), (3, 1))


def kat_image(w, h, channels=3):
    """pixel(x,y,c) = (7x + 13y + 29c + (x*y mod 11)) & 0xFF   (SURVEY.md section 8c)."""
    y, x, c = np.meshgrid(np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64), np.arange(channels, dtype=np.int64), indexing="ij")
    return ((7 * x + 13 * y + 29 * c + ((x * y) % 11)) & 0xFF).astype(np.uint8)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def gen_image(kind, w, h, channels, image_index=0):
    """Generators of SURVEY.md section 8d: noise / smooth / const, seed = 0xF7A5E000 + image_index."""
    seed = np.uint64(0xF7A5E000 + image_index)
    n = h * w * channels
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        r = _splitmix64(seed + idx * np.uint64(0x9E3779B97F4A7C15))
    if kind == "noise":
        v = r & np.uint64(0xFF)
    elif kind == "smooth":
        pix = idx // np.uint64(channels)
        x = pix % np.uint64(w)
        y = pix // np.uint64(w)
        v = (((x + np.uint64(2) * y) >> np.uint64(3)) + (r & np.uint64(7))) & np.uint64(0xFF)
    elif kind == "const":
        v = np.full(n, 128, np.uint64)
    else:
        raise ValueError(kind)
    return v.astype(np.uint8).reshape(h, w, channels)


def random_params(seed, scale=0.3):
    rng = np.random.default_rng(seed)
    vp = rng.normal(0.0, scale, (3, 6)).astype(np.float32)
    wp = np.abs(rng.normal(0.0, scale, (3, 6))).astype(np.float32)
    wp[:, 0] += 1.0
    return vp, wp
