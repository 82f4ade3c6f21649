"""The oracle's per-cell and per-patch forms (oracle/fri_oracle.c: fri_oracle_cell, fri_oracle_from_raster_cells) against the full-lattice oracle.
They exist for config 5 (16384 x 16384), whose whole lattice the hash-map-shaped restatement cannot hold: the GPU test compares the device with them on
sampled cells, so they must themselves be the full oracle's answer. CPU only."""
import numpy as np
import pytest

from oracle import fri_oracle
from tests.common import KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, gen_image, random_params


@pytest.mark.parametrize("shape", [(200, 120, 3), (129, 257, 1), (10, 10, 3), (1, 700, 1)])
def test_single_cell_form_is_the_full_oracle(shape):
    w, h, c = shape
    img = gen_image("noise", w, h, c, 21)
    W = fri_oracle.Wavelet(img, h, w, c)
    cs, co = W.centers(), W.coefficients()
    out, kept = fri_oracle.cell_coefficients(img, h, w, c, cs)
    assert kept.all() and np.array_equal(out.transpose(1, 0, 2), co)
    # a centre far outside the image: every leaf is None, the retain rule drops the cell (wavelet_transform.rs:415-416)
    out, kept = fri_oracle.cell_coefficients(img, h, w, c, [[w + 4000, h + 4000]])
    assert not kept[0] and (out == -(2 ** 31)).all()


def _two_hop_patch(W, cell):
    """centres of `cell`, its lattice neighbours and theirs (the oracle's own nearby vectors, wavelet_transform.rs:71-90)"""
    v = fri_oracle.nearby_vectors(9)
    cs = {tuple(W.centers()[cell])}
    for _ in range(2):
        cs |= {(re + dx, im + dy) for re, im in cs for dx, dy in v}
    return np.array(sorted(cs), np.int32)


@pytest.mark.parametrize("shape,kind", [((320, 200, 1), "smooth"), ((160, 130, 3), "noise")])
def test_patch_form_gives_the_full_images_contexts(shape, kind):
    w, h, c = shape
    img = gen_image(kind, w, h, c, 4)
    W = fri_oracle.Wavelet(img, h, w, c)
    full_centres = {tuple(x): i for i, x in enumerate(W.centers())}
    W.quantize(np.ones(32, np.int32))
    vp, wp = random_params(9)
    vpa, wpa = np.ascontiguousarray(vp, np.float32).reshape(3, 6), np.ascontiguousarray(wp, np.float32).reshape(3, 6)
    ch = c - 1
    b, p, _, _ = W.predict(ch, vp, wp)
    co = W.coefficients()
    rng = np.random.default_rng(5)
    for cell in [0, W.num_cells - 1] + list(rng.integers(0, W.num_cells, 6)):
        patch = _two_hop_patch(W, int(cell))
        Wp = fri_oracle.Wavelet(img, h, w, c, centers=patch)  # cells outside the image are dropped by the retain rule, as in the full lattice
        idx = {tuple(x): i for i, x in enumerate(Wp.centers())}
        assert set(idx) <= set(full_centres)
        k = idx[tuple(W.centers()[cell])]
        assert np.array_equal(Wp.coefficients()[:, k], co[:, cell])
        for heap in range(512):
            r = Wp.context_at(ch, k, heap, vpa, wpa)
            if r is None:
                assert co[ch, cell, heap] == -(2 ** 31)
            else:
                assert r == (b[cell, heap], p[cell, heap]), (cell, heap)
        Wp.close()
    W.close()
