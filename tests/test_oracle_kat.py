"""Pins the CPU oracle (oracle/fri_oracle.c) to the known answers recorded in SURVEY.md section 8c
and Appendix A. Those were derived by an independent reading of the reference (the survey's scratch
model), so agreement = two independent restatements of the Rust agree; the reference itself ships
no golden vectors (PARITY UNPINNED, see oracle/fri_oracle.h)."""
import numpy as np
import pytest

from tests.common import KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS, kat_image

# (w, h) -> F, Some/channel, fnv64(coefs), ch0 bucket totals, fnv64(hist) ch0/1/2   [SURVEY.md section 8c]
KATS = {
    (10, 10): (3, 151, 0x832681AF504F7BDA, [8, 5, 2, 10, 22, 22, 7, 5, 4, 66],
               (0x3A78097CB9707890, 0xCD08BB4048C09132, 0xB2F1E3D71EAA85A2)),
    (64, 48): (15, 3298, 0x04797D84397D36B5, [11, 117, 156, 440, 705, 240, 41, 41, 45, 1502],
               (0x9129D0C3C7F3FF19, 0x5C4A90388CFDB90F, 0x7FB236DCC5C08695)),
    (100, 37): (14, 3960, 0x1B479D3D891D0FF2, [13, 147, 195, 480, 854, 283, 47, 62, 79, 1800],
                (0x7AA85FF452EBF979, 0x0C417F955AE443A1, 0xAB565518CCC1BA45)),
}

# Appendix A order KATs: level -> (count, fnv64 of (re,im) int32-LE pairs)
ORDER_KATS = {
    (10, 10): {0: (3, 0xE9B23773861A10CC), 1: (6, 0x05A97F05D28AFC3B), 2: (12, 0x343BA5EFAF0C25EE), 3: (24, 0xEEC42D9F2D527222),
               4: (48, 0xDC4D72C1588B1F36), 5: (96, 0xA3ED9D8AB1731FA6), 6: (192, 0x3B2A0331A9615C1D), 7: (384, 0x714F94B3B37A3E95),
               8: (768, 0x9F32274CC369EEFE)},
    (64, 48): {0: (15, 0x1CED0DE3ABD6DC2C), 1: (30, 0x15974D309C356130), 2: (60, 0xA616E9FA3101F0E6), 3: (120, 0x04122965CB19CFC2),
               4: (240, 0xA644A2854B850B36), 5: (480, 0xC279972E2F991A96), 6: (960, 0xEF21B60E4D9EB645), 7: (1920, 0x37279FC400E262ED),
               8: (3840, 0xDB81EC164E03FD4D)},
}


@pytest.mark.parametrize("wh", list(KATS))
def test_coefficient_and_histogram_kats(oracle, wh):
    w, h = wh
    F, some, fnv_coef, buckets, fnv_hist = KATS[wh]
    img = kat_image(w, h)
    W = oracle.Wavelet(img, h, w, 3)
    assert W.num_cells == F
    co = W.coefficients()
    assert int((co[0] != oracle.NONE).sum()) == some
    # stream = cells ascending (im, re), per cell channels 0,1,2, 512 x int32-LE heap order
    assert oracle.fnv1a64_np(co.transpose(1, 0, 2)) == fnv_coef
    assert W.quantize(np.ones(32, np.int32)) == 0
    for ch in range(3):
        _, _, hist, oob = W.predict(ch, KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS)
        assert oob == 0
        assert int(hist.sum()) == some
        if ch == 0:
            assert hist.sum(1).tolist() == buckets
        assert oracle.fnv1a64_np(hist) == fnv_hist[ch]
    assert np.array_equal(W.to_raster(), img.reshape(-1))  # lossless identity (bench.rs:97-101)


def test_10x10_details(oracle):
    W = oracle.Wavelet(kat_image(10, 10), 10, 10, 3)
    cen = W.centers().tolist()
    assert cen == [[-21, -9], [5, 5], [-26, 8]]
    co = W.coefficients()
    N = oracle.NONE
    k = cen.index([5, 5])
    assert co[0, k, :16].tolist() == [17, 33, 35, 2, 79, -33, -3, N, -4, -12, N, -47, N, 6, N, N]
    assert co[0, k, 256:264].tolist() == [-18, -17, -20, -19, -15, -14, -17, -16]
    assert int((co[0, k] != N).sum()) == 126
    k2 = cen.index([-21, -9])
    assert int((co[0, k2] != N).sum()) == 10
    nz = {i: int(co[0, k2, i]) for i in range(512) if co[0, k2, i] not in (N, 0)}
    assert nz == {6: 1, 12: -1, 25: -1, 51: 2, 102: -3, 205: 7, 410: -13}
    b, p, _, _ = W.predict(0, KAT_VALUE_PARAMS, KAT_WIDTH_PARAMS)
    assert [(int(b[k, i]), int(p[k, i])) for i in (0, 1, 2, 3, 5)] == [(0, 0), (0, 0), (1, 0), (5, 1), (5, 2)]


@pytest.mark.parametrize("wh", list(ORDER_KATS))
def test_symbol_order_kats(oracle, wh):
    w, h = wh
    W = oracle.Wavelet(kat_image(w, h), h, w, 3)
    for level, (n, fnv) in ORDER_KATS[wh].items():
        s = W.sorted_level(level)
        assert len(s) == n == W.num_cells << level  # the reference's own assertion, wavelet_transform.rs:701
        assert oracle.fnv1a64_np(s) == fnv


def test_pair_kats(oracle):
    # SURVEY.md section 8a row 3
    assert oracle.pair(10, 10) == (0, 10)
    assert oracle.pair(5, 8) == (-3, 7)
    assert oracle.pair(8, 5) == (3, 6)
    assert oracle.pair(None, 8) == (-8, 4)
    assert oracle.pair(7, None) == (7, 3)
    assert oracle.pair(0, 255) == (-255, 128)
    assert oracle.pair(255, 0) == (255, 127)
    assert oracle.pair(None, None) is None


def test_nearby_vectors_table(oracle):
    # SURVEY.md section 8a row 5 [tabulated]
    tab = {
        1: [(-1, 1), (-1, -1), (0, -2), (1, -1), (1, 1), (0, 2)],
        2: [(-2, 0), (-2, 2), (0, 2), (2, 0), (2, -2), (0, -2)],
        3: [(-3, -1), (-2, 2), (1, 3), (3, 1), (2, -2), (-1, -3)],
        4: [(5, -1), (-1, -3), (-6, -2), (-5, 1), (1, 3), (6, 2)],
        5: [(1, 3), (11, 1), (10, -2), (-1, -3), (-11, -1), (-10, 2)],
        6: [(-11, -1), (-9, 5), (2, 6), (11, 1), (9, -5), (-2, -6)],
        7: [(9, -5), (-13, -7), (-22, -2), (-9, 5), (13, 7), (22, 2)],
        8: [(13, 7), (31, -3), (18, -10), (-13, -7), (-31, 3), (-18, 10)],
        9: [(-31, 3), (-5, 17), (26, 14), (31, -3), (5, -17), (-26, -14)],
    }
    for d, v in tab.items():
        assert oracle.nearby_vectors(d) == v


def test_small_scalars(oracle):
    L = oracle.lib()
    assert [L.fri_oracle_quant_layer(i) for i in (0, 1, 2, 3, 6, 7, 255, 256, 510, 511)] == [0, 1, 1, 2, 2, 3, 8, 8, 8, 9]
    assert [L.fri_oracle_pack_signed(k) for k in (0, 1, -1, 2, -2, 511, -512)] == [0, 2, 1, 4, 3, 1022, 1023]
    assert all(L.fri_oracle_unpack_signed(L.fri_oracle_pack_signed(k)) == k for k in range(-600, 600))
    edges = [(0.0, 0), (2.99, 0), (3.0, 1), (4.9, 1), (5.0, 2), (6.0, 3), (7.99, 3), (8.0, 4), (12.0, 5), (16.0, 6), (20.0, 7), (25.0, 8),
             (29.9, 8), (30.0, 9), (1e12, 9), (-5.0, 0), (float("nan"), 0)]
    for wv, b in edges:
        assert L.fri_oracle_assign_bucket(wv) == b


@pytest.mark.parametrize("wh,expect", [((512, 512), (617, 578)), ((1920, 1080), (4317, 4221))])
def test_cell_counts(oracle, wh, expect):
    # SURVEY.md section 8 size table (BFS cells, retained cells)
    w, h = wh
    W = oracle.Wavelet(np.zeros((h, w, 3), np.uint8), h, w, 3)
    assert (W.num_bfs_cells, W.num_cells) == expect


def test_luma_matches_rgb_channel0(oracle):
    # a C=1 plane and the same plane replicated to RGB give the same channel 0
    rng = np.random.default_rng(5)
    g = rng.integers(0, 256, (48, 64, 1), dtype=np.uint8)
    W1 = oracle.Wavelet(g, 48, 64, 1)
    W3 = oracle.Wavelet(np.repeat(g, 3, axis=2), 48, 64, 3)
    assert np.array_equal(W1.centers(), W3.centers())
    assert np.array_equal(W1.coefficients()[0], W3.coefficients()[0])
    assert np.array_equal(W1.to_raster(), g.reshape(-1))
