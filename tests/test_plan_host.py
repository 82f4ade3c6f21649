"""Host logic of the plan (geometry.cpp) against the CPU oracle: cell lattice, canonical order, Some/None masks, neighbour
cells, the static neighbour table of the gather kernel, and the work decomposition of the forward kernel. No GPU needed."""
import numpy as np
import pytest

from tests.common import gen_image

SHAPES = [(10, 10, 3), (64, 48, 3), (100, 37, 3), (1, 1, 1), (1, 700, 1), (700, 1, 3), (33, 17, 1), (512, 512, 3), (777, 333, 1), (1000, 700, 3)]
V9 = [(-31, 3), (-5, 17), (26, 14), (31, -3), (5, -17), (-26, -14)]  # get_nearby_vectors(9), wavelet_transform.rs:71-90


def plan(w, h, c):
    import frave_amd

    return frave_amd.Plan(None, w, h, c)


@pytest.mark.parametrize("shape", SHAPES)
def test_lattice_and_masks_match_oracle(oracle, shape):
    w, h, c = shape
    P = plan(w, h, c)
    W = oracle.Wavelet(np.zeros((h, w, c), np.uint8), h, w, c)
    assert (P.num_cells, P.num_bfs_cells) == (W.num_cells, W.num_bfs_cells)
    assert np.array_equal(P.centers(), W.centers())
    some = W.coefficients()[0] != oracle.NONE
    assert np.array_equal(P.valid_bits(), some)
    assert P.num_some == int(some.sum())
    assert P.coef_count == c * P.num_cells * 512 and P.pixel_bytes == w * h * c
    cen = [tuple(int(v) for v in r) for r in P.centers()]
    assert cen == sorted(cen, key=lambda p: (p[1], p[0]))  # utils.rs:17-32


@pytest.mark.parametrize("shape", [(64, 48, 3), (512, 512, 1), (1000, 700, 3)])
def test_neighbour_cells(shape):
    w, h, c = shape
    P = plan(w, h, c)
    cen = [tuple(int(v) for v in r) for r in P.centers()]
    index = {p: i for i, p in enumerate(cen)}
    nb = P.neighbour_cells()
    for k, (x, y) in enumerate(cen):
        assert nb[k, 0] == k
        for i, (dx, dy) in enumerate(V9):
            assert nb[k, 1 + i] == index.get((x + dx, y + dy), -1)


@pytest.mark.parametrize("shape", [(10, 10, 3), (100, 37, 3), (300, 200, 1)])
def test_neighbour_table_replays_get_neighbour_values(oracle, shape):
    """Gather through the plan's static table + neighbour ids on the CPU and compare with the oracle's restatement of
    ContextModeler::get_neighbour_values (context_modeling.rs:25-77), including the level-7 quirk."""
    w, h, c = shape
    img = gen_image("noise", w, h, c, 4)
    P = plan(w, h, c)
    W = oracle.Wavelet(img, h, w, c)
    co = W.coefficients()[0].astype(np.int64)
    co0 = np.where(co == oracle.NONE, 0, co)  # unwrap_or(0)
    want = W.neighbour_values(0)  # [F][512][6]
    tab = P.neighbour_table().astype(np.int64)  # [512][6]
    nb = P.neighbour_cells().astype(np.int64)  # [F][8]
    heap, slot, never = tab & 511, (tab >> 9) & 7, (tab >> 15) & 1
    F = P.num_cells
    got = np.zeros((F, 512, 6), np.int64)
    for k in range(6):
        cell = nb[:, slot[:, k]]  # [F][512]
        val = co0[np.clip(cell, 0, None), heap[None, :, k]]
        got[:, :, k] = np.where((cell >= 0) & (never[None, :, k] == 0), val, 0)
    assert np.array_equal(got[:, 2:, :], want[:, 2:, :])
    # heap index 0 and 1: left / up_left / up_right cells' same index (prediction.rs:96-112)
    assert (slot[:2, :3] == np.array([[5, 6, 1]] * 2)).all() and (heap[:2, :3] == np.array([[0] * 3, [1] * 3])).all()
    assert (never[:2, 3:] == 1).all()


def test_neighbour_table_static_facts():
    tab = plan(64, 48, 3).neighbour_table().astype(np.int64)
    never = (tab >> 15) & 1
    assert int(never[2:].sum()) == 248  # 248 of the 3060 (node, neighbour) pairs are never a node of that level
    assert int(((tab >> 9) & 7)[never == 0].max()) <= 6


@pytest.mark.parametrize("shape", [(10, 10, 3), (512, 512, 1), (777, 333, 3), (1920, 1080, 1), (4096, 4096, 1), (4096, 4096, 3)])
def test_forward_decomposition(shape):
    w, h, c = shape
    P = plan(w, h, c)
    t = P.tiling()
    tiles, cells, wg = P.tile_table()
    assert sorted(cells.tolist()) == list(range(P.num_cells))  # every cell exactly once
    assert wg[0] == 0 and wg[-1] == t["n_tiles"] and (np.diff(wg) >= 1).all()
    assert tiles[0, 4] == 0 and (tiles[1:, 4] == tiles[:-1, 4] + tiles[:-1, 5]).all() and tiles[-1, 4] + tiles[-1, 5] == P.num_cells
    assert 1 <= tiles[:, 5].min() and tiles[:, 5].max() == t["max_tile_cells"] <= t["cells_per_tile"]
    assert t["max_tile_cells"] * c <= 16  # 4 waves x 2 pairs x 2 items
    cen = P.centers()
    for x_lo, y_lo, wpx, rows, b, n in tiles[:: max(1, len(tiles) // 200)]:
        cs = cen[cells[b:b + n]]
        # the staged rectangle covers the in-image part of every cell's 46 x 21 leaf bounding box
        assert x_lo <= max(0, cs[:, 0].min() - 15) and x_lo + wpx - 1 >= min(w - 1, cs[:, 0].max() + 30)
        assert y_lo <= max(0, cs[:, 1].min() - 8) and y_lo + rows - 1 >= min(h - 1, cs[:, 1].max() + 12)
        assert 0 <= x_lo and x_lo + wpx <= w and 0 <= y_lo and y_lo + rows <= h
    assert t["lds_pitch"] % 16 == 0 and t["lds_pitch"] >= tiles[:, 2].max() * c + 15
    assert t["lds_rows"] == tiles[:, 3].max()
    assert 2 * (t["lds_pitch"] * t["lds_rows"] + 16 * t["max_tile_cells"]) < 160 * 1024
    per_share = np.diff(wg)  # whole tiles are dealt to the shares (a tile costs one iteration whatever it holds)
    assert per_share.max() - per_share.min() <= 1  # balanced to one tile (host-only plans carry no rank weights)
    assert (tiles[:, 5] >= t["cells_per_tile"] - 1).mean() > 0.9 or len(tiles) < 20  # bands are cut evenly: tiles are full or one short


def test_survey_size_table():
    # (BFS cells, retained, interior) at the BASELINE sizes
    assert (lambda p: (p.num_bfs_cells, p.num_cells, p.num_interior_cells))(plan(4096, 4096, 1)) == (33559, 33289, 32249)
    assert (lambda p: (p.num_bfs_cells, p.num_cells))(plan(1920, 1080, 3)) == (4317, 4221)
    assert (lambda p: (p.num_bfs_cells, p.num_cells))(plan(512, 512, 3)) == (617, 578)


@pytest.mark.parametrize("shape", [(10, 10, 1), (64, 48, 3), (100, 37, 3), (512, 512, 1), (777, 333, 3), (1920, 1080, 1), (1, 300, 1), (300, 1, 3)])
def test_inverse_write_out_lists_cover_every_owned_byte_once(shape):
    """K3's static lists (whole quads, whole dwords, masked dwords) partition exactly the bytes that belong to retained cells:
    all of the image when the lattice covers it, fewer for images thinner than a cell (the reference's BFS leaves holes there)"""
    w, h, c = shape
    L = plan(w, h, c).inverse_lists()
    assert L["built"] == 1
    owned = 16 * L["quads"] + 4 * L["dwords"] + L["part_bytes"]
    if min(w, h) >= 46:
        assert owned == w * h * c
    else:
        assert 0 < owned <= w * h * c and owned % c == 0
