// geometry.cpp -- see geometry.hpp. Host only, no HIP.
//
// Citations are relative to /root/reference/crates/libfri/src/.
#include "geometry.hpp"

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <deque>

namespace fri {

namespace {

// fractal.rs:51-86, entries 0..10 (all that depth 9 uses: digits 0..8, plus 9 and 10 for the cell lattice).
constexpr Int2 kLiterals[11] = {{0, 1}, {-1, 1}, {2, 0}, {-3, -1}, {5, -1}, {1, 3}, {-11, -1}, {9, -5}, {13, 7}, {-31, 3}, {5, -17}};

inline Int2 add(Int2 a, Int2 b) { return {a.x + b.x, a.y + b.y}; }
inline Int2 sub(Int2 a, Int2 b) { return {a.x - b.x, a.y - b.y}; }
inline Int2 neg(Int2 a) { return {-a.x, -a.y}; }
inline int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
inline int pos_mod(int a, int m) {
    int r = a % m;
    return r < 0 ? r + m : r;
}

struct LatticeBasis {
    Int2 zl, zmd;
    int det;
    // (dx, dy) = a*zl + b*zmd  ->  (a, b); exact iff (dx, dy) is a lattice point
    bool coords(Int2 d, int &a, int &b) const {
        long na = (long)d.x * zmd.y - (long)d.y * zmd.x;
        long nb = (long)zl.x * d.y - (long)zl.y * d.x;
        if (na % det || nb % det) return false;
        a = (int)(na / det);
        b = (int)(nb / det);
        return true;
    }
};

StaticTables make_static() {
    StaticTables t{};
    std::memcpy(t.literals, kLiterals, sizeof(kLiterals));
    // Fractal::new (wavelet_transform.rs:42-69): leaf 512+s sits at centre + sum_k bit_k(s) * LITERALS[k]
    for (int s = 0; s < kCell; s++) {
        Int2 o{0, 0};
        for (int k = 0; k < kDepth; k++)
            if (s >> k & 1) o = add(o, kLiterals[k]);
        t.leaf_off[s] = o;
    }
    nearby_vectors(kDepth, t.v9);
    LatticeBasis lb{t.v9[0], t.v9[5], 0};
    lb.det = lb.zl.x * lb.zmd.y - lb.zl.y * lb.zmd.x;
    if (lb.det != kCell) {
        t.error = "cell lattice determinant is not 512";
        return t;
    }
    // Z^2 / Lambda is cyclic: r = (dx + 181*dy) mod 512 separates the 512 leaves of a cell.
    if (pos_mod(lb.zl.x + 181 * lb.zl.y, kCell) != 0 || pos_mod(lb.zmd.x + 181 * lb.zmd.y, kCell) != 0) {
        t.error = "residue functional does not vanish on the cell lattice";
        return t;
    }
    bool seen[kCell] = {};
    for (int s = 0; s < kCell; s++) {
        int r = pos_mod(t.leaf_off[s].x + 181 * t.leaf_off[s].y, kCell);
        if (seen[r]) {
            t.error = "leaf offsets are not a complete residue system";
            return t;
        }
        seen[r] = true;
        t.residue_lut[r] = (uint16_t)s;
    }
    t.nbr_delta[0] = {0, 0};
    for (int i = 0; i < 6; i++) {
        int a, b;
        if (!lb.coords(t.v9[i], a, b)) {
            t.error = "V9 is not in the lattice";
            return t;
        }
        t.nbr_delta[1 + i] = {a, b};
    }
    t.nbr_delta[7] = {0, 0};

    // locate(q): q (relative to the cell centre) = delta + leaf_off[s]  ->  (neighbour slot, s)
    auto locate = [&](Int2 q, int &slot, int &s) -> bool {
        s = t.residue_lut[pos_mod(q.x + 181 * q.y, kCell)];
        int a, b;
        if (!lb.coords(sub(q, t.leaf_off[s]), a, b)) return false;
        for (int i = 0; i < 7; i++)
            if (t.nbr_delta[i].x == a && t.nbr_delta[i].y == b) {
                slot = i;
                return true;
            }
        return false;
    };

    for (int p = 0; p < kCell; p++)
        for (int k = 0; k < 6; k++) t.nbr_table[p][k] = 0x8000;
    // Heap index 0 (DC) and 1 (root): get_lf_context_bucket (prediction.rs:86-149) with current_depth 0:
    // left = centre + V9[4], up_left = centre + V9[5], up_right = centre + V9[0], same heap index in that cell.
    for (int p = 0; p < 2; p++) {
        const int slots[3] = {1 + 4, 1 + 5, 1 + 0};
        for (int k = 0; k < 3; k++) t.nbr_table[p][k] = (uint16_t)(p | (slots[k] << 9));
    }
    // Levels 1..8: ContextModeler::get_neighbour_values (context_modeling.rs:25-77).
    for (int level = 1; level < kDepth; level++) {
        const int d = kDepth - level;
        Int2 V[6];
        nearby_vectors(d, V);
        const int low_mask = (1 << d) - 1;
        // gpm[lvl] membership of a position, static part: it is a node of level `lvl` of cell `slot`
        auto is_node = [&](Int2 q, int lvl, int &slot, int &s) -> bool {
            if (!locate(q, slot, s)) {
                slot = -1;
                return false;
            }
            return (s & ((1 << (kDepth - lvl)) - 1)) == 0;
        };
        for (int p = 1 << level; p < (2 << level); p++) {
            const Int2 P = t.leaf_off[(p - (1 << level)) << d];
            Int2 left = add(P, V[4]), right = add(P, V[1]);
            Int2 up_left = add(P, V[5]), up_right = add(P, V[0]);
            Int2 down_left = add(P, V[3]), down_right = add(P, V[2]);
            if (d == 2) {
                // wavelet_transform.rs:115-177: the probes index the global map by `depth` (= 2), i.e. level 2's key set.
                int s0, sl0, s1, sl1;
                bool a0 = is_node(add(P, V[0]), 2, sl0, s0), a1 = is_node(add(P, Int2{-1, -1}), 2, sl1, s1);
                if ((a0 && sl0 != 0) || (a1 && sl1 != 0)) {
                    t.error = "level-7 up probe depends on a neighbouring cell";
                    return t;
                }
                if (!a0 && a1) {
                    up_right = add(P, Int2{-1, -1});
                    up_left = add(up_right, V[4]);
                }
                bool b0 = is_node(add(P, V[3]), 2, sl0, s0), b1 = is_node(add(P, Int2{1, 1}), 2, sl1, s1);
                if ((b0 && sl0 != 0) || (b1 && sl1 != 0)) {
                    t.error = "level-7 down probe depends on a neighbouring cell";
                    return t;
                }
                if (!b0 && b1) {
                    down_left = add(P, Int2{1, 1});
                    down_right = add(down_left, V[1]);
                }
            }
            const Int2 q[6] = {left, up_left, up_right, right, down_left, down_right};
            for (int k = 0; k < 6; k++) {
                int slot, s;
                if (!locate(q[k], slot, s)) {
                    t.error = "neighbour position outside the 7-cell neighbourhood";
                    return t;
                }
                if (s & low_mask) continue; // not a level-`level` node anywhere: global_position_map[level].get() is None -> 0
                int heap = (kCell + s) >> d;
                if (k >= 3) heap >>= 1; // above-level values read the parent (context_modeling.rs:66)
                t.nbr_table[p][k] = (uint16_t)(heap | (slot << 9));
            }
        }
    }
    return t;
}

} // namespace

void nearby_vectors(int depth, Int2 out[6]) {
    Int2 zl, zmd;
    if (depth == 1) {
        zl = {-1, 1};
        zmd = {0, 2};
    } else if (depth == 2) {
        zl = {-2, 0};
        zmd = {0, -2};
    } else if (depth == 3) {
        zl = {-3, -1};
        zmd = {-1, -3};
    } else {
        zl = kLiterals[depth];
        zmd = add(kLiterals[depth + 1], zl);
    }
    out[0] = zl;
    out[1] = sub(zl, zmd);
    out[2] = neg(zmd);
    out[3] = neg(zl);
    out[4] = sub(zmd, zl);
    out[5] = zmd;
}

const StaticTables &static_tables() {
    static const StaticTables t = make_static();
    return t;
}

std::string build_geometry(uint32_t width, uint32_t height, uint32_t channels, const TilingParams &tp, Geometry &g) {
    const StaticTables &st = static_tables();
    if (!st.error.empty()) return st.error;
    if (!width || !height || (channels != 1 && channels != 3)) return "width/height must be > 0 and channels 1 or 3";
    if ((uint64_t)width * height * channels >= (1ull << 32)) return "image larger than the reference's u32 pixel index (images.rs:94)";
    g = Geometry{};
    g.width = width;
    g.height = height;
    g.channels = channels;
    const int W = (int)width, H = (int)height;
    const Int2 c0{W / 2, H / 2}; // wavelet_transform.rs:452
    const Int2 zl = st.v9[0], zmd = st.v9[5];
    // Lattice-coordinate window that holds every in-bounds centre plus one ring of neighbours.
    int amin = INT_MAX, amax = INT_MIN, bmin = INT_MAX, bmax = INT_MIN;
    for (int cx : {0 - c0.x, W - c0.x})
        for (int cy : {0 - c0.y, H - c0.y}) {
            long na = (long)cx * zmd.y - (long)cy * zmd.x, nb = (long)zl.x * cy - (long)zl.y * cx;
            int a = (int)floor_div((int)na, kCell), b = (int)floor_div((int)nb, kCell);
            amin = std::min(amin, a);
            amax = std::max(amax, a + 1);
            bmin = std::min(bmin, b);
            bmax = std::max(bmax, b + 1);
        }
    amin -= 2;
    bmin -= 2;
    amax += 2;
    bmax += 2;
    const int adim = amax - amin + 1, bdim = bmax - bmin + 1;
    std::vector<int32_t> grid((size_t)adim * bdim, -2); // -2 unseen, -1 seen/not retained, >=0 cell id
    auto at = [&](int a, int b) -> int32_t & { return grid[(size_t)(a - amin) * bdim + (b - bmin)]; };

    // fractal_divide (wavelet_transform.rs:450-484) in lattice coordinates.
    struct AB {
        int a, b;
    };
    std::deque<AB> to_add;
    std::vector<AB> found;
    to_add.push_back({0, 0});
    at(0, 0) = -1;
    while (!to_add.empty()) {
        AB cur = to_add.front();
        to_add.pop_front();
        found.push_back(cur);
        Int2 pos{c0.x + cur.a * zl.x + cur.b * zmd.x, c0.y + cur.a * zl.y + cur.b * zmd.y};
        if (pos.x < 0 || pos.y < 0 || pos.x > W || pos.y > H) continue; // boundary cell: kept, not expanded (:459-466)
        for (int i = 1; i <= 6; i++) {
            int na = cur.a + st.nbr_delta[i].x, nb = cur.b + st.nbr_delta[i].y;
            if (na < amin || na > amax || nb < bmin || nb > bmax) return "internal: lattice window too small";
            if (at(na, nb) == -2) {
                at(na, nb) = -1;
                to_add.push_back({na, nb});
            }
        }
    }
    g.n_bfs_cells = (uint32_t)found.size();

    // retain(): keep cells with a Some DC <=> at least one leaf inside the image (wavelet_transform.rs:415-416).
    struct Cand {
        Int2 c;
        AB ab;
    };
    std::vector<Cand> kept;
    kept.reserve(found.size());
    for (AB ab : found) {
        Int2 c{c0.x + ab.a * zl.x + ab.b * zmd.x, c0.y + ab.a * zl.y + ab.b * zmd.y};
        bool any = false;
        if (c.x - 15 >= 0 && c.x + 30 < W && c.y - 8 >= 0 && c.y + 12 < H) {
            any = true;
        } else {
            for (int s = 0; s < kCell && !any; s++) {
                int x = c.x + st.leaf_off[s].x, y = c.y + st.leaf_off[s].y;
                any = x >= 0 && y >= 0 && x < W && y < H; // images.rs:90
            }
        }
        if (any) kept.push_back({c, ab});
    }
    if (kept.empty()) return "empty lattice";
    std::sort(kept.begin(), kept.end(), [](const Cand &l, const Cand &r) { return l.c.y != r.c.y ? l.c.y < r.c.y : l.c.x < r.c.x; }); // utils.rs:17-32
    const size_t F = kept.size();
    // the kernels index coefficients with 32-bit element offsets ((channel * F + cell) * 512)
    if ((uint64_t)F * kCell * channels >= (1ull << 32)) return "image too large: more than 2^32 coefficient slots";
    g.centers.resize(F);
    g.interior.assign(F, 0);
    g.valid_mask.assign(F * 16, 0);
    g.nbr_cells.assign(F * kNbr, -1);
    for (size_t k = 0; k < F; k++) {
        g.centers[k] = kept[k].c;
        at(kept[k].ab.a, kept[k].ab.b) = (int32_t)k;
    }
    for (size_t k = 0; k < F; k++) {
        const Int2 c = kept[k].c;
        uint8_t nodev[2 * kCell];
        int nvalid = 0;
        for (int s = 0; s < kCell; s++) {
            int x = c.x + st.leaf_off[s].x, y = c.y + st.leaf_off[s].y;
            bool v = x >= 0 && y >= 0 && x < W && y < H;
            nodev[kCell + s] = v;
            nvalid += v;
        }
        for (int p = kCell - 1; p >= 1; p--) nodev[p] = nodev[2 * p] | nodev[2 * p + 1];
        uint32_t *m = &g.valid_mask[k * 16];
        if (nodev[1]) m[0] |= 1u; // DC = low_pass[1] (wavelet_transform.rs:221)
        for (int p = 1; p < kCell; p++)
            if (nodev[p]) m[p >> 5] |= 1u << (p & 31);
        for (int i = 0; i < 16; i++) g.n_some += (uint64_t)__builtin_popcount(m[i]);
        g.n_valid_leaves += (uint64_t)nvalid;
        g.interior[k] = nvalid == kCell;
        g.n_interior += nvalid == kCell;
        for (int i = 0; i < 7; i++) {
            int na = kept[k].ab.a + st.nbr_delta[i].x, nb = kept[k].ab.b + st.nbr_delta[i].y;
            int32_t id = -1;
            if (na >= amin && na <= amax && nb >= bmin && nb <= bmax) id = at(na, nb);
            g.nbr_cells[k * kNbr + i] = id >= 0 ? id : -1;
        }
    }

    // Gather-kernel tiles: kPredBlock x kPredBlock blocks in lattice coordinates (a, b) with a halo ring of one cell, so
    // that all 7 neighbour cells of every block cell are among the tile's slots.
    {
        const int na = (amax - amin) / kPredBlock + 1, nb = (bmax - bmin) / kPredBlock + 1;
        std::vector<int32_t> tile_of((size_t)na * nb, -1);
        for (size_t k = 0; k < F; k++) {
            const int ta = (kept[k].ab.a - amin) / kPredBlock, tb = (kept[k].ab.b - bmin) / kPredBlock;
            int32_t &t = tile_of[(size_t)ta * nb + tb];
            if (t < 0) {
                t = (int32_t)g.n_pred_tiles++;
                g.pred_slots.resize((size_t)g.n_pred_tiles * kPredSlots, -1);
                const int a0 = amin + ta * kPredBlock - 1, b0 = bmin + tb * kPredBlock - 1;
                for (int i = 0; i < kPredSide; i++)
                    for (int j = 0; j < kPredSide; j++) {
                        const int a = a0 + i, b = b0 + j;
                        int32_t id = -1;
                        if (a >= amin && a <= amax && b >= bmin && b <= bmax) id = at(a, b);
                        g.pred_slots[(size_t)t * kPredSlots + i * kPredSide + j] = id >= 0 ? (id | (g.interior[id] ? kPredSlotInterior : 0)) : -1;
                    }
            }
        }
    }

    // Work decomposition for the LDS-staged forward kernel (see geometry.hpp). Every 512-pixel-wide window holds
    // exactly one centre per row (centres satisfy x + 181*y = const mod 512), so a tile of n cells of a band spans
    // about 512 * n / band_rows pixels in x (+45 for the cell footprint) and band_rows + 20 rows.
    int band_rows = tp.band_rows > 0 ? tp.band_rows : (channels == 1 ? 32 : 16);
    int cells_per_tile = tp.cells_per_tile > 0 ? tp.cells_per_tile : (channels == 1 ? 8 : 5);
    if (cells_per_tile * (int)channels > 16) cells_per_tile = 16 / (int)channels; // 4 waves x at most 2 pairs of (cell, channel) items each
    // Workgroup shares: as many as the device keeps resident at once (so the launch is a single round with no tail),
    // each with the same number of cells (+-1), but never less than one tile's worth.
    const int target_wgs = tp.target_wgs > 0 ? tp.target_wgs : 1024;
    size_t n_wg = std::min<size_t>((size_t)target_wgs, (F + cells_per_tile - 1) / cells_per_tile);
    if (tp.cells_per_wg > 0) n_wg = (F + tp.cells_per_wg - 1) / tp.cells_per_wg;
    if (n_wg < 1) n_wg = 1;
    g.band_rows = band_rows;
    g.cells_per_tile = cells_per_tile;
    g.cells_per_wg = (int32_t)((F + n_wg - 1) / n_wg);
    const int cy_min = g.centers.front().y;
    std::vector<int32_t> order(F);
    for (size_t k = 0; k < F; k++) order[k] = (int32_t)k;
    auto band = [&](int32_t k) { return (g.centers[k].y - cy_min) / band_rows; };
    std::sort(order.begin(), order.end(), [&](int32_t l, int32_t r) {
        int bl = band(l), br = band(r);
        if (bl != br) return bl < br;
        if (g.centers[l].x != g.centers[r].x) return g.centers[l].x < g.centers[r].x;
        return g.centers[l].y < g.centers[r].y;
    });
    g.tile_cells.assign(order.begin(), order.end());
    g.tile_meta.resize(F);
    for (size_t k = 0; k < F; k++) g.tile_meta[k] = TileCell{g.centers[order[k]].x, g.centers[order[k]].y, order[k], g.interior[order[k]]};
    int max_w = 0, max_r = 0;
    auto emit_tile = [&](size_t i, size_t j) {
        int x0 = INT_MAX, x1 = INT_MIN, y0 = INT_MAX, y1 = INT_MIN;
        for (size_t k = i; k < j; k++) {
            Int2 c = g.centers[order[k]];
            x0 = std::min(x0, c.x - 15);
            x1 = std::max(x1, c.x + 30);
            y0 = std::min(y0, c.y - 8);
            y1 = std::max(y1, c.y + 12);
        }
        x0 = std::max(x0, 0);
        y0 = std::max(y0, 0);
        x1 = std::min(x1, W - 1);
        y1 = std::min(y1, H - 1);
        Tile t{x0, y0, x1 - x0 + 1, y1 - y0 + 1, (int32_t)i, (int32_t)(j - i)};
        max_w = std::max(max_w, t.width_px);
        max_r = std::max(max_r, t.n_rows);
        g.max_tile_cells = std::max(g.max_tile_cells, t.cell_count);
        g.tiles.push_back(t);
    };
    // Widest tile the LDS budget allows: the largest pitch = 8 or 24 (mod 32 dwords) with rows * pitch <= budget, minus the
    // 30 bytes of lead-in / round-up. Cells of a run are sorted by x.
    int width_cap = INT_MAX;
    if (tp.tile_buffer_bytes > 0) {
        int pitch = (tp.tile_buffer_bytes / (band_rows + 20)) / 16 * 16;
        while (pitch > 0 && ((pitch / 4) % 32) != 8 && ((pitch / 4) % 32) != 24) pitch -= 16;
        width_cap = std::max(46, (pitch - 30) / (int)channels); // a single cell (46 px) must always fit
    }
    auto width_of = [&](size_t a, size_t b2) {
        const int x0 = std::max(g.centers[order[a]].x - 15, 0), x1 = std::min(g.centers[order[b2 - 1]].x + 30, W - 1);
        return x1 - x0 + 1;
    };
    // Tiles first: every band (a maximal run of the ordered cells) is cut evenly into tiles of <= cells_per_tile cells, so all
    // but a few tiles are full or one cell short. A tile costs a workgroup one iteration whatever it holds: cutting the bands
    // *after* dealing cells to the shares (as an earlier version did) left 6- and 7-cell tiles everywhere (4672 tiles for the 33 289
    // cells of a 4096^2 plane instead of 4290).
    for (size_t i = 0; i < F;) {
        size_t j = i;
        const int b = band(order[i]);
        while (j < F && band(order[j]) == b) j++;
        const size_t m = j - i, n_t = (m + cells_per_tile - 1) / cells_per_tile;
        bool fits = true;
        for (size_t t = 0; t < n_t && fits; t++) fits = width_of(i + m * t / n_t, i + m * (t + 1) / n_t) <= width_cap;
        if (fits) {
            for (size_t t = 0; t < n_t; t++) emit_tile(i + m * t / n_t, i + m * (t + 1) / n_t);
        } else { // a sparse band: greedy, as many cells as the width cap allows (always at least one)
            for (size_t a = i; a < j;) {
                size_t b2 = a + 1;
                while (b2 < j && b2 - a < (size_t)cells_per_tile && width_of(a, b2 + 1) <= width_cap) b2++;
                emit_tile(a, b2);
                a = b2;
            }
        }
        i = j;
    }
    // Then whole tiles to shares. Equal tile counts (+-1) unless rank weights are given: share sh is run by workgroup b(sh) (the
    // inverse of the kernels' XCD-contiguous block -> share map), whose dispatch rank is b / (target_wgs / ranks).
    const size_t T = g.tiles.size();
    if (tp.cells_per_wg <= 0) n_wg = (size_t)target_wgs; // one share per resident workgroup, or one per tile if there are fewer tiles
    n_wg = std::max<size_t>(1, std::min(n_wg, T));
    g.cells_per_wg = (int32_t)((F + n_wg - 1) / n_wg);
    {
        const int ranks = std::min(std::max(tp.ranks, 1), 4);
        bool weighted = true;
        for (int i = 0; i < ranks; i++) weighted = weighted && tp.rank_weight[i] > 0;
        std::vector<double> cum(n_wg + 1, 0.0);
        const size_t q = n_wg >> 3, r = n_wg & 7, per_rank = std::max<size_t>(1, (size_t)target_wgs / ranks);
        for (size_t x = 0, sh = 0; x < 8; x++) {
            const size_t n_x = q + (x < r ? 1 : 0);
            for (size_t idx = 0; idx < n_x; idx++, sh++) {
                const size_t b = idx * 8 + x;
                cum[sh + 1] = cum[sh] + (weighted ? (double)tp.rank_weight[std::min<size_t>(b / per_rank, ranks - 1)] : 1.0);
            }
        }
        g.wg_tiles.assign(n_wg + 1, 0);
        for (size_t sh = 0; sh <= n_wg; sh++) g.wg_tiles[sh] = (int32_t)((double)T * cum[sh] / cum[n_wg] + 0.5);
        g.wg_tiles[0] = 0;
        g.wg_tiles[n_wg] = (int32_t)T;
        for (size_t sh = 1; sh <= n_wg; sh++) // every share keeps at least one tile (n_wg <= T)
            g.wg_tiles[sh] = (int32_t)std::min<size_t>(std::max<size_t>((size_t)g.wg_tiles[sh], (size_t)g.wg_tiles[sh - 1] + 1), T - (n_wg - sh));
        for (size_t sh = 0; sh < n_wg; sh++) {
            const Tile &first = g.tiles[g.wg_tiles[sh]], &last = g.tiles[g.wg_tiles[sh + 1] - 1];
            g.max_wg_cells = std::max(g.max_wg_cells, last.cell_begin + last.cell_count - first.cell_begin);
            g.max_wg_tiles = std::max(g.max_wg_tiles, g.wg_tiles[sh + 1] - g.wg_tiles[sh]);
        }
    }
    if (tp.strided_shares && n_wg > 1) {
        // Interleaved shares: the tiles are dealt to the shares like cards, round after round in share order, instead of giving every share one contiguous
        // run. At any moment of a one-round launch the resident workgroups then work on ONE window of consecutive tiles that slides over the image - the
        // pixel rows they read and the coefficient range they write are a compact, moving region of memory, not n_wg fronts spread over all of it - and
        // neighbouring tiles still run side by side on one XCD (consecutive shares belong to one XCD: xcd_contiguous_share). A share's tiles and cells stay
        // contiguous in the arrays (the kernels walk [wg_tiles[sh], wg_tiles[sh + 1]) and load a share's cell records as one range): the arrays are permuted.
        std::vector<std::vector<int32_t>> dealt(n_wg);
        size_t next = 0;
        for (bool any = true; any && next < T;) {
            any = false;
            for (size_t sh = 0; sh < n_wg && next < T; sh++)
                if (dealt[sh].size() < (size_t)(g.wg_tiles[sh + 1] - g.wg_tiles[sh])) {
                    dealt[sh].push_back((int32_t)next++);
                    any = true;
                }
        }
        std::vector<Tile> tiles2;
        std::vector<TileCell> meta2;
        std::vector<int32_t> cells2;
        tiles2.reserve(T), meta2.reserve(F), cells2.reserve(F);
        g.max_wg_cells = 0;
        for (size_t sh = 0; sh < n_wg; sh++) {
            int32_t share_cells = 0;
            for (int32_t t_old : dealt[sh]) {
                Tile t = g.tiles[t_old];
                const int32_t begin = (int32_t)meta2.size();
                for (int32_t k = 0; k < t.cell_count; k++) {
                    meta2.push_back(g.tile_meta[t.cell_begin + k]);
                    cells2.push_back(g.tile_cells[t.cell_begin + k]);
                }
                t.cell_begin = begin;
                share_cells += t.cell_count;
                tiles2.push_back(t);
            }
            g.max_wg_cells = std::max(g.max_wg_cells, share_cells);
        }
        g.tiles.swap(tiles2);
        g.tile_meta.swap(meta2);
        g.tile_cells.swap(cells2);
    }
    {
        const size_t cells_target = (size_t)(tp.batch_share_tiles > 0 ? tp.batch_share_tiles : 4) * cells_per_tile;
        size_t merge = 1;
        while (merge * (size_t)g.cells_per_wg < cells_target && merge < n_wg) merge++;
        for (size_t sh = 0; sh < n_wg; sh += merge) g.wg_tiles_batch.push_back(g.wg_tiles[sh]);
        g.wg_tiles_batch.push_back(g.wg_tiles[n_wg]);
        for (size_t k = 0; k + 1 < g.wg_tiles_batch.size(); k++) g.max_wg_tiles_batch = std::max(g.max_wg_tiles_batch, g.wg_tiles_batch[k + 1] - g.wg_tiles_batch[k]);
    }
    // A staged row starts at the 16-byte boundary at or below its first byte: up to 15 bytes of lead-in.
    // The pitch (in dwords) is kept = 8 or 24 mod 32: with the cell footprint that gives the fewest LDS bank
    // conflicts for the byte gather (2-way instead of 4-way at a multiple of 64 bytes).
    int pitch = ((max_w * (int)channels + 15 + 15) / 16) * 16;
    while (((pitch / 4) % 32) != 8 && ((pitch / 4) % 32) != 24) pitch += 16;
    g.lds_pitch = pitch;
    g.lds_rows = max_r;
    return "";
}

void build_inverse_lists(Geometry &g, size_t max_bytes) {
    const StaticTables &st = static_tables();
    const int W = (int)g.width, H = (int)g.height, C = (int)g.channels;
    g.inv_lists.clear();
    g.inv_quads.clear();
    g.inv_dwords.clear();
    g.inv_parts.clear();
    g.inv_rect_bytes = 0;
    auto give_up = [&]() {
        g.inv_quads.clear();
        g.inv_dwords.clear();
        g.inv_parts.clear();
    };
    std::vector<InvTileLists> lists(g.tiles.size());
    std::vector<uint16_t> bitmap;
    for (size_t ti = 0; ti < g.tiles.size(); ti++) {
        const Tile &t = g.tiles[ti];
        const int a0 = (t.x_lo * C) & ~15;
        const int rq = ((t.x_lo + t.width_px) * C - 1 - a0) / 16 + 1;
        if (t.n_rows > 256 || rq > 64) return give_up(); // entries hold the row and the dword column in 8 bits each
        g.inv_rect_bytes = std::max(g.inv_rect_bytes, t.n_rows * rq * 16);
        bitmap.assign((size_t)t.n_rows * rq, 0);
        for (int c = 0; c < t.cell_count; c++) {
            const Int2 cen = g.centers[g.tile_cells[t.cell_begin + c]];
            for (int s2 = 0; s2 < kCell; s2++) {
                const int x = cen.x + st.leaf_off[s2].x, y = cen.y + st.leaf_off[s2].y;
                if (x < 0 || y < 0 || x >= W || y >= H) continue; // set_pixel, images.rs:104
                for (int ch = 0; ch < C; ch++) {
                    const int col = x * C + ch - a0;
                    bitmap[(size_t)(y - t.y_lo) * rq + (col >> 4)] |= (uint16_t)(1u << (col & 15));
                }
            }
        }
        lists[ti].quad_begin = (uint32_t)g.inv_quads.size();
        lists[ti].dword_begin = (uint32_t)g.inv_dwords.size();
        lists[ti].part_begin = (uint32_t)g.inv_parts.size();
        for (int r = 0; r < t.n_rows; r++)
            for (int k = 0; k < rq; k++) {
                const uint16_t m = bitmap[(size_t)r * rq + k];
                if (m == 0xFFFFu) {
                    g.inv_quads.push_back((uint16_t)(r << 8 | k));
                } else if (m) {
                    for (int d = 0; d < 4; d++) {
                        const uint32_t nib = (m >> (4 * d)) & 15u;
                        if (nib == 15u)
                            g.inv_dwords.push_back((uint16_t)(r << 8 | (4 * k + d)));
                        else if (nib)
                            g.inv_parts.push_back((uint32_t)(r << 8 | (4 * k + d)) << 4 | nib);
                    }
                }
            }
        lists[ti].quad_count = (uint32_t)g.inv_quads.size() - lists[ti].quad_begin;
        lists[ti].dword_count = (uint32_t)g.inv_dwords.size() - lists[ti].dword_begin;
        lists[ti].part_count = (uint32_t)g.inv_parts.size() - lists[ti].part_begin;
        if ((g.inv_quads.size() + g.inv_dwords.size()) * 2 + g.inv_parts.size() * 4 > max_bytes) return give_up();
    }
    g.inv_lists.swap(lists);
}

} // namespace fri
