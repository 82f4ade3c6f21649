// geometry.hpp -- host-side geometry of the tame-twindragon cell lattice for one (width, height).
//
// Everything the reference re-derives per image with hash maps (Fractal::new, fractal_divide, the
// retain() filter, get_global_position_map; stages/wavelet_transform.rs:42-69, 405-484 of
// /root/reference/crates/libfri/src) is a function of (width, height) only. It is computed once
// here as dense tables and uploaded to HBM by the plan.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace fri {

struct Int2 {
    int32_t x, y;  // x = Complex.re, y = Complex.im (wavelet_transform.rs:198-199)
};

constexpr int kDepth = 9;          // BASE_FRAC_DEPTH, wavelet_transform.rs:39
constexpr int kCell = 1 << kDepth; // 512 leaves / coefficients per cell
constexpr int kNbr = 8;            // neighbour-cell list stride: self + 6 lattice neighbours (+1 pad)

// Static (image independent) tables derived from LITERALS (fractal.rs:51-86).
struct StaticTables {
    Int2 literals[11];
    Int2 leaf_off[kCell];       // leaf suffix s -> pixel offset from the cell centre
    uint16_t residue_lut[kCell]; // ((dx + 181*dy) mod 512) -> leaf suffix s
    Int2 v9[6];                 // get_nearby_vectors(9): cell lattice neighbours
    Int2 nbr_delta[kNbr];       // lattice-coordinate deltas (da, db) of the neighbour-cell list (slot delta = da * kPredSide + db)
    // Neighbour map for the gather (context_modeling.rs:25-77 + wavelet_transform.rs:97-177):
    // entry [p][k], k = left, up_left, up_right, right, down_left, down_right.
    // bits 0-8 heap index, bits 9-11 neighbour-cell slot, bit 15 = position is not a node of that level.
    uint16_t nbr_table[kCell][6];
    std::string error; // non-empty if a structural assumption failed
};
const StaticTables &static_tables();
void nearby_vectors(int depth, Int2 out[6]); // wavelet_transform.rs:71-90

// One K1/K3 workgroup: a group of cells whose pixel footprint is staged through LDS.
struct Tile {
    int32_t x_lo, y_lo;     // top-left pixel of the staged rectangle (already clipped to the image)
    int32_t width_px, n_rows;
    int32_t cell_begin, cell_count; // range in tile_cells
};

// Write-out lists of the inverse kernel (geometry only). A tile's pixels are staged in LDS as rows of 16-byte quads; image row
// y_lo + r occupies LDS row r, byte column c of the LDS row is image byte a0 + c of that row, a0 = (x_lo * C) rounded down to 16.
// Which bytes the tile's own cells write is known at plan creation, so the kernel neither tracks ownership nor scans the
// rectangle. Three lists per tile, each walked with densely packed lanes (a store instruction costs the same with 2 or 64
// active lanes, so the shapes are kept apart instead of branching per lane):
//   quads : 16-byte quads written whole                       entry r << 8 | quad
//   dwords: whole dwords inside partly owned quads              entry r << 8 | dword   (dword = column / 4 < 256)
//   parts : partly owned dwords with their 4-bit byte mask      entry (r << 8 | dword) << 4 | mask
struct InvTileLists {
    uint32_t quad_begin, quad_count;
    uint32_t dword_begin, dword_count;
    uint32_t part_begin, part_count;
};

// One cell of a tile as the forward/inverse kernels see it (16 bytes, staged into LDS per workgroup).
struct TileCell {
    int32_t cx, cy, cell, interior;
};

struct Geometry {
    uint32_t width = 0, height = 0, channels = 0;
    uint32_t n_bfs_cells = 0;
    uint32_t n_interior = 0;
    uint64_t n_some = 0;                // Some coefficients per channel
    uint64_t n_valid_leaves = 0;        // pixels that are a leaf of a retained cell (== width * height unless the lattice has holes)
    std::vector<Int2> centers;          // [F] canonical order (ascending im, then re)
    std::vector<uint8_t> interior;      // [F] 1 = all 512 leaves inside the image
    std::vector<uint32_t> valid_mask;   // [F][16]
    std::vector<int32_t> nbr_cells;     // [F][kNbr]
    // forward/inverse tiling
    // Work decomposition: the cells of a band (band_rows rows of centres), sorted by x, are split evenly into
    // workgroup shares of about cells_per_wg cells; a share is split evenly into tiles of <= cells_per_tile cells
    // that the workgroup stages through LDS one after the other (double buffered).
    std::vector<Tile> tiles;
    std::vector<int32_t> tile_cells;
    std::vector<TileCell> tile_meta;    // same order as tile_cells
    std::vector<int32_t> wg_tiles;      // [n_wg + 1] tile range of each workgroup share
    // Shares for batch launches: runs of consecutive shares merged until a share holds >= 4 tiles' worth of cells. A small image
    // is cut into one-tile shares to fill the machine on its own; with hundreds of frames in one launch that only multiplies the
    // per-workgroup start-up cost.
    std::vector<int32_t> wg_tiles_batch; // [n_wg_batch + 1]
    int32_t max_wg_tiles_batch = 0;
    std::vector<InvTileLists> inv_lists; // [n_tiles]; empty if the lists were not built (budget)
    std::vector<uint16_t> inv_quads, inv_dwords;
    std::vector<uint32_t> inv_parts;
    int32_t inv_rect_bytes = 0; // largest n_rows * quads-per-row * 16 over all tiles
    int32_t lds_pitch = 0;   // bytes per staged row (multiple of 16)
    int32_t lds_rows = 0;    // max rows per tile
    int32_t band_rows = 0, cells_per_tile = 0, cells_per_wg = 0;
    int32_t max_tile_cells = 0; // largest cell_count over all tiles (<= cells_per_tile)
    int32_t max_wg_tiles = 0;   // most tiles in one workgroup share
    int32_t max_wg_cells = 0;   // most cells in one workgroup share
    // Gather kernel (K2): blocks of kPredBlock x kPredBlock cells in lattice coordinates plus a one-cell halo ring.
    // pred_slots[t][(kPredBlock+2)^2] = cell id held by each LDS slot of tile t (-1 = no retained cell there), with
    // kPredSlotInterior set when all 512 leaves of the cell are inside the image (so the gather kernels need no second lookup).
    std::vector<int32_t> pred_slots;
    uint32_t n_pred_tiles = 0;
};

constexpr int32_t kPredSlotInterior = 0x40000000;          // flag bit in a pred_slots entry; cell ids stay below it
constexpr int kPredBlock = 4;                              // cells per block edge
constexpr int kPredSide = kPredBlock + 2;                  // with halo
constexpr int kPredSlots = kPredSide * kPredSide;          // 36

struct TilingParams {
    int band_rows = 0, cells_per_tile = 0; // 0 = default
    int target_wgs = 0;   // workgroup shares to aim for (resident workgroups of the device); 0 = 1024
    int cells_per_wg = 0; // if > 0 overrides target_wgs
    // Relative share size by dispatch rank (workgroup b has rank min(ranks * b / target_wgs, ranks - 1): the r-th workgroup placed
    // on its CU). The CU's arbiter favours older waves, so later workgroups progress more slowly and get fewer cells.
    // {0,..} = equal shares.
    float rank_weight[4] = {0, 0, 0, 0};
    int ranks = 4; // resident workgroups per CU (<= 4)
    // Bytes one LDS tile buffer may take (rows * pitch); tiles of sparse bands (the image's last rows) are cut narrower
    // instead of sizing every buffer for them. 0 = no cap.
    int tile_buffer_bytes = 0;
    int batch_share_tiles = 0; // tiles' worth of cells per merged batch share (0 = 4)
    bool strided_shares = false; // deal the tiles to the shares round-robin (the resident set works on one sliding window of the image) instead of one contiguous run each
};

// Returns "" on success, else an error string.
std::string build_geometry(uint32_t width, uint32_t height, uint32_t channels, const TilingParams &tp, Geometry &out);
// Fills g.inv_* (call after build_geometry); skipped (lists stay empty) if they would exceed max_bytes.
void build_inverse_lists(Geometry &g, size_t max_bytes);

} // namespace fri
