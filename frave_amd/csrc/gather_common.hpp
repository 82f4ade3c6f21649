// gather_common.hpp -- what the neighbour-gather kernels (K2 predict + histogram, K4 fit sums) share: tile walk, LDS staging of
// the 36 cells of a tile as int16, the packed neighbour offsets. Included inside their translation units.
#pragma once
#include "device_common.hpp"

namespace fri {
namespace {

constexpr int kPredThreads = 512; // 8 waves
constexpr int kPredWaves = kPredThreads / 64;
constexpr int kHistBins = 10 * 1024;
constexpr int kSlotStride = 1040; // bytes per staged cell: 512 int16 + 8 zero halfwords (what "never a node" entries read); 16-byte multiple

// ---- shared by the gather kernels (K2 and the fit accumulators) -----------------------------------------------------
// Neighbour halfword offsets of node p relative to the own LDS slot, two per register: out[0] = {k0, k1}, out[1] = {k2, k3},
// out[2] = {k4, k5} (k = left, up_left, up_right, right, down_left, down_right; context_modeling.rs:37-71).
// A pred_slots entry: cell id, -1 = no cell, kPredSlotInterior set for interior cells.
__device__ __forceinline__ int pred_slot_cell(int raw) { return raw < 0 ? -1 : raw & (kPredSlotInterior - 1); }
__device__ __forceinline__ bool pred_slot_interior(int raw) { return raw >= 0 && (raw & kPredSlotInterior) != 0; }

// Packed offsets of one node from its 12-byte row of the neighbour table (host: build_pred_offsets at plan creation; the kernels
// load the result - the 48 entries of a lane used to be 48 serialised round trips at the start of every workgroup).
__host__ __device__ __forceinline__ void pred_offsets_from_row(const uint32_t (&row)[3], uint32_t (&out)[3]) {
    uint32_t h[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const uint32_t e = (row[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        const int slot = (e >> 9) & 7; // index into {self, +V9[0..5]} = lattice deltas (0,0),(1,0),(1,-1),(0,-1),(-1,0),(-1,1),(0,1)
        // da = +1 for slots 1, 2; -1 for 4, 5.  db = -1 for slots 2, 3; +1 for 5, 6.  As 2-bit fields of constants (0 -> 0, 1 -> +1, 3 -> -1).
        const int da = (int)((0x0F14u >> (2 * slot)) & 3u), db = (int)((0x14F0u >> (2 * slot)) & 3u);
        const int sa = (da & 1) - (da & 2), sb = (db & 1) - (db & 2);
        const int rel = (sa * kPredSide + sb) * (kSlotStride / 2) + (int)(e & 511u);
        const int o = (e & 0x8000u) ? 512 : rel; // 512 = the slot's zero pad ("never a node")
        h[k] = (uint32_t)o & 0xFFFFu;
    }
    out[0] = h[0] | (h[1] << 16);
    out[1] = h[2] | (h[3] << 16);
    out[2] = h[4] | (h[5] << 16);
}
__device__ __forceinline__ void pred_gather(const uint8_t *own, const uint32_t (&o)[3], int (&v)[6]) {
    const int h[6] = {(int)(short)(o[0] & 0xFFFFu), (int)o[0] >> 16, (int)(short)(o[1] & 0xFFFFu), (int)o[1] >> 16, (int)(short)(o[2] & 0xFFFFu), (int)o[2] >> 16};
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = *reinterpret_cast<const short *>(own + 2 * h[k]);
}

// Tile walk: blocks are dealt round-robin over the 8 XCDs; XCD x gets the contiguous eighth [x n/8, (x+1) n/8) of the
// tiles and its workgroups stride through it together, so concurrently staged tiles are neighbours in the image and
// their shared halo cells hit in that XCD's L2 (placement only affects speed).
struct PredTileWalk {
    uint32_t first, end, step;
    __device__ explicit PredTileWalk(uint32_t n_tiles) {
        const uint32_t groups = gridDim.x < 8u ? gridDim.x : 8u; // a grid smaller than 8 blocks: every block is its own group
        const uint32_t xcd = blockIdx.x % groups, wg_in_xcd = blockIdx.x / groups;
        step = (gridDim.x - xcd + groups - 1u) / groups;
        first = (uint32_t)((uint64_t)n_tiles * xcd / groups) + wg_in_xcd;
        end = (uint32_t)((uint64_t)n_tiles * (xcd + 1u) / groups);
    }
};

// The LDS images hold coefficients as int16. Every coefficient the forward kernel produces fits (|v| <= 255 before the
// quantiser divides), but the ABI takes any int32 array: a Some value outside [-32768, 32767] cannot be staged, so it is
// counted as out of alphabet (what the caller must treat as "the reference would not have produced a stream": its symbol
// |value - prediction| could only stay below 1024 if the predictor tracked such values). Cheap common case: one and-or per
// value; only a wave that sees a None or an outlier does the exact count.
__device__ __forceinline__ uint32_t pred_count_outliers(const int (&v)[8]) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) m |= ((uint32_t)v[i] + 0x8000u) & 0xFFFF0000u;
    if (!__any(m != 0)) return 0;
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) n += (v[i] != kNone && v[i] != (int)(short)v[i]) ? 1u : 0u;
    return n;
}
__device__ __forceinline__ bool pred_is_block_slot(int slot) {
    const int r = slot / kPredSide, c = slot - r * kPredSide;
    return r >= 1 && r <= kPredBlock && c >= 1 && c <= kPredBlock;
}

// Stages the 36 cells of a tile: 64 lanes x 8 coefficients per cell, int32 -> int16 by truncation. Every Some coefficient
// fits, and None (INT32_MIN = 0x80000000) truncates to 0, which is what the reference's .unwrap_or(0) reads; a slot without
// a retained cell is all zeros. One v_perm_b32 packs two low halves.
// range_counter (may be NULL): incremented once per wave that stages, into a BLOCK slot, a Some coefficient outside [-256, 255] - the
// precondition of the fit kernels' 32-bit partial sums (every cell is a block cell of exactly one tile).
__device__ __forceinline__ void pred_stage_tile(const int32_t *__restrict__ coefs, const int32_t *s_slot_cell, uint8_t *s_cells, int lane, int wave,
                                                uint32_t *range_counter = nullptr) {
    for (int slot = wave; slot < kPredSlots; slot += kPredWaves) {
        const int cell = s_slot_cell[slot];
        int4 lo = make_int4(0, 0, 0, 0), hi = lo;
        if (cell >= 0) {
            const int4 *src = reinterpret_cast<const int4 *>(coefs + (size_t)cell * kCell + 8 * lane);
            lo = src[0];
            hi = src[1];
            if (range_counter && pred_is_block_slot(slot)) {
                const int v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                uint32_t m = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) m |= v[i] == kNone ? 0u : ((uint32_t)v[i] + 256u) & 0xFFFFFE00u;
                if (__any(m != 0) && lane == 0) atomicAdd(range_counter, 1u);
            }
        }
        auto pk = [](int lo16, int hi16) -> uint32_t { return __builtin_amdgcn_perm((uint32_t)hi16, (uint32_t)lo16, 0x05040100u); };
        uint4 packed;
        packed.x = pk(lo.x, lo.y);
        packed.y = pk(lo.z, lo.w);
        packed.z = pk(hi.x, hi.y);
        packed.w = pk(hi.z, hi.w);
        uint8_t *dst = s_cells + slot * kSlotStride;
        *reinterpret_cast<uint4 *>(dst + 16 * lane) = packed;
        if (lane == 0) *reinterpret_cast<uint4 *>(dst + 1024) = make_uint4(0, 0, 0, 0);
    }
}

// ---- the tile skeleton of round 2's gather kernel (K2 kernel3) -------------------------------------------------------------------------
// One 1024-thread workgroup per CU, 16 waves. A tile's 36 cells sit in LDS as 16-bit values, 1 KiB per cell, halfword pairs permuted
// inside their tree level's region (gather_layout.inc, tools/lds_layout_search.py). Two images: tile i + 1 is staged while tile i is worked on.
constexpr int kP3Threads = 1024;
constexpr int kP3Waves = kP3Threads / 64;
static_assert(kP3Waves == kPredBlock * kPredBlock, "one wave per block cell");
constexpr int kP3SlotBytes = 1024;
constexpr int kP3ZeroOff = kPredSlots * kP3SlotBytes;          // zero words behind the 36 cells: what "never a node" entries read (one per block cell of a wave, 1 KiB apart)
constexpr int kP3ImageBytes = kP3ZeroOff + kP3SlotBytes + 64;  // 37 952
constexpr int kP3Halo = kPredSlots - kPredBlock * kPredBlock;  // 20 halo slots: one whole cell per wave + a quarter of one of the last four
static_assert(kP3Halo == kP3Waves + kP3Waves / 4, "halo staging: a whole cell per wave, the remaining cells in quarters");
static_assert(kP3ImageBytes + kP3SlotBytes < 65536, "image + cell offset must fit a DS instruction's 16-bit offset field");

// h-th halo slot of a tile (h < 20): top row, bottom row, left column, right column
__device__ __forceinline__ int p3_halo_slot(int h) {
    return h < 6 ? h : h < 12 ? 5 * kPredSide + (h - 6) : h < 16 ? (h - 11) * kPredSide : (h - 15) * kPredSide + 5;
}

} // namespace
} // namespace fri
