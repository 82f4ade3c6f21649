// gather_common.hpp -- what the neighbour-gather kernels (K2 predict + histogram, K4 fit sums) share: the slot-list entry format, the tile walk and the cell
// geometry of a tile's LDS image (1 KiB per cell, pairs at gather_layout.inc's positions). Included inside their translation units.
#pragma once
#include "device_common.hpp"

namespace fri {
namespace {

constexpr int kPredThreads = 512; // 8 waves
constexpr int kPredWaves = kPredThreads / 64;
constexpr int kHistBins = 10 * 1024;

// ---- shared by the gather kernels (K2 and the fit accumulators) -----------------------------------------------------
// A pred_slots entry: cell id, -1 = no cell, kPredSlotInterior set for interior cells.
__device__ __forceinline__ int pred_slot_cell(int raw) { return raw < 0 ? -1 : raw & (kPredSlotInterior - 1); }
__device__ __forceinline__ bool pred_slot_interior(int raw) { return raw >= 0 && (raw & kPredSlotInterior) != 0; }

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

// ---- the tile skeleton of round 2's gather kernel (K2 kernel3) -------------------------------------------------------------------------
// One 1024-thread workgroup per CU, 16 waves. A tile's 36 cells sit in LDS as 16-bit values, 1 KiB per cell, halfword pairs permuted
// inside their tree level's region (gather_layout.inc, tools/lds_layout_search.py). Two images: tile i + 1 is staged while tile i is worked on.
constexpr int kP3Threads = 1024;
constexpr int kP3Waves = kP3Threads / 64;
static_assert(kP3Waves == kPredBlock * kPredBlock, "one wave per block cell");
constexpr int kP3SlotBytes = 1024;
constexpr int kP3ZeroOff = kPredSlots * kP3SlotBytes;          // zero words behind the 36 cells: what "never a node" entries read (one per block cell of a wave, 1 KiB apart)
constexpr int kP3ImageBytes = kP3ZeroOff + kP3SlotBytes + 64;  // 37 952
static_assert(kP3ImageBytes + kP3SlotBytes < 65536, "image + cell offset must fit a DS instruction's 16-bit offset field");

} // namespace
} // namespace fri
