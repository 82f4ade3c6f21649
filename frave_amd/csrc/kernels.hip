// kernels.hip -- gfx950 (CDNA4, wave64) kernels of the libfri hot path.
//
//   K1 fwd_transform_quant   address-map gather + 9-level residue (S-)transform + per-layer quantiser
//                            (Fractal::extract_coefficients, stages/wavelet_transform.rs:179-225;
//                             quantization::encode, stages/quantization.rs:7-25)
//   K2 predict_histogram     6-neighbour gather + context bucket + prediction + ANS symbol histogram
//                            (context_modeling.rs:25-77; stages/prediction.rs:86-207, 237-298)
//   K3 inverse_transform     dequantisation + inverse residue transform + clamp
//                            (stages/quantization.rs:27-45; stages/wavelet_transform.rs:358-381; images.rs:103-111)
//
// Citations are relative to /root/reference/crates/libfri/src/. All three are byte/integer gather-scan
// kernels bounded by HBM traffic; there is no dense contraction here and no MFMA.
//
// Work decomposition shared by K1 and K3: one 64-lane wavefront owns one (cell, channel). Lane L owns the
// eight leaves 8L..8L+7 of the cell's digit tree; their pixel offsets from the lane base are the subset
// sums of LITERALS[0..2] (a fixed 4x3 footprint), the lane base is the subset sum of LITERALS[3..8]
// selected by the bits of L. Tree levels 8,7,6 are register arithmetic inside the lane, levels 5..0 are
// six cross-lane butterfly rounds (lane ^ 1, 2, 4, 8, 16, 32). The cell's 512 int32 coefficients leave
// as four fully coalesced store instructions (1 KiB + 512 B + 256 B + 256 B).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

#include <cstring>

namespace fri {
namespace {

constexpr int kNone = INT32_MIN; // wire encoding of Option::None
constexpr int kFwdThreads = 256; // 4 waves per workgroup
constexpr int kFwdWaves = kFwdThreads / 64;

// Leaf j (0..7) of a lane: bit0 -> LITERALS[0]=(0,1), bit1 -> LITERALS[1]=(-1,1), bit2 -> LITERALS[2]=(2,0).
__host__ __device__ constexpr int leaf_dx(int j) { return ((j & 2) ? -1 : 0) + ((j & 4) ? 2 : 0); }
__host__ __device__ constexpr int leaf_dy(int j) { return (j & 1) + ((j >> 1) & 1); }
// Lane base: bits 0..5 of the lane select LITERALS[3..8] = (-3,-1),(5,-1),(1,3),(-11,-1),(9,-5),(13,7).
__host__ __device__ constexpr int lane_dx(int l) {
    return -3 * (l & 1) + 5 * ((l >> 1) & 1) + ((l >> 2) & 1) - 11 * ((l >> 3) & 1) + 9 * ((l >> 4) & 1) + 13 * ((l >> 5) & 1);
}
__host__ __device__ constexpr int lane_dy(int l) {
    return -(l & 1) - ((l >> 1) & 1) + 3 * ((l >> 2) & 1) - ((l >> 3) & 1) - 5 * ((l >> 4) & 1) + 7 * ((l >> 5) & 1);
}

// Source lane whose butterfly result belongs at heap index `lane` (0..63) of the coefficient array:
// round j (xor 1<<j) produces the level 5-j node m = lane >> (j+1) in every lane of its group; the lane
// (2m+1) << j of the group is the designated holder. Heap index 0 (DC) comes from lane 0.
__device__ __forceinline__ int low_source_lane(int lane) {
    if (lane == 0) return 0;
    const int lv = 31 - __clz(lane);
    const int m = lane - (1 << lv);
    return (2 * m + 1) << (5 - lv);
}

__device__ __forceinline__ int quant_layer(int heap_index) { return 31 - __clz(heap_index + 1); } // quantization.rs:13

constexpr int kMaxPairsPerWave = 2;    // pairs of (cell, channel) items a wave carries in registers per tile
constexpr int kMaxItemsPerTile = 2 * kMaxPairsPerWave * kFwdWaves;
constexpr int kMaxChunksPerThread = 6; // 16-byte chunks a thread stages per tile (tile <= 6 * 256 * 16 B = 24 KiB)

struct FwdArgs {
    const uint8_t *pixels;
    size_t pixel_stride;
    int32_t *coefs;
    size_t coef_stride;
    const Tile *tiles;
    const TileCell *tile_meta; // per tile-cell {cx, cy, cell id, interior}, in tile order
    const int32_t *wg_tiles;   // [n_wg + 1]
    uint32_t n_wg;
    int32_t width, height;
    uint32_t F;
    int32_t pitch;
    int32_t meta_off;  // byte offset of the cell records inside one LDS buffer
    int32_t buf_bytes; // bytes of one LDS buffer (pixel rows + cell records)
    uint32_t cpr, cpr_magic;
    int32_t q_identity;
    int32_t ablate; // timing-only ablation (FRI_HIP_K1_ABLATE): 1 = skip staging, 2 = skip the cell loop, 4 = skip stores. 0 in production.
    unsigned long long *trace; // diagnostic timeline, null in production
    QMatrix q;
};

// ---- packed arithmetic: two (cell, channel) items per register ------------------------------------------------------
// Every value of the transform fits 16 bits (differences in [-255, 255], low-pass values in [0, 255]) and K1 is bound by
// VALU issue (a wave64 integer instruction occupies its SIMD for 4 cycles), so one wave transforms TWO items at once:
// item A in the low half, item B in the high half of each VGPR, v_pk_*_i16 arithmetic. DPP / permlane moves carry both.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

// d = l - r;  s = r + trunc(d / 2) = (l + r + (l < r)) >> 1  for l, r >= 0   (wavelet_transform.rs:211-218).
// Missing (None) operands enter as 0, which is exactly what try_apply substitutes (wavelet_transform.rs:14-26); which
// outputs are None is decided separately from the validity tree (boundary cells only).
__device__ __forceinline__ void pk_pair(u16x2 l, u16x2 r, s16x2 &d, u16x2 &s) {
    d = (s16x2)l - (s16x2)r;
    s = (l + r + ((u16x2)d >> 15)) >> 1;
}

// Round J of the cross-lane part of the tree (level 5-J): combine the two child groups of 2^J lanes.
//   J = 0,1 : DPP quad_perm (lane^1, lane^2)
//   J = 2,3 : DPP row_half_mirror / row_mirror (lane -> 7-lane / 15-lane: lands in the sibling group)
//   J = 4   : v_permlane16_swap  (rows 0<->1, 2<->3)     } called with both operands = s they return
//   J = 5   : v_permlane32_swap  (lower 32 <-> upper 32) } (left child's s, right child's s) in every lane
// All lanes of a child group hold the same low-pass value, so fetching from ANY lane of the sibling group is
// enough -- that is what lets every round be a VALU cross-lane op instead of an LDS permute.
// Bit J of the lane clear = left child (the right child is the one that adds the LITERAL).
template <int J>
__device__ __forceinline__ void xlane_children(int lane, int v, int &l, int &r) {
    if constexpr (J < 4) {
        constexpr int ctrl = J == 0 ? 0xB1 : J == 1 ? 0x4E : J == 2 ? 0x141 : 0x140;
        const int other = __builtin_amdgcn_update_dpp(v, v, ctrl, 0xF, 0xF, false);
        const bool hi = (lane >> J) & 1;
        l = hi ? other : v;
        r = hi ? v : other;
    } else if constexpr (J == 4) {
        const uint2v w = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
        l = (int)w.x;
        r = (int)w.y;
    } else {
        const uint2v w = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        l = (int)w.x;
        r = (int)w.y;
    }
}

template <int J>
__device__ __forceinline__ void pk_cross(int lane, int tz, u16x2 &s, int &vlow) {
    int l, r;
    xlane_children<J>(lane, __builtin_bit_cast(int, s), l, r);
    s16x2 d;
    pk_pair(__builtin_bit_cast(u16x2, l), __builtin_bit_cast(u16x2, r), d, s);
    if (tz == J) vlow = __builtin_bit_cast(int, d);
}

// The 9-level transform of two items held by one wave. leaf[j] = pixels of leaf 8*lane + j of item A (.x) and B (.y).
// res[0..3] = coef[256+4L .. +3], res[4..5] = coef[128+2L ..], res[6] = coef[64+L], res[7] = coef[L], both items packed.
__device__ __forceinline__ void fwd_wave_pk(const u16x2 (&leaf)[8], int lane, int (&res)[8]) {
    s16x2 d8[4], d7[2], d6;
    u16x2 s8[4], s7[2], s;
#pragma unroll
    for (int i = 0; i < 4; i++) pk_pair(leaf[2 * i], leaf[2 * i + 1], d8[i], s8[i]); // level 8: nodes 256 + 4L + i
#pragma unroll
    for (int i = 0; i < 2; i++) pk_pair(s8[2 * i], s8[2 * i + 1], d7[i], s7[i]); // level 7: nodes 128 + 2L + i
    pk_pair(s7[0], s7[1], d6, s);                                                 // level 6: node 64 + L
    const int tz = lane ? __builtin_ctz(lane) : 6;
    int vlow = 0;
    pk_cross<0>(lane, tz, s, vlow); // level 5
    pk_cross<1>(lane, tz, s, vlow);
    pk_cross<2>(lane, tz, s, vlow);
    pk_cross<3>(lane, tz, s, vlow);
    pk_cross<4>(lane, tz, s, vlow);
    pk_cross<5>(lane, tz, s, vlow); // level 0 (root)
    if (lane == 0) vlow = __builtin_bit_cast(int, s); // coefficients[0] = low_pass_values[1] (wavelet_transform.rs:221)
#pragma unroll
    for (int i = 0; i < 4; i++) res[i] = __builtin_bit_cast(int, d8[i]);
    res[4] = __builtin_bit_cast(int, d7[0]);
    res[5] = __builtin_bit_cast(int, d7[1]);
    res[6] = __builtin_bit_cast(int, d6);
    res[7] = __shfl(vlow, low_source_lane(lane));
}

// Which outputs of a boundary cell are None: a node is Some iff at least one leaf below it is inside the image
// (try_apply returns None only for (None, None)). Same tree, OR instead of arithmetic; bits of item A in the low half,
// item B in the high half. m = leaf validity, bit j (+16) = leaf 8*lane + j. Returns the per-lane validity of the 8
// outputs in the layout of res[]: bits 0..3 level 8, bits 4,5 level 7, bit 6 level 6, bit 7 the low-64 coefficient.
__device__ __forceinline__ uint32_t validity_tree_pk(uint32_t m, int lane) {
    const uint32_t v8 = (m | (m >> 1)) & 0x00550055u;   // nodes 256+4L+i at bit 2i
    const uint32_t v7 = (v8 | (v8 >> 2)) & 0x00110011u; // nodes 128+2L+i at bit 4i
    int sv = (int)((v7 | (v7 >> 4)) & 0x00010001u);     // node 64+L
    const uint32_t v6 = (uint32_t)sv;
    const int tz = lane ? __builtin_ctz(lane) : 6;
    int vlow = 0, l, r;
    xlane_children<0>(lane, sv, l, r); sv = l | r; if (tz == 0) vlow = sv;
    xlane_children<1>(lane, sv, l, r); sv = l | r; if (tz == 1) vlow = sv;
    xlane_children<2>(lane, sv, l, r); sv = l | r; if (tz == 2) vlow = sv;
    xlane_children<3>(lane, sv, l, r); sv = l | r; if (tz == 3) vlow = sv;
    xlane_children<4>(lane, sv, l, r); sv = l | r; if (tz == 4) vlow = sv;
    xlane_children<5>(lane, sv, l, r); sv = l | r; if (tz == 5) vlow = sv;
    if (lane == 0) vlow = sv;
    const uint32_t low = (uint32_t)__shfl(vlow, low_source_lane(lane));
    // compact: level 8 bits 0,2,4,6 -> 0..3 ; level 7 bits 0,4 -> 4,5 ; level 6 -> 6 ; low -> 7   (per 16-bit half)
    uint32_t out = (v8 & 0x00010001u) | ((v8 >> 1) & 0x00020002u) | ((v8 >> 2) & 0x00040004u) | ((v8 >> 3) & 0x00080008u);
    out |= ((v7 & 0x00010001u) << 4) | ((v7 & 0x00100010u) << 1);
    out |= (v6 << 6) | (low << 7);
    return out;
}

__device__ __forceinline__ int quant_one(int v, int heap_index, const FwdArgs &a) { // quantization.rs:13-17, None untouched
    return (a.q_identity || v == kNone) ? v : v / a.q.q[quant_layer(heap_index)];
}

// Unpacks one item (HALF = 0: low halves, 1: high halves), applies the None mask and the quantiser, and stores the
// cell's 512 int32 coefficients as four fully coalesced store instructions (1 KiB + 512 B + 256 B + 256 B).
template <int HALF, bool MASKED>
__device__ __forceinline__ void store_item(int32_t *__restrict__ coefs, uint32_t elem_off, int lane, const int (&res)[8], uint32_t valid,
                                           const FwdArgs &a) {
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v[i] = HALF ? (res[i] >> 16) : (int)(short)(res[i] & 0xFFFF);
        if (MASKED && !((valid >> (16 * HALF + i)) & 1u)) v[i] = kNone;
    }
    if (!a.q_identity) {
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = quant_one(v[i], 256 + 4 * lane + i, a);
        v[4] = quant_one(v[4], 128 + 2 * lane, a);
        v[5] = quant_one(v[5], 128 + 2 * lane + 1, a);
        v[6] = quant_one(v[6], 64 + lane, a);
        v[7] = quant_one(v[7], lane, a);
    }
    int32_t *out = coefs + elem_off; // 32-bit element offset off a wave-uniform base: saddr + voffset addressing
    // Streaming (nontemporal) stores: the coefficients are written once and not read back by this kernel. Regular stores leave
    // up to 32 MB of dirty lines in the eight L2s, which the end-of-kernel release then has to write back while nothing else
    // runs. A/B on one box, us per 4096^2 launch: plain 27.5, nt 22.0-23.6, sc1 26.0, sc0 sc1 25.5, sc0 sc1 nt 22.0-23.2.
    __builtin_nontemporal_store(i32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<i32x4 *>(out + 256 + 4 * lane));
    __builtin_nontemporal_store(i32x2{v[4], v[5]}, reinterpret_cast<i32x2 *>(out + 128 + 2 * lane));
    __builtin_nontemporal_store(v[6], out + 64 + lane);
    __builtin_nontemporal_store(v[7], out + lane);
}

// Native vector type on purpose: HIP's uint4 is a struct whose copies become llvm.memcpy, which kept the staging
// array in scratch memory (and put a vmcnt(0) behind every load).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// A tile is staged as a stream of 16-byte chunks, chunk i handled by thread i % 256, NCH chunks per thread:
//   i <  total = n_rows * cpr : pixel chunk, row r = i / cpr, column chunk k = i % cpr. LDS byte r * pitch + 16 k
//                               (= 16 i, because pitch = 16 cpr) holds global byte (row_start(r) & ~15) + 16 k, so
//                               every global access is an aligned 16-byte vector load whatever the image width and
//                               pointer alignment are.
//   total <= i < limit        : cell record i - total of the tile (TileCell is 16 bytes), LDS byte meta_off + 16 (i - total).
// Both halves are branch-free on purpose. A thread with nothing to fetch for a slot loads the tile's first cell record
// (always valid) and commits it to a private junk slot. If a load could be skipped, or a commit bypassed, on some path,
// the compiler's waitcnt pass would see a load still pending around the loop back-edge and drain vmcnt(0) at the top of
// the next iteration -- which also waits for the coefficient stores just issued (vmcnt counts stores on CDNA4).
// EDGE = false (picked by the host when the image base is 16-byte aligned and its size a multiple of 16): no chunk can
// straddle the ends of the caller's buffer, so every chunk is exactly one global_load_dwordx4.
// The chunk -> (row, column) map depends on the plan only; each thread computes its NCH pairs once (ChunkMap).
template <int NCH>
struct ChunkMap {
    uint32_t row_off[NCH]; // r * width * C : byte offset of the chunk's row from the tile's first row
    uint32_t k16[NCH];     // 16 * k
    uint32_t r[NCH];
};

// FAST = !EDGE && (width * C) % 16 == 0: every row of the image starts at the same offset modulo 16, so the lead-in of
// a tile's rows is one scalar and chunk c of a thread sits at (scalar tile base) + row_off[c] + k16[c] -- the address
// arithmetic per chunk collapses to a compare and a select. The cell records are fetched by their own load (slot NCH).
template <int C, bool EDGE, bool FAST, int NCH>
__device__ __forceinline__ void stage_issue(const FwdArgs &a, const Tile &t, const uint8_t *__restrict__ img, int tid, const ChunkMap<NCH> &cm,
                                            u32x4 (&sv)[NCH + 1]) {
    // 32-bit byte offsets from the image base (an image is < 4 GiB, images.rs:94); the loads go through `img` so they stay
    // global_load (a flat_load would also count on lgkmcnt and tie the prefetch to every LDS wait of the transform).
    const uint32_t a16 = (uint32_t)reinterpret_cast<uintptr_t>(img);
    const uint32_t img_bytes = (uint32_t)a.width * (uint32_t)a.height * C;
    const uint32_t n_rows = (a.ablate & 1) ? 0u : (uint32_t)t.n_rows;
    const uint32_t row_bytes = (uint32_t)t.width_px * C;
    const uint32_t tile_off = ((uint32_t)t.y_lo * (uint32_t)a.width + (uint32_t)t.x_lo) * C;
    const u32x4 *meta = reinterpret_cast<const u32x4 *>(a.tile_meta + t.cell_begin);
    sv[NCH] = meta[tid < t.cell_count ? tid : 0];
    if (FAST) {
        const uint32_t lead = (a16 + tile_off) & 15u; // same for every row of the tile
        const uint8_t *base = img + (tile_off - lead); // 16-byte aligned, wave-uniform
        const uint32_t span = lead + row_bytes;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const bool need = cm.r[c] < n_rows && cm.k16[c] < span;
            sv[c] = *reinterpret_cast<const u32x4 *>(base + (need ? cm.row_off[c] + cm.k16[c] : 0u)); // dummy = the tile's first chunk
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint32_t off = tile_off + cm.row_off[c];     // first byte of the row segment
        const uint32_t lead = (a16 + off) & 15u;           // bytes between the 16-byte boundary below and that byte
        const uint32_t ca = off - lead + cm.k16[c];        // chunk's byte offset from img (may wrap below 0 when EDGE)
        const bool need = cm.r[c] < n_rows && cm.k16[c] < lead + row_bytes;
        if (!EDGE) {
            sv[c] = *(need ? reinterpret_cast<const u32x4 *>(img + ca) : meta);
        } else {
            const long long cas = (long long)off - (long long)lead + (long long)cm.k16[c]; // signed chunk offset from img
            const bool whole = need && cas >= 0 && cas + 16 <= (long long)img_bytes;
            u32x4 v = *(whole ? reinterpret_cast<const u32x4 *>(img + ca) : meta);
            if (need && !whole) { // first/last chunk of the buffer: stay inside the caller's allocation
                uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int b = 0; b < 16; b++) { // fully unrolled: w[] stays in registers
                    const uint32_t p = ca + (uint32_t)b; // wraps for bytes below the buffer start -> >= img_bytes
                    if (p < img_bytes) w[b >> 2] |= (uint32_t)img[p] << (8 * (b & 3));
                }
                v = u32x4{w[0], w[1], w[2], w[3]};
            }
            sv[c] = v;
        }
    }
}

// Branch-free like stage_issue: chunks a thread did not need go to its private junk slot.
template <int NCH>
__device__ __forceinline__ void stage_commit(const FwdArgs &a, const Tile &t, uint8_t *buf, uint8_t *junk, int tid, const u32x4 (&sv)[NCH + 1]) {
    const uint32_t total = (a.ablate & 1) ? 0u : (uint32_t)t.n_rows * a.cpr;
    uint8_t *mine = junk + 16 * tid;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)c * kFwdThreads;
        *reinterpret_cast<u32x4 *>(i < total ? buf + 16u * i : mine) = sv[c];
    }
    *reinterpret_cast<u32x4 *>(tid < t.cell_count ? buf + a.meta_off + 16 * tid : mine) = sv[NCH];
}

// Blocks are dealt round-robin over the 8 XCDs (each with its own L2). Give XCD x one contiguous range of
// workgroup shares, so that shares sharing pixel rows (their tile halos overlap) hit in the same L2.
// Placement only changes speed, never results.
__device__ __forceinline__ uint32_t xcd_contiguous_share(uint32_t b, uint32_t n) {
    const uint32_t x = b & 7u, idx = b >> 3, q = n >> 3, r = n & 7u;
    return x * q + min(x, r) + idx;
}

// Diagnostic timeline (instrumented build + FRI_HIP_TRACE=1; compiled out of the product): thread 0 of a workgroup stamps the
// 100 MHz constant clock into slot `slot` of its share's record. Slot 0 = entry, 1 = prologue done, 2 + i = tile i done
// (i < 12), 14 = hardware id (HW_ID | XCC_ID << 32), 15 = exit.
#ifndef FRI_HIP_ENABLE_TRACE
#define FRI_HIP_ENABLE_TRACE 0 // `make trace` builds the instrumented library; the stamps are compiled out of the product
#endif
constexpr bool kTraceBuild = FRI_HIP_ENABLE_TRACE != 0;
constexpr int kTraceSlots = 16;
__device__ __forceinline__ void trace_stamp(unsigned long long *trace, uint32_t wg, int slot, int tid) {
    if (kTraceBuild && trace && tid == 0) trace[(size_t)wg * kTraceSlots + min(slot, 13)] = wall_clock64();
}
__device__ __forceinline__ void trace_exit(unsigned long long *trace, uint32_t wg, int tid) {
    if (kTraceBuild && trace && tid == 0) {
        const unsigned long long hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        trace[(size_t)wg * kTraceSlots + 14] = hw | (xcc << 32);
        trace[(size_t)wg * kTraceSlots + 15] = wall_clock64();
    }
}

// Workgroup barrier that orders LDS traffic only. __syncthreads() would also emit s_waitcnt vmcnt(0), i.e. wait
// for every coefficient store of the tile to drain (vmcnt counts stores on CDNA4) -- exactly the latency the
// pipeline is built to hide. Global memory is never exchanged between the waves of a workgroup here.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- leaf fetch for single-channel planes ----------------------------------------------------------------------------
// rb[dy] = LDS byte offset of pixel (x0, y0 + dy). w[0] = pixels x0..x0+3 of row 0, w[1], w[2] = pixels x0-1..x0+2 of rows 1, 2.
__device__ __forceinline__ void fetch_windows(const uint8_t *cur, const int (&rb)[3], uint32_t (&w)[3]) {
#pragma unroll
    for (int row = 0; row < 3; row++) {
        const uint32_t b = (uint32_t)(rb[row] - (row ? 1 : 0));
        const uint32_t *p = reinterpret_cast<const uint32_t *>(cur + (b & ~3u));
        w[row] = __builtin_amdgcn_alignbyte(p[1], p[0], b & 3u);
    }
}
// leaf_mask bit j = leaf j is inside the image; clears the window bytes of the leaves that are not.
__device__ __forceinline__ void mask_windows(uint32_t leaf_mask, uint32_t (&w)[3]) {
    auto ff = [&](int j, int byte) { return ((leaf_mask >> j) & 1u) ? (0xFFu << (8 * byte)) : 0u; };
    w[0] &= ff(0, 0) | ff(4, 2);
    w[1] &= ff(2, 0) | ff(1, 1) | ff(6, 2) | ff(5, 3);
    w[2] &= ff(3, 0) | ff(7, 2);
}
// {byte I of a, byte I of b} zero-extended into the two 16-bit halves.
template <int I>
__device__ __forceinline__ u16x2 pair_bytes(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(u16x2, __builtin_amdgcn_perm(b, a, 0x0C000C00u | ((4u + I) << 16) | (unsigned)I));
}

// One item of a pair: where its 8 leaves sit in the staged rectangle, and which of them are inside the image.
struct ItemAddr {
    int rb[3];          // LDS byte offset of (x0, y0 + dy, ch)
    uint32_t leaf_mask; // bit j = leaf j inside the image (0xFF for interior cells)
    uint32_t elem_off;  // (ch * F + cell) * 512
};

template <int C, bool FAST>
__device__ __forceinline__ ItemAddr item_addr(const FwdArgs &a, const Tile &t, const TileCell *meta, int it, int ldx, int ldy, uint32_t sh_base,
                                               uint32_t wc16) {
    const int cl = it / C, ch = it - cl * C;
    const TileCell m = meta[cl];
    ItemAddr r;
    r.elem_off = ((uint32_t)ch * a.F + (uint32_t)__builtin_amdgcn_readfirstlane(m.cell)) * kCell;
    const int x0 = m.cx + ldx, y0 = m.cy + ldy;
    // row y of the staged rectangle starts (a16 + (y * width + x_lo) * C) & 15 bytes into its LDS row
    const int col = __mul24(x0 - t.x_lo, C) + ch;
    if (FAST) { // width * C is a multiple of 16: the same lead-in for every row
        r.rb[0] = __mul24(y0 - t.y_lo, a.pitch) + col + (int)(sh_base & 15u);
        r.rb[1] = r.rb[0] + a.pitch;
        r.rb[2] = r.rb[1] + a.pitch;
    } else {
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int y = y0 + dy;
            const uint32_t sh = (sh_base + __umul24((uint32_t)y & 15u, wc16)) & 15u;
            r.rb[dy] = __mul24(y - t.y_lo, a.pitch) + (int)sh + col;
        }
    }
    r.leaf_mask = 0xFFu;
    if (__builtin_amdgcn_readfirstlane(m.interior) == 0) { // wave-uniform: boundary cell
        r.leaf_mask = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int x = x0 + leaf_dx(j), y = y0 + leaf_dy(j);
            if (x >= 0 && y >= 0 && x < a.width && y < a.height) r.leaf_mask |= 1u << j; // get_pixel, images.rs:90
        }
    }
    return r;
}

// K1. grid = (workgroup shares, images), block = 256 (4 waves). Each workgroup walks the tiles of its share:
// while tile i is being transformed out of one LDS buffer (and its coefficient stores drain), the pixel
// rectangle of tile i+1 is already in flight from HBM/L2 into registers and is committed to the other buffer.
template <int C, bool EDGE, bool FAST, int NCH>
__global__ void __launch_bounds__(kFwdThreads) fwd_transform_quant_kernel(const FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wg = xcd_contiguous_share(blockIdx.x, a.n_wg);
    const int tb = a.wg_tiles[wg], te = a.wg_tiles[wg + 1];
    trace_stamp(a.trace, wg, 0, tid);
    // The share's tile descriptors go to LDS once, before any store is issued: fetching them inside the loop would be a
    // vector load behind s_waitcnt vmcnt(0) per tile (scalar loads are off the table once the kernel has stored), and that
    // wait would also drain the previous tile's coefficient stores.
    uint8_t *junk = lds + 2 * a.buf_bytes; // 16 bytes per thread, written, never read
    Tile *lds_tiles = reinterpret_cast<Tile *>(junk + 16 * kFwdThreads);
    if (tid < te - tb) lds_tiles[tid] = a.tiles[tb + tid];
    const uint8_t *img = a.pixels + (size_t)blockIdx.y * a.pixel_stride;
    int32_t *coefs = a.coefs + (size_t)blockIdx.y * a.coef_stride;
    const int ldx = lane_dx(lane), ldy = lane_dy(lane);
    const uint32_t a16 = (uint32_t)reinterpret_cast<uintptr_t>(img);
    const uint32_t wc = (uint32_t)a.width * C, wc16 = wc & 15u;

    ChunkMap<NCH> cm;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)c * kFwdThreads;
        const uint32_t r = a.cpr == 1 ? i : __umulhi(i, a.cpr_magic); // i / cpr (the magic for cpr == 1 would be 2^32)
        cm.r[c] = r;
        cm.k16[c] = 16u * (i - r * a.cpr);
        cm.row_off[c] = r * wc;
    }

    u32x4 st[NCH + 1]; // one tile's worth of in-flight 16-byte chunks of this thread (+ its cell record)
    Tile t = a.tiles[tb];
    stage_issue<C, EDGE, FAST, NCH>(a, t, img, tid, cm, st);
    stage_commit<NCH>(a, t, lds, junk, tid, st);
    __syncthreads();
    trace_stamp(a.trace, wg, 1, tid);

    for (int ti = tb; ti < te; ti++) {
        const uint8_t *cur = lds + ((ti - tb) & 1) * a.buf_bytes;
        uint8_t *nxt = lds + (((ti - tb) & 1) ^ 1) * a.buf_bytes;
        const bool more = ti + 1 < te;
        Tile tn = t;
        if (more) {
            const Tile &l = lds_tiles[ti + 1 - tb];
            tn.x_lo = __builtin_amdgcn_readfirstlane(l.x_lo);
            tn.y_lo = __builtin_amdgcn_readfirstlane(l.y_lo);
            tn.width_px = __builtin_amdgcn_readfirstlane(l.width_px);
            tn.n_rows = __builtin_amdgcn_readfirstlane(l.n_rows);
            tn.cell_begin = __builtin_amdgcn_readfirstlane(l.cell_begin);
            tn.cell_count = __builtin_amdgcn_readfirstlane(l.cell_count);
            stage_issue<C, EDGE, FAST, NCH>(a, tn, img, tid, cm, st); // loads stay in flight across the transform below
        }

        const TileCell *meta = reinterpret_cast<const TileCell *>(cur + a.meta_off);
        const int n_items = (a.ablate & 2) ? 0 : t.cell_count * C;
        const uint32_t sh_base = a16 + (uint32_t)t.x_lo * C;
        int res[kMaxPairsPerWave][8];
        uint32_t offA[kMaxPairsPerWave], offB[kMaxPairsPerWave], valid[kMaxPairsPerWave];
#pragma unroll
        for (int c = 0; c < kMaxPairsPerWave; c++) {
            const int itA = 2 * (wave + kFwdWaves * c);
            offA[c] = offB[c] = 0;
            valid[c] = 0xFFFFFFFFu;
            if (itA < n_items) {
                const int itB = itA + 1 < n_items ? itA + 1 : itA; // odd tail: item B mirrors A and is not stored
                const ItemAddr A = item_addr<C, FAST>(a, t, meta, itA, ldx, ldy, sh_base, wc16);
                const ItemAddr B = item_addr<C, FAST>(a, t, meta, itB, ldx, ldy, sh_base, wc16);
                offA[c] = A.elem_off;
                offB[c] = B.elem_off;
                u16x2 leaf[8];
                if constexpr (C == 1) {
                    // Sub-dword LDS reads are slow (a ds_read_u8 wave-instruction holds the LDS pipe for ~16 cycles), and a
                    // lane's 8 leaves sit in three 4-byte windows: row 0 = x..x+3, rows 1 and 2 = x-1..x+2. Fetch each window
                    // as two aligned dwords + v_alignbyte, then v_perm pairs item A's and item B's bytes into 16-bit halves.
                    uint32_t wA[3], wB[3];
                    fetch_windows(cur, A.rb, wA);
                    fetch_windows(cur, B.rb, wB);
                    if ((A.leaf_mask & B.leaf_mask) != 0xFFu) { // leaves outside the image enter as 0
                        mask_windows(A.leaf_mask, wA);
                        mask_windows(B.leaf_mask, wB);
                    }
                    // window byte of leaf j: row 0: leaf0 -> 0, leaf4 -> 2; row 1: leaf2 -> 0, leaf1 -> 1, leaf6 -> 2, leaf5 -> 3; row 2: leaf3 -> 0, leaf7 -> 2
                    leaf[0] = pair_bytes<0>(wA[0], wB[0]);
                    leaf[1] = pair_bytes<1>(wA[1], wB[1]);
                    leaf[2] = pair_bytes<0>(wA[1], wB[1]);
                    leaf[3] = pair_bytes<0>(wA[2], wB[2]);
                    leaf[4] = pair_bytes<2>(wA[0], wB[0]);
                    leaf[5] = pair_bytes<3>(wA[1], wB[1]);
                    leaf[6] = pair_bytes<2>(wA[1], wB[1]);
                    leaf[7] = pair_bytes<2>(wA[2], wB[2]);
                } else if ((A.leaf_mask & B.leaf_mask) == 0xFFu) {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        leaf[j].x = cur[A.rb[leaf_dy(j)] + leaf_dx(j) * C];
                        leaf[j].y = cur[B.rb[leaf_dy(j)] + leaf_dx(j) * C];
                    }
                } else { // some lane has a leaf outside the image: it enters as 0, the None outputs come from the validity tree
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        leaf[j].x = ((A.leaf_mask >> j) & 1u) ? cur[A.rb[leaf_dy(j)] + leaf_dx(j) * C] : (uint8_t)0;
                        leaf[j].y = ((B.leaf_mask >> j) & 1u) ? cur[B.rb[leaf_dy(j)] + leaf_dx(j) * C] : (uint8_t)0;
                    }
                }
                fwd_wave_pk(leaf, lane, res[c]);
                if (__builtin_amdgcn_readfirstlane(meta[itA / C].interior & meta[itB / C].interior) == 0)
                    valid[c] = validity_tree_pk(A.leaf_mask | (B.leaf_mask << 16), lane);
            }
        }
        if (more) stage_commit<NCH>(a, tn, nxt, junk, tid, st);
#pragma unroll
        for (int c = 0; c < kMaxPairsPerWave; c++) {
            const int itA = 2 * (wave + kFwdWaves * c);
            if (itA < n_items) {
                bool go = !(a.ablate & 4);
                if (!go) go = (res[c][0] ^ res[c][1] ^ res[c][2] ^ res[c][3] ^ res[c][4] ^ res[c][5] ^ res[c][6] ^ res[c][7]) == 0x12345678; // keeps the arithmetic alive
                if (go) {
                    if (__builtin_amdgcn_readfirstlane(valid[c] == 0xFFFFFFFFu ? 1 : 0) && __all(valid[c] == 0xFFFFFFFFu)) { // no None anywhere in the pair
                        store_item<0, false>(coefs, offA[c], lane, res[c], valid[c], a);
                        if (itA + 1 < n_items) store_item<1, false>(coefs, offB[c], lane, res[c], valid[c], a);
                    } else {
                        store_item<0, true>(coefs, offA[c], lane, res[c], valid[c], a);
                        if (itA + 1 < n_items) store_item<1, true>(coefs, offB[c], lane, res[c], valid[c], a);
                    }
                }
            }
        }
        trace_stamp(a.trace, wg, 2 + ti - tb, tid);
        lds_barrier();
        t = tn;
    }
    trace_exit(a.trace, wg, tid);
}

// ------------------------------------------------------------------------------------------------
// K2: prediction + bucket + histogram for one channel plane.
// ------------------------------------------------------------------------------------------------
constexpr int kPredThreads = 512; // 8 waves
constexpr int kPredWaves = kPredThreads / 64;
constexpr int kHistBins = 10 * 1024;
constexpr int kSlotStride = 1040; // bytes per staged cell: 512 int16 + 8 zero halfwords (what "never a node" entries read); 16-byte multiple

// Rust `f32 as u32` / `f32 as i32` (prediction.rs:56, :206): truncation toward zero, saturating, NaN -> 0. That is exactly
// what gfx950's v_cvt_u32_f32 / v_cvt_i32_f32 do in hardware; a C++ cast would be undefined out of range, so the
// instructions are named explicitly (pure VALU, no memory, no wait states to manage).
__device__ __forceinline__ uint32_t f32_as_u32(float x) {
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ int f32_as_i32(float x) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
// assign_bucket, prediction.rs:55-68: 0..3->0, 3..5->1, 5..6->2, 6..8->3, 8..12->4, 12..16->5, 16..20->6, 20..25->7, 25..30->8, 30..->9
// as four 32-entry bit planes indexed by min(width as u32, 31).
__host__ __device__ constexpr uint32_t bucket_of(uint32_t w) {
    return w < 3 ? 0 : w < 5 ? 1 : w < 6 ? 2 : w < 8 ? 3 : w < 12 ? 4 : w < 16 ? 5 : w < 20 ? 6 : w < 25 ? 7 : w < 30 ? 8 : 9;
}
__host__ __device__ constexpr uint32_t bucket_plane(int bit) {
    uint32_t m = 0;
    for (uint32_t w = 0; w < 32; w++) m |= ((bucket_of(w) >> bit) & 1u) << w;
    return m;
}
__device__ __forceinline__ uint32_t bucket_of_rt(uint32_t width_u32) {
    const uint32_t w = min(width_u32, 31u);
    constexpr uint32_t P0 = bucket_plane(0), P1 = bucket_plane(1), P2 = bucket_plane(2), P3 = bucket_plane(3);
    return __builtin_amdgcn_ubfe(P0, w, 1) | (__builtin_amdgcn_ubfe(P1, w, 1) << 1) | (__builtin_amdgcn_ubfe(P2, w, 1) << 2) |
           (__builtin_amdgcn_ubfe(P3, w, 1) << 3);
}
__device__ __forceinline__ uint32_t assign_bucket(float width) { return bucket_of_rt(f32_as_u32(width)); }
__device__ __forceinline__ int iabs_w(int a) { return a < 0 ? (int)(0u - (unsigned)a) : a; }
__device__ __forceinline__ int sub_w(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int add_w(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
// pack_signed, utils.rs:34-40 (k >= 0 -> 2k, k < 0 -> -2k - 1; wrapping arithmetic like a release build) = zig-zag
__device__ __forceinline__ uint32_t pack_signed(int k) { return ((uint32_t)k << 1) ^ (uint32_t)(k >> 31); }

struct PredArgs {
    const int32_t *coefs;      // one channel plane [F][512]
    const int32_t *pred_slots; // [n_tiles][kPredSlots]
    const uint16_t *nbr_table; // [512][6]
    const uint32_t *pred_off;  // [512][4] packed neighbour halfword offsets of every node (build_pred_offsets)
    const uint8_t *interior;   // [F]
    const uint32_t *valid_mask; // [F][16]
    uint8_t *bucket;
    int32_t *prediction;
    uint32_t *hist;
    unsigned long long *n_oob;
    uint8_t *junk;             // plan scratch, kPredJunkBytes per wave of the pipelined K2: output lines of block slots without a cell
    unsigned long long *trace; // diagnostic timeline, null in production
    uint32_t *acc;             // plan scratch, all zero between launches: [kHistBins] counts, then kAccOob (u64), kAccTicket
    uint32_t n_tiles;
    PredictParams pp;
};

// Histogram hand-over without a memset in front of the kernel (two fill kernels cost ~6 us per call): every workgroup adds
// its LDS table into the plan's accumulator, then takes a ticket; the workgroup that draws the last ticket moves the totals to
// the caller's arrays with atomic exchanges, which leaves the accumulator zero for the next launch.
constexpr int kAccOob = kHistBins, kAccTicket = kHistBins + 2;
static_assert(kHistBins + 4 == (int)kPredAccWords, "accumulator layout");
__device__ __forceinline__ void pred_hand_over(const PredArgs &a, const uint32_t *s_hist, uint32_t *s_flag, int tid, int n_threads) {
    for (int i = tid; i < kHistBins; i += n_threads) {
        const uint32_t c = s_hist[i];
        if (c) __hip_atomic_fetch_add(a.acc + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0 && s_hist[kHistBins])
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.acc + kAccOob), (unsigned long long)s_hist[kHistBins], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Order without fences: an agent-scope fence on this multi-XCD part writes back and invalidates the whole L2 (measured:
    // +80 us per launch). The adds above are device-scope atomics, executed at the coherence point and acknowledged through
    // vmcnt; __syncthreads() waits for vmcnt(0) in every wave, so all of this workgroup's adds are performed before thread 0
    // draws the ticket. The last workgroup then reads with device-scope loads, which do not hit a stale L2 line.
    __syncthreads();
    if (tid == 0) *s_flag = __hip_atomic_fetch_add(a.acc + kAccTicket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (*s_flag == 0) return;
    // all other workgroups have finished (their adds precede their tickets): plain coherent loads, all in flight together
    // (an atomic exchange per bin, one after the other, took 30-60 us), then the zeros for the next launch
    static_assert(kHistBins % 512 == 0, "unrolled by 512-thread strides");
    if (n_threads == 1024) {
        uint32_t v[kHistBins / 1024];
#pragma unroll
        for (int k = 0; k < kHistBins / 1024; k++) v[k] = __hip_atomic_load(a.acc + tid + 1024 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < kHistBins / 1024; k++) {
            a.hist[tid + 1024 * k] = v[k];
            __hip_atomic_store(a.acc + tid + 1024 * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        uint32_t v[kHistBins / 512];
#pragma unroll
        for (int k = 0; k < kHistBins / 512; k++) v[k] = __hip_atomic_load(a.acc + tid + 512 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < kHistBins / 512; k++) {
            a.hist[tid + 512 * k] = v[k];
            __hip_atomic_store(a.acc + tid + 512 * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) {
        *a.n_oob = __hip_atomic_exchange(reinterpret_cast<unsigned long long *>(a.acc + kAccOob), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.acc + kAccTicket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

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
__device__ __forceinline__ void pred_stage_tile(const int32_t *__restrict__ coefs, const int32_t *s_slot_cell, uint8_t *s_cells, int lane, int wave,
                                                uint32_t *outlier_counter = nullptr) {
    for (int slot = wave; slot < kPredSlots; slot += kPredWaves) {
        const int cell = s_slot_cell[slot];
        int4 lo = make_int4(0, 0, 0, 0), hi = lo;
        if (cell >= 0) {
            const int4 *src = reinterpret_cast<const int4 *>(coefs + (size_t)cell * kCell + 8 * lane);
            lo = src[0];
            hi = src[1];
            if (outlier_counter && pred_is_block_slot(slot)) { // every cell is a block cell of exactly one tile
                const int v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const uint32_t n = pred_count_outliers(v);
                if (n) atomicAdd(outlier_counter, n);
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

// One node of the gather/predict/histogram loop. P_HI = p >> 6 is compile time, so the parameter group
// (prediction.rs:165-179: level 8 -> 0, level 7 -> 1, levels 1..6 -> 2) is too and the parameters stay in SGPRs.
template <int I>
__device__ __forceinline__ void predict_node(const uint8_t *own, int lane, uint32_t o01, uint32_t o23, uint32_t o45, bool some, const PredictParams &pp,
                                             uint32_t *s_hist, uint8_t *bucket_dst, int32_t *pred_dst) {
    constexpr int g = I >= 4 ? 0 : I >= 2 ? 1 : 2;
    const float *wp = pp.width[g], *vp = pp.value[g];
    const int value = *reinterpret_cast<const short *>(own + 2 * (lane + 64 * I));
    // neighbour halfword offsets relative to the own slot, two per register
    const int o[6] = {(int)(short)(o01 & 0xFFFFu), (int)o01 >> 16, (int)(short)(o23 & 0xFFFFu), (int)o23 >> 16, (int)(short)(o45 & 0xFFFFu), (int)o45 >> 16};
    float f[6];
    int v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        v[k] = *reinterpret_cast<const short *>(own + 2 * o[k]);
        f[k] = (float)v[k];
    }
    // get_hf_context_bucket, prediction.rs:165-206: f32, left to right, one rounding per op. The reference takes |a - b|
    // on i32 and converts; for these magnitudes |f32(a) - f32(b)| is the same exact value, and the absolute value rides
    // on the multiply as a source modifier.
    float width = wp[0];
    width = __fadd_rn(width, __fmul_rn(wp[1], fabsf(__fsub_rn(f[0], f[3]))));
    width = __fadd_rn(width, __fmul_rn(wp[2], fabsf(__fsub_rn(f[1], f[2]))));
    width = __fadd_rn(width, __fmul_rn(wp[3], fabsf(__fsub_rn(f[4], f[5]))));
    width = __fadd_rn(width, __fmul_rn(wp[4], fabsf(__fsub_rn(f[1], f[5]))));
    width = __fadd_rn(width, __fmul_rn(wp[5], fabsf(__fsub_rn(f[2], f[4]))));
    uint32_t bucket = assign_bucket(width);
    float pf = __fmul_rn(f[0], vp[0]);
    pf = __fadd_rn(pf, __fmul_rn(f[1], vp[1]));
    pf = __fadd_rn(pf, __fmul_rn(f[2], vp[2]));
    pf = __fadd_rn(pf, __fmul_rn(f[3], vp[3]));
    pf = __fadd_rn(pf, __fmul_rn(f[4], vp[4]));
    pf = __fadd_rn(pf, __fmul_rn(f[5], vp[5]));
    int prediction = f32_as_i32(pf);
    if (I == 0) { // heap index 0 (DC) and 1 (root) live in lanes 0, 1: get_lf_context_bucket, prediction.rs:134-144
        const uint32_t w = (uint32_t)iabs_w(sub_w(v[0], v[2]));
        const int mx = max(v[0], v[2]), mn = min(v[0], v[2]);
        const int lf_pred = v[1] >= mx ? mx : v[1] <= mn ? mn : sub_w(add_w(v[0], v[2]), v[1]);
        const bool lf = lane < 2;
        bucket = lf ? bucket_of_rt(w) : bucket;
        prediction = lf ? lf_pred : prediction;
    }
    // an out-of-alphabet symbol (the reference would panic, entropy_coding.rs:99) goes to the counter bin behind the 10 x 1024 table
    const uint32_t sym = pack_signed(sub_w(value, prediction));
    const uint32_t bin = sym < 1024u ? bucket * 1024u + sym : (uint32_t)kHistBins;
    if (some) atomicAdd(&s_hist[bin], 1u); // bump_freq, entropy_coding.rs:98-100. None nodes are masked off, not sent to a common trash
                                           // bin: 64 lanes adding to one LDS address take ~0.7 us per instruction
    // a None node is never written by the reference and stays (0, 0) (wavelet_transform.rs:60-64)
    if (bucket_dst) bucket_dst[64 * I] = (uint8_t)(some ? bucket : 0u);
    if (pred_dst) pred_dst[64 * I] = some ? prediction : 0;
}

// K2. Persistent workgroups (2 per CU), each walks tiles = 4 x 4 blocks of cells in lattice coordinates. Per tile the 36
// cells of the block plus its halo ring are staged into LDS as int16 (every coefficient fits; None and missing cells are
// stored as 0, which is what the reference's unwrap_or(0) yields), so the 6-neighbour gather of
// ContextModeler::get_neighbour_values (context_modeling.rs:25-77) is an LDS gather: the neighbour of node p sits at
// (own slot + slot delta) * kSlotStride + 2 * heap, and both are image independent -- each lane keeps the 48 offsets of
// its 8 nodes in registers (two per VGPR) for the whole kernel. Lane L owns nodes L, L + 64, ..., L + 448 of a cell:
// neighbouring lanes touch neighbouring halfwords (no structural bank conflict).
__global__ void __launch_bounds__(kPredThreads, 4) predict_histogram_kernel(const PredArgs a) {
    __shared__ uint32_t s_hist[kHistBins + 2]; // + out-of-alphabet counter + trash bin
    __shared__ __attribute__((aligned(16))) uint8_t s_cells[kPredSlots * kSlotStride];
    __shared__ int32_t s_slot_cell[kPredSlots];
    __shared__ uint32_t s_flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kHistBins + 2; i += kPredThreads) s_hist[i] = 0;

    uint32_t off[8][3]; // neighbour halfword offsets relative to the own slot, loop invariant
#pragma unroll
    for (int i = 0; i < 8; i++) { // precomputed at plan creation (build_pred_offsets): 8 loads, no arithmetic
        const u32x4 o = reinterpret_cast<const u32x4 *>(a.pred_off)[lane + 64 * i];
        off[i][0] = o.x, off[i][1] = o.y, off[i][2] = o.z;
    }

    const PredTileWalk walk(a.n_tiles);
    for (uint32_t tile = walk.first; tile < walk.end; tile += walk.step) {
        __syncthreads(); // everyone is done with the previous tile's LDS image (and the histogram is zeroed on the first pass)
        if (tid < kPredSlots) s_slot_cell[tid] = pred_slot_cell(a.pred_slots[(size_t)tile * kPredSlots + tid]);
        __syncthreads();
        pred_stage_tile(a.coefs, s_slot_cell, s_cells, lane, wave, &s_hist[kHistBins]);
        __syncthreads();

        for (int r = wave; r < kPredBlock * kPredBlock; r += kPredWaves) { // two block cells per wave
            const int slot = (1 + r / kPredBlock) * kPredSide + 1 + (r % kPredBlock);
            const int cell = s_slot_cell[slot];
            if (cell < 0) continue;
            const uint8_t *own = s_cells + slot * kSlotStride;
            // Some/None of this lane's 8 nodes: node lane + 64 i is bit (lane & 31) of mask word 2 i + (lane >> 5)
            uint32_t some_bits = 0xFFu;
            if (__builtin_amdgcn_readfirstlane((int)a.interior[cell]) == 0) { // wave-uniform: boundary cell
                some_bits = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) some_bits |= ((a.valid_mask[(size_t)cell * 16 + 2 * i + (lane >> 5)] >> (lane & 31)) & 1u) << i;
            }
            const size_t base = (size_t)cell * kCell + lane;
            uint8_t *bd = a.bucket ? a.bucket + base : nullptr;
            int32_t *pd = a.prediction ? a.prediction + base : nullptr;
            predict_node<0>(own, lane, off[0][0], off[0][1], off[0][2], some_bits & 1u, a.pp, s_hist, bd, pd);
            predict_node<1>(own, lane, off[1][0], off[1][1], off[1][2], some_bits & 2u, a.pp, s_hist, bd, pd);
            predict_node<2>(own, lane, off[2][0], off[2][1], off[2][2], some_bits & 4u, a.pp, s_hist, bd, pd);
            predict_node<3>(own, lane, off[3][0], off[3][1], off[3][2], some_bits & 8u, a.pp, s_hist, bd, pd);
            predict_node<4>(own, lane, off[4][0], off[4][1], off[4][2], some_bits & 16u, a.pp, s_hist, bd, pd);
            predict_node<5>(own, lane, off[5][0], off[5][1], off[5][2], some_bits & 32u, a.pp, s_hist, bd, pd);
            predict_node<6>(own, lane, off[6][0], off[6][1], off[6][2], some_bits & 64u, a.pp, s_hist, bd, pd);
            predict_node<7>(own, lane, off[7][0], off[7][1], off[7][2], some_bits & 128u, a.pp, s_hist, bd, pd);
        }
    }
    __syncthreads();
    pred_hand_over(a, s_hist, &s_flag, tid, kPredThreads);
}

// K2, pipelined form. One 1024-thread workgroup per CU (16 waves = the 16 block cells of a tile), two LDS cell images:
// while tile i is gathered / predicted out of one image, the 36 cells of tile i + 1 are in flight from HBM/L2 into registers
// (2-3 cells per wave) and are committed to the other image at the end of the iteration, so a tile costs one barrier and
// the staging latency overlaps the arithmetic (the single-buffered kernel above leaves the VALU idle 44 % of the time).
// The slot lists (which cell sits in which LDS slot) run two tiles ahead through a three-entry ring. Bucket and prediction
// are written once and never read here: nontemporal stores.
constexpr int kPred2Threads = 1024;
constexpr int kPred2Waves = kPred2Threads / 64;
constexpr int kPred2Stage = (kPredSlots + kPred2Waves - 1) / kPred2Waves; // cells staged per wave
constexpr int kPredCellsBytes = kPredSlots * kSlotStride;
static_assert(kPred2Threads / 64 == (int)kPredJunkWaves && 512 + 2048 == (int)kPredJunkBytes, "junk layout");
constexpr int kPredHistBytes = ((kHistBins + 2) * 4 + 15) & ~15;
constexpr int kPredMaskWords = kPredSlots * 16; // Some/None masks of the staged cells, per image
constexpr int kPred2LdsBytes = kPredHistBytes + 2 * kPredCellsBytes + 3 * kPredSlots * 4 + 32 * 2 + 2 * kPredMaskWords * 4;
static_assert(kPred2Waves == kPredBlock * kPredBlock, "one wave per block cell");

template <int I, bool INTERIOR>
__device__ __forceinline__ void predict_node2(const uint8_t *own, int lane, uint32_t o01, uint32_t o23, uint32_t o45, bool some, const PredictParams &pp,
                                              uint32_t *s_hist, const uint16_t *s_bkt, uint8_t *bucket_dst, int32_t *pred_dst) {
    constexpr int g = I >= 4 ? 0 : I >= 2 ? 1 : 2;
    const float *wp = pp.width[g], *vp = pp.value[g];
    const int value = *reinterpret_cast<const short *>(own + 2 * (lane + 64 * I));
    const int o[6] = {(int)(short)(o01 & 0xFFFFu), (int)o01 >> 16, (int)(short)(o23 & 0xFFFFu), (int)o23 >> 16, (int)(short)(o45 & 0xFFFFu), (int)o45 >> 16};
    float f[6];
    int v[6];
#pragma unroll
    for (int k = 0; k < 6; k++) {
        v[k] = *reinterpret_cast<const short *>(own + 2 * o[k]);
        f[k] = (float)v[k];
    }
    // get_hf_context_bucket, prediction.rs:165-206 (see predict_node)
    float width = wp[0];
    width = __fadd_rn(width, __fmul_rn(wp[1], fabsf(__fsub_rn(f[0], f[3]))));
    width = __fadd_rn(width, __fmul_rn(wp[2], fabsf(__fsub_rn(f[1], f[2]))));
    width = __fadd_rn(width, __fmul_rn(wp[3], fabsf(__fsub_rn(f[4], f[5]))));
    width = __fadd_rn(width, __fmul_rn(wp[4], fabsf(__fsub_rn(f[1], f[5]))));
    width = __fadd_rn(width, __fmul_rn(wp[5], fabsf(__fsub_rn(f[2], f[4]))));
    // assign_bucket (prediction.rs:55-68) as a 32-entry LDS table of bucket << 10: the kernel is bound by instruction issue and
    // the LDS pipe has room (one ds_read_u16 instead of nine VALU instructions)
    uint32_t b10 = s_bkt[min(f32_as_u32(width), 31u)];
    float pf = __fmul_rn(f[0], vp[0]);
    pf = __fadd_rn(pf, __fmul_rn(f[1], vp[1]));
    pf = __fadd_rn(pf, __fmul_rn(f[2], vp[2]));
    pf = __fadd_rn(pf, __fmul_rn(f[3], vp[3]));
    pf = __fadd_rn(pf, __fmul_rn(f[4], vp[4]));
    pf = __fadd_rn(pf, __fmul_rn(f[5], vp[5]));
    int prediction = f32_as_i32(pf);
    if (I == 0) { // heap index 0 (DC) and 1 (root) live in lanes 0, 1: get_lf_context_bucket, prediction.rs:134-144
        const uint32_t w = (uint32_t)iabs_w(sub_w(v[0], v[2]));
        const int mx = max(v[0], v[2]), mn = min(v[0], v[2]);
        const int lf_pred = v[1] >= mx ? mx : v[1] <= mn ? mn : sub_w(add_w(v[0], v[2]), v[1]);
        const bool lf = lane < 2;
        b10 = lf ? bucket_of_rt(w) << 10 : b10;
        prediction = lf ? lf_pred : prediction;
    }
    const uint32_t sym = pack_signed(sub_w(value, prediction));
    uint32_t bin = sym < 1024u ? b10 + sym : (uint32_t)kHistBins;
    uint32_t bucket = b10 >> 10;
    if (!INTERIOR) { // a None node is not counted and stays (0, 0) in the outputs (wavelet_transform.rs:60-64). Skipped under the
                     // exec mask, not sent to a trash bin: 64 lanes adding to ONE LDS address take ~0.7 us per instruction (measured)
        bucket = some ? bucket : 0u;
        prediction = some ? prediction : 0;
        if (some) atomicAdd(&s_hist[bin], 1u);
    } else {
        atomicAdd(&s_hist[bin], 1u); // bump_freq, entropy_coding.rs:98-100
    }
    __builtin_nontemporal_store((uint8_t)bucket, bucket_dst + 64 * I);
    __builtin_nontemporal_store(prediction, pred_dst + 64 * I);
}

__global__ void __launch_bounds__(kPred2Threads) predict_histogram_kernel2(const PredArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(lds); // 10 x 1024 + out-of-alphabet counter + trash bin
    uint8_t *s_cells = lds + kPredHistBytes;              // [2][kPredCellsBytes]
    int32_t *s_ring = reinterpret_cast<int32_t *>(s_cells + 2 * kPredCellsBytes); // [3][kPredSlots]
    uint16_t *s_bkt = reinterpret_cast<uint16_t *>(s_ring + 3 * kPredSlots);       // [32] bucket_of(w) << 10
    uint32_t *s_masks = reinterpret_cast<uint32_t *>(s_bkt + 32);                  // [2][kPredSlots][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    trace_stamp(a.trace, blockIdx.x, 0, tid);
    for (int i = tid; i < kHistBins + 2; i += kPred2Threads) s_hist[i] = 0;
    if (tid < 32) s_bkt[tid] = (uint16_t)(bucket_of((uint32_t)tid) << 10);

    uint32_t off[8][3]; // neighbour halfword offsets relative to the own slot, loop invariant
#pragma unroll
    for (int i = 0; i < 8; i++) { // precomputed at plan creation (build_pred_offsets): 8 loads, no arithmetic
        const u32x4 o = reinterpret_cast<const u32x4 *>(a.pred_off)[lane + 64 * i];
        off[i][0] = o.x, off[i][1] = o.y, off[i][2] = o.z;
    }

    const PredTileWalk walk(a.n_tiles);
    if (walk.first < walk.end) { // (a workgroup without a tile still takes part in the hand-over below)
    const uint32_t last = walk.first + ((walk.end - 1 - walk.first) / walk.step) * walk.step; // this workgroup's last tile
    const int slot_lane = tid % kPredSlots;
    if (tid < kPredSlots) {
        s_ring[tid] = a.pred_slots[(size_t)walk.first * kPredSlots + tid];
        s_ring[kPredSlots + tid] = a.pred_slots[(size_t)min(walk.first + walk.step, last) * kPredSlots + tid];
    }
    __syncthreads();
    // stage tile 0 straight into image 0
    for (int sl = wave; sl < kPredSlots; sl += kPred2Waves) {
        const int cell = pred_slot_cell(s_ring[sl]);
        i32x4 lo = i32x4{0, 0, 0, 0}, hi = lo;
        if (lane < 16) s_masks[sl * 16 + lane] = a.valid_mask[(size_t)max(cell, 0) * 16 + lane];
        if (cell >= 0) {
            const i32x4 *src = reinterpret_cast<const i32x4 *>(a.coefs + (size_t)cell * kCell + 8 * lane);
            lo = src[0], hi = src[1];
            if (pred_is_block_slot(sl)) {
                const int v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const uint32_t n = pred_count_outliers(v);
                if (n) atomicAdd(&s_hist[kHistBins], n);
            }
        }
        uint8_t *dst = s_cells + sl * kSlotStride;
        *reinterpret_cast<u32x4 *>(dst + 16 * lane) = u32x4{__builtin_amdgcn_perm((uint32_t)lo.y, (uint32_t)lo.x, 0x05040100u), __builtin_amdgcn_perm((uint32_t)lo.w, (uint32_t)lo.z, 0x05040100u),
                                                            __builtin_amdgcn_perm((uint32_t)hi.y, (uint32_t)hi.x, 0x05040100u), __builtin_amdgcn_perm((uint32_t)hi.w, (uint32_t)hi.z, 0x05040100u)};
        if (lane == 0) *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    trace_stamp(a.trace, blockIdx.x, 1, tid);

    int it = 0;
    for (uint32_t tile = walk.first; tile < walk.end; tile += walk.step, it++) {
        const bool more = tile + walk.step < walk.end;
        const int32_t *cur_slots = s_ring + (it % 3) * kPredSlots, *nxt_slots = s_ring + ((it + 1) % 3) * kPredSlots;
        const uint8_t *cur = s_cells + (it & 1) * kPredCellsBytes;
        uint8_t *nxt = s_cells + ((it & 1) ^ 1) * kPredCellsBytes;
        const uint32_t *cur_masks = s_masks + (it & 1) * kPredMaskWords;
        uint32_t *nxt_masks = s_masks + ((it & 1) ^ 1) * kPredMaskWords;
        // in flight across the arithmetic below: the slot list of tile i + 2 and the cells of tile i + 1
        const int32_t slot_pre = a.pred_slots[(size_t)min(tile + 2 * walk.step, last) * kPredSlots + slot_lane];
        i32x4 st_lo[kPred2Stage], st_hi[kPred2Stage];
        int st_cell[kPred2Stage];
        uint32_t st_mask[kPred2Stage];
        if (more) {
#pragma unroll
            for (int j = 0; j < kPred2Stage; j++) {
                const int sl = wave + kPred2Waves * j;
                if (sl < kPredSlots) {
                    st_cell[j] = pred_slot_cell(__builtin_amdgcn_readfirstlane(nxt_slots[sl]));
                    const i32x4 *src = reinterpret_cast<const i32x4 *>(a.coefs + (size_t)max(st_cell[j], 0) * kCell + 8 * lane);
                    st_lo[j] = src[0], st_hi[j] = src[1];
                    st_mask[j] = a.valid_mask[(size_t)max(st_cell[j], 0) * 16 + (lane & 15)]; // the cell's Some/None bits travel with it
                }
            }
        }

        { // One block cell per wave. Every path issues exactly 16 stores (a wave without a retained cell at its block slot
          // writes zeros to the plan's junk lines), so the compiler can count them: the commit below waits for the staging
          // loads with vmcnt(16) instead of vmcnt(0) and does not sit out the acknowledgement of the stores just issued.
            const int slot = (1 + wave / kPredBlock) * kPredSide + 1 + (wave % kPredBlock);
            const int raw = __builtin_amdgcn_readfirstlane(cur_slots[slot]); // everything this phase needs is in LDS: a global load here
            const int cell = pred_slot_cell(raw);                            // would have to wait for the staging loads just issued
            const bool has = cell >= 0;
            const uint8_t *own = cur + slot * kSlotStride;
            // (junk lines are private to the wave: one shared line would be a write hot spot for every edge tile of the image)
            const size_t junk = ((size_t)blockIdx.x * kPred2Waves + wave) * kPredJunkBytes;
            uint8_t *bd = (has ? a.bucket + (size_t)cell * kCell : a.junk + junk) + lane;
            int32_t *pd = (has ? a.prediction + (size_t)cell * kCell : reinterpret_cast<int32_t *>(a.junk + junk + 512)) + lane;
            if (pred_slot_interior(raw)) {
                predict_node2<0, true>(own, lane, off[0][0], off[0][1], off[0][2], true, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<1, true>(own, lane, off[1][0], off[1][1], off[1][2], true, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<2, true>(own, lane, off[2][0], off[2][1], off[2][2], true, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<3, true>(own, lane, off[3][0], off[3][1], off[3][2], true, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<4, true>(own, lane, off[4][0], off[4][1], off[4][2], true, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<5, true>(own, lane, off[5][0], off[5][1], off[5][2], true, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<6, true>(own, lane, off[6][0], off[6][1], off[6][2], true, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<7, true>(own, lane, off[7][0], off[7][1], off[7][2], true, a.pp, s_hist, s_bkt, bd, pd);
            } else if (!has) { // no retained cell at this block slot (image edge): only the fixed number of stores, to the wave's junk lines
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    __builtin_nontemporal_store((uint8_t)0, bd + 64 * i);
                    __builtin_nontemporal_store(0, pd + 64 * i);
                }
            } else { // boundary cell: Some/None of node lane + 64 i is bit (lane & 31) of mask word 2 i + (lane >> 5)
                uint32_t some_bits = 0;
                {
#pragma unroll
                    for (int i = 0; i < 8; i++) some_bits |= ((cur_masks[slot * 16 + 2 * i + (lane >> 5)] >> (lane & 31)) & 1u) << i;
                }
                predict_node2<0, false>(own, lane, off[0][0], off[0][1], off[0][2], some_bits & 1u, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<1, false>(own, lane, off[1][0], off[1][1], off[1][2], some_bits & 2u, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<2, false>(own, lane, off[2][0], off[2][1], off[2][2], some_bits & 4u, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<3, false>(own, lane, off[3][0], off[3][1], off[3][2], some_bits & 8u, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<4, false>(own, lane, off[4][0], off[4][1], off[4][2], some_bits & 16u, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<5, false>(own, lane, off[5][0], off[5][1], off[5][2], some_bits & 32u, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<6, false>(own, lane, off[6][0], off[6][1], off[6][2], some_bits & 64u, a.pp, s_hist, s_bkt, bd, pd);
                predict_node2<7, false>(own, lane, off[7][0], off[7][1], off[7][2], some_bits & 128u, a.pp, s_hist, s_bkt, bd, pd);
            }
        }

        if (more) {
#pragma unroll
            for (int j = 0; j < kPred2Stage; j++) {
                const int sl = wave + kPred2Waves * j;
                if (sl < kPredSlots) {
                    i32x4 lo = st_lo[j], hi = st_hi[j];
                    if (st_cell[j] < 0) {
                        lo = hi = i32x4{0, 0, 0, 0}; // no retained cell at this slot: the reference reads 0 there
                    } else if (pred_is_block_slot(sl)) {
                        const int v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        const uint32_t n = pred_count_outliers(v);
                        if (n) atomicAdd(&s_hist[kHistBins], n);
                    }
                    uint8_t *dst = nxt + sl * kSlotStride;
                    if (lane < 16) nxt_masks[sl * 16 + lane] = st_mask[j];
                    *reinterpret_cast<u32x4 *>(dst + 16 * lane) = u32x4{__builtin_amdgcn_perm((uint32_t)lo.y, (uint32_t)lo.x, 0x05040100u), __builtin_amdgcn_perm((uint32_t)lo.w, (uint32_t)lo.z, 0x05040100u),
                                                                        __builtin_amdgcn_perm((uint32_t)hi.y, (uint32_t)hi.x, 0x05040100u), __builtin_amdgcn_perm((uint32_t)hi.w, (uint32_t)hi.z, 0x05040100u)};
                    if (lane == 0) *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        if (tid < kPredSlots) s_ring[((it + 2) % 3) * kPredSlots + tid] = slot_pre;
        lds_barrier();
        trace_stamp(a.trace, blockIdx.x, 2 + it, tid);
    }
    }
    __syncthreads();
    trace_stamp(a.trace, blockIdx.x, 13, tid);
    pred_hand_over(a, s_hist, reinterpret_cast<uint32_t *>(s_ring), tid, kPred2Threads);
    trace_exit(a.trace, blockIdx.x, tid);
}

// ------------------------------------------------------------------------------------------------
// Fit accumulators: the sums behind ContextModeler::optimize_parameters
// (context_modeling.rs:79-213), so that the host solves two 6 x 6 systems per layer group instead of running an SVD over
// n x 6 f32 matrices (n = 8.5 M rows at 4096^2). Same tiles, staging and LDS gather as K2.
//   MODE 0 (value fit, :175-202): per layer group g, the Gram matrix of u = [v0..v5, value] over the Some nodes of levels
//           1..8: gram[g][28] (upper triangle, row major) -- A^T A, A^T b and b^T b in exact integers.
//   MODE 1 (width fit, :144-173): with the value parameters x: r = |f32(value) - f32 prediction| (the same left-to-right f32
//           evaluation as K2 / nalgebra's gemv), w = [1, |v0-v3|, |v1-v2|, |v4-v5|, |v1-v5|, |v2-v4|]:
//           wtw[g][21] = sum w w^T (exact integers), wtr[g][6] = sum w r (f64).
// Lanes are bound to layer groups so that a lane needs one set of accumulators: lanes 0..31 take the level-8 nodes
// (group 0), 32..47 level 7 (group 1), 48..63 levels 0..6 (group 2; heap index 0, 1 are not part of the fit).
// ------------------------------------------------------------------------------------------------
struct FitArgs {
    const int32_t *coefs;
    const int32_t *pred_slots;
    const uint16_t *nbr_table;
    const uint8_t *interior;
    const uint32_t *valid_mask;
    uint32_t n_tiles;
    PredictParams pp;
    const uint32_t *pred_off;  // [512][4] packed neighbour offsets per node (build_pred_offsets)
    unsigned long long *gram; // [3][28]   (MODE 0)
    unsigned long long *wtw;  // [3][21]   (MODE 1)
    double *wtr;              // [3][6]    (MODE 1)
    unsigned long long *acc;  // plan scratch, all zero between launches: [kFitAccInt] integer sums, [18] f64 bit patterns, then the ticket
};
constexpr int kFitAccInt = 3 * 28, kFitAccDbl = kFitAccInt, kFitAccTicket = kFitAccInt + 18;
static_assert(kFitAccTicket + 1 == (int)kFitAccWords, "fit accumulator layout");

template <int MODE>
__global__ void __launch_bounds__(kPredThreads, 4) fit_accumulate_kernel(const FitArgs a) {
    constexpr int NI = MODE == 0 ? 28 : 21;
    __shared__ __attribute__((aligned(16))) uint8_t s_cells[kPredSlots * kSlotStride];
    __shared__ int32_t s_slot_cell[kPredSlots];
    __shared__ int32_t s_slot_interior[kPredSlots];
    __shared__ uint32_t s_flag;
    __shared__ unsigned long long s_int[3][28];
    __shared__ double s_dbl[3][6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 3 * 28) (&s_int[0][0])[tid] = 0;
    if (tid < 18) (&s_dbl[0][0])[tid] = 0.0;

    const int g = lane < 32 ? 0 : lane < 48 ? 1 : 2;
    const int p0 = lane < 32 ? 256 + lane : lane < 48 ? 128 + (lane - 32) : lane - 48;
    const int pstep = lane < 32 ? 32 : 16;
    // The neighbour offsets of a lane's 8 nodes live in LDS here (not in registers as in K2): together with 21-28 accumulators
    // they would not fit 128 VGPRs, and this kernel is not on the critical path. The map is per lane, identical in all waves.
    __shared__ uint32_t s_off[8][64][3];
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const u32x4 o = reinterpret_cast<const u32x4 *>(a.pred_off)[p0 + pstep * i];
            s_off[i][lane][0] = o.x;
            s_off[i][lane][1] = o.y;
            s_off[i][lane][2] = o.z;
        }
    }
    float vp[6];
#pragma unroll
    for (int k = 0; k < 6; k++) vp[k] = a.pp.value[g][k];

    int acc[NI];
    double dacc[6];
#pragma unroll
    for (int k = 0; k < NI; k++) acc[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) dacc[k] = 0.0;
    int cells_since_flush = 0;
    auto flush = [&]() { // per-lane int32 sums -> workgroup int64 sums (sign-extended two's complement adds)
#pragma unroll
        for (int k = 0; k < NI; k++) {
            atomicAdd(&s_int[g][k], (unsigned long long)(long long)acc[k]);
            acc[k] = 0;
        }
        cells_since_flush = 0;
    };

    const PredTileWalk walk(a.n_tiles);
    for (uint32_t tile = walk.first; tile < walk.end; tile += walk.step) {
        __syncthreads();
        if (tid < kPredSlots) {
            const int raw = a.pred_slots[(size_t)tile * kPredSlots + tid];
            s_slot_cell[tid] = pred_slot_cell(raw);
            s_slot_interior[tid] = pred_slot_interior(raw) ? 1 : 0;
        }
        __syncthreads();
        pred_stage_tile(a.coefs, s_slot_cell, s_cells, lane, wave);
        __syncthreads();
        for (int r = wave; r < kPredBlock * kPredBlock; r += kPredWaves) {
            const int slot = (1 + r / kPredBlock) * kPredSide + 1 + (r % kPredBlock);
            const int cell = s_slot_cell[slot];
            if (cell < 0) continue;
            const uint8_t *own = s_cells + slot * kSlotStride;
            const bool boundary = __builtin_amdgcn_readfirstlane(s_slot_interior[slot]) == 0;
#pragma unroll 1
            for (int i = 0; i < 8; i++) {
                const int p = p0 + pstep * i;
                bool use = p >= 2; // heap index 0 and 1 are coded by the LF predictor and are not rows of the fit
                if (boundary) use = use && ((a.valid_mask[(size_t)cell * 16 + (p >> 5)] >> (p & 31)) & 1u);
                int u[7];
                int v[6];
                const uint32_t off_i[3] = {s_off[i][lane][0], s_off[i][lane][1], s_off[i][lane][2]};
                pred_gather(own, off_i, v);
                const int value = *reinterpret_cast<const short *>(own + 2 * p);
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 6; k++) u[k] = use ? v[k] : 0; // a None row is all zeros in the reference (:109-134)
                    u[6] = use ? value : 0;
                    int n = 0;
#pragma unroll
                    for (int r0 = 0; r0 < 7; r0++)
#pragma unroll
                        for (int c0 = r0; c0 < 7; c0++) acc[n++] += __mul24(u[r0], u[c0]);
                } else {
                    float pf = __fmul_rn((float)v[0], vp[0]);
#pragma unroll
                    for (int k = 1; k < 6; k++) pf = __fadd_rn(pf, __fmul_rn((float)v[k], vp[k]));
                    const float res = fabsf(__fsub_rn((float)value, pf));
                    int w[6] = {1, iabs_w(v[0] - v[3]), iabs_w(v[1] - v[2]), iabs_w(v[4] - v[5]), iabs_w(v[1] - v[5]), iabs_w(v[2] - v[4])};
#pragma unroll
                    for (int k = 0; k < 6; k++) w[k] = use ? w[k] : 0;
                    int n = 0;
#pragma unroll
                    for (int r0 = 0; r0 < 6; r0++)
#pragma unroll
                        for (int c0 = r0; c0 < 6; c0++) acc[n++] += __mul24(w[r0], w[c0]);
                    const double rd = (double)res;
#pragma unroll
                    for (int k = 0; k < 6; k++) dacc[k] += (double)w[k] * rd;
                }
            }
            if (++cells_since_flush >= 1024) flush(); // 8 nodes x 255^2 x 1024 cells < 2^31
        }
    }
    // Final reduction through LDS scratch instead of atomics: all 32 (16) lanes of a layer group would add to ONE LDS address,
    // which costs ~0.7 us per instruction (see K2), 28 + 6 times per wave. The cell image is free now: every lane parks its sums
    // at its own address, then one thread per (wave, sum, group) adds a group's lanes up - in a fixed order, so the f64 sums of
    // MODE 1 no longer depend on the arrival order of atomics.
    __syncthreads();
    {
        constexpr int NS = NI + (MODE == 1 ? 12 : 0); // int sums + 6 doubles as 12 words
        int32_t *scr = reinterpret_cast<int32_t *>(s_cells); // [4 waves][NS][64] words at a time: the 36-slot image holds 37 440 B
        static_assert(4 * NS * 64 * 4 <= kPredSlots * kSlotStride, "scratch fits the cell image");
        for (int half = 0; half < 2; half++) {
            const bool mine = (wave >> 2) == half;
            const int w4 = wave & 3;
            if (mine) {
#pragma unroll
                for (int k = 0; k < NI; k++) scr[(w4 * NS + k) * 64 + lane] = acc[k];
                if (MODE == 1) {
#pragma unroll
                    for (int k = 0; k < 6; k++) {
                        const unsigned long long u = __builtin_bit_cast(unsigned long long, dacc[k]);
                        scr[(w4 * NS + NI + 2 * k) * 64 + lane] = (int32_t)(uint32_t)u;
                        scr[(w4 * NS + NI + 2 * k + 1) * 64 + lane] = (int32_t)(uint32_t)(u >> 32);
                    }
                }
            }
            __syncthreads();
            for (int t = tid; t < 4 * NI * 3; t += kPredThreads) { // integer sums: (wave of this half, k, group)
                const int ww = t / (NI * 3), k = (t / 3) % NI, gg = t % 3;
                const int l0 = gg == 0 ? 0 : gg == 1 ? 32 : 48, l1 = gg == 0 ? 32 : gg == 1 ? 48 : 64;
                long long sum = 0;
                for (int l = l0; l < l1; l++) sum += scr[(ww * NS + k) * 64 + l];
                atomicAdd(&s_int[gg][k], (unsigned long long)sum); // <= 8 adds per address in the whole kernel
            }
            if (MODE == 1) {
                for (int t = tid; t < 4 * 6 * 3; t += kPredThreads) {
                    const int ww = t / 18, k = (t / 3) % 6, gg = t % 3;
                    const int l0 = gg == 0 ? 0 : gg == 1 ? 32 : 48, l1 = gg == 0 ? 32 : gg == 1 ? 48 : 64;
                    double sum = 0.0;
                    for (int l = l0; l < l1; l++) {
                        const unsigned long long u = (unsigned long long)(uint32_t)scr[(ww * NS + NI + 2 * k) * 64 + l] |
                                                     (unsigned long long)(uint32_t)scr[(ww * NS + NI + 2 * k + 1) * 64 + l] << 32;
                        sum += __builtin_bit_cast(double, u);
                    }
                    atomicAdd(&s_dbl[gg][k], sum);
                }
            }
            __syncthreads();
        }
    }
    // Hand-over like K2's: add into the plan accumulator, draw a ticket, the last workgroup moves the totals out and re-zeroes.
    if (tid < 3 * NI) {
        const int gg = tid / NI, k = tid % NI;
        __hip_atomic_fetch_add(a.acc + tid, s_int[gg][k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 1 && tid < 18) __hip_atomic_fetch_add(reinterpret_cast<double *>(a.acc + kFitAccDbl) + tid, (&s_dbl[0][0])[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads(); // vmcnt(0) in every wave: the adds are performed
    if (tid == 0) s_flag = __hip_atomic_fetch_add(a.acc + kFitAccTicket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_flag == 0) return;
    if (tid < 3 * NI) {
        (MODE == 0 ? a.gram : a.wtw)[tid] = __hip_atomic_load(a.acc + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.acc + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 1 && tid < 18) {
        const unsigned long long u = __hip_atomic_load(a.acc + kFitAccDbl + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.wtr[tid] = __builtin_bit_cast(double, u);
        __hip_atomic_store(a.acc + kFitAccDbl + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) __hip_atomic_store(a.acc + kFitAccTicket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// K3: inverse transform. A workgroup walks the forward kernel's tiles of its share (<= cells_per_tile cells of one
// band each), one wave per (cell, channel) item; mirror of fwd_wave. A cell's 512 pixels are scattered over ~50
// rows, so byte stores straight from registers cost one L2 write request per pixel (measured: 41 us). Instead the
// waves scatter into an LDS image of the tile's pixel rectangle -- 16 bit per byte: value | 0x100, the ninth bit
// says "a cell of this tile owns the byte" -- and the workgroup then writes the rectangle out row by row: one
// aligned dword store where all four bytes are owned (90 % of the bytes: the interior of the tile's footprint);
// the dwords on its fractal rim are queued and written with byte stores from densely packed lanes afterwards
// (a store instruction costs the same with 2 or 64 active lanes). Every pixel has exactly one owning cell, so no
// byte is written twice and none is skipped. The coefficients of tile i + 1 are loaded while tile i is processed.
// ------------------------------------------------------------------------------------------------
constexpr int kInvThreads = 256;
constexpr int kInvWaves = kInvThreads / 64;
constexpr int kInvMaxItemsPerWave = 4; // (cell, channel) items of one tile per transform wave

struct InvArgs {
    const int32_t *coefs;
    uint8_t *pixels;
    const Tile *tiles;
    const TileCell *tile_meta; // in tile order
    const int32_t *wg_tiles;   // [n_wg + 1]
    int32_t width, height, channels;
    uint32_t F, n_wg;
    int32_t buf_bytes;   // LDS pixel rectangle (16 bit per byte), multiple of 16
    int32_t queue_bytes; // LDS rim queue per wave
    int32_t max_wg_tiles;
    int32_t q_identity;
    int32_t ablate; // timing experiments only (FRI_HIP_K3_ABLATE): 1 = no global stores, 2 = no LDS scatter
    unsigned long long *trace; // diagnostic timeline, null in production
    QMatrix q;
};

__device__ __forceinline__ int dequant_ref(int v, int heap_index, const InvArgs &a) {
    // quantization::decode divides like encode (quantization.rs:37); reproduced bit for bit.
    if (a.q_identity || v == kNone) return v;
    return v / a.q.q[quant_layer(heap_index)];
}

// The eight coefficient dwords a lane holds of one (cell, channel) item: heap nodes lane (levels 0-5 and the DC), 64 + lane,
// 128 + 2 lane + {0, 1}, 256 + 4 lane + {0..3}.
struct InvRegs {
    int32_t d8[4], d7[2], d6, low;
};

__device__ __forceinline__ InvRegs inv_load(const int32_t *in, int lane) {
    // read once: streaming loads keep the eight L2s for the pixel lines that neighbouring tiles complete
    const i32x4 c8 = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(in + 256 + 4 * lane));
    const i32x2 c7 = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(in + 128 + 2 * lane));
    InvRegs r;
    r.d8[0] = c8.x, r.d8[1] = c8.y, r.d8[2] = c8.z, r.d8[3] = c8.w;
    r.d7[0] = c7.x, r.d7[1] = c7.y;
    r.d6 = __builtin_nontemporal_load(in + 64 + lane);
    r.low = __builtin_nontemporal_load(in + lane);
    return r;
}

// extract_values for one item (wavelet_transform.rs:358-380) on the lane-distributed tree: six cross-lane levels top-down,
// then levels 6-8 in registers. leaf[j] = value of leaf 8 lane + j (0 where the last difference is None: `if let Some(dif)`
// at :365 leaves those pixels at the raster's initial 0).
// SOME = the wave has checked that none of the item's 512 coefficients is None (every interior cell of an encoder's
// output): the butterflies then need no None test. Wrapping arithmetic like a release build of the reference.
template <bool SOME>
__device__ __forceinline__ void unpair_t(int low, int d, int &left, int &right) {
    if (!SOME && d == kNone) {
        left = 0;
        right = 0;
    } else {
        right = sub_w(low, d / 2);
        left = add_w(d, right);
    }
}
template <bool SOME>
__device__ __forceinline__ void inv_wave(InvRegs c, int lane, const InvArgs &a, int (&leaf)[8]) {
    if (!a.q_identity) {
#pragma unroll
        for (int i = 0; i < 4; i++) c.d8[i] = dequant_ref(c.d8[i], 256 + 4 * lane + i, a);
#pragma unroll
        for (int i = 0; i < 2; i++) c.d7[i] = dequant_ref(c.d7[i], 128 + 2 * lane + i, a);
        c.d6 = dequant_ref(c.d6, 64 + lane, a);
        c.low = dequant_ref(c.low, lane, a);
    }
    int s = __shfl(c.low, 0); // low_pass_values[1] = coefficients[0].unwrap()  (:361)
#pragma unroll
    for (int j = 5; j >= 0; j--) { // levels 0..5
        const int lv = 5 - j;
        const int d = __shfl(c.low, (1 << lv) + (lane >> (j + 1)));
        int l, r;
        unpair_t<SOME>(s, d, l, r);
        s = ((lane >> j) & 1) ? r : l;
    }
    int s7[2], s8[4];
    unpair_t<SOME>(s, c.d6, s7[0], s7[1]);
#pragma unroll
    for (int i = 0; i < 2; i++) unpair_t<SOME>(s7[i], c.d7[i], s8[2 * i], s8[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 4; i++) unpair_t<SOME>(s8[i], c.d8[i], leaf[2 * i], leaf[2 * i + 1]);
}
__device__ __forceinline__ bool inv_has_none(const InvRegs &c) {
    const bool n = c.d8[0] == kNone || c.d8[1] == kNone || c.d8[2] == kNone || c.d8[3] == kNone || c.d7[0] == kNone || c.d7[1] == kNone ||
                   c.d6 == kNone || c.low == kNone;
    return __any(n);
}

// The loads of one tile's items for this wave. Slots past the tile's items re-load its last item (no conditional loads:
// a load the compiler cannot prove executed costs an immediate wait).
template <int NI>
__device__ __forceinline__ void inv_prefetch(const InvArgs &a, const Tile &t, const TileCell *cells, int wave, int lane, InvRegs (&r)[NI]) {
    const int C = a.channels, n_items = t.cell_count * C;
#pragma unroll
    for (int s = 0; s < NI; s++) {
        const int item = min(wave + kInvWaves * s, n_items - 1);
        const int cl = item / C, ch = item - cl * C;
        r[s] = inv_load(a.coefs + ((size_t)ch * a.F + (uint32_t)cells[cl].cell) * kCell, lane);
    }
}

template <int NI>
__global__ void __launch_bounds__(kInvThreads) inverse_transform_kernel(const InvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint16_t *img16 = reinterpret_cast<uint16_t *>(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t *queue = reinterpret_cast<uint16_t *>(lds + a.buf_bytes + wave * a.queue_bytes);
    Tile *lds_tiles = reinterpret_cast<Tile *>(lds + a.buf_bytes + kInvWaves * a.queue_bytes);
    TileCell *lds_cells = reinterpret_cast<TileCell *>(lds_tiles + a.max_wg_tiles);
    // Blocks of one XCD take a contiguous range of shares: the rim bytes of neighbouring tiles complete their lines in ONE L2.
    const uint32_t wg = xcd_contiguous_share(blockIdx.x, a.n_wg);
    const int tb = a.wg_tiles[wg], te = a.wg_tiles[wg + 1];
    const int C = a.channels;
    const uint32_t base_lo = (uint32_t)reinterpret_cast<uintptr_t>(a.pixels);
    const uint32_t wc = (uint32_t)a.width * (uint32_t)C;
    trace_stamp(a.trace, wg, 0, tid);
    {
        const Tile first = a.tiles[tb], last = a.tiles[te - 1];
        const int n_cells = last.cell_begin + last.cell_count - first.cell_begin;
        if (tid < te - tb) lds_tiles[tid] = a.tiles[tb + tid];
        for (int i = tid; i < n_cells; i += kInvThreads) {
            TileCell tc = a.tile_meta[first.cell_begin + i];
            lds_cells[i] = tc;
        }
        u32x4 *z = reinterpret_cast<u32x4 *>(lds);
        for (int i = tid; i < a.buf_bytes / 16; i += kInvThreads) z[i] = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    const int cell0 = lds_tiles[0].cell_begin;
    trace_stamp(a.trace, wg, 1, tid);

    InvRegs pre[NI];
    inv_prefetch<NI>(a, lds_tiles[0], lds_cells, wave, lane, pre);
    for (int ti = tb; ti < te; ti++) {
        const Tile t = lds_tiles[ti - tb];
        InvRegs cur[NI];
#pragma unroll
        for (int s = 0; s < NI; s++) cur[s] = pre[s];
        {
            const Tile tn = lds_tiles[min(ti + 1, te - 1) - tb];
            inv_prefetch<NI>(a, tn, lds_cells + (tn.cell_begin - cell0), wave, lane, pre);
        }
        const int n_items = t.cell_count * C;
        // Staged rows start at the 16-byte boundary at or below their first byte (lead-in 0..15), so every global quad is an
        // aligned 16-byte store; rq quads (16 output bytes = 16 LDS halfwords each) per row.
        const int rq = (t.width_px * C + 30) >> 4;
        const int pitch16 = rq * 16; // LDS halfwords per staged row
        int leaf[NI][8];
#pragma unroll
        for (int s = 0; s < NI; s++) {
            if (inv_has_none(cur[s])) inv_wave<false>(cur[s], lane, a, leaf[s]);
            else inv_wave<true>(cur[s], lane, a, leaf[s]);
        }
#pragma unroll
        for (int s = 0; s < NI; s++) {
            const int item = wave + kInvWaves * s;
            if (item < n_items && !(a.ablate & 2)) {
                const int cl = item / C, ch = item - cl * C;
                const TileCell tc = lds_cells[t.cell_begin - cell0 + cl];
                const int x0 = tc.cx + lane_dx(lane), y0 = tc.cy + lane_dy(lane);
                int rowbase[3];
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const int y = y0 + dy;
                    const uint32_t g = (uint32_t)y * wc + (uint32_t)(t.x_lo * C); // byte offset of the staged row (only bits 0-3 matter)
                    rowbase[dy] = (y - t.y_lo) * pitch16 + (int)((base_lo + g) & 15u) + (x0 - t.x_lo) * C + ch;
                }
                if (__builtin_amdgcn_readfirstlane(tc.interior)) {
#pragma unroll
                    for (int j = 0; j < 8; j++) img16[rowbase[leaf_dy(j)] + leaf_dx(j) * C] = (uint16_t)(0x100 | min(max(leaf[s][j], 0), 255));
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int x = x0 + leaf_dx(j), y = y0 + leaf_dy(j);
                        if (x >= 0 && y >= 0 && x < a.width && y < a.height) // set_pixel, images.rs:104
                            img16[rowbase[leaf_dy(j)] + leaf_dx(j) * C] = (uint16_t)(0x100 | min(max(leaf[s][j], 0), 255));
                    }
                }
            }
        }
        lds_barrier(); // the rectangle is complete

        // Quad pass over the flattened rectangle: a quad whose 16 bytes are all owned goes out as one store, a partly owned
        // one is queued as (row << 8 | quad).
        int qn = 0;
        const int n_quads = t.n_rows * rq;
        const float inv_rq = 1.0f / (float)rq;
        for (int q0 = 0; q0 < n_quads; q0 += kInvThreads) {
            const int qi = q0 + tid;
            u32x4 lo = u32x4{0u, 0u, 0u, 0u}, hi = lo;
            u32x4 *src = reinterpret_cast<u32x4 *>(img16) + 2 * qi;
            int r = 0, k = 0;
            if (qi < n_quads) {
                lo = src[0], hi = src[1];
                r = (int)(((float)qi + 0.5f) * inv_rq); // exact: qi < 2^16
                k = qi - r * rq;
            }
            const uint32_t all = lo.x & lo.y & lo.z & lo.w & hi.x & hi.y & hi.z & hi.w & 0x01000100u;
            const uint32_t any = (lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) & 0x01000100u;
            const bool full = all == 0x01000100u, rim = any != 0 && !full;
            if (full) {
                src[0] = u32x4{0u, 0u, 0u, 0u};
                src[1] = u32x4{0u, 0u, 0u, 0u};
                const size_t g = ((size_t)(t.y_lo + r) * (size_t)a.width + (size_t)t.x_lo) * (size_t)C;
                uint8_t *p = a.pixels + g - (int)((base_lo + (uint32_t)g) & 15u) + 16 * k; // 16-byte aligned
                const u32x4 out{__builtin_amdgcn_perm(lo.y, lo.x, 0x06040200u), __builtin_amdgcn_perm(lo.w, lo.z, 0x06040200u),
                                __builtin_amdgcn_perm(hi.y, hi.x, 0x06040200u), __builtin_amdgcn_perm(hi.w, hi.z, 0x06040200u)};
                if (!(a.ablate & 1)) *reinterpret_cast<u32x4 *>(p) = out;
            }
            const unsigned long long m = __ballot(rim);
            if (rim) queue[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(r << 8 | k);
            qn += __popcll(m);
        }
        // Rim pass: one queued quad per lane; owned dwords as dword stores, the rest byte by byte.
        for (int e0 = 0; e0 < qn; e0 += 64) {
            const int e = e0 + lane;
            if (e < qn) {
                const int rk = queue[e], r = rk >> 8, k = rk & 255;
                const size_t g = ((size_t)(t.y_lo + r) * (size_t)a.width + (size_t)t.x_lo) * (size_t)C;
                uint8_t *p = a.pixels + g - (int)((base_lo + (uint32_t)g) & 15u) + 16 * k;
                u32x4 *src = reinterpret_cast<u32x4 *>(img16 + r * pitch16) + 2 * k;
                const u32x4 lo = src[0], hi = src[1];
                src[0] = u32x4{0u, 0u, 0u, 0u};
                src[1] = u32x4{0u, 0u, 0u, 0u};
                const uint32_t u[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (!(a.ablate & 1)) {
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const uint32_t v0 = u[2 * d], v1 = u[2 * d + 1];
                        const uint32_t own = (v0 & 0x01000100u) | ((v1 & 0x01000100u) << 1);
                        if (own == 0x03000300u) {
                            *reinterpret_cast<uint32_t *>(p + 4 * d) = __builtin_amdgcn_perm(v1, v0, 0x06040200u);
                        } else if (own != 0) {
                            if (v0 & 0x00000100u) p[4 * d + 0] = (uint8_t)v0;
                            if (v0 & 0x01000000u) p[4 * d + 1] = (uint8_t)(v0 >> 16);
                            if (v1 & 0x00000100u) p[4 * d + 2] = (uint8_t)v1;
                            if (v1 & 0x01000000u) p[4 * d + 3] = (uint8_t)(v1 >> 16);
                        }
                    }
                }
            }
        }
        lds_barrier(); // the rectangle is all zero again
        trace_stamp(a.trace, wg, 2 + ti - tb, tid);
    }
    trace_exit(a.trace, wg, tid);
}

} // namespace

static size_t fwd_meta_offset(const DevicePlan &p) { return ((size_t)p.lds_pitch * p.lds_rows + 15) & ~(size_t)15; }
static size_t fwd_buf_bytes(const DevicePlan &p) { return fwd_meta_offset(p) + (size_t)p.max_tile_cells * sizeof(TileCell); }
// 16 bit per staged byte; a staged row holds <= lds_pitch bytes including its lead-in (lds_pitch >= widest row + 15).
static size_t inv_buf_bytes(const DevicePlan &p) { return (size_t)p.lds_rows * (size_t)p.lds_pitch * 2; }
// worst case: every quad a wave looks at is a rim quad (2 bytes per entry), rounded to 16
static size_t inv_queue_bytes(const DevicePlan &p) {
    const size_t quads = (size_t)p.lds_rows * (size_t)(p.lds_pitch / 16);
    return (((quads + kInvThreads - 1) / kInvThreads) * 64 * 2 + 15) & ~(size_t)15;
}
size_t inv_lds_bytes(const DevicePlan &p) { return inv_buf_bytes(p) + kInvWaves * inv_queue_bytes(p) + (size_t)p.max_wg_tiles * sizeof(Tile) + (size_t)p.max_wg_cells * sizeof(TileCell); }
size_t fwd_lds_bytes(const DevicePlan &p) { return 2 * fwd_buf_bytes(p) + 16 * kFwdThreads + (size_t)p.max_wg_tiles * sizeof(Tile); }
static size_t fwd_chunks(const DevicePlan &p) { return (size_t)p.lds_rows * (p.lds_pitch / 16); }

bool device_footprint_matches(const StaticTables &st) {
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++) {
            const Int2 o = st.leaf_off[8 * l + j];
            if (o.x != lane_dx(l) + leaf_dx(j) || o.y != lane_dy(l) + leaf_dy(j)) return false;
        }
    return true;
}

bool fwd_plan_fits(const DevicePlan &p) {
    return fwd_chunks(p) <= (size_t)kMaxChunksPerThread * kFwdThreads &&
           (size_t)p.max_tile_cells * p.channels <= (size_t)kMaxItemsPerTile && p.max_tile_cells <= kFwdThreads &&
           p.max_wg_tiles <= kFwdThreads && fwd_lds_bytes(p) <= 160 * 1024;
}

hipError_t launch_fwd_transform_quant(const DevicePlan &p, uint32_t n_images, const uint8_t *pixels, size_t pixel_stride, int32_t *coefs,
                                      size_t coef_stride, const QMatrix &q, hipStream_t stream) {
    if (!fwd_plan_fits(p)) return hipErrorInvalidConfiguration;
    FwdArgs a{};
    a.pixels = pixels;
    a.pixel_stride = pixel_stride;
    a.coefs = coefs;
    a.coef_stride = coef_stride;
    a.tiles = p.tiles;
    a.tile_meta = p.tile_meta;
    // many images per launch: merged shares (the machine is full anyway; fewer, longer workgroups amortise their start-up)
    const bool batch = n_images >= 8 && p.n_wg_batch > 0 && p.n_wg_batch < p.n_wg && p.k1_batch_shares;
    a.wg_tiles = batch ? p.wg_tiles_batch : p.wg_tiles;
    a.n_wg = batch ? p.n_wg_batch : p.n_wg;
    a.width = p.width;
    a.height = p.height;
    a.F = p.F;
    a.pitch = p.lds_pitch;
    a.meta_off = (int32_t)fwd_meta_offset(p);
    a.buf_bytes = (int32_t)fwd_buf_bytes(p);
    a.cpr = (uint32_t)p.lds_pitch / 16u;
    a.cpr_magic = (uint32_t)(((1ull << 32) + a.cpr - 1) / a.cpr);
    a.q = q;
    a.q_identity = 1;
    for (int i = 0; i <= 9; i++) a.q_identity &= (q.q[i] == 1); // layers 0..9 are the only ones a 512-node cell uses
    a.ablate = p.k1_ablate;
    a.trace = p.trace;
    const size_t lds = fwd_lds_bytes(p);
    const dim3 grid(a.n_wg, n_images), block(kFwdThreads);
    // EDGE variant only when a 16-byte chunk could straddle the ends of one of the caller's image buffers
    const size_t img_bytes = (size_t)p.width * p.height * p.channels;
    const bool edge = (reinterpret_cast<uintptr_t>(pixels) & 15) || (img_bytes & 15) || (n_images > 1 && (pixel_stride & 15));
    const bool fast = !edge && (((size_t)p.width * p.channels) & 15) == 0; // every image row starts at the same offset mod 16
    const bool small = fwd_chunks(p) <= 4 * (size_t)kFwdThreads;           // 4 chunks per thread suffice (the common, tuned case)
    void (*kern)(FwdArgs);
#define FRI_PICK_N(CH, E, FA) (small ? fwd_transform_quant_kernel<CH, E, FA, 4> : fwd_transform_quant_kernel<CH, E, FA, kMaxChunksPerThread>)
#define FRI_PICK(CH) (edge ? FRI_PICK_N(CH, true, false) : fast ? FRI_PICK_N(CH, false, true) : FRI_PICK_N(CH, false, false))
    kern = p.channels == 1 ? FRI_PICK(1) : FRI_PICK(3);
#undef FRI_PICK
#undef FRI_PICK_N
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    return hipGetLastError();
}

void build_pred_offsets(const uint16_t *nbr_table, uint32_t *out) {
    for (int p = 0; p < kCell; p++) {
        uint32_t row[3], o[3];
        std::memcpy(row, nbr_table + p * 6, sizeof(row));
        pred_offsets_from_row(row, o);
        out[4 * p] = o[0], out[4 * p + 1] = o[1], out[4 * p + 2] = o[2], out[4 * p + 3] = 0;
    }
}

hipError_t launch_predict_histogram(const DevicePlan &p, const int32_t *coefs_channel, const PredictParams &pp, uint8_t *bucket,
                                    int32_t *prediction, uint32_t *hist, unsigned long long *n_oob, hipStream_t stream) {
    if (!p.pred_acc) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    PredArgs a{};
    a.acc = p.pred_acc + (size_t)(p.pred_seq++ % kPredAccRing) * kPredAccWords; // one accumulator per launch in flight
    a.coefs = coefs_channel;
    a.pred_slots = p.pred_slots;
    a.nbr_table = p.nbr_table;
    a.pred_off = p.pred_off;
    a.interior = p.interior;
    a.valid_mask = p.valid_mask;
    a.bucket = bucket;
    a.prediction = prediction;
    a.hist = hist;
    a.n_oob = n_oob;
    a.n_tiles = p.n_pred_tiles;
    a.pp = pp;
    if (p.k2_single_buffered || !bucket || !prediction) { // optional outputs: the single-buffered kernel skips the stores of a NULL output
        uint32_t blocks = p.n_pred_tiles < p.hist_blocks ? p.n_pred_tiles : p.hist_blocks;
        if (!blocks) blocks = 1;
        hipLaunchKernelGGL(predict_histogram_kernel, dim3(blocks), dim3(kPredThreads), 0, stream, a);
        return hipGetLastError();
    }
    uint32_t blocks = p.n_pred_tiles < p.pred_blocks ? p.n_pred_tiles : p.pred_blocks;
    if (!blocks) blocks = 1;
    a.junk = p.junk;
    a.trace = p.trace;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(predict_histogram_kernel2), hipFuncAttributeMaxDynamicSharedMemorySize, kPred2LdsBytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(predict_histogram_kernel2, dim3(blocks), dim3(kPred2Threads), kPred2LdsBytes, stream, a);
    return hipGetLastError();
}

hipError_t launch_fit_accumulate(const DevicePlan &p, int mode, const int32_t *coefs_channel, const PredictParams &pp, unsigned long long *sums_int,
                                 double *sums_dbl, hipStream_t stream) {
    if (!p.fit_acc) return hipErrorInvalidValue;
    FitArgs a{};
    a.coefs = coefs_channel;
    a.pred_slots = p.pred_slots;
    a.nbr_table = p.nbr_table;
    a.pred_off = p.pred_off;
    a.interior = p.interior;
    a.valid_mask = p.valid_mask;
    a.n_tiles = p.n_pred_tiles;
    a.pp = pp;
    a.acc = p.fit_acc + (size_t)(p.fit_seq++ % kPredAccRing) * kFitAccWords;
    a.gram = sums_int;
    a.wtw = sums_int;
    a.wtr = sums_dbl;
    uint32_t blocks = p.n_pred_tiles < p.hist_blocks ? p.n_pred_tiles : p.hist_blocks;
    if (!blocks) blocks = 1;
    if (mode == 0)
        hipLaunchKernelGGL(fit_accumulate_kernel<0>, dim3(blocks), dim3(kPredThreads), 0, stream, a);
    else
        hipLaunchKernelGGL(fit_accumulate_kernel<1>, dim3(blocks), dim3(kPredThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_inverse_transform(const DevicePlan &p, const int32_t *coefs, const QMatrix &q, uint8_t *pixels, hipStream_t stream) {
    // RasterImage::from_wavelet starts from an all-zero raster (wavelet_transform.rs:309-317). When every pixel belongs to a
    // retained cell the kernel writes all of them (zeros included); only a lattice with holes (very thin images) needs the fill.
    if (!p.covers_image) {
        hipError_t e = hipMemsetAsync(pixels, 0, (size_t)p.width * p.height * p.channels, stream);
        if (e != hipSuccess) return e;
    }
    InvArgs a{};
    a.coefs = coefs;
    a.pixels = pixels;
    a.tiles = p.tiles;
    a.tile_meta = p.tile_meta;
    a.wg_tiles = p.wg_tiles;
    a.width = p.width;
    a.height = p.height;
    a.channels = p.channels;
    a.F = p.F;
    a.n_wg = p.n_wg;
    a.buf_bytes = (int32_t)inv_buf_bytes(p);
    a.max_wg_tiles = p.max_wg_tiles;
    a.ablate = p.k3_ablate;
    a.trace = p.trace;
    a.q = q;
    a.q_identity = 1;
    for (int i = 0; i <= 9; i++) a.q_identity &= (q.q[i] == 1);
    a.queue_bytes = (int32_t)inv_queue_bytes(p);
    const int items_per_wave = (p.max_tile_cells * p.channels + kInvWaves - 1) / kInvWaves;
    if (items_per_wave > kInvMaxItemsPerWave || p.max_wg_tiles > kInvThreads || p.lds_rows > 256 || p.lds_pitch / 16 > 256) return hipErrorInvalidConfiguration;
    const size_t lds = inv_lds_bytes(p);
    void (*kern)(const InvArgs) = items_per_wave <= 1 ? inverse_transform_kernel<1> : items_per_wave == 2 ? inverse_transform_kernel<2> : inverse_transform_kernel<4>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(p.n_wg), dim3(kInvThreads), lds, stream, a);
    return hipGetLastError();
}

} // namespace fri
