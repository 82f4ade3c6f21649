// k1_forward.hip -- K1 fwd_transform_quant: address-map gather + 9-level residue (S-)transform + per-layer quantiser
// (Fractal::extract_coefficients, stages/wavelet_transform.rs:179-225; quantization::encode, stages/quantization.rs:7-25).
// A byte/integer gather-scan kernel bounded by HBM traffic and instruction issue; no dense contraction, no MFMA.
#include "device_common.hpp"

namespace fri {
namespace {

constexpr int kFwdThreads = 256; // 4 waves per workgroup
constexpr int kFwdWaves = kFwdThreads / 64;

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
    int32_t ablate; // timing-only ablation (FRI_HIP_K1_ABLATE): 1 = skip staging, 2 = skip the cell loop, 4 = skip stores, 8 = return at entry. 0 in production.
    unsigned long long *trace; // diagnostic timeline, null in production
    unsigned long long *xcd_stat; // fri_hip_plan_tune_forward only (null otherwise): [8][2] = per XCD {sum of workgroup lifetimes in 100 MHz ticks, workgroups}
    QMatrix q;
};

// ---- packed arithmetic: two (cell, channel) items per register ------------------------------------------------------
// Every value of the transform fits 16 bits (differences in [-255, 255], low-pass values in [0, 255]), so one wave transforms TWO
// items at once: item A in the low half, item B in the high half of each VGPR; DPP / permlane moves carry both.
// d = l - r;  s = r + trunc(d / 2) = (l + r + (l < r)) >> 1  for l, r >= 0   (wavelet_transform.rs:211-218).
// Missing (None) operands enter as 0, which is exactly what try_apply substitutes (wavelet_transform.rs:14-26); which
// outputs are None is decided separately from the validity tree (boundary cells only).
//
// Round 5: the butterfly without packed-16 instructions (tools/micro/valu_rate2.hip: every v_pk_* holds the SIMD for 4 cycles, plain
// 32-bit add / sub / shift / xor for 2; rounds 1-4 spent five v_pk_* per butterfly and two more instructions to unpack each d):
//   x = l - r            one plain 32-bit subtraction. Low half: d_A mod 2^16, exact. High half: d_B - (l_A < r_A), off by the borrow.
//   s = v_lerp_u8(l, r, x >> 15)      per BYTE (l + r + (c & 1)) >> 1: bytes 0 and 2 hold the two items' values, bytes 1 and 3 are 0 and
//                        stay 0. The rounding bit only matters when l + r is odd, i.e. d != 0, and then the sign of (d_B - borrow) IS
//                        the sign of d_B (d_B >= 1 -> >= 0, d_B <= -1 -> < 0): bit 15 / bit 31 of x are good enough.
//   d_A, d_B             where a difference is an output, it is formed straight as int32 from the 16-bit fields (v_sub_u32_sdwa
//                        WORD_0 / WORD_1: the compiler's choice for the expressions below) - nothing to unpack in front of the stores.
// Three instructions per butterfly on the low-pass chain, five where both d leave (were five + two).
__device__ __forceinline__ uint32_t pk_low(uint32_t l, uint32_t r, uint32_t x) { return __builtin_amdgcn_lerp(l, r, x >> 15); }
__device__ __forceinline__ void pk_pair(uint32_t l, uint32_t r, int &dA, int &dB, uint32_t &s) {
    s = pk_low(l, r, l - r);
    dA = (int)(l & 0xFFFFu) - (int)(r & 0xFFFFu);
    dB = (int)(l >> 16) - (int)(r >> 16);
}
// x = l - r as one 32-bit word (see above) -> the two int32 differences: the low half sign-extended; the high half + the borrow the
// low half took (adding 0x8000 carries into the high half exactly when bit 15, the low difference's sign, is set).
__device__ __forceinline__ void unpack_x(uint32_t x, int &dA, int &dB) {
    dA = (int)(short)(x & 0xFFFFu);
    dB = (int)(x + 0x8000u) >> 16;
}

// Round J of the cross-lane part of the tree (level 5-J): combine the two child groups of 2^J lanes.
//   J = 0,1 : DPP quad_perm (lane^1, lane^2)
//   J = 2,3 : DPP row_half_mirror / row_mirror (lane -> 7-lane / 15-lane: lands in the sibling group)
//   J = 4   : v_permlane16_swap  (rows 0<->1, 2<->3)     } called with both operands = s they return
//   J = 5   : v_permlane32_swap  (lower 32 <-> upper 32) } (left child's s, right child's s) in every lane
// All lanes of a child group hold the same low-pass value, so fetching from ANY lane of the sibling group is
// enough -- that is what lets every round be a VALU cross-lane op instead of an LDS permute.
// Bit J of the lane clear = left child (the right child is the one that adds the LITERAL).
// J < 4: no selects into (l, r). x = other - own is d in the lanes of the right child (own = r) and -d in the left child's; the
// low-pass value is symmetric in its operands except for the rounding bit, (l < r) = sign(x) in right-child lanes and its
// complement in left-child lanes (d = 0: the bit does not matter) - one xor with a per-lane constant. The detail a lane keeps
// (tz == J: bit J is its lowest set bit) is always taken in a right-child lane, where x = d.
template <int J>
__device__ __forceinline__ uint32_t dpp_sibling(uint32_t v) {
    constexpr int ctrl = J == 0 ? 0xB1 : J == 1 ? 0x4E : J == 2 ? 0x141 : 0x140;
    // every lane has a source under these controls, so the `old` operand is never used: mov_dpp leaves it undefined and the
    // compiler needs no copy of v in front of each v_mov_b32_dpp
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, ctrl, 0xF, 0xF, true);
}
template <int J>
__device__ __forceinline__ void pk_cross(int lane, int tz, uint32_t &s, uint32_t &vlow) {
    uint32_t x;
    if constexpr (J < 4) {
        const uint32_t other = dpp_sibling<J>(s);
        x = other - s;
        const uint32_t flip = ((lane >> J) & 1) ? 0u : 0x00010001u; // loop invariant: a register per round
        s = __builtin_amdgcn_lerp(s, other, (x >> 15) ^ flip);
    } else {
        const uint2v w = J == 4 ? __builtin_amdgcn_permlane16_swap(s, s, false, false) : __builtin_amdgcn_permlane32_swap(s, s, false, false);
        x = w.x - w.y;
        s = pk_low(w.x, w.y, x);
    }
    if (tz == J) vlow = x;
}
// the same routing for the validity OR-tree (symmetric: no roles)
template <int J>
__device__ __forceinline__ int xlane_or(int v) {
    if constexpr (J < 4) {
        return v | (int)dpp_sibling<J>((uint32_t)v);
    } else {
        const uint2v w = J == 4 ? __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false) : __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
        return (int)(w.x | w.y);
    }
}

// The 9-level transform of two items held by one wave. leaf[j] = pixels of leaf 8*lane + j of item A (low half) and B (high half).
// rA / rB[0..3] = coef[256+4L .. +3], [4..5] = coef[128+2L ..], [6] = coef[64+L], [7] = coef[L] of item A / B, as int32.
__device__ __forceinline__ void fwd_wave_pk(const uint32_t (&leaf)[8], int lane, int (&rA)[8], int (&rB)[8]) {
    uint32_t s8[4], s7[2], s;
#pragma unroll
    for (int i = 0; i < 4; i++) pk_pair(leaf[2 * i], leaf[2 * i + 1], rA[i], rB[i], s8[i]); // level 8: nodes 256 + 4L + i
#pragma unroll
    for (int i = 0; i < 2; i++) pk_pair(s8[2 * i], s8[2 * i + 1], rA[4 + i], rB[4 + i], s7[i]); // level 7: nodes 128 + 2L + i
    pk_pair(s7[0], s7[1], rA[6], rB[6], s);                                                        // level 6: node 64 + L
    const int tz = lane ? __builtin_ctz(lane) : 6;
    uint32_t vlow = 0;
    pk_cross<0>(lane, tz, s, vlow); // level 5
    pk_cross<1>(lane, tz, s, vlow);
    pk_cross<2>(lane, tz, s, vlow);
    pk_cross<3>(lane, tz, s, vlow);
    pk_cross<4>(lane, tz, s, vlow);
    pk_cross<5>(lane, tz, s, vlow); // level 0 (root)
    if (lane == 0) vlow = s; // coefficients[0] = low_pass_values[1] (wavelet_transform.rs:221); values <= 255 in both halves: unpack_x leaves them alone
    unpack_x((uint32_t)__shfl((int)vlow, low_source_lane(lane)), rA[7], rB[7]);
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
    int vlow = 0;
    sv = xlane_or<0>(sv); if (tz == 0) vlow = sv;
    sv = xlane_or<1>(sv); if (tz == 1) vlow = sv;
    sv = xlane_or<2>(sv); if (tz == 2) vlow = sv;
    sv = xlane_or<3>(sv); if (tz == 3) vlow = sv;
    sv = xlane_or<4>(sv); if (tz == 4) vlow = sv;
    sv = xlane_or<5>(sv); if (tz == 5) vlow = sv;
    if (lane == 0) vlow = sv;
    const uint32_t low = (uint32_t)__shfl(vlow, low_source_lane(lane));
    // compact: level 8 bits 0,2,4,6 -> 0..3 ; level 7 bits 0,4 -> 4,5 ; level 6 -> 6 ; low -> 7   (per 16-bit half)
    uint32_t out = (v8 & 0x00010001u) | ((v8 >> 1) & 0x00020002u) | ((v8 >> 2) & 0x00040004u) | ((v8 >> 3) & 0x00080008u);
    out |= ((v7 & 0x00010001u) << 4) | ((v7 & 0x00100010u) << 1);
    out |= (v6 << 6) | (low << 7);
    return out;
}

__device__ __forceinline__ int quant_one(int v, int heap_index, const FwdArgs &a) { // quantization.rs:13-17, None untouched
    return v == kNone ? v : v / a.q.q[quant_layer(heap_index)];
}

// One item (HALF = 0: item A, 1: item B of the pair - selects the half of `valid`): applies the None mask and the quantiser, and stores the
// cell's 512 int32 coefficients as four fully coalesced store instructions (1 KiB + 512 B + 256 B + 256 B).
// C16 (round 5, the chains' compact coefficient planes): the item leaves as 512 int16 - every coefficient of the transform fits nine bits - with None as 0, which is
// what every reader of such a plane takes a None for (the scan and the fit read neighbours with unwrap_or(0) and know Some from None by the plan's masks); `coefs`
// is then the int16 plane's base and elem_off counts halfwords. Plain stores: the plane is read back by the next kernels of the chain.
template <int HALF, bool MASKED, bool QID, bool NT, bool C16 = false>
__device__ __forceinline__ void store_item(int32_t *__restrict__ coefs, uint32_t elem_off, int lane, const int (&res)[8], uint32_t valid,
                                           const FwdArgs &a) {
    int v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v[i] = res[i];
        if (MASKED && !((valid >> (16 * HALF + i)) & 1u)) v[i] = kNone;
    }
    if (!QID) { // QID: the all-ones matrix of today's reference (quantization.rs:3-5) - no division code in that kernel at all
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
    if constexpr (C16) {
        int16_t *o16 = reinterpret_cast<int16_t *>(coefs) + elem_off;
        if (MASKED) {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = v[i] == kNone ? 0 : v[i]; // (the quantiser leaves a None alone)
        }
        const uint32_t p0 = ((uint32_t)v[0] & 0xFFFFu) | ((uint32_t)v[1] << 16), p1 = ((uint32_t)v[2] & 0xFFFFu) | ((uint32_t)v[3] << 16);
        const uint32_t p2 = ((uint32_t)v[4] & 0xFFFFu) | ((uint32_t)v[5] << 16);
        *reinterpret_cast<i32x2 *>(o16 + 256 + 4 * lane) = i32x2{(int)p0, (int)p1};
        *reinterpret_cast<uint32_t *>(o16 + 128 + 2 * lane) = p2;
        o16[64 + lane] = (int16_t)v[6];
        o16[lane] = (int16_t)v[7];
        return;
    }
    if constexpr (NT) {
        __builtin_nontemporal_store(i32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<i32x4 *>(out + 256 + 4 * lane));
        __builtin_nontemporal_store(i32x2{v[4], v[5]}, reinterpret_cast<i32x2 *>(out + 128 + 2 * lane));
        __builtin_nontemporal_store(v[6], out + 64 + lane);
        __builtin_nontemporal_store(v[7], out + lane);
    } else { // plain stores: the coefficients may stay in the L2 / the Infinity Cache for the kernels that read them next (see launch_fwd_transform_quant)
        *reinterpret_cast<i32x4 *>(out + 256 + 4 * lane) = i32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<i32x2 *>(out + 128 + 2 * lane) = i32x2{v[4], v[5]};
        out[64 + lane] = v[6];
        out[lane] = v[7];
    }
}


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
    const uint32_t n_rows = (ablate_flags(a.ablate) & 1) ? 0u : (uint32_t)t.n_rows;
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
    const uint32_t total = (ablate_flags(a.ablate) & 1) ? 0u : (uint32_t)t.n_rows * a.cpr;
    uint8_t *mine = junk + 16 * tid;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const uint32_t i = (uint32_t)tid + (uint32_t)c * kFwdThreads;
        *reinterpret_cast<u32x4 *>(i < total ? buf + 16u * i : mine) = sv[c];
    }
    *reinterpret_cast<u32x4 *>(tid < t.cell_count ? buf + a.meta_off + 16 * tid : mine) = sv[NCH];
}

// ---- leaf fetch for single-channel planes ----------------------------------------------------------------------------
// wb[row] = byte offset (in LDS address space) of the row's 4-byte window: w[0] = pixels x0..x0+3 of row 0,
// w[1], w[2] = pixels x0-1..x0+2 of rows 1, 2.
typedef const uint32_t __attribute__((address_space(3))) *LdsDwordPtr;
__device__ __forceinline__ void fetch_windows(const int (&wb)[3], uint32_t (&w)[3]) {
#pragma unroll
    for (int row = 0; row < 3; row++) {
        const uint32_t b = (uint32_t)wb[row];
        const LdsDwordPtr p = (LdsDwordPtr)(uintptr_t)(b & ~3u); // the address is complete: nothing left to add per window
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
// The same for interleaved RGB: a lane's 8 leaves of one channel sit 3 bytes apart - row 0: bytes 0 and 6 from (x0, y0, ch), rows 1
// and 2: bytes 0, 3, 6, 9 and 0, 6 from (x0 - 1, y0 + 1 / + 2, ch). Seven 4-byte windows out of ten aligned dwords (six LDS
// instructions) instead of eight ds_read_u8, which hold the LDS pipe ~12 cycles each (PMC: 200 LDS cycles per pair of items,
// half of them bank conflicts). w[0], w[1]: row 0 from byte 0 / byte 4; w[2..4]: row 1 from byte 0 / 4 / 8; w[5], w[6]: row 2.
__device__ __forceinline__ void fetch_windows_rgb(int buf_off, const int (&rb)[3], uint32_t (&w)[7]) {
    const uint32_t b0 = (uint32_t)(buf_off + rb[0]), b1 = (uint32_t)(buf_off + rb[1] - 3), b2 = (uint32_t)(buf_off + rb[2] - 3);
    const LdsDwordPtr p0 = (LdsDwordPtr)(uintptr_t)(b0 & ~3u), p1 = (LdsDwordPtr)(uintptr_t)(b1 & ~3u), p2 = (LdsDwordPtr)(uintptr_t)(b2 & ~3u);
    const uint32_t d00 = p0[0], d01 = p0[1], d02 = p0[2];
    const uint32_t d10 = p1[0], d11 = p1[1], d12 = p1[2], d13 = p1[3];
    const uint32_t d20 = p2[0], d21 = p2[1], d22 = p2[2];
    w[0] = __builtin_amdgcn_alignbyte(d01, d00, b0 & 3u);
    w[1] = __builtin_amdgcn_alignbyte(d02, d01, b0 & 3u);
    w[2] = __builtin_amdgcn_alignbyte(d11, d10, b1 & 3u);
    w[3] = __builtin_amdgcn_alignbyte(d12, d11, b1 & 3u);
    w[4] = __builtin_amdgcn_alignbyte(d13, d12, b1 & 3u);
    w[5] = __builtin_amdgcn_alignbyte(d21, d20, b2 & 3u);
    w[6] = __builtin_amdgcn_alignbyte(d22, d21, b2 & 3u);
}
// {byte I of a, byte I of b} zero-extended into the two 16-bit halves.
template <int I>
__device__ __forceinline__ uint32_t pair_bytes(uint32_t a, uint32_t b) {
    return __builtin_amdgcn_perm(b, a, 0x0C000C00u | ((4u + I) << 16) | (unsigned)I);
}

// One item of a pair: where its 8 leaves sit in the staged rectangle, and which of them are inside the image.
struct ItemAddr {
    int rb[3];          // C = 3: byte offset of (x0, y0 + dy, ch) in the tile buffer. C = 1: offsets of the three leaf windows
                        // (fetch_windows) as LDS addresses: buffer address and the rows' -1 already folded in
    uint32_t leaf_mask; // bit j = leaf j inside the image (0xFF for interior cells)
    uint32_t elem_off;  // (ch * F + cell) * 512
};

template <int C, bool FAST>
__device__ __forceinline__ ItemAddr item_addr(const FwdArgs &a, const Tile &t, const TileCell *meta, int it, int ldx, int ldy, int lane_rb, int buf_off,
                                               uint32_t sh_base, uint32_t wc16) {
    const int cl = it / C, ch = it - cl * C;
    const TileCell m = meta[cl];
    ItemAddr r;
    r.elem_off = ((uint32_t)ch * a.F + (uint32_t)__builtin_amdgcn_readfirstlane(m.cell)) * kCell;
    const int x0 = m.cx + ldx, y0 = m.cy + ldy;
    // row y of the staged rectangle starts (a16 + (y * width + x_lo) * C) & 15 bytes into its LDS row
    const int col = __mul24(x0 - t.x_lo, C) + ch;
    if (FAST) { // width * C is a multiple of 16: the same lead-in for every row
        // = (cell part, wave-uniform: scalar ALU) + (lane part ldy * pitch + ldx * C, loop invariant): one vector add per item
        const int cell_base = (__builtin_amdgcn_readfirstlane(m.cy) - t.y_lo) * a.pitch + (__builtin_amdgcn_readfirstlane(m.cx) - t.x_lo) * C + ch + (int)(sh_base & 15u);
        r.rb[0] = cell_base + lane_rb + (C == 1 ? buf_off : 0);
        r.rb[1] = r.rb[0] + (C == 1 ? a.pitch - 1 : a.pitch);
        r.rb[2] = r.rb[0] + (C == 1 ? 2 * a.pitch - 1 : 2 * a.pitch);
    } else {
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int y = y0 + dy;
            const uint32_t sh = (sh_base + __umul24((uint32_t)y & 15u, wc16)) & 15u;
            r.rb[dy] = __mul24(y - t.y_lo, a.pitch) + (int)sh + col + (C == 1 ? buf_off - (dy ? 1 : 0) : 0);
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
// MEASURE: the same kernel under another name - the launches fri_hip_plan_tune_forward times on its scratch buffers (candidate tilings, most of them slower than
// the one kept) must not sit in the same row of a kernel trace's statistics as the caller's launches.
template <int C, bool EDGE, bool FAST, int NCH, bool QID, bool NT, bool MEASURE = false, bool C16 = false>
__global__ void __launch_bounds__(kFwdThreads) fwd_transform_quant_kernel(const FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    if (ablate_flags(a.ablate) & 8) return; // timing only: what dispatching the grid alone costs
    const int tid = threadIdx.x, lane = tid & 63;
    // scalar: the per-pair branches become uniform and the items' store bases stay in SGPRs (global_store ... s[base:base+1]) instead
    // of 64-bit vector address arithmetic per store: 117 -> 108 VGPRs, ~16 fewer vector instructions per pair
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t wg = xcd_contiguous_share(blockIdx.x, a.n_wg);
    const int tb = a.wg_tiles[wg], te = a.wg_tiles[wg + 1];
    trace_stamp(a.trace, wg, 0, tid);
    const unsigned long long t_entry = a.xcd_stat ? wall_clock64() : 0ull;
    // The share's tile descriptors go to LDS once, before any store is issued: fetching them inside the loop would be a
    // vector load behind s_waitcnt vmcnt(0) per tile (scalar loads are off the table once the kernel has stored), and that
    // wait would also drain the previous tile's coefficient stores.
    uint8_t *junk = lds + 2 * a.buf_bytes; // 16 bytes per thread, written, never read
    Tile *lds_tiles = reinterpret_cast<Tile *>(junk + 16 * kFwdThreads);
    if (tid < te - tb) lds_tiles[tid] = a.tiles[tb + tid];
    const uint8_t *img = a.pixels + (size_t)blockIdx.y * a.pixel_stride;
    int32_t *coefs = C16 ? reinterpret_cast<int32_t *>(reinterpret_cast<int16_t *>(a.coefs) + (size_t)blockIdx.y * a.coef_stride) // (a compact plane: see store_item)
                         : a.coefs + (size_t)blockIdx.y * a.coef_stride;
    const int ldx = lane_dx(lane), ldy = lane_dy(lane);
    const int lane_rb = ldy * a.pitch + ldx * C; // the lane's share of a leaf window's LDS offset (item_addr)
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
        const int buf_off = (int)(uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)cur; // LDS address of the buffer
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

        // Issue priority falls with a wave's progress through the tile (3 -> 2 -> 1, 0 for the stores), so that at every conflict the wave that is furthest
        // behind - of any of the CU's workgroups - is the preferred one (the arbiter otherwise prefers the oldest): planes -1 % (16.43-16.54 against 16.61-16.69 us
        // in three interleaved pairs), RGB +2 % (three workgroups per CU with five-cell tiles: not for C = 3).
        if constexpr (C == 1) __builtin_amdgcn_s_setprio(3);
        const TileCell *meta = reinterpret_cast<const TileCell *>(cur + a.meta_off);
        const int n_items = (ablate_flags(a.ablate) & 2) ? 0 : t.cell_count * C;
        const uint32_t sh_base = a16 + (uint32_t)t.x_lo * C;
        int resA[kMaxPairsPerWave][8], resB[kMaxPairsPerWave][8];
        uint32_t offA[kMaxPairsPerWave], offB[kMaxPairsPerWave], valid[kMaxPairsPerWave];
#pragma unroll
        for (int c = 0; c < kMaxPairsPerWave; c++) {
            const int itA = 2 * (wave + kFwdWaves * c);
            offA[c] = offB[c] = 0;
            valid[c] = 0xFFFFFFFFu;
            if (itA < n_items) {
                const int itB = itA + 1 < n_items ? itA + 1 : itA; // odd tail: item B mirrors A and is not stored
                const ItemAddr A = item_addr<C, FAST>(a, t, meta, itA, ldx, ldy, lane_rb, buf_off, sh_base, wc16);
                const ItemAddr B = item_addr<C, FAST>(a, t, meta, itB, ldx, ldy, lane_rb, buf_off, sh_base, wc16);
                offA[c] = A.elem_off;
                offB[c] = B.elem_off;
                uint32_t leaf[8];
                if constexpr (C == 1) {
                    // Sub-dword LDS reads are slow (a ds_read_u8 wave-instruction holds the LDS pipe for ~16 cycles), and a
                    // lane's 8 leaves sit in three 4-byte windows: row 0 = x..x+3, rows 1 and 2 = x-1..x+2. Fetch each window
                    // as two aligned dwords + v_alignbyte, then v_perm pairs item A's and item B's bytes into 16-bit halves.
                    uint32_t wA[3], wB[3];
                    fetch_windows(A.rb, wA);
                    fetch_windows(B.rb, wB);
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
                } else {
                    // window byte of leaf j: row 0: leaf0 -> w[0] byte 0, leaf4 -> w[1] byte 2; row 1: leaf2 -> w[2] byte 0, leaf1 -> w[2] byte 3,
                    // leaf6 -> w[3] byte 2, leaf5 -> w[4] byte 1; row 2: leaf3 -> w[5] byte 0, leaf7 -> w[6] byte 2
                    uint32_t wA[7], wB[7];
                    fetch_windows_rgb(buf_off, A.rb, wA);
                    fetch_windows_rgb(buf_off, B.rb, wB);
                    leaf[0] = pair_bytes<0>(wA[0], wB[0]);
                    leaf[4] = pair_bytes<2>(wA[1], wB[1]);
                    leaf[2] = pair_bytes<0>(wA[2], wB[2]);
                    leaf[1] = pair_bytes<3>(wA[2], wB[2]);
                    leaf[6] = pair_bytes<2>(wA[3], wB[3]);
                    leaf[5] = pair_bytes<1>(wA[4], wB[4]);
                    leaf[3] = pair_bytes<0>(wA[5], wB[5]);
                    leaf[7] = pair_bytes<2>(wA[6], wB[6]);
                    if ((A.leaf_mask & B.leaf_mask) != 0xFFu) { // some lane has a leaf outside the image: it enters as 0, the None outputs come from the validity tree
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            if (!((A.leaf_mask >> j) & 1u)) leaf[j] &= 0xFFFF0000u;
                            if (!((B.leaf_mask >> j) & 1u)) leaf[j] &= 0x0000FFFFu;
                        }
                    }
                }
                fwd_wave_pk(leaf, lane, resA[c], resB[c]);
                if constexpr (C == 1) {
                    if (c == 0)
                        __builtin_amdgcn_s_setprio(2);
                    else
                        __builtin_amdgcn_s_setprio(1);
                }
                if (__builtin_amdgcn_readfirstlane(meta[itA / C].interior & meta[itB / C].interior) == 0)
                    valid[c] = validity_tree_pk(A.leaf_mask | (B.leaf_mask << 16), lane);
            }
        }
        if (more) stage_commit<NCH>(a, tn, nxt, junk, tid, st);
        if constexpr (C == 1) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int c = 0; c < kMaxPairsPerWave; c++) {
            const int itA = 2 * (wave + kFwdWaves * c);
            if (itA < n_items) {
                bool go = !(ablate_flags(a.ablate) & 4);
                if (!go) go = (resA[c][0] ^ resA[c][1] ^ resA[c][2] ^ resA[c][3] ^ resA[c][4] ^ resA[c][5] ^ resA[c][6] ^ resA[c][7] ^ resB[c][0] ^ resB[c][1] ^ resB[c][2] ^ resB[c][3] ^
                               resB[c][4] ^ resB[c][5] ^ resB[c][6] ^ resB[c][7]) == 0x12345678; // keeps the arithmetic alive
                if (go) {
                    if (__builtin_amdgcn_readfirstlane(valid[c] == 0xFFFFFFFFu ? 1 : 0) && __all(valid[c] == 0xFFFFFFFFu)) { // no None anywhere in the pair
                        store_item<0, false, QID, NT, C16>(coefs, offA[c], lane, resA[c], valid[c], a);
                        if (itA + 1 < n_items) store_item<1, false, QID, NT, C16>(coefs, offB[c], lane, resB[c], valid[c], a);
                    } else {
                        store_item<0, true, QID, NT, C16>(coefs, offA[c], lane, resA[c], valid[c], a);
                        if (itA + 1 < n_items) store_item<1, true, QID, NT, C16>(coefs, offB[c], lane, resB[c], valid[c], a);
                    }
                }
            }
        }
        trace_stamp(a.trace, wg, 2 + ti - tb, tid);
        lds_barrier();
        t = tn;
    }
    trace_exit(a.trace, wg, tid);
    if (a.xcd_stat && tid == 0) { // how long this XCD's workgroups live: what the tuner balances the XCDs' shares by (two fire-and-forget atomics, tuning launches only)
        const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; // XCC_ID[3:0]
        (void)__hip_atomic_fetch_add(a.xcd_stat + 2 * xcc, wall_clock64() - t_entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (void)__hip_atomic_fetch_add(a.xcd_stat + 2 * xcc + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


} // namespace

static size_t fwd_meta_offset(const DevicePlan &p) { return ((size_t)p.lds_pitch * p.lds_rows + 15) & ~(size_t)15; }
static size_t fwd_buf_bytes(const DevicePlan &p) { return fwd_meta_offset(p) + (size_t)p.max_tile_cells * sizeof(TileCell); }
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
                                      size_t coef_stride, const QMatrix &q, hipStream_t stream, bool cached_stores, int16_t *coefs16) {
    if (!fwd_plan_fits(p)) return hipErrorInvalidConfiguration;
    const bool plain = p.k1_cached_stores >= 0 ? p.k1_cached_stores > 0 : cached_stores;
    FwdArgs a{};
    a.pixels = pixels;
    a.pixel_stride = pixel_stride;
    a.coefs = coefs16 ? reinterpret_cast<int32_t *>(coefs16) : coefs; // (coefs16: the chains' compact planes - int16, None as 0, coef_stride in halfwords; store_item<.., C16>)
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
    a.xcd_stat = p.k1_xcd_stat;
    const size_t lds = fwd_lds_bytes(p);
    const dim3 grid(a.n_wg, n_images), block(kFwdThreads);
    // EDGE variant only when a 16-byte chunk could straddle the ends of one of the caller's image buffers
    const size_t img_bytes = (size_t)p.width * p.height * p.channels;
    const bool edge = (reinterpret_cast<uintptr_t>(pixels) & 15) || (img_bytes & 15) || (n_images > 1 && (pixel_stride & 15));
    const bool fast = !edge && (((size_t)p.width * p.channels) & 15) == 0; // every image row starts at the same offset mod 16
    const bool small = fwd_chunks(p) <= 4 * (size_t)kFwdThreads;           // 4 chunks per thread suffice (the common, tuned case)
    void (*kern)(FwdArgs);
#define FRI_PICK_T(CH, E, FA, N, QI) (plain ? fwd_transform_quant_kernel<CH, E, FA, N, QI, false> : fwd_transform_quant_kernel<CH, E, FA, N, QI, true>)
#define FRI_PICK_Q(CH, E, FA, N) (a.q_identity ? FRI_PICK_T(CH, E, FA, N, true) : FRI_PICK_T(CH, E, FA, N, false))
#define FRI_PICK_N(CH, E, FA) (small ? FRI_PICK_Q(CH, E, FA, 4) : FRI_PICK_Q(CH, E, FA, kMaxChunksPerThread))
#define FRI_PICK(CH) (edge ? FRI_PICK_N(CH, true, false) : fast ? FRI_PICK_N(CH, false, true) : FRI_PICK_N(CH, false, false))
    kern = p.channels == 1 ? FRI_PICK(1) : FRI_PICK(3);
    if (p.k1_measuring && !edge && !plain && a.q_identity) { // the tuner's launches (aligned scratch, all-ones matrix, nontemporal stores): the MEASURE instance
#define FRI_PICK_M(CH) (fast ? (small ? fwd_transform_quant_kernel<CH, false, true, 4, true, true, true> : fwd_transform_quant_kernel<CH, false, true, kMaxChunksPerThread, true, true, true>) \
                             : (small ? fwd_transform_quant_kernel<CH, false, false, 4, true, true, true> : fwd_transform_quant_kernel<CH, false, false, kMaxChunksPerThread, true, true, true>))
        kern = p.channels == 1 ? FRI_PICK_M(1) : FRI_PICK_M(3);
#undef FRI_PICK_M
    }
    if (coefs16) { // (plain stores, no MEASURE instance)
#define FRI_PICK_C(CH, E, FA, N) (a.q_identity ? fwd_transform_quant_kernel<CH, E, FA, N, true, false, false, true> : fwd_transform_quant_kernel<CH, E, FA, N, false, false, false, true>)
#define FRI_PICK_CN(CH, E, FA) (small ? FRI_PICK_C(CH, E, FA, 4) : FRI_PICK_C(CH, E, FA, kMaxChunksPerThread))
#define FRI_PICK_CC(CH) (edge ? FRI_PICK_CN(CH, true, false) : fast ? FRI_PICK_CN(CH, false, true) : FRI_PICK_CN(CH, false, false))
        kern = p.channels == 1 ? FRI_PICK_CC(1) : FRI_PICK_CC(3);
#undef FRI_PICK_CC
#undef FRI_PICK_CN
#undef FRI_PICK_C
    }
#undef FRI_PICK
#undef FRI_PICK_N
#undef FRI_PICK_Q
#undef FRI_PICK_T
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    (void)hipGetLastError(); // the check behind the launch must not pick up an error an earlier, unrelated call left behind
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    return hipGetLastError();
}

} // namespace fri
