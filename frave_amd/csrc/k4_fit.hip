// k4_fit.hip -- the normal-equation sums behind the context-model fit (next row 8f-3).
#include "gather_common.hpp"
#include "solve6.hpp"

namespace fri {
namespace {

__device__ __forceinline__ uint32_t f32_to_u32_sat(float x) { // Rust-style `as u32`: truncation toward zero, saturating, NaN -> 0 (what v_cvt_u32_f32 does)
    uint32_t r;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
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
    const uint32_t *pred_off;  // [512][4] packed neighbour BYTE offsets per node in K2's permuted 1 KiB cell layout (build_gather_tables)
    const uint16_t *pair_pos;  // [256] dword position of a pair of sibling nodes inside a cell (gather_layout.inc)
    const uint32_t *halo_list; // [1024] the halo values a tile needs (build_halo_list), two per thread of kernel2
    unsigned long long *gram; // [3][28]   (MODE 0)
    unsigned long long *wtw;  // [3][21]   (MODE 1)
    double *wtr;              // [3][6]    (MODE 1)
    unsigned long long *out_range; // [n_planes] or NULL: waves that staged a Some coefficient outside [-256, 255] (the sums are then not to be trusted)
    unsigned long long *acc;  // plan scratch, all zero between launches: [kFitAccInt] integer sums, [18] fixed-point sums (W^T r), then the ticket
    // planes of a batch (grid.y): plane k reads coefs + k * coef_stride, takes its parameters from params[k] (NULL: pp), hands over through
    // acc + k * kFitAccWords and writes gram / wtw + k * 3 * NI, wtr + k * 18
    size_t coef_stride;
    const PredictParams *params;
    PredictParams pp3[3]; // plane k < 3 of a launch without a params array
    int32_t ablate;       // timing-only (tuning build, FRI_HIP_K4_ABLATE): 1 = no sums, 2 = tiles after the first are staged without their global loads
    unsigned long long *trace; // diagnostic timeline (tuning build + FRI_HIP_TRACE=1), null in production
    int32_t older_eighths;     // kernel2 with a full grid: eighths of a CU's tiles its first-dispatched workgroup walks (0: equal shares; see the kernel)
    // kernel2, the solve in the tail (NULL: sums only): the workgroup that moves a plane's totals out also solves its three 6 x 6 systems and writes
    // the parameters - MODE 0: .value, MODE 1: .width of solve_params[plane] - where the next kernel of the chain reads them
    float *solve_params;            // PredictParams[n_planes] in device memory
    float *host_params;             // the same into mapped host memory (or NULL), with the plane's out-of-range count next to it:
    unsigned long long *host_range; // [n_planes] mapped host memory (or NULL)
    unsigned long long rows[3];     // MODE 1: heights of the reference's matrices (F * {256, 128, 128})
};
constexpr int kFitAccInt = 3 * 28, kFitAccDbl = kFitAccInt, kFitAccTicket = kFitAccInt + 18, kFitAccRange = kFitAccTicket + 1;
static_assert(kFitAccRange + 1 == (int)kFitAccWords, "fit accumulator layout");

// ---------------------------------------------------------------------------------------------------------------------
// Round 2: the same sums with a third of the instructions. What the kernel above spends per pair of nodes: 14 sign-extending LDS
// reads, 7 packs, the unpacking and adding of 12 neighbour offsets kept in LDS (~30 instructions), 7 masks, 28 (21) multiply-adds.
// Here a wave owns ONE pair of a lane's nodes - PG = wave & 3: heap 4 lane + {0, 1}, 4 lane + {2, 3} (levels 0..7),
// 256 + 4 lane + {0, 1}, 256 + 4 lane + {2, 3} (level 8) - in the eight block cells of two block rows, whose LDS slots lie at fixed
// distances: twelve precomputed LDS addresses serve all eight cells (the distance rides in the instructions' offset field), and the
// two nodes of the pair share every register - the first is gathered with ds_read_u16_d16 into the low halves, the second with
// ds_read_u16_d16_hi into the high halves of the SAME six registers, so the packed operands of v_dot2c_i32_i16 arrive without a pack
// instruction, and the pair's own values are one aligned dword of the staged cell. Lanes stay bound to layer groups (one set of
// accumulators per lane): level 8 is group 0; below, lanes 0..31 are group 2 (levels 0..6), lanes 32..63 group 1 (level 7).
// Staging, tile walk and hand-over are the kernel above's.
// ---------------------------------------------------------------------------------------------------------------------
// The six neighbour values of the lane's two nodes in one cell, packed {first node, second node} per register (the value pass's int16 image). A d16 load into
// one half of a register does not keep the other half on this chip (the register's unused half comes back as zero - d16 writes are whole-register when the
// memories run with ECC), so the first node's value arrives through ds_read_u16 (zero-extended), the second node's through ds_read_u16_d16_hi (already shifted)
// and one v_or_b32 (two issue cycles; v_perm_b32 or v_lshl_or_b32: four) makes the packed operand. One asm statement with its wait inside: the compiler does
// not track these loads, and nothing of its own can come between them and their wait.
template <int OFFSET>
__device__ __forceinline__ void fit2_gather(uint32_t (&g)[6], const uint32_t (&a)[6], const uint32_t (&b)[6]) {
    uint32_t lo[6], hi[6];
    asm volatile("ds_read_u16 %0, %12 offset:%24\n\t"
                 "ds_read_u16 %1, %13 offset:%24\n\t"
                 "ds_read_u16 %2, %14 offset:%24\n\t"
                 "ds_read_u16 %3, %15 offset:%24\n\t"
                 "ds_read_u16 %4, %16 offset:%24\n\t"
                 "ds_read_u16 %5, %17 offset:%24\n\t"
                 "ds_read_u16_d16_hi %6, %18 offset:%24\n\t"
                 "ds_read_u16_d16_hi %7, %19 offset:%24\n\t"
                 "ds_read_u16_d16_hi %8, %20 offset:%24\n\t"
                 "ds_read_u16_d16_hi %9, %21 offset:%24\n\t"
                 "ds_read_u16_d16_hi %10, %22 offset:%24\n\t"
                 "ds_read_u16_d16_hi %11, %23 offset:%24\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(lo[5]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4]),
                   "=&v"(hi[5])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "n"(OFFSET)
                 : "memory");
#pragma unroll
    for (int k = 0; k < 6; k++) g[k] = lo[k] | hi[k];
}

// The width pass's gather: twelve halfwords that ARE the upper halves of floats (fit2_pack<true>), loaded straight into the upper halves of twelve registers
// (ds_read_u16_d16_hi, the other half zero: see above); one asm statement with its wait inside, like fit2_gather.
template <int OFFSET>
__device__ __forceinline__ void fit2_gather_floats(uint32_t (&lo)[6], uint32_t (&hi)[6], const uint32_t (&a)[6], const uint32_t (&b)[6]) {
    asm volatile("ds_read_u16_d16_hi %0, %12 offset:%24\n\t"
                 "ds_read_u16_d16_hi %1, %13 offset:%24\n\t"
                 "ds_read_u16_d16_hi %2, %14 offset:%24\n\t"
                 "ds_read_u16_d16_hi %3, %15 offset:%24\n\t"
                 "ds_read_u16_d16_hi %4, %16 offset:%24\n\t"
                 "ds_read_u16_d16_hi %5, %17 offset:%24\n\t"
                 "ds_read_u16_d16_hi %6, %18 offset:%24\n\t"
                 "ds_read_u16_d16_hi %7, %19 offset:%24\n\t"
                 "ds_read_u16_d16_hi %8, %20 offset:%24\n\t"
                 "ds_read_u16_d16_hi %9, %21 offset:%24\n\t"
                 "ds_read_u16_d16_hi %10, %22 offset:%24\n\t"
                 "ds_read_u16_d16_hi %11, %23 offset:%24\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(lo[5]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4]),
                   "=&v"(hi[5])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "n"(OFFSET)
                 : "memory");
}

// One pair of nodes of one cell, value fit: the 28 sums of [v0..v5, value] [v0..v5, value]^T. g: the six packed int16 neighbour values; own: the pair's own
// values packed the same way. A None row is all zeros in the reference (:109-134): the caller has masked g and own (fit2_cell: boundary cells only - an
// interior cell has no None, and the one lane whose pair is not a row of the fit gathers zeros).
__device__ __forceinline__ void fit2_pair_value(const uint32_t (&g)[6], uint32_t own, int (&acc)[28]) {
    s16x2 u[7];
#pragma unroll
    for (int k = 0; k < 6; k++) u[k] = __builtin_bit_cast(s16x2, g[k]);
    u[6] = __builtin_bit_cast(s16x2, own);
    int n = 0;
#pragma unroll
    for (int r0 = 0; r0 < 7; r0++)
#pragma unroll
        for (int c0 = r0; c0 < 7; c0++, n++) acc[n] = __builtin_amdgcn_sdot2(u[r0], u[c0], acc[n], false);
}

// One pair of nodes of one cell, width fit: 21 integer sums W^T W and six f32 sums W^T r. The width pass stages the UPPER HALVES OF THE VALUES' FLOATS (K2's
// image format: exact for |v| <= 256, the fit's precondition), so a gathered halfword is the float after one shift - the pass needs every value as a float
// (the residual r = |f32(value) - A x| in f32, left to right like nalgebra's gemv, context_modeling.rs:150-160) and a conversion costs four issue cycles where a
// shift, an f32 add or multiply cost two (tools/micro/valu_rate2.hip; with the int16 image of the value pass a pair spent 24 of its ~100 vector instructions
// on v_cvt_f32_i32). The features w = [1, |v0-v3|, |v1-v2|, |v4-v5|, |v1-v5|, |v2-v4|] are exact f32 differences; their packed int16 form for v_dot2c comes out
// of the mantissa: |d| + 2^23 holds the integer |d| <= 511 in its low bits. lo / hi: the first / second node's six neighbour values as floats (fit2_gather_floats); own: both
// nodes' own halfwords; one_bits: 1 per Some node. A None row: everything masked to 0 by the caller, its residual is |0 - 0|.
__device__ __forceinline__ void fit2_pair_width(const uint32_t (&lo)[6], const uint32_t (&hi)[6], uint32_t own, uint32_t one_bits, const float (&vp)[6], int (&acc)[28],
                                                float (&facc)[6]) {
    float fa[7], fb[7];
#pragma unroll
    for (int k = 0; k < 6; k++) fa[k] = __builtin_bit_cast(float, lo[k]), fb[k] = __builtin_bit_cast(float, hi[k]);
    fa[6] = __builtin_bit_cast(float, own << 16), fb[6] = __builtin_bit_cast(float, own & 0xFFFF0000u);
    float pa = __fmul_rn(fa[0], vp[0]), pb = __fmul_rn(fb[0], vp[0]);
#pragma unroll
    for (int k = 1; k < 6; k++) pa = __fadd_rn(pa, __fmul_rn(fa[k], vp[k])), pb = __fadd_rn(pb, __fmul_rn(fb[k], vp[k]));
    const float ra = fabsf(__fsub_rn(fa[6], pa)), rb = fabsf(__fsub_rn(fb[6], pb));
    constexpr int kD0[5] = {0, 1, 4, 1, 2}, kD1[5] = {3, 2, 5, 5, 4};
    float da[5], db[5];
#pragma unroll
    for (int j = 0; j < 5; j++) da[j] = fabsf(__fsub_rn(fa[kD0[j]], fa[kD1[j]])), db[j] = fabsf(__fsub_rn(fb[kD0[j]], fb[kD1[j]]));
    s16x2 w[6];
    w[0] = __builtin_bit_cast(s16x2, one_bits);
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint32_t ia = __builtin_bit_cast(uint32_t, __fadd_rn(da[j], 8388608.0f)), ib = __builtin_bit_cast(uint32_t, __fadd_rn(db[j], 8388608.0f));
        w[1 + j] = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(ib, ia, 0x05040100u));
    }
    int n = 0;
#pragma unroll
    for (int r0 = 0; r0 < 6; r0++)
#pragma unroll
        for (int c0 = r0; c0 < 6; c0++, n++) acc[n] = __builtin_amdgcn_sdot2(w[r0], w[c0], acc[n], false);
    // W^T r: f32 partial sums over the nodes a lane has in a tile, 64-bit fixed point from there on (fit_f32_to_fixed); every product enters its sum through a
    // fused multiply-add (a rounded product and an add are two instructions more per feature and node and no closer to the real sum)
    facc[0] = __fadd_rn(__fadd_rn(facc[0], ra), rb);
#pragma unroll
    for (int j = 0; j < 5; j++) {
        facc[1 + j] = __builtin_fmaf(da[j], ra, facc[1 + j]);
        facc[1 + j] = __builtin_fmaf(db[j], rb, facc[1 + j]);
    }
}

// Staging in two steps, so that a tile's global loads fly while the tile before it is worked on: the loads (a block cell and a halo value per
// step), and - behind half of the current tile's sums - the conversion and the LDS writes into the OTHER of two cell images.
// K2's cell geometry since round 4: 1 KiB per cell, pairs at gather_layout.inc's positions. (Rounds 2-3: 1040 bytes per cell - sixteen zero bytes behind each cell
// for the "never a node" entries - which puts neighbouring cells four banks apart: 216 LDS cycles for the 48 gather instructions of a cell in heap order, 199 with
// K2's pair positions, against 135 / 110 with cells 1 KiB apart, tools/lds_layout_search.py --k4. The zero those entries read is now heap node 0's halfword: the
// DC value, which no node ever gathers and the fit has no row for, staged as 0.)
constexpr int kFit2Slot = 1024;
constexpr int kFit2Image = kPredSlots * kFit2Slot; // 36 864 B; image 1 sits behind image 0, inside the gathers' 16-bit offset field
static_assert(kFit2Image + (kPredSide + kPredBlock) * kFit2Slot < 65536, "image + cell offset must fit a DS instruction's offset field");
// Round 3: the same sparse halo as K2 (k2_predict.hip, build_halo_list). A tile's 16 block cells are staged whole - two per wave, one behind each
// half of the current tile's sums - but of its 20 halo cells only the 902 values a 4 x 4 block ever gathers, two per thread: 45 KB of loads per
// tile instead of 72, a third of the conversions. Values land at K2's pair positions (gather_layout.inc; this kernel's waves gather exactly the node patterns
// of K2's roles), two bytes each, None as 0. (Round 3 tried those positions with cells 1040 bytes apart and found nothing - 37.2 / 55.1 us either way: at that
// distance they are worth 8 % of the gathers' LDS cycles, see kFit2Slot; with cells 1 KiB apart, bank conflicts went from 44 % to 18 % of the LDS-active cycles
// and the value pass from 33.0 to 30.6 us.)
struct Fit2Block {
    int4 lo, hi;
};
__device__ __forceinline__ int fit2_block_slot(int wave, int i) {
    const int c = 2 * wave + i;
    return (1 + c / kPredBlock) * kPredSide + 1 + (c % kPredBlock);
}
// C16 (round 5): `coefs` is a chain's compact plane - int16, None as 0 (k1_forward.hip, store_item) - and a lane's eight values are ONE 16-byte load whose four dwords
// are already the pairs the value pass stages (r.lo; r.hi unused).
template <bool C16 = false>
__device__ __forceinline__ void fit2_block_load(Fit2Block &r, const int32_t *__restrict__ coefs, const int32_t *s_slot_cell, int slot, int lane, bool skip) {
    r.lo = make_int4(0, 0, 0, 0), r.hi = r.lo;
    const int cell = skip ? -1 : s_slot_cell[slot];
    if (cell >= 0) {
        if (C16) {
            r.lo = *reinterpret_cast<const int4 *>(reinterpret_cast<const int16_t *>(coefs) + (size_t)cell * kCell + 8 * lane);
        } else {
            const int4 *src = reinterpret_cast<const int4 *>(coefs + (size_t)cell * kCell + 8 * lane);
            r.lo = src[0];
            r.hi = src[1];
        }
    }
}
// FLOATS = false (value pass): a staged halfword is the value's int16 (the packed operands of v_dot2c_i32_i16). FLOATS = true (width pass): the upper half of
// the value's float, K2's image format (fit2_pair_width); the conversion reads the low half of the dword sign-extended, so None (0x80000000) is 0 here too.
template <bool FLOATS>
__device__ __forceinline__ uint32_t fit2_pack(int first, int second) {
    if (!FLOATS) return __builtin_amdgcn_perm((uint32_t)second, (uint32_t)first, 0x05040100u);
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, (float)(short)second), __builtin_bit_cast(uint32_t, (float)(short)first), 0x07060302u);
}
template <bool CHECK, bool FLOATS, bool C16 = false>
__device__ __forceinline__ void fit2_block_commit(const Fit2Block &r, const int32_t *s_slot_cell, uint8_t *image, int slot, int lane, uint32_t stage_pos, uint32_t *range_counter) {
    auto pk = [](int lo16, int hi16) -> uint32_t { return fit2_pack<FLOATS>(lo16, hi16); };
    if (C16) { // a dword of the compact plane = two int16: the value pass stages it as it is, the width pass as two float halves (fit2_pack reads an operand's low half)
        static_assert(!(C16 && CHECK), "compact planes come from this library's forward kernel: nothing to check");
        auto pair = [&](int d) -> uint32_t { return FLOATS ? fit2_pack<true>(d, d >> 16) : (uint32_t)d; };
        uint8_t *dst = image + slot * kFit2Slot;
        *reinterpret_cast<uint32_t *>(dst + 4u * (stage_pos & 255u)) = pair(r.lo.x) & (lane == 0 ? 0xFFFF0000u : 0xFFFFFFFFu); // heap node 0: the cell's zero (see kFit2Slot)
        *reinterpret_cast<uint32_t *>(dst + 4u * ((stage_pos >> 8) & 255u)) = pair(r.lo.y);
        *reinterpret_cast<uint32_t *>(dst + 4u * ((stage_pos >> 16) & 255u)) = pair(r.lo.z);
        *reinterpret_cast<uint32_t *>(dst + 4u * (stage_pos >> 24)) = pair(r.lo.w);
        return;
    }
    const int4 lo = r.lo, hi = r.hi;
    if (CHECK && s_slot_cell[slot] >= 0) { // a None is 0x80000000: its low half stages as 0, and it is not an outlier
        const int v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) m |= v[j] == kNone ? 0u : ((uint32_t)v[j] + 256u) & 0xFFFFFE00u;
        if (__any(m != 0) && lane == 0) atomicAdd(range_counter, 1u);
    }
    uint8_t *dst = image + slot * kFit2Slot;
    *reinterpret_cast<uint32_t *>(dst + 4u * (stage_pos & 255u)) = pk(lo.x, lo.y) & (lane == 0 ? 0xFFFF0000u : 0xFFFFFFFFu); // heap node 0: the cell's zero (see kFit2Slot)
    *reinterpret_cast<uint32_t *>(dst + 4u * ((stage_pos >> 8) & 255u)) = pk(lo.z, lo.w);
    *reinterpret_cast<uint32_t *>(dst + 4u * ((stage_pos >> 16) & 255u)) = pk(hi.x, hi.y);
    *reinterpret_cast<uint32_t *>(dst + 4u * (stage_pos >> 24)) = pk(hi.z, hi.w);
}
// one halo value: entry = slot | heap << 8 | byte position inside the cell << 20 (build_halo_list; threads without an entry stage into the unused corner slot 0)
template <bool C16 = false>
__device__ __forceinline__ int fit2_halo_load(const int32_t *__restrict__ coefs, const int32_t *s_slot_cell, uint32_t entry, bool skip) {
    const int cell = skip ? -1 : s_slot_cell[entry & 63u];
    if (C16) return cell >= 0 ? (int)reinterpret_cast<const int16_t *>(coefs)[(size_t)cell * kCell + ((entry >> 8) & 511u)] : 0;
    return cell >= 0 ? coefs[(size_t)cell * kCell + ((entry >> 8) & 511u)] : 0;
}
template <bool FLOATS>
__device__ __forceinline__ void fit2_halo_commit(int v, uint8_t *image, uint32_t entry) {
    const uint16_t h = FLOATS ? (uint16_t)(__builtin_bit_cast(uint32_t, (float)(short)v) >> 16) : (uint16_t)v; // None = INT32_MIN: low half 0 (unwrap_or(0))
    *reinterpret_cast<uint16_t *>(image + (entry & 63u) * kFit2Slot + (entry >> 20)) = h; // (bits 20+: the value's byte position inside its cell)
}

// has_bits / interior_bits: bit (C >> 2) * kPredSide + (C & 3) = "the wave's block cell C holds a retained cell" / "... an interior one" (the tile's slot table as
// two wave-uniform words, read once per tile: as two LDS reads per cell - cell index, interior flag, each waited for before a branch - they were two of the
// three LDS round trips in a row that a cell's ~100 vector instructions had to hide behind, with four waves per SIMD).
template <int MODE, int IMG, int C>
__device__ __forceinline__ void fit2_cell(const int32_t *s_slot_cell, uint32_t has_bits, uint32_t interior_bits, const uint32_t *mask_word, uint32_t mask_shift, uint32_t keep, int slot0,
                                          const uint32_t (&addr)[2][6], uint32_t own_addr, const float (&vp)[6], int (&acc)[28], float (&facc)[6]) {
    constexpr int kBit = (C >> 2) * kPredSide + (C & 3);
    constexpr int kOff = IMG * kFit2Image + kBit * kFit2Slot;
    if (!((has_bits >> kBit) & 1u)) return;
    uint32_t own = *(__attribute__((address_space(3))) const uint32_t *)(uintptr_t)(own_addr + kOff) & keep; // (keep: 0 for the one lane whose pair is heap nodes 0 and 1)
    uint32_t one = keep & 0x00010001u;
    const bool boundary = !((interior_bits >> kBit) & 1u); // boundary cell: node p is bit (p & 31) of mask word p >> 5; a None node's row is all zeros
    if (MODE == 0) {
        uint32_t g[6];
        fit2_gather<kOff>(g, addr[0], addr[1]);
        if (boundary) {
            const int cell = __builtin_amdgcn_readfirstlane(s_slot_cell[slot0 + kBit]);
            const uint32_t bits = mask_word[(size_t)cell * 16] >> mask_shift;
            const uint32_t mask = ((bits & 1u) ? 0x0000FFFFu : 0u) | ((bits & 2u) ? 0xFFFF0000u : 0u);
#pragma unroll
            for (int k = 0; k < 6; k++) g[k] &= mask;
            own &= mask;
        }
        fit2_pair_value(g, own, acc);
    } else {
        uint32_t lo[6], hi[6];
        fit2_gather_floats<kOff>(lo, hi, addr[0], addr[1]);
        if (boundary) {
            const int cell = __builtin_amdgcn_readfirstlane(s_slot_cell[slot0 + kBit]);
            const uint32_t bits = mask_word[(size_t)cell * 16] >> mask_shift;
            const uint32_t first = (bits & 1u) ? 0xFFFFFFFFu : 0u, second = (bits & 2u) ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int k = 0; k < 6; k++) lo[k] &= first, hi[k] &= second;
            own &= (first & 0xFFFFu) | (second << 16), one &= (first & 0xFFFFu) | (second << 16);
        }
        fit2_pair_width(lo, hi, own, one, vp, acc, facc);
    }
    asm volatile("" ::: "memory"); // one cell's gathers at a time: the next cell's would cost 13 more registers
}
template <int MODE, int IMG, int C0>
__device__ __forceinline__ void fit2_cells(const int32_t *s_slot_cell, uint32_t has_bits, uint32_t interior_bits, const uint32_t *mask_word, uint32_t mask_shift, uint32_t keep, int slot0,
                                           const uint32_t (&addr)[2][6], uint32_t own_addr, const float (&vp)[6], int (&acc)[28], float (&facc)[6]) {
    fit2_cell<MODE, IMG, C0 + 0>(s_slot_cell, has_bits, interior_bits, mask_word, mask_shift, keep, slot0, addr, own_addr, vp, acc, facc);
    fit2_cell<MODE, IMG, C0 + 1>(s_slot_cell, has_bits, interior_bits, mask_word, mask_shift, keep, slot0, addr, own_addr, vp, acc, facc);
    fit2_cell<MODE, IMG, C0 + 2>(s_slot_cell, has_bits, interior_bits, mask_word, mask_shift, keep, slot0, addr, own_addr, vp, acc, facc);
    fit2_cell<MODE, IMG, C0 + 3>(s_slot_cell, has_bits, interior_bits, mask_word, mask_shift, keep, slot0, addr, own_addr, vp, acc, facc);
}

// A wave's sums on their way to the workgroup's: across the 16 lanes of a DPP row in registers - stage by stage over ALL sums, so that no add waits for the
// one before it (sum by sum, every one of the 4 x NI dependent DPP adds sat behind two wait states) - then one lane per row adds the row's total to the row's copy of the workgroup's sums in LDS as a
// 64-bit integer (four lanes of one ds_add_u64, four addresses; s_int: [4 rows][3 groups][28]). 32 bits hold a row for 16 tiles in both passes (value: |u| <= 256, a lane adds at most 8 cells x 2 x 256^2 =
// 2^20 per tile; width: features up to 511, four times that).
template <int NI>
__device__ __forceinline__ void fit2_wave_sums(int (&acc)[28], int group, int lane, unsigned long long (*s_int)[28]) {
    int r[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) r[k] = acc[k], acc[k] = 0;
#pragma unroll
    for (int k = 0; k < NI; k++) r[k] += __builtin_amdgcn_update_dpp(0, r[k], 0xB1, 0xF, 0xF, true);  // quad_perm(1,0,3,2)
#pragma unroll
    for (int k = 0; k < NI; k++) r[k] += __builtin_amdgcn_update_dpp(0, r[k], 0x4E, 0xF, 0xF, true);  // quad_perm(2,3,0,1)
#pragma unroll
    for (int k = 0; k < NI; k++) r[k] += __builtin_amdgcn_update_dpp(0, r[k], 0x141, 0xF, 0xF, true); // row_half_mirror
#pragma unroll
    for (int k = 0; k < NI; k++) r[k] += __builtin_amdgcn_update_dpp(0, r[k], 0x140, 0xF, 0xF, true); // row_mirror: every lane holds its row's total
    if ((lane & 15) == 0) { // (a copy of the sums per DPP row: with one address for the four lanes the compiler's atomic optimiser turns every add into a scalar loop over the lanes)
        unsigned long long *const mine = &s_int[(lane >> 4) * 3 + group][0];
#pragma unroll
        for (int k = 0; k < NI; k++) atomicAdd(mine + k, (unsigned long long)(long long)r[k]);
    }
}

// W^T r (the width fit's right-hand side) is a sum of f32 values: per (tile, lane) the products feature x residual of the lane's sixteen nodes in that tile,
// added in f32 in a fixed order - the reference's whole fit is f32 (context_modeling.rs:144-173). From there on the sum is INTEGER: every such f32 value
// (>= 0, < 2^24: features <= 511, residuals of fitted parameters <= ~1024, sixteen of them) becomes a 64-bit fixed-point number with kFitFixBits fraction bits
// by a function of the value alone (truncation below 2^-20: < 1e-6 per value), and integer adds commute. So the total does not depend on which workgroup walks which tile, on how many planes a launch
// holds or on who finishes when: the same bits from every entry point and every run, hence the same f32 parameters, buckets and bytes. (Rounds 1-3 added
// doubles in arrival order) A value of 2^24 or more - parameters from nowhere - is clamped to 2^24 and counted (below). 2^24 x 2^20 = 2^44 per partial sum:
// the 64-bit total cannot wrap below 2^19 saturated partial sums, and a single one already makes the call report out-of-range.
constexpr int kFitFixBits = 20; // a plane's total stays below 2^63 up to sums of 8.8e12 (a 16384^2 noise plane: ~1.6e12)
// `saturated` is raised for a value that does not fit (>= 2^24, infinite or NaN: value parameters from nowhere); it enters as 2^24 and the plane's out-of-range count
// reports it (FRI_HIP_ERR_OUT_OF_RANGE from the host forms), so a wrapped or meaningless W^T r never leaves silently.
__device__ __forceinline__ unsigned long long fit_f32_to_fixed(float v0, bool &saturated) {
    saturated = saturated || !(v0 < 16777216.0f);
    const float v = fminf(v0, 16777216.0f);                      // v_min_f32: NaN -> 2^24
    const float t = __builtin_truncf(v);                         // v >= 0
    const uint32_t hi = f32_to_u32_sat(t);                       // <= 2^24: a plane's 2 M partial sums stay below 2^45 before the fraction bits, 2^65 is never reached... see kFitFixBits
    const uint32_t lo = f32_to_u32_sat((v - t) * (float)(1u << kFitFixBits)); // exact difference, exact scaling, truncation below 2^-20 (only values < 16 have such bits)
    return ((unsigned long long)hi << kFitFixBits) + lo;
}

// One layer group's solve at the end of the sums kernel. A function of its own, not inlined: as part of the kernel's body its ~90 registers' worth of f64
// temporaries pushed the tile loop of the width pass (122 registers of 128) into spilling.
template <int MODE>
__device__ __attribute__((noinline)) void fit2_tail_solve(const long long *sums_int, const double *sums_dbl, unsigned long long rows, Solve6Work *w, float *params,
                                                          float *host_params) {
    float out[6];
    if (MODE == 0)
        fit_value_group(sums_int, out, *w);
    else
        fit_width_group(sums_int, sums_dbl, rows, out, *w);
#pragma unroll
    for (int k = 0; k < 6; k++) params[k] = out[k];
    if (host_params) {
#pragma unroll
        for (int k = 0; k < 6; k++) host_params[k] = out[k];
    }
}

// CHECK = false: the coefficients were written by this library's forward kernel earlier in the same call (kPredForwardOutput): every magnitude is <= 255 by
// construction, the staging does not look (40 of a block cell's ~75 vector instructions) and the out-of-range count stays 0.
template <int MODE, bool CHECK, bool C16 = false>
__global__ void __launch_bounds__(kPredThreads, 4) fit_accumulate_kernel2(const FitArgs a0) {
    constexpr int NI = MODE == 0 ? 28 : 21;
    const uint32_t plane = blockIdx.y;
    const int32_t *const coefs = C16 ? reinterpret_cast<const int32_t *>(reinterpret_cast<const int16_t *>(a0.coefs) + plane * a0.coef_stride) : a0.coefs + plane * a0.coef_stride;
    unsigned long long *const accp = a0.acc + (size_t)plane * kFitShards * kFitAccWords;           // copy 0: ticket, out-of-range count
    unsigned long long *const accs = accp + (size_t)(blockIdx.x % kFitShards) * kFitAccWords;       // this workgroup's copy of the sums
    PredictParams pp; // static indices only: a dynamic index into the argument struct would keep all of it in scratch memory
    if (a0.params)
        pp = a0.params[plane];
    else if (plane == 0)
        pp = a0.pp3[0];
    else if (plane == 1)
        pp = a0.pp3[1];
    else
        pp = a0.pp3[2];
    __shared__ __attribute__((aligned(16))) uint8_t s_cells[2 * kFit2Image]; // two cell images: tile i + 1 is staged while tile i is worked on
    __shared__ int32_t s_slot_cell[2][kPredSlots];
    __shared__ uint32_t s_slot_bits[2][2]; // per image: bit s = slot s holds a retained cell / an interior one (every block slot is below 32: fit2_cell)
    __shared__ uint32_t s_flag, s_range;
    __shared__ unsigned long long s_int[4 * 3][28]; // [DPP row][layer group][sum], see fit2_wave_sums; the hand-over adds the four rows
    __shared__ unsigned long long s_fix[16 * 3][6]; // [copy][layer group][sum]: W^T r in fixed point (kFitFixBits fraction bits): integer adds commute, so the sums do not depend on who arrives when
    __shared__ double s_dbl[3][6];              // (the totals, for the solve in the tail)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pg = wave & 3, half = wave >> 2;
    const int n0 = pg == 0 ? 128 + 2 * lane : pg == 1 ? 2 * lane : 256 + 4 * lane + 2 * (pg & 1);
    const int ablate = ablate_flags(a0.ablate);
    PredTileWalk walk(a0.n_tiles);
    // Two workgroups share a CU, and the one dispatched first keeps most of its issue slots (older waves win the arbitration): per-workgroup time
    // stamps showed the second one taking 8-9 us per tile beside 4.7 us for the first, then running its last tiles alone on the CU - at the lower
    // rate of a half-empty CU - long after the first had finished. As in K1 the split follows the dispatch rank: workgroups b and b + gridDim / 2
    // (the pair of one CU under in-order dispatch; a wrong guess costs speed, never correctness) walk ONE strided sequence of tiles, the older one
    // its first older_eighths / 8, the younger one the rest.
    if (a0.older_eighths > 0 && gridDim.x >= 16u && gridDim.x % 16u == 0u) {
        const uint32_t per_xcd = gridDim.x / 8u, pairs = per_xcd / 2u;                     // workgroups / pairs per XCD range
        const uint32_t xcd = blockIdx.x % 8u, in_xcd = blockIdx.x / 8u, pair = in_xcd % pairs, younger = in_xcd / pairs;
        const uint32_t lo = (uint32_t)((uint64_t)a0.n_tiles * xcd / 8u), hi = (uint32_t)((uint64_t)a0.n_tiles * (xcd + 1u) / 8u);
        const uint32_t n_pair = lo + pair < hi ? (hi - lo - pair + pairs - 1u) / pairs : 0u; // tiles of the pair: lo + pair + pairs * k
        const uint32_t n_older = min(n_pair, (n_pair * (uint32_t)a0.older_eighths + 3u) / 8u);
        walk.step = pairs;
        walk.first = lo + pair + (younger ? n_older * pairs : 0u);
        walk.end = younger ? hi : min(hi, lo + pair + n_older * pairs);
        if (walk.first > walk.end) walk.first = walk.end;
    }
    // Entry `tid` of a tile's slot list: uniform row address + the lane's 32-bit offset, made opaque so that it is formed where it is used - as a loop
    // invariant it is a 64-bit per-lane pointer, two registers the width pass does not have: spilled, and reloaded once per tile behind an s_waitcnt vmcnt(0),
    // i.e. behind the staging loads wave 0 had just issued.
    auto slot_entry = [&](uint32_t t) {
        uint32_t off = (uint32_t)tid * 4u;
        asm volatile("" : "+v"(off));
        return *reinterpret_cast<const int32_t *>(reinterpret_cast<const uint8_t *>(a0.pred_slots + (size_t)t * kPredSlots) + off);
    };
    // EVERYTHING the prologue reads from global memory that does not depend on other loads is requested here, in one go: the lane's neighbour offsets and pair
    // positions, its halo entries, the slot lists of the workgroup's first two tiles. Read where they were first used they were three dependent round trips in
    // front of the first tile's coefficients (offsets; halo entries; the first slot list, waited for on the spot): the prologue's 4.9 us are 3.9 now.
    const u32x4 off_a = reinterpret_cast<const u32x4 *>(a0.pred_off)[n0], off_b = reinterpret_cast<const u32x4 *>(a0.pred_off)[n0 + 1];
    const uint32_t own_pos = a0.pair_pos[n0 >> 1];
    // the thread's two halo values per tile; entries >= 1024 - 122 of the list stage into the unused corner slot, where two of them may meet: harmless
    const uint32_t halo_e0 = a0.halo_list[tid], halo_e1 = a0.halo_list[tid + kPredThreads];
    // where the four pairs a lane stages of a block cell (heap nodes 8 lane .. 8 lane + 7) live inside the cell: dword positions, one byte each
    const uint32_t stage_pos = (uint32_t)a0.pair_pos[4 * lane] | (uint32_t)a0.pair_pos[4 * lane + 1] << 8 | (uint32_t)a0.pair_pos[4 * lane + 2] << 16 | (uint32_t)a0.pair_pos[4 * lane + 3] << 24;
    int first_raw = -1, next_raw = -1; // thread t < 36: slot t of the workgroup's first tile / of the tile after the current one, requested a tile ahead
    if (tid < kPredSlots && walk.first < walk.end) first_raw = slot_entry(walk.first);
    if (tid < kPredSlots && walk.first + walk.step < walk.end) next_raw = slot_entry(walk.first + walk.step);
    __builtin_amdgcn_sched_barrier(0); // (all of the above is in flight before anything below looks at a loaded value)
    if (tid < 4 * 3 * 28) (&s_int[0][0])[tid] = 0;
    if (tid < 16 * 18) (&s_fix[0][0])[tid] = 0ull;
    if (tid == 0) s_range = 0;
    trace_stamp(a0.trace, blockIdx.x, 0, tid);

    // The wave's pair of nodes per lane: wave & 3 = 0: level 7 (128 + 2 lane, + 1), 1: levels 0..6 (2 lane, + 1), 2 and 3: level 8 (256 + 4 lane +
    // {0, 1} and {2, 3}) - one layer group per wave, so the value parameters are scalars and a wave's sums have one destination.
    // Loop invariants of the lane: LDS addresses (first of the wave's eight cells) of the six neighbours of its two nodes, and of the pair itself.
    const int group = pg == 0 ? 1 : pg == 1 ? 2 : 0;
    const int slot0 = (1 + 2 * half) * kPredSide + 1; // first block cell of the wave's two block rows
    const uint32_t cells_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s_cells;
    uint32_t addr[2][6];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const u32x4 o = j == 0 ? off_a : off_b;
        const uint32_t rel[3] = {o.x, o.y, o.z};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            // K2's table: byte offset from the own cell, 0x7FFF = never a node: the own cell's first halfword (heap node 0, staged as 0)
            const int r = (int)(short)((rel[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
            addr[j][k] = cells_lds + (uint32_t)(slot0 * kFit2Slot + (r == 0x7FFF ? 0 : r));
        }
    }
    uint32_t own_addr = cells_lds + (uint32_t)(slot0 * kFit2Slot) + 4u * own_pos;
    if (n0 == 0) { // heap nodes 0 and 1 are not rows of the fit: that lane gathers the zero at the head of its cell, and its own pair is masked (keep)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int k = 0; k < 6; k++) addr[j][k] = cells_lds + (uint32_t)(slot0 * kFit2Slot);
    }
    // the pair's two bits of a boundary cell's mask (node p is bit p & 31 of word p >> 5); heap index 0 and 1 are coded by the LF predictor
    // and are not rows of the fit
    const uint32_t *const mask_word = a0.valid_mask + (n0 >> 5);
    const uint32_t mask_shift = n0 & 31;
    const uint32_t keep = n0 == 0 ? 0u : 0xFFFFFFFFu;
    float vp[6];
#pragma unroll
    for (int k = 0; k < 6; k++) vp[k] = group == 0 ? pp.value[0][k] : group == 1 ? pp.value[1][k] : pp.value[2][k]; // selects, not a dynamic index
    int acc[28];
    unsigned long long fx[6]; // W^T r of this lane, fixed point (kFitFixBits fraction bits)
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) fx[k] = 0ull;
    bool fx_saturated = false; // a W^T r partial sum of this lane did not fit the fixed-point format (fit_f32_to_fixed)
    int tiles_since_flush = 0, trace_it = 0;
    const int block_a = fit2_block_slot(wave, 0), block_b = fit2_block_slot(wave, 1);

    uint32_t tile = walk.first;
    static_assert((kPredBlock * kPredSide + kPredBlock) < 32 && kPredSlots <= 64, "the block slots' bits fit one word; the slot table is written by one wave");
    auto slot_table = [&](int img, int raw) { // (threads 0..kPredSlots-1: lanes of wave 0, so the ballots are the table)
        s_slot_cell[img][tid] = pred_slot_cell(raw);
        const unsigned long long has = __ballot(pred_slot_cell(raw) >= 0), interior = __ballot(pred_slot_interior(raw));
        if (tid == 0) s_slot_bits[img][0] = (uint32_t)has, s_slot_bits[img][1] = (uint32_t)interior;
    };
    // One tile with image IMG current: publish the next tile's slot table, request its coefficients, do this tile's sums, convert and
    // write the next tile into the other image. Two barriers per tile.
#define FRI_FIT2_PHASE(IMG)                                                                                                                       \
    {                                                                                                                                            \
        /* behind the last tile an empty table: the staging steps run unconditionally (a conditional definition would make the staged */       \
        /* registers loop-carried - 80 of them live through the whole loop) and find no cell */                                                  \
        if (tid < kPredSlots) slot_table(IMG ^ 1, tile + walk.step < walk.end ? next_raw : -1);                                                  \
        __syncthreads();                                                                                                                         \
        float facc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                                                                                          \
        const uint32_t has_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_slot_bits[IMG][0]) >> slot0;                                   \
        const uint32_t interior_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_slot_bits[IMG][1]) >> slot0;                              \
        {                                                                                                                                        \
            Fit2Block st;                                                                                                                        \
            fit2_block_load<C16>(st, coefs, s_slot_cell[IMG ^ 1], block_a, lane, ablate & 2);                                                         \
            const int hv = fit2_halo_load<C16>(coefs, s_slot_cell[IMG ^ 1], halo_e0, ablate & 2);                                                     \
            if (tid < kPredSlots && tile + 2 * walk.step < walk.end) next_raw = slot_entry(tile + 2 * walk.step);                                 \
            if (!(ablate & 1)) fit2_cells<MODE, IMG, 0>(s_slot_cell[IMG], has_bits, interior_bits, mask_word, mask_shift, keep, slot0, addr, own_addr, vp, acc, facc); \
            fit2_block_commit<CHECK, MODE == 1, C16>(st, s_slot_cell[IMG ^ 1], s_cells + (IMG ^ 1) * kFit2Image, block_a, lane, stage_pos, &s_range);                              \
            fit2_halo_commit<MODE == 1>(hv, s_cells + (IMG ^ 1) * kFit2Image, halo_e0);                                                                     \
        }                                                                                                                                        \
        {                                                                                                                                        \
            Fit2Block st;                                                                                                                        \
            fit2_block_load<C16>(st, coefs, s_slot_cell[IMG ^ 1], block_b, lane, ablate & 2);                                                         \
            const int hv = fit2_halo_load<C16>(coefs, s_slot_cell[IMG ^ 1], halo_e1, ablate & 2);                                                     \
            if (!(ablate & 1)) fit2_cells<MODE, IMG, 4>(s_slot_cell[IMG], has_bits, interior_bits, mask_word, mask_shift, keep, slot0, addr, own_addr, vp, acc, facc); \
            fit2_block_commit<CHECK, MODE == 1, C16>(st, s_slot_cell[IMG ^ 1], s_cells + (IMG ^ 1) * kFit2Image, block_b, lane, stage_pos, &s_range);                              \
            fit2_halo_commit<MODE == 1>(hv, s_cells + (IMG ^ 1) * kFit2Image, halo_e1);                                                                     \
        }                                                                                                                                        \
        if (MODE == 1) {                                                                                                                         \
            _Pragma("unroll") for (int k = 0; k < 6; k++) fx[k] += fit_f32_to_fixed(facc[k], fx_saturated);                                     \
        }                                                                                                                                        \
        if (++tiles_since_flush >= 16) { /* 16 nodes x 256^2 x 2 per tile and lane: a row of 16 lanes stays below 2^31 for 16 tiles */            \
            fit2_wave_sums<NI>(acc, group, lane, s_int);                                                                                         \
            tiles_since_flush = 0;                                                                                                               \
        }                                                                                                                                        \
        __syncthreads();                                                                                                                         \
        trace_stamp(a0.trace, blockIdx.x, 2 + trace_it++, tid);                                                                                  \
        tile += walk.step;                                                                                                                       \
    }
    if (tile < walk.end) {
        if (tid < kPredSlots) slot_table(0, first_raw);
        __syncthreads();
        {
            Fit2Block sa, sb;
            fit2_block_load<C16>(sa, coefs, s_slot_cell[0], block_a, lane, false);
            fit2_block_load<C16>(sb, coefs, s_slot_cell[0], block_b, lane, false);
            const int h0 = fit2_halo_load<C16>(coefs, s_slot_cell[0], halo_e0, false), h1 = fit2_halo_load<C16>(coefs, s_slot_cell[0], halo_e1, false);
            fit2_block_commit<CHECK, MODE == 1, C16>(sa, s_slot_cell[0], s_cells, block_a, lane, stage_pos, &s_range);
            fit2_block_commit<CHECK, MODE == 1, C16>(sb, s_slot_cell[0], s_cells, block_b, lane, stage_pos, &s_range);
            fit2_halo_commit<MODE == 1>(h0, s_cells, halo_e0);
            fit2_halo_commit<MODE == 1>(h1, s_cells, halo_e1);
        }
        __syncthreads();
        trace_stamp(a0.trace, blockIdx.x, 1, tid);
        while (true) {
            FRI_FIT2_PHASE(0)
            if (tile >= walk.end) break;
            FRI_FIT2_PHASE(1)
            if (tile >= walk.end) break;
        }
    }
#undef FRI_FIT2_PHASE
    // A wave's sums go to the workgroup's through fit2_wave_sums (DPP row totals, one LDS add per row). (Round 1's kernel parked all lanes' sums in LDS and let
    // one thread per sum walk 32-64 of them: ~6 us per workgroup; round 2 read the row totals out through readlane into scalar adds: ~250 instructions per wave.)
    trace_stamp(a0.trace, blockIdx.x, 13, tid);
    fit2_wave_sums<NI>(acc, group, lane, s_int);
    if (MODE == 1) { // a lane's six fixed-point sums go to copy (lane & 15) of the workgroup's: six LDS adds (four lanes per address and instruction) instead of a 72-step shuffle tree
        unsigned long long *const mine = &s_fix[(lane & 15) * 3 + group][0];
#pragma unroll
        for (int k = 0; k < 6; k++) atomicAdd(mine + k, fx[k]);
        if (__any(fx_saturated) && lane == 0) atomicAdd(&s_range, 1u); // counted with the staging's out-of-range values: the sums are not to be trusted
    }
    __syncthreads();
    // hand-over: add into this workgroup's copy of the plane's accumulator, draw a ticket, the last workgroup sums the copies, moves the totals out
    // and re-zeroes
    if (tid < 3 * NI) {
        const int gg = tid / NI, k = tid % NI;
        __hip_atomic_fetch_add(accs + tid, (s_int[gg][k] + s_int[3 + gg][k]) + (s_int[6 + gg][k] + s_int[9 + gg][k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (MODE == 1 && tid < 18) {
        unsigned long long v = 0;
#pragma unroll
        for (int c = 0; c < 16; c++) v += (&s_fix[0][0])[c * 18 + tid];
        __hip_atomic_fetch_add(accs + kFitAccDbl + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0 && s_range) __hip_atomic_fetch_add(accp + kFitAccRange, (unsigned long long)s_range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_for_own_memory_ops_then_barrier(); // every wave has waited for its adds before the ticket is drawn
    if (tid == 0) s_flag = __hip_atomic_fetch_add(accp + kFitAccTicket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    trace_exit(a0.trace, blockIdx.x, tid);
    if (s_flag == 0) return;
    // one wave moves everything out
    if (tid >= 64) return;
    unsigned long long *const out_int = (MODE == 0 ? a0.gram : a0.wtw) + (size_t)plane * 3 * NI;
    for (int i = tid; i < 3 * NI; i += 64) {
        unsigned long long part[kFitShards], sum = 0;
#pragma unroll
        for (uint32_t sh = 0; sh < kFitShards; sh++) part[sh] = __hip_atomic_load(accp + sh * kFitAccWords + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // all in flight together
#pragma unroll
        for (uint32_t sh = 0; sh < kFitShards; sh++) {
            sum += part[sh];
            __hip_atomic_store(accp + sh * kFitAccWords + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        out_int[i] = sum;
        s_int[i / NI][i % NI] = sum; // (for the solve below)
    }
    if (MODE == 1 && tid < 18) {
        unsigned long long part[kFitShards], total = 0;
#pragma unroll
        for (uint32_t sh = 0; sh < kFitShards; sh++) part[sh] = __hip_atomic_load(accp + sh * kFitAccWords + kFitAccDbl + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (uint32_t sh = 0; sh < kFitShards; sh++) {
            total += part[sh];
            __hip_atomic_store(accp + sh * kFitAccWords + kFitAccDbl + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double sum = (double)total * (1.0 / (double)(1ull << kFitFixBits)); // (unsigned: a sum of non-negative terms)
        a0.wtr[(size_t)plane * 18 + tid] = sum;
        (&s_dbl[0][0])[tid] = sum;
    }
    if (tid == 0) {
        const unsigned long long r = __hip_atomic_exchange(accp + kFitAccRange, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a0.out_range) a0.out_range[plane] = r;
        if (a0.host_range) a0.host_range[plane] = r;
        __hip_atomic_store(accp + kFitAccTicket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ContextModeler::optimize_value_prediction / optimize_width_prediction (context_modeling.rs:144-202) right here: lane g solves layer group g from the
    // totals this wave has just parked in LDS, its workspace a piece of the (now idle) cell images. As kernels of their own between the sums kernels and
    // the scan the two solves were 12 us each of the 175 us chain (launch, three lanes walking arrays in scratch memory, drain).
    if (!a0.solve_params) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tid < 3) {
        static_assert(3 * sizeof(Solve6Work) <= 2 * kFit2Image, "the solve's workspace lives in the cell images");
        const unsigned long long rows = tid == 0 ? a0.rows[0] : tid == 1 ? a0.rows[1] : a0.rows[2]; // (selects: no dynamic index into the argument struct)
        const size_t at = (size_t)plane * (sizeof(PredictParams) / sizeof(float)) + (MODE ? 18 : 0) + tid * 6;
        fit2_tail_solve<MODE>(reinterpret_cast<const long long *>(s_int[tid]), s_dbl[tid], rows, reinterpret_cast<Solve6Work *>(s_cells) + tid, a0.solve_params + at,
                              a0.host_params ? a0.host_params + at : nullptr);
    }
}

// The 6 x 6 solves of the fit on the device (ContextModeler::optimize_value_prediction / optimize_width_prediction,
// context_modeling.rs:144-202, behind prediction.rs:232-235): one thread per (plane, layer group) turns the sums a fit_accumulate launch
// left in device memory into that group's six parameters, written into the device parameter array the next kernel of the chain reads
// (the width pass of the fit, then K2) - so the chain K1 -> sums -> solve -> sums -> solve -> K2 is enqueued without the host in between.
// The arithmetic is solve6.hpp's, shared with the host entry points. The chain itself solves in the tail of fit_accumulate_kernel2 (FitArgs::solve_params);
// this kernel serves the entry points that take sums from the caller (fri_hip_fit_value_params_batch_dev, fri_hip_fit_width_params_batch_dev).
struct SolveArgs {
    const unsigned long long *sums_int; // [n_planes][3][28] (MODE 0) / [n_planes][3][21] (MODE 1)
    const double *sums_dbl;             // [n_planes][3][6] (MODE 1)
    float *params;                      // PredictParams[n_planes]: MODE 0 writes .value, MODE 1 .width
    uint32_t n_planes;
    unsigned long long rows[3];         // MODE 1: heights of the reference's matrices (F * {256, 128, 128})
    // for callers that want the parameters on the host (fri_hip_encode_image_dev with fit): a second copy into pinned host memory, written by the
    // solving threads themselves (no copy command behind the kernel), and the planes' out-of-range counts next to it
    float *host_params;                 // PredictParams[n_planes] in mapped host memory, or NULL
    const unsigned long long *range;    // [n_planes] device, or NULL
    unsigned long long *host_range;     // [n_planes] mapped host memory, or NULL
};
constexpr uint32_t kSolveThreads = 32; // a thread's workspace is 1.4 KB of LDS
template <int MODE>
__global__ void __launch_bounds__(kSolveThreads) fit_solve_kernel(const SolveArgs a) {
    __shared__ Solve6Work s_work[kSolveThreads];
    Solve6Work &w = s_work[threadIdx.x];
    const uint32_t t = blockIdx.x * kSolveThreads + threadIdx.x;
    if (t >= a.n_planes * 3u) return;
    const uint32_t plane = t / 3u, g = t % 3u;
    float *out = a.params + (size_t)plane * (sizeof(PredictParams) / sizeof(float)) + (MODE ? 18 : 0) + g * 6;
    if (MODE == 0) {
        fit_value_group(reinterpret_cast<const long long *>(a.sums_int) + ((size_t)plane * 3 + g) * 28, out, w);
    } else {
        const unsigned long long rows = g == 0 ? a.rows[0] : g == 1 ? a.rows[1] : a.rows[2]; // (selects: no dynamic index into the argument struct)
        fit_width_group(reinterpret_cast<const long long *>(a.sums_int) + ((size_t)plane * 3 + g) * 21, a.sums_dbl + ((size_t)plane * 3 + g) * 6, rows, out, w);
    }
    if (a.host_params) {
        float *h = a.host_params + (out - a.params);
#pragma unroll
        for (int k = 0; k < 6; k++) h[k] = out[k];
        if (g == 0 && a.range && a.host_range) a.host_range[plane] = a.range[plane];
    }
}

} // namespace

hipError_t launch_fit_solve(int mode, uint32_t n_planes, const unsigned long long *sums_int, const double *sums_dbl, const unsigned long long rows[3], float *params,
                            hipStream_t stream, float *host_params, const unsigned long long *range, unsigned long long *host_range) {
    if (!n_planes || !sums_int || !params || (mode == 1 && !sums_dbl)) return hipErrorInvalidValue;
    SolveArgs a{};
    a.sums_int = sums_int;
    a.sums_dbl = sums_dbl;
    a.params = params;
    a.n_planes = n_planes;
    a.host_params = host_params, a.range = range, a.host_range = host_range;
    for (int g = 0; g < 3; g++) a.rows[g] = rows ? rows[g] : 0;
    const uint32_t blocks = (n_planes * 3u + kSolveThreads - 1u) / kSolveThreads;
    (void)hipGetLastError();
    if (mode == 0)
        hipLaunchKernelGGL(fit_solve_kernel<0>, dim3(blocks), dim3(kSolveThreads), 0, stream, a);
    else
        hipLaunchKernelGGL(fit_solve_kernel<1>, dim3(blocks), dim3(kSolveThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_fit_accumulate(const DevicePlan &p, unsigned long long *acc, int mode, const PredBatch &b, unsigned long long *sums_int, double *sums_dbl,
                                 unsigned long long *out_of_range, hipStream_t stream, const FitSolve *solve, int trust) {
    if (!acc || !b.n_planes || b.n_planes > 65535u) return hipErrorInvalidValue;
    FitArgs a{};
    if (b.coefs16 && trust != kPredForwardOutput) return hipErrorInvalidValue; // compact planes exist inside this library's chains only
    a.coefs = b.coefs16 ? reinterpret_cast<const int32_t *>(b.coefs16) : b.coefs;
    a.coef_stride = b.coef_stride;
    a.params = b.params;
    for (int k = 0; k < 3; k++) a.pp3[k] = b.pp[k];
    a.pred_slots = p.pred_slots;
    a.nbr_table = p.nbr_table;
    a.pred_off = p.gather_off;
    a.pair_pos = p.pair_pos;
    a.halo_list = p.halo_list;
    a.interior = p.interior;
    a.valid_mask = p.valid_mask;
    a.n_tiles = p.n_pred_tiles;
    a.acc = acc;
    a.gram = sums_int;
    a.wtw = sums_int;
    a.wtr = sums_dbl;
    a.out_range = out_of_range;
    if (solve) {
        if (!solve->params) return hipErrorInvalidValue;
        a.solve_params = solve->params, a.host_params = solve->host_params, a.host_range = solve->host_range;
        for (int g = 0; g < 3; g++) a.rows[g] = solve->rows[g];
    }
    a.ablate = p.k4_ablate;
    a.trace = p.trace;
    a.older_eighths = b.n_planes == 1 && p.hist_blocks <= p.n_pred_tiles ? p.k4_older_eighths : 0; // a plane on the whole machine: two co-resident workgroups per CU
    uint32_t blocks = p.n_pred_tiles < p.hist_blocks ? p.n_pred_tiles : p.hist_blocks;
    if (b.n_planes > 1) { // as in launch_predict_histogram: a plane on an eighth of the machine, eight planes side by side
        const uint32_t share = (p.n_pred_tiles + 7) / 8, eighth = p.hist_blocks / 8 ? p.hist_blocks / 8 : 1;
        blocks = share < eighth ? eighth : share;
        if (blocks > p.hist_blocks) blocks = p.hist_blocks;
        if (blocks > p.n_pred_tiles) blocks = p.n_pred_tiles;
    }
    if (!blocks) blocks = 1;
    (void)hipGetLastError(); // the check behind the launch must not pick up an error an earlier, unrelated call left behind
    const bool check = trust != kPredForwardOutput;
    void (*kern)(const FitArgs) = mode == 0 ? (check ? fit_accumulate_kernel2<0, true> : fit_accumulate_kernel2<0, false>) : (check ? fit_accumulate_kernel2<1, true> : fit_accumulate_kernel2<1, false>);
    if (b.coefs16) kern = mode == 0 ? fit_accumulate_kernel2<0, false, true> : fit_accumulate_kernel2<1, false, true>;
    hipLaunchKernelGGL(kern, dim3(blocks, b.n_planes), dim3(kPredThreads), 0, stream, a);
    return hipGetLastError();
}

} // namespace fri
