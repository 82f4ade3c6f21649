// k2_predict.hip -- K2 predict_histogram: 6-neighbour gather + context bucket + prediction + ANS symbol histogram for one
// channel plane (context_modeling.rs:25-77; stages/prediction.rs:86-207, 237-298).
#include "gather_common.hpp"

namespace fri {
namespace {

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
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    // stage tile 0 straight into image 0: all of a wave's (up to three) cells are requested before the first is converted - one
    // global round trip instead of three in a row (the prologue took 4.7 us of the kernel's 54)
    {
        int cell0[kPred2Stage];
        i32x4 lo0[kPred2Stage], hi0[kPred2Stage];
        uint32_t mask0[kPred2Stage];
#pragma unroll
        for (int j0 = 0; j0 < kPred2Stage; j0++) {
            const int sl = min(wave + kPred2Waves * j0, kPredSlots - 1);
            cell0[j0] = pred_slot_cell(__builtin_amdgcn_readfirstlane(s_ring[sl]));
            const i32x4 *src = reinterpret_cast<const i32x4 *>(a.coefs + (size_t)max(cell0[j0], 0) * kCell + 8 * lane);
            lo0[j0] = src[0], hi0[j0] = src[1];
            mask0[j0] = a.valid_mask[(size_t)max(cell0[j0], 0) * 16 + (lane & 15)];
        }
#pragma unroll
        for (int j0 = 0; j0 < kPred2Stage; j0++) {
            const int sl = wave + kPred2Waves * j0;
            if (sl < kPredSlots) {
                i32x4 lo = lo0[j0], hi = hi0[j0];
                if (lane < 16) s_masks[sl * 16 + lane] = mask0[j0];
                if (cell0[j0] < 0) {
                    lo = hi = i32x4{0, 0, 0, 0};
                } else if (pred_is_block_slot(sl)) {
                    const int v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    const uint32_t n = pred_count_outliers(v);
                    if (n) atomicAdd(&s_hist[kHistBins], n);
                }
                uint8_t *dst = s_cells + sl * kSlotStride;
                *reinterpret_cast<u32x4 *>(dst + 16 * lane) = u32x4{__builtin_amdgcn_perm((uint32_t)lo.y, (uint32_t)lo.x, 0x05040100u), __builtin_amdgcn_perm((uint32_t)lo.w, (uint32_t)lo.z, 0x05040100u),
                                                                    __builtin_amdgcn_perm((uint32_t)hi.y, (uint32_t)hi.x, 0x05040100u), __builtin_amdgcn_perm((uint32_t)hi.w, (uint32_t)hi.z, 0x05040100u)};
                if (lane == 0) *reinterpret_cast<u32x4 *>(dst + 1024) = u32x4{0u, 0u, 0u, 0u};
            }
        }
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


} // namespace

void build_pred_offsets(const uint16_t *nbr_table, uint32_t *out) {
    for (int p = 0; p < kCell; p++) {
        uint32_t row[3], o[3];
        std::memcpy(row, nbr_table + p * 6, sizeof(row));
        pred_offsets_from_row(row, o);
        out[4 * p] = o[0], out[4 * p + 1] = o[1], out[4 * p + 2] = o[2], out[4 * p + 3] = 0;
    }
}

hipError_t launch_predict_histogram(const DevicePlan &p, uint32_t acc_slot, const int32_t *coefs_channel, const PredictParams &pp, uint8_t *bucket,
                                    int32_t *prediction, uint32_t *hist, unsigned long long *n_oob, hipStream_t stream) {
    if (!p.pred_acc || acc_slot >= kPredAccRing) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    PredArgs a{};
    a.acc = p.pred_acc + (size_t)acc_slot * kPredAccWords;
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

} // namespace fri
