// k2_predict.hip -- K2 predict_histogram: 6-neighbour gather + context bucket + prediction + ANS symbol histogram for one
// channel plane (context_modeling.rs:25-77; stages/prediction.rs:86-207, 237-298).
#include "gather_common.hpp"
#include "solve6.hpp"

#include <algorithm>
#include <vector>

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
    const uint32_t *pred_off;  // [512][4] packed neighbour offsets of every node: halfwords from the own slot at stride 1040 for the earlier kernels
                               // (build_pred_offsets), bytes from the own slot in the permuted 1 KiB layout for kernel3 (build_gather_tables)
    const uint16_t *pair_pos;  // [256] dword position of halfword pair q inside a 1 KiB slot (gather_layout.inc)
    const uint16_t *heap_of_pos; // [512] its inverse per halfword: heap index stored at halfword position i
    const uint32_t *halo_list; // [kP3Threads] the halo values a tile needs, one per thread (build_halo_list)
    uint32_t *inexact;         // (set per plane by the kernel: acc + kAccInexact) raised when a staged value does not fit the LDS image; the exact kernel then redoes the plane
    // planes of a batch (grid.y): plane k reads coefs + k * coef_stride, writes bucket / prediction + k * out_stride, hist + k * 10 * 1024, n_oob + k,
    // hands over through acc + k * kPredAccWords and takes its parameters from params[k] (NULL: pp)
    size_t coef_stride, out_stride;
    const PredictParams *params;
    int32_t ablate;            // timing-only (tuning build, FRI_HIP_K2_ABLATE): 1 = no predict phase (zeros are stored), 2 = tiles after the first are not staged,
                               // 4 = no histogram update, 32 = no output stores (8 = no bucket-table read and 16 = no gathers are retired: see p3_half)
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
    uint16_t *words;      // predict_histogram_kernel3<., true>: bucket << 10 | symbol per node, [n_planes] planes out_stride apart, INSTEAD of bucket / prediction
    int32_t trusted;      // the caller vouches for |coefficient| <= 256 (this library's forward kernel wrote them) and enqueues no exact kernel: a plane that
                          // raises `inexact` all the same reports n_oob = ~0 (an error the host maps to FRI_HIP_ERR_OUT_OF_RANGE) and lowers the flag itself
    PredictParams pp;     // this plane's parameters (filled per plane inside the kernel)
    PredictParams pp3[3]; // plane k < 3 of a launch without a params array
};

// Histogram hand-over without a memset in front of the kernel (two fill kernels cost ~6 us per call): every workgroup adds
// its LDS table into the plan's accumulator, then takes a ticket; the workgroup that draws the last ticket moves the totals to
// the caller's arrays with atomic exchanges, which leaves the accumulator zero for the next launch.
constexpr int kAccOob = kHistBins, kAccTicket = kHistBins + 2, kAccInexact = kHistBins + 4; // + the exact kernel's ticket at kAccInexact + 1
static_assert(kHistBins + 8 == (int)kPredAccWords, "accumulator layout");
__device__ __forceinline__ void pred_hand_over(const PredArgs &a, const uint32_t *s_hist, uint32_t *s_flag, int tid, int n_threads) {
    // (the counts go to one of kPredShards copies of the accumulator: all workgroups of a launch adding into the same hot bins serialise at ~12 ns
    // per add and address at the memory side; the last workgroup sums the copies. Copy 0 also holds the counters behind the bins.)
    uint32_t *const acc_s = a.acc + (size_t)(blockIdx.x % kPredShards) * kPredAccWords;
    for (int i = tid; i < kHistBins; i += n_threads) {
        const uint32_t c = s_hist[i];
        if (c) __hip_atomic_fetch_add(acc_s + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0 && s_hist[kHistBins])
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(a.acc + kAccOob), (unsigned long long)s_hist[kHistBins], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // Order without fences: an agent-scope fence on this multi-XCD part writes back and invalidates the whole L2 (measured:
    // +80 us per launch). The adds above are device-scope atomics, executed at the coherence point and acknowledged through
    // vmcnt: every wave waits for its own vmcnt(0) - explicitly, see wait_for_own_memory_ops_then_barrier - so all of this workgroup's
    // adds are performed before thread 0 draws the ticket. The last workgroup then reads with device-scope loads, which do not hit a
    // stale L2 line.
    wait_for_own_memory_ops_then_barrier();
    if (tid == 0) *s_flag = __hip_atomic_fetch_add(a.acc + kAccTicket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (*s_flag == 0) return;
    // all other workgroups have finished (their adds precede their tickets): plain coherent loads, all in flight together
    // (an atomic exchange per bin, one after the other, took 30-60 us), then the zeros for the next launch
    static_assert(kHistBins % 512 == 0, "unrolled by 512-thread strides");
    // (a plane kernel3 could not represent - a.inexact raised - hands over an all-zero histogram: the exact kernel behind it adds the real one)
    const bool inexact = a.inexact && __hip_atomic_load(a.inexact, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (n_threads == 1024) {
        uint32_t v[kHistBins / 1024][kPredShards];
#pragma unroll
        for (int k = 0; k < kHistBins / 1024; k++)
#pragma unroll
            for (uint32_t sh = 0; sh < kPredShards; sh++) v[k][sh] = __hip_atomic_load(a.acc + sh * kPredAccWords + tid + 1024 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < kHistBins / 1024; k++) {
            uint32_t sum = 0;
#pragma unroll
            for (uint32_t sh = 0; sh < kPredShards; sh++) {
                sum += v[k][sh];
                __hip_atomic_store(a.acc + sh * kPredAccWords + tid + 1024 * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            a.hist[tid + 1024 * k] = inexact ? 0u : sum;
        }
    } else { // (the 512-thread kernel of round 1, tuning builds only: it adds into copy blockIdx % kPredShards like everyone else)
        for (int k = 0; k < kHistBins / 512; k++) {
            uint32_t sum = 0;
            for (uint32_t sh = 0; sh < kPredShards; sh++) {
                sum += __hip_atomic_load(a.acc + sh * kPredAccWords + tid + 512 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.acc + sh * kPredAccWords + tid + 512 * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            a.hist[tid + 512 * k] = sum;
        }
    }
    if (tid == 0) {
        const unsigned long long oob = __hip_atomic_exchange(reinterpret_cast<unsigned long long *>(a.acc + kAccOob), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *a.n_oob = inexact ? (a.trusted ? ~0ull : 0ull) : oob;
        if (inexact && a.trusted) __hip_atomic_store(a.inexact, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // no exact kernel follows to lower it
        __hip_atomic_store(a.acc + kAccTicket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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



// ---------------------------------------------------------------------------------------------------------------------
// K2, third form: the kernel the product launches. What the counters and microbenchmarks of round 2 said about the pipelined
// kernel above (tools/micro/valu_rate.hip, tools/lds_layout_search.py, profiles/r01_k2_pmc_summary.txt):
//  * a random 16-bit LDS gather costs ~7 LDS cycles per wave-instruction (32-lane groups on 32 banks), the structured gathers
//    of this kernel 3.7 in heap order - against 2 when conflict-free - and the LDS pipe was as busy as the vector ALU;
//  * v_cvt_f32_i32 and every integer instruction hold the SIMD twice as long as v_mul_f32 / v_add_f32; six conversions and
//    six address computations per node were a third of the instruction stream;
//  * 16 store instructions per cell (8 of them one byte per lane).
// Hence:
//  * LDS image: 1 KiB per cell, halfword pairs permuted inside their tree level's region (gather_layout.inc, found by annealing
//    on the static neighbour table): 108-112 LDS cycles for the 48 gathers of a cell instead of 178;
//  * values are stored as the upper half of their f32 bit pattern (every coefficient the forward kernel produces, |v| <= 255,
//    is exact in 8 significant bits) and gathered with ds_read_u16_d16_hi into registers whose lower half stays zero: the
//    gather delivers f32 directly, no conversion; a tile that holds a value outside [-256, 256] takes a slow path that
//    gathers int32 from global memory and evaluates exactly what the reference does for any i32 (prediction.rs:86-207);
//  * the 48 LDS addresses of a lane are loop invariant (each wave owns one block slot; the two images differ by a constant
//    that rides in the instruction's offset field, the tile loop is unrolled by two), so a gather costs no vector ALU work;
//  * lane L owns heap nodes 4L..4L+3 and 256+4L..256+4L+3: its own values come straight from the registers of the staging
//    loads (exact int32), and a cell leaves as 2 x dwordx4 (predictions) + 2 x dword (bucket bytes) store instructions.
// ---------------------------------------------------------------------------------------------------------------------
struct P3Lds { // static LDS: every address below is a compile-time constant that folds into the DS instructions' offset fields
    uint32_t hist[kHistBins + 4];                       // 10 x 1024 counters + the out-of-alphabet counter (+ pad)
    uint8_t cells[2][kP3ImageBytes];                    // two images of a tile's 36 cells (+ zero words)
    int32_t ring[3][kPredSlots];                        // slot lists of tiles i, i + 1, i + 2
    uint16_t bkt[32];                                   // bucket_of(w) << 12
    uint32_t masks[2][kPredSlots][16];                  // Some/None masks of the staged cells
};
static_assert(sizeof(P3Lds) <= 160 * 1024 && kP3ImageBytes + kP3SlotBytes < 65536, "LDS budget / image + cell offset must fit a DS instruction's 16-bit offset field");
struct P3Group { // the parameters of one layer group (prediction.rs:165-179)
    float w[6], v[6];
};
__device__ __forceinline__ P3Group p3_group(const PredictParams &pp, int g) {
    P3Group q;
#pragma unroll
    for (int k = 0; k < 6; k++) q.w[k] = pp.width[g][k], q.v[k] = pp.value[g][k];
    return q;
}
// Six gathers of one node: f32 values straight out of the bf16-style LDS image. The destination registers' low halves are zero and
// stay zero (d16_hi loads write bits 31:16 only). The compiler does not track these loads, so the waits are explicit: p3_issue_wait
// starts the gathers of the NEXT node into `nxt` and waits for those of the current node in `cur` - LDS operations complete in
// order, so lgkmcnt(6) leaves exactly the six just issued in flight while everything older has landed (an LDS operation the
// compiler slips in behind them only makes the wait longer). Between two blocks nothing may touch the registers in flight:
// tools/check_k2_isa.py verifies that on the generated code.
template <int OFFSET>
__device__ __forceinline__ void p3_issue(float (&g)[6], const uint32_t (&a)[6]) {
    asm volatile("ds_read_u16_d16_hi %0, %6 offset:%12\n\t"
                 "ds_read_u16_d16_hi %1, %7 offset:%12\n\t"
                 "ds_read_u16_d16_hi %2, %8 offset:%12\n\t"
                 "ds_read_u16_d16_hi %3, %9 offset:%12\n\t"
                 "ds_read_u16_d16_hi %4, %10 offset:%12\n\t"
                 "ds_read_u16_d16_hi %5, %11 offset:%12"
                 : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "n"(OFFSET));
}
template <int OFFSET>
__device__ __forceinline__ void p3_issue_wait(float (&nxt)[6], const uint32_t (&a)[6], float (&cur)[6]) {
    asm volatile("ds_read_u16_d16_hi %0, %12 offset:%18\n\t"
                 "ds_read_u16_d16_hi %1, %13 offset:%18\n\t"
                 "ds_read_u16_d16_hi %2, %14 offset:%18\n\t"
                 "ds_read_u16_d16_hi %3, %15 offset:%18\n\t"
                 "ds_read_u16_d16_hi %4, %16 offset:%18\n\t"
                 "ds_read_u16_d16_hi %5, %17 offset:%18\n\t"
                 "s_waitcnt lgkmcnt(6)"
                 : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]), "+v"(nxt[4]), "+v"(nxt[5]), "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]),
                   "+v"(cur[5])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "n"(OFFSET));
}
// The bucket-table read of the PREVIOUS node (assign_bucket as a 32-entry LDS table) rides in front of the block's gathers: it is older than the six
// loads lgkmcnt(6) leaves in flight, so it has landed when the block ends. As a plain C++ load the compiler tracked it with its own counter - which does
// not know the asm's loads - and put `s_waitcnt lgkmcnt(0)` in front of its first use, directly behind the block that had just issued the next node's
// gathers: every node then waited for the full latency of the gathers meant to fly during its arithmetic.
template <int OFFSET>
__device__ __forceinline__ void p3_issue_wait_tbl(float (&nxt)[6], const uint32_t (&a)[6], float (&cur)[6], uint32_t &tbl, uint32_t tbl_addr) {
    asm volatile("ds_read_u16 %12, %19\n\t"
                 "ds_read_u16_d16_hi %0, %13 offset:%20\n\t"
                 "ds_read_u16_d16_hi %1, %14 offset:%20\n\t"
                 "ds_read_u16_d16_hi %2, %15 offset:%20\n\t"
                 "ds_read_u16_d16_hi %3, %16 offset:%20\n\t"
                 "ds_read_u16_d16_hi %4, %17 offset:%20\n\t"
                 "ds_read_u16_d16_hi %5, %18 offset:%20\n\t"
                 "s_waitcnt lgkmcnt(6)"
                 : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]), "+v"(nxt[4]), "+v"(nxt[5]), "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]),
                   "+v"(cur[5]), "=&v"(tbl)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(tbl_addr), "n"(OFFSET));
}
__device__ __forceinline__ void p3_wait_tbl(float (&cur)[6], uint32_t &tbl, uint32_t tbl_addr) {
    asm volatile("ds_read_u16 %6, %7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "=&v"(tbl)
                 : "v"(tbl_addr));
}
// the last node's table read: issued, landed a few instructions later (p3_tbl_land) - nothing else of the wave is in flight in LDS then
__device__ __forceinline__ void p3_tbl_issue(uint32_t &tbl, uint32_t tbl_addr) { asm volatile("ds_read_u16 %0, %1" : "=&v"(tbl) : "v"(tbl_addr)); }
__device__ __forceinline__ void p3_tbl_land(uint32_t &tbl) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(tbl)); }

// get_hf_context_bucket (prediction.rs:151-207) on f32 neighbour values: left to right, one rounding per operation. The reference
// takes |a - b| on i32 and converts; for |a|, |b| <= 256 the difference of the two floats is the same exact value, and the
// absolute value rides on the multiply as a source modifier.
__device__ __forceinline__ void p3_node_math(const float (&f)[6], const P3Group &q, float &width, float &pf) {
    width = q.w[0];
    width = __fadd_rn(width, __fmul_rn(q.w[1], fabsf(__fsub_rn(f[0], f[3]))));
    width = __fadd_rn(width, __fmul_rn(q.w[2], fabsf(__fsub_rn(f[1], f[2]))));
    width = __fadd_rn(width, __fmul_rn(q.w[3], fabsf(__fsub_rn(f[4], f[5]))));
    width = __fadd_rn(width, __fmul_rn(q.w[4], fabsf(__fsub_rn(f[1], f[5]))));
    width = __fadd_rn(width, __fmul_rn(q.w[5], fabsf(__fsub_rn(f[2], f[4]))));
    pf = __fmul_rn(f[0], q.v[0]);
    pf = __fadd_rn(pf, __fmul_rn(f[1], q.v[1]));
    pf = __fadd_rn(pf, __fmul_rn(f[2], q.v[2]));
    pf = __fadd_rn(pf, __fmul_rn(f[3], q.v[3]));
    pf = __fadd_rn(pf, __fmul_rn(f[4], q.v[4]));
    pf = __fadd_rn(pf, __fmul_rn(f[5], q.v[5]));
}

// get_lf_context_bucket (prediction.rs:134-144); on wave-uniform values this is scalar ALU work
__device__ __forceinline__ void p3_lf(int v0, int v1, int v2, uint32_t &b12, int &prediction) {
    const uint32_t w = (uint32_t)iabs_w(sub_w(v0, v2));
    const int mx = max(v0, v2), mn = min(v0, v2);
    prediction = v1 >= mx ? mx : v1 <= mn ? mn : sub_w(add_w(v0, v2), v1);
    b12 = bucket_of_rt(w) << 12;
}

// The heap node behind node slot n (0..3) of a lane. Waves of role 1 own level 8: 256 + 4 lane + n. Waves of role 0 own levels 0..7 as
// 2 lane, 2 lane + 1 (levels 0..6) and 128 + 2 lane, 128 + 2 lane + 1 (level 7): every instruction then works on ONE parameter group
// (prediction.rs:165-179), so the parameters stay in scalar registers for both roles.
template <int ROLE>
__device__ __forceinline__ int p3_node_of(int lane, int n) {
    return ROLE ? 256 + 4 * lane + n : 128 * (n >> 1) + 2 * lane + (n & 1);
}

// Four nodes of one of the wave's two block cells (CELL = 0, 1: neighbouring slots, 1 KiB apart - the same address registers serve
// both, the distance rides in the offset field) out of LDS image IMG. One straight-line body for every kind of cell: interior,
// boundary (some4 = which of the lane's nodes are Some) and absent (some4 = 0, the results go to the wave's junk lines) - the gather
// registers then never pass through a control-flow merge, where the compiler would copy all twelve. Heap nodes 0 and 1 (lane 0 of
// role 0) belong to the LF predictor, which the prologue evaluates for all of the workgroup's tiles (p3_lf_finish): here they are computed like any node and masked.
// Role 1 leaves as one dwordx4 + one dword store, role 0 as two dwordx2 + two short stores.
// WORDS: the cell leaves as one halfword per node, bucket << 10 | symbol - the counter index the node bumped, which is what the emitter codes
// (k5_stream.hip takes it from there) - instead of 5 bytes of bucket and prediction: role 1 one dwordx2, role 0 two dword stores to `wd`.
template <int IMG, int ROLE, int CELL, bool WORDS>
__device__ __forceinline__ void p3_half(const uint32_t (&addr)[4][6], float (&ga)[6], float (&gb)[6], const int (&own)[4], const PredictParams &pp, bool interior, uint32_t some4,
                                        int lane, uint32_t *s_hist, const uint16_t *s_bkt, uint8_t *bd, int32_t *pd, uint16_t *wd, uint8_t *junk, int ablate) {
    constexpr int kOff = IMG * kP3ImageBytes + CELL * kP3SlotBytes;
    uint32_t b12[4], bin[4], sym[4];
    int pred[4];
    const uint32_t bkt_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const uint16_t *)s_bkt;
    uint32_t tbl_addr[4];
    p3_issue<kOff>(ga, addr[0]);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // the next node's gathers fly while this node is evaluated; the two register sets alternate. (No timing-only switch in here: a runtime branch
        // around a block makes the compiler copy gather registers at the merge - while their loads are in flight.)
        float(&g)[6] = (j & 1) ? gb : ga;
        float(&gn)[6] = (j & 1) ? ga : gb;
        // Issue priority falls with the wave's progress through the tile (3, 3, 2, 2 | 1, 1, 0, 0 over its eight nodes): the arbiter otherwise prefers the
        // OLDER of a SIMD's four waves at every conflict, waves 0-3 reach the tile's barrier 1.2-1.5 us before waves 12-15 (tools/trace_k2_waves.py), and
        // each tile ends with SIMDs issuing from a single wave. With the laggard always the preferred one the four advance together (K2 -2.3 us at 4096^2;
        // a static priority per wave only changes who waits for whom and gained nothing).
        if (j == 0) __builtin_amdgcn_s_setprio(CELL ? 1 : 3);
        if (j == 2) __builtin_amdgcn_s_setprio(CELL ? 0 : 2);
        if (j == 0)
            p3_issue_wait<kOff>(gn, addr[1], g);
        else if (j < 3)
            p3_issue_wait_tbl<kOff>(gn, addr[j + 1], g, b12[j - 1], tbl_addr[j - 1]);
        else
            p3_wait_tbl(g, b12[2], tbl_addr[2]);
        if (j > 0) bin[j - 1] = b12[j - 1] + (sym[j - 1] << 2); // byte offset of the counter in the 10 x 1024 table
        const P3Group q = p3_group(pp, ROLE ? 0 : j < 2 ? 2 : 1); // uniform: scalar registers
        float width, pf;
        p3_node_math(g, q, width, pf);
        tbl_addr[j] = bkt_lds + 2u * min(f32_as_u32(width), 31u); // assign_bucket (prediction.rs:55-68) as a 32-entry table of bucket << 12
        pred[j] = f32_as_i32(pf);
        sym[j] = pack_signed(sub_w(own[j], pred[j]));
    }
    p3_tbl_issue(b12[3], tbl_addr[3]);
    p3_tbl_land(b12[3]);
    bin[3] = b12[3] + (sym[3] << 2);
    // out of alphabet (entropy_coding.rs:99 would panic): counted apart. One test for the four nodes.
    if (__builtin_expect(__any((sym[0] | sym[1] | sym[2] | sym[3]) >= 1024u), 0)) {
#pragma unroll
        for (int j = 0; j < 4; j++) bin[j] = sym[j] < 1024u ? bin[j] : (uint32_t)kHistBins * 4u;
    }
    if (ablate & 4) pred[0] ^= (int)(bin[0] ^ bin[1] ^ bin[2] ^ bin[3]); // (timing only: keeps the bin arithmetic alive)
    // How the four nodes leave. Called at the end of BOTH branches below rather than behind their merge: the boundary branch changes predictions and
    // buckets, and behind a merge the common path paid six register copies per cell for that.
    auto leave = [&](const int (&pr)[4], const uint32_t (&bk)[4]) {
        if (WORDS) {
            // bin = 4 x (bucket << 10 | symbol) (out of alphabet: 4 x kHistBins, "bucket 10" - no such symbol may be emitted, n_out_of_alphabet says so);
            // two nodes per dword: the low two bits of a bin are zero, so bin1 << 14 lands on bit 16
            const uint32_t w01 = (bin[1] << 14) | (bin[0] >> 2), w23 = (bin[3] << 14) | (bin[2] >> 2);
            if (ROLE) {
                if (!(ablate & 32) || ((w01 ^ w23) == 0x12345678)) __builtin_nontemporal_store(i32x2{(int)w01, (int)w23}, reinterpret_cast<i32x2 *>(wd + 256) + lane);
            } else {
                uint32_t *q01 = lane == 0 ? reinterpret_cast<uint32_t *>(junk) : reinterpret_cast<uint32_t *>(wd) + lane; // nodes 0 and 1: the LF pass writes them
                if (!(ablate & 32) || ((w01 ^ w23) == 0x12345678)) {
                    __builtin_nontemporal_store(w01, q01);
                    __builtin_nontemporal_store(w23, reinterpret_cast<uint32_t *>(wd + 128) + lane);
                }
            }
            return;
        }
        // bucket << 12 in each: byte 1 holds bucket << 4; collect the byte-1s, then one shift moves all the nibbles down.
        // Wave-uniform bases + a 32-bit lane offset: the stores address as saddr + voffset, no 64-bit pointer arithmetic per cell.
        if (ROLE) {
            const uint32_t lo = __builtin_amdgcn_perm(bk[1], bk[0], 0x0C0C0501u), hi = __builtin_amdgcn_perm(bk[3], bk[2], 0x05010C0Cu);
            if (!(ablate & 32) || ((lo ^ hi ^ pr[0] ^ pr[1] ^ pr[2] ^ pr[3]) == 0x12345678)) { // (32: timing only, no stores; the test keeps the arithmetic alive)
                __builtin_nontemporal_store(i32x4{pr[0], pr[1], pr[2], pr[3]}, reinterpret_cast<i32x4 *>(pd + 256) + lane);
                __builtin_nontemporal_store((lo | hi) >> 4, reinterpret_cast<uint32_t *>(bd + 256) + lane);
            }
        } else {
            const uint32_t lo = __builtin_amdgcn_perm(bk[1], bk[0], 0x0C0C0501u) >> 4, hi = __builtin_amdgcn_perm(bk[3], bk[2], 0x0C0C0501u) >> 4;
            // Nodes 0 and 1 are written by the LF pass: lane 0's pair goes to the wave's junk lines instead. By address, not under a branch -
            // the number of stores per cell must be the same on every path, or the wait for the staging loads turns into a wait for stores.
            i32x2 *p01 = lane == 0 ? reinterpret_cast<i32x2 *>(junk + 512) : reinterpret_cast<i32x2 *>(pd) + lane;
            uint16_t *b01 = lane == 0 ? reinterpret_cast<uint16_t *>(junk) : reinterpret_cast<uint16_t *>(bd) + lane;
            if (!(ablate & 32) || ((lo ^ hi ^ pr[0] ^ pr[1] ^ pr[2] ^ pr[3]) == 0x12345678)) {
                __builtin_nontemporal_store(i32x2{pr[0], pr[1]}, p01);
                __builtin_nontemporal_store((uint16_t)lo, b01);
                __builtin_nontemporal_store(i32x2{pr[2], pr[3]}, reinterpret_cast<i32x2 *>(pd + 128) + lane);
                __builtin_nontemporal_store((uint16_t)hi, reinterpret_cast<uint16_t *>(bd + 128) + lane);
            }
        }
    };
    if (__builtin_expect(interior, 1)) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool lf = ROLE == 0 && j < 2 && lane == 0;
            if (!lf && !(ablate & 4)) atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(s_hist) + bin[j]), 1u); // bump_freq, entropy_coding.rs:98-100
        }
        leave(pred, b12);
    } else { // a None node is not counted and stays (0, 0) in the outputs (wavelet_transform.rs:60-64)
        int pm[4];
        uint32_t bm[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool some = (some4 >> j) & 1u; // (the LF nodes' bits are already cleared)
            if (some && !(ablate & 4)) atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(s_hist) + bin[j]), 1u);
            pm[j] = some ? pred[j] : 0;
            bm[j] = some ? b12[j] : 0u;
        }
        leave(pm, bm);
    }
}

// get_lf_context_bucket (prediction.rs:86-149) for heap nodes 0 (DC) and 1 (root), OUTSIDE the tile loop. The two nodes of a cell see the
// same heap node of three neighbouring CELLS (left, up-left, up-right: entries 0..2 of their rows of the static neighbour table), 66 K
// nodes of a 4096^2 plane's 17 M - but as a pass of one wave inside every tile iteration (round 2) they sat on the critical path of the
// tile's barrier: ~95 vector instructions and three LDS round trips on one of sixteen waves, ~0.3 us of every 4.3 us tile. Here one thread
// takes one (tile of the workgroup's walk, block cell, node): slot entries first (hop A: the cell and its three neighbours come out of the
// tile's slot list), then four int32 coefficients straight from global memory (hop B; exact for any int32, no LDS image involved), and both
// hops ride on round trips the prologue makes anyway (slot lists of the first tiles; the first tile's staging loads).
struct P3LfItem {
    int raw;          // slot entry of the own cell (-1: no item / no cell)
    int nb[3];        // slot entries of the three neighbour cells (-1: absent, or the position is never a node: the reference reads 0)
    int node;         // heap index 0 / 1
};
__device__ __forceinline__ void p3_lf_hop_a(const PredArgs &a, uint32_t tile, bool active, int tid, P3LfItem &it) {
    const int c = (tid & 31) >> 1;
    it.node = tid & 1;
    const int slot = (1 + c / kPredBlock) * kPredSide + 1 + (c % kPredBlock);
    const int32_t *row = a.pred_slots + (size_t)(active ? tile : 0) * kPredSlots;
    it.raw = row[slot];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const uint32_t e = a.nbr_table[it.node * 6 + k];
        const int s7 = (e >> 9) & 7; // index into {self, +V9[0..5]}: lattice deltas as in pred_offsets_from_row
        const int da = (int)((0x0F14u >> (2 * s7)) & 3u), db = (int)((0x14F0u >> (2 * s7)) & 3u);
        const int delta = ((da & 1) - (da & 2)) * kPredSide + ((db & 1) - (db & 2));
        const int r = row[slot + delta];
        it.nb[k] = (e & 0x8000u) ? -1 : r;
    }
    if (!active) it.raw = -1;
}
struct P3LfValues {
    int v[3], value;
    uint32_t mask0;
};
__device__ __forceinline__ void p3_lf_hop_b(const PredArgs &a, const int32_t *plane, const P3LfItem &it, P3LfValues &x) {
    const uint32_t n = (uint32_t)it.node;
#pragma unroll
    for (int k = 0; k < 3; k++) x.v[k] = (plane + (size_t)max(pred_slot_cell(it.nb[k]), 0) * kCell)[n]; // unconditional loads: an absent neighbour reads cell 0 and is zeroed below
    const int cell = max(pred_slot_cell(it.raw), 0);
    x.value = (plane + (size_t)cell * kCell)[n];
    x.mask0 = a.valid_mask[(size_t)cell * 16];
}
template <bool WORDS>
__device__ __forceinline__ void p3_lf_finish(const PredArgs &a, uint32_t *s_hist, const P3LfItem &it, const P3LfValues &x) {
    const int cell = pred_slot_cell(it.raw);
    if (cell < 0) return;
    int v[3];
#pragma unroll
    for (int k = 0; k < 3; k++) v[k] = it.nb[k] < 0 || x.v[k] == kNone ? 0 : x.v[k]; // no cell there / Option::None: unwrap_or(0)
    uint32_t b12;
    int prediction;
    p3_lf(v[0], v[1], v[2], b12, prediction);
    const bool some = pred_slot_interior(it.raw) || ((x.mask0 >> it.node) & 1u);
    const uint32_t sym = pack_signed(sub_w(x.value, prediction));
    const uint32_t counter = sym < 1024u ? (b12 >> 2) + sym : (uint32_t)kHistBins;
    if (some) atomicAdd(&s_hist[counter], 1u);
    if (WORDS) {
        a.words[(size_t)cell * kCell + it.node] = (uint16_t)counter;
        return;
    }
    if (a.prediction) a.prediction[(size_t)cell * kCell + it.node] = some ? prediction : 0;
    if (a.bucket) a.bucket[(size_t)cell * kCell + it.node] = (uint8_t)(some ? b12 >> 12 : 0u);
}

// The same outputs the way the reference computes them for ANY int32 input (prediction.rs:86-207, context_modeling.rs:25-77): one
// thread per node, neighbour values gathered from global memory as int32 through the static neighbour table and the cells' neighbour
// lists, |a - b| in wrapping i32 before the conversion. kernel3's LDS image holds magnitudes up to 256 only (everything the forward
// kernel produces); when it meets a larger value it raises the plan's `inexact` flag, its hand-over then writes an all-zero histogram,
// and this kernel - enqueued behind it by the entry points that take coefficients of unknown origin - redoes the plane. With the flag
// down it returns at once.
struct ExactArgs {
    const int32_t *coefs;      // one channel plane [F][512]
    const uint16_t *nbr_table; // [512][6]
    const int32_t *nbr_cells;  // [F][kNbr]
    const uint8_t *interior;   // [F]
    const uint32_t *valid_mask; // [F][16]
    uint8_t *bucket;
    int32_t *prediction;
    uint32_t *hist;
    unsigned long long *n_oob;
    uint32_t *acc;     // plane k's flag and ticket sit at acc + k * kPredAccWords + kAccInexact, + 1
    uint32_t F;
    PredictParams pp;
    size_t coef_stride, out_stride; // planes of a batch, as in PredArgs
    const PredictParams *params;
    PredictParams pp3[3]; // plane k < 3 of a launch without a params array
};
__global__ void __launch_bounds__(kCell) exact_predict_kernel(const ExactArgs a0) {
    ExactArgs a = a0;
    {
        const uint32_t plane = blockIdx.y;
        a.coefs += plane * a0.coef_stride;
        if (a.bucket) a.bucket += plane * a0.out_stride;
        if (a.prediction) a.prediction += plane * a0.out_stride;
        a.hist += (size_t)plane * kHistBins;
        a.n_oob += plane;
        a.acc += (size_t)plane * kPredShards * kPredAccWords;
    }
    // this plane's parameters as scalars (static indices only: a dynamic index into the argument struct would keep all of it in scratch memory);
    // a thread then picks its layer group's set with selects
    PredictParams pp;
    if (a0.params)
        pp = a0.params[blockIdx.y];
    else if (blockIdx.y == 0)
        pp = a0.pp3[0];
    else if (blockIdx.y == 1)
        pp = a0.pp3[1];
    else
        pp = a0.pp3[2];
    uint32_t *const inexact = a.acc + kAccInexact;
    __shared__ uint32_t s_go;
    if (threadIdx.x == 0) s_go = __hip_atomic_load(inexact, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_go == 0) return;
    const int p = threadIdx.x;
    for (uint32_t cell = blockIdx.x; cell < a.F; cell += gridDim.x) {
        int v[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const uint32_t e = a.nbr_table[p * 6 + k];
            v[k] = 0;
            if (!(e & 0x8000u)) { // else: the position is never a node of that level, the reference finds nothing and reads 0
                const int c = a.nbr_cells[(size_t)cell * kNbr + ((e >> 9) & 7u)];
                if (c >= 0) {
                    const int x = a.coefs[(size_t)c * kCell + (e & 511u)];
                    v[k] = x == kNone ? 0 : x; // unwrap_or(0)
                }
            }
        }
        uint32_t bucket;
        int prediction;
        if (p < 2) { // get_lf_context_bucket, prediction.rs:134-144
            const uint32_t w = (uint32_t)iabs_w(sub_w(v[0], v[2]));
            const int mx = max(v[0], v[2]), mn = min(v[0], v[2]);
            prediction = v[1] >= mx ? mx : v[1] <= mn ? mn : sub_w(add_w(v[0], v[2]), v[1]);
            bucket = bucket_of_rt(w);
        } else { // get_hf_context_bucket, prediction.rs:165-206
            const int g = p >= 256 ? 0 : p >= 128 ? 1 : 2;
            float wp[6], vp[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                wp[k] = g == 0 ? pp.width[0][k] : g == 1 ? pp.width[1][k] : pp.width[2][k];
                vp[k] = g == 0 ? pp.value[0][k] : g == 1 ? pp.value[1][k] : pp.value[2][k];
            }
            float width = wp[0];
            width = __fadd_rn(width, __fmul_rn(wp[1], (float)iabs_w(sub_w(v[0], v[3]))));
            width = __fadd_rn(width, __fmul_rn(wp[2], (float)iabs_w(sub_w(v[1], v[2]))));
            width = __fadd_rn(width, __fmul_rn(wp[3], (float)iabs_w(sub_w(v[4], v[5]))));
            width = __fadd_rn(width, __fmul_rn(wp[4], (float)iabs_w(sub_w(v[1], v[5]))));
            width = __fadd_rn(width, __fmul_rn(wp[5], (float)iabs_w(sub_w(v[2], v[4]))));
            bucket = assign_bucket(width);
            float pf = __fmul_rn((float)v[0], vp[0]);
#pragma unroll
            for (int k = 1; k < 6; k++) pf = __fadd_rn(pf, __fmul_rn((float)v[k], vp[k]));
            prediction = f32_as_i32(pf);
        }
        const bool some = a.interior[cell] || ((a.valid_mask[(size_t)cell * 16 + (p >> 5)] >> (p & 31)) & 1u);
        const int value = a.coefs[(size_t)cell * kCell + p];
        const uint32_t sym = pack_signed(sub_w(value, prediction));
        if (some) {
            if (sym < 1024u)
                atomicAdd(a.hist + bucket * 1024u + sym, 1u); // bump_freq, entropy_coding.rs:98-100
            else
                atomicAdd(a.n_oob, 1ull);
        }
        if (a.prediction) a.prediction[(size_t)cell * kCell + p] = some ? prediction : 0;
        if (a.bucket) a.bucket[(size_t)cell * kCell + p] = (uint8_t)(some ? bucket : 0u);
    }
    // the last block to finish lowers the flag for the next launch
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(inexact + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __hip_atomic_store(inexact + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(inexact, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ int p3_some_or_zero(int v) { return v == kNone ? 0 : v; }
// int32 -> upper half of the f32 pattern, two per dword; with CHECK returns the larger magnitude (as f32), else 0
template <bool CHECK>
__device__ __forceinline__ float p3_pack2(int x, int y, uint32_t &d) {
    const float f0 = (float)x, f1 = (float)y;
    d = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, f1), __builtin_bit_cast(uint32_t, f0), 0x07060302u);
    return CHECK ? __builtin_fmaxf(__builtin_fabsf(f0), __builtin_fabsf(f1)) : 0.f;
}
// Four consecutive heap nodes (two pairs) of a cell into its slot at byte positions pos0, pos1. `raw` is the slot-list entry.
// Returns the largest magnitude written (CHECK). The fix-up of a boundary cell's None entries / of a slot without a cell happens IN PLACE
// (wave-uniform and rare): a by-value copy made the compiler keep two versions of the lane's own values around the branch - eight register
// moves per tile on the common path; the own values with None -> 0 serve the residuals just as well (a None node's outputs are masked).
template <bool CHECK>
__device__ __forceinline__ float p3_commit4(int raw, i32x4 v, uint8_t *dst, uint32_t pos0, uint32_t pos1) {
    if (!pred_slot_interior(raw)) { // a slot without a retained cell reads as 0, a boundary cell's None entries too (unwrap_or(0))
        if (raw < 0)
            v = i32x4{0, 0, 0, 0};
        else
            v = i32x4{p3_some_or_zero(v.x), p3_some_or_zero(v.y), p3_some_or_zero(v.z), p3_some_or_zero(v.w)};
    }
    uint32_t d0, d1;
    const float m = __builtin_fmaxf(p3_pack2<CHECK>(v.x, v.y, d0), p3_pack2<CHECK>(v.z, v.w, d1));
    *reinterpret_cast<uint32_t *>(dst + pos0) = d0;
    *reinterpret_cast<uint32_t *>(dst + pos1) = d1;
    return m;
}
// every integer of magnitude <= 256 is exact in the 8 significant bits of the stored half; a plane that holds anything larger is redone by
// the exact kernel (exact_predict_kernel), which the library launches behind this one whenever the caller's coefficients are not known
// to come from the forward kernel
__device__ __forceinline__ void p3_check(float m, int lane, uint32_t *inexact) {
    if (__builtin_expect(__any(m > 256.0f), 0) && lane == 0) __hip_atomic_store(inexact, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One halo value per thread and tile (build_halo_list): the slot-list entry of its halo cell comes out of the LDS ring, the value straight from
// the plane (wave-uniform base + 32-bit offset), and it lands as one halfword - the upper half of its f32 pattern - in the next image.
struct P3Halo {
    int raw, v;
};
__device__ __forceinline__ void p3_issue_halo(const int32_t *plane, const int32_t *slots, uint32_t ring_off, uint32_t heap_off, P3Halo &h) {
    h.raw = *reinterpret_cast<const int32_t *>(reinterpret_cast<const uint8_t *>(slots) + ring_off);
    const uint32_t cell = (uint32_t)max(h.raw, 0) & (uint32_t)(kPredSlotInterior - 1); // a slot without a cell reads cell 0 and is zeroed at the commit
    h.v = *reinterpret_cast<const int32_t *>(reinterpret_cast<const uint8_t *>(plane) + ((cell << 11) + heap_off));
}
template <bool CHECK>
__device__ __forceinline__ float p3_commit_halo(const P3Halo &h, uint8_t *image, uint32_t lds_off) {
    const int v = (h.raw < 0 || h.v == kNone) ? 0 : h.v; // no cell there / Option::None: unwrap_or(0)
    const float f = (float)v;
    *reinterpret_cast<uint16_t *>(image + lds_off) = (uint16_t)(__builtin_bit_cast(uint32_t, f) >> 16);
    return CHECK ? __builtin_fabsf(f) : 0.f;
}

// Wave w of a workgroup works on the block cells 2 (w >> 1) and 2 (w >> 1) + 1 of a tile - neighbours in a block row, so their LDS slots
// are 1 KiB apart - and on one half of their nodes (p3_node_of): role 1 = level 8, role 0 = levels 0..7. One set of 24 neighbour
// addresses serves both cells.
struct P3Lane {          // loop invariants of a lane
    uint32_t addr[4][6]; // LDS addresses (image 0, first cell) of the six neighbours of its four nodes
    uint32_t opos[2];    // byte positions inside a slot of the two pairs it stages of its own cells: role 1 = pairs 128 + 2 lane, + 1; role 0 = pairs lane and 64 + lane
    uint32_t halo_ring;  // the thread's halo value (build_halo_list): byte offset of its cell's entry in a slot list,
    uint32_t halo_heap;  // byte offset of the value inside the cell's 2 KiB,
    uint32_t halo_lds;   // byte offset of its halfword inside an LDS image
};

// what a wave stages of one of its own block cells: its four nodes (role 1: one dwordx4, role 0: two dwordx2). Wave-uniform base
// + 32-bit lane offset: saddr + voffset addressing, no vector pointer arithmetic.
template <int ROLE>
__device__ __forceinline__ i32x4 p3_load_own(const int32_t *cell_base, uint32_t lane) {
    if (ROLE) return reinterpret_cast<const i32x4 *>(cell_base + 256)[lane];
    const i32x2 a = reinterpret_cast<const i32x2 *>(cell_base)[lane], b = reinterpret_cast<const i32x2 *>(cell_base + 128)[lane];
    return i32x4{a.x, a.y, b.x, b.y};
}

// OWN_CUR / OWN_NXT: the lane's own values (exact int32, for the residuals) of this tile's two cells and, loaded here, of the next
// tile's; the two register sets swap roles from tile to tile (IMG), so nothing is copied.
template <int IMG, int ROLE, bool CHECK, bool WORDS>
__device__ __forceinline__ void p3_tile(const PredArgs &a, const int32_t *plane, P3Lds &lds, int it, bool more, uint32_t next2_tile, int tid, int lane, int wave, int slot_a,
                                        const P3Lane &L, float (&ga)[6], float (&gb)[6], const i32x4 (&own_cur)[2], i32x4 (&own_nxt)[2]) {
    uint32_t *s_hist = lds.hist;
    const uint16_t *s_bkt = lds.bkt;
    const int32_t *cur_slots = lds.ring[it % 3], *nxt_slots = lds.ring[(it + 1) % 3];
    // in flight across the arithmetic below: the slot list of tile i + 2 and what this wave stages of tile i + 1 - its half of its two
    // block cells and one halo value per lane
    const int32_t slot_pre = a.pred_slots[(size_t)next2_tile * kPredSlots + tid % kPredSlots];
    uint32_t st_own_mask[2];
    P3Halo st_halo;
    int raw_own[2] = {-1, -1};
    if (ablate_flags(a.ablate) & 2) more = false;
    if (more) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
            raw_own[c] = __builtin_amdgcn_readfirstlane(nxt_slots[slot_a + c]);
            const int cell = max(pred_slot_cell(raw_own[c]), 0);
            own_nxt[c] = p3_load_own<ROLE>(plane + (size_t)cell * kCell, (uint32_t)lane);
            if (ROLE == 1) st_own_mask[c] = (a.valid_mask + (size_t)cell * 16)[(uint32_t)lane & 15u]; // the level-8 waves have registers to spare
        }
        p3_issue_halo(plane, nxt_slots, L.halo_ring, L.halo_heap, st_halo);
    }

    // two block cells per wave; every cell issues the same number of stores (without a retained cell at the block slot they go to the
    // wave's junk lines), so the commit below waits for the staging loads with a counted vmcnt and not for the stores just issued
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int raw = (ablate_flags(a.ablate) & 1) ? -1 : __builtin_amdgcn_readfirstlane(cur_slots[slot_a + c]); // all this phase needs is in LDS or registers
        const int cell = pred_slot_cell(raw);
        const bool has = cell >= 0, interior = pred_slot_interior(raw);
        const size_t junk = ((size_t)blockIdx.x * kP3Waves + wave) * kPredJunkBytes;
        uint8_t *bd = has && a.bucket ? a.bucket + (size_t)cell * kCell : a.junk + junk;
        int32_t *pd = has && a.prediction ? a.prediction + (size_t)cell * kCell : reinterpret_cast<int32_t *>(a.junk + junk + 512);
        uint16_t *wd = WORDS && has ? a.words + (size_t)cell * kCell : reinterpret_cast<uint16_t *>(a.junk + junk + 512);
        uint32_t some4 = 0;
        if (!interior && has) { // boundary cell: node p is bit (p & 31) of mask word p >> 5
            const uint32_t *m = lds.masks[IMG][slot_a + c];
            if (ROLE) {
                some4 = (m[8 + (lane >> 3)] >> (4 * (lane & 7))) & 15u;
            } else {
                some4 = ((m[lane >> 4] >> (2 * (lane & 15))) & 3u) | (((m[4 + (lane >> 4)] >> (2 * (lane & 15))) & 3u) << 2);
                if (lane == 0) some4 &= ~3u; // heap nodes 0 and 1: p3_lf_finish
            }
        }
        const int own4[4] = {own_cur[c].x, own_cur[c].y, own_cur[c].z, own_cur[c].w};
        if (WORDS && (ablate_flags(a.ablate) & 1)) {
            if (ROLE) {
                __builtin_nontemporal_store(i32x2{0, 0}, reinterpret_cast<i32x2 *>(wd + 256) + lane);
            } else {
                __builtin_nontemporal_store(0u, reinterpret_cast<uint32_t *>(wd) + lane);
                __builtin_nontemporal_store(0u, reinterpret_cast<uint32_t *>(wd + 128) + lane);
            }
        } else if (ablate_flags(a.ablate) & 1) {
            if (ROLE) {
                __builtin_nontemporal_store(i32x4{0, 0, 0, 0}, reinterpret_cast<i32x4 *>(pd + 256) + lane);
                __builtin_nontemporal_store(0u, reinterpret_cast<uint32_t *>(bd + 256) + lane);
            } else {
                __builtin_nontemporal_store(i32x2{0, 0}, reinterpret_cast<i32x2 *>(pd) + lane);
                __builtin_nontemporal_store((uint16_t)0, reinterpret_cast<uint16_t *>(bd) + lane);
                __builtin_nontemporal_store(i32x2{0, 0}, reinterpret_cast<i32x2 *>(pd + 128) + lane);
                __builtin_nontemporal_store((uint16_t)0, reinterpret_cast<uint16_t *>(bd + 128) + lane);
            }
        } else if (c == 0) {
            p3_half<IMG, ROLE, 0, WORDS>(L.addr, ga, gb, own4, a.pp, interior, some4, lane, s_hist, s_bkt, bd, pd, wd, a.junk + junk, ablate_flags(a.ablate));
        } else {
            p3_half<IMG, ROLE, 1, WORDS>(L.addr, ga, gb, own4, a.pp, interior, some4, lane, s_hist, s_bkt, bd, pd, wd, a.junk + junk, ablate_flags(a.ablate));
        }
    }

    if (more) {
        // The staged registers are consumed from here on, not earlier: left alone, the compiler hoists uses of the next tile's own
        // values to the top of the iteration and waits for the loads there - in front of the arithmetic they are meant to hide behind.
        asm volatile("" : "+v"(own_nxt[0]), "+v"(own_nxt[1]), "+v"(st_halo.v));
        // The image the staging writes into is a compile-time constant per unrolled phase, so the nine write addresses of a lane (image + slot +
        // lane position) are loop invariants to the compiler: it hoists them out of the tile loop into registers the loop does not have - and
        // reloads the spilled ones from scratch memory behind an s_waitcnt vmcnt(0), i.e. behind the tile's stores. The image offset is made
        // opaque (a scalar: no instruction), so the additions stay where they are.
        uint32_t nxt_img = IMG ^ 1;
        asm volatile("" : "+s"(nxt_img));
        uint8_t *nxt = lds.cells[0] + nxt_img * kP3ImageBytes;
        uint32_t *nxt_masks = &lds.masks[0][0][0] + nxt_img * (kPredSlots * 16);
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            m = __builtin_fmaxf(m, p3_commit4<CHECK>(raw_own[c], own_nxt[c], nxt + (slot_a + c) * kP3SlotBytes, L.opos[0], L.opos[1]));
            if (ROLE == 1 && lane < 16) nxt_masks[(slot_a + c) * 16 + lane] = st_own_mask[c];
        }
        m = __builtin_fmaxf(m, p3_commit_halo<CHECK>(st_halo, nxt, L.halo_lds));
        if (CHECK) p3_check(m, lane, a.inexact);
    }
    if (tid < kPredSlots) lds.ring[(it + 2) % 3][tid] = slot_pre;
    lds_barrier();
    trace_stamp(a.trace, blockIdx.x, 2 + it, tid);
}

template <int ROLE, bool CHECK, bool WORDS>
__device__ __forceinline__ void p3_run(const PredArgs &a, P3Lds &lds, int tid, int lane, int wave) {
    uint8_t *s_cells = lds.cells[0];
    int32_t *s_ring = &lds.ring[0][0];
    uint32_t *s_masks = &lds.masks[0][0][0];
    const int pair = wave >> 1; // block cells 2 pair, 2 pair + 1: block row pair >> 1, columns 2 (pair & 1), + 1
    const int slot_a = (1 + (pair >> 1)) * kPredSide + 1 + 2 * (pair & 1);
    const int32_t *plane = a.coefs;

    P3Lane L;
    const uint32_t cells_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s_cells;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const u32x4 o = reinterpret_cast<const u32x4 *>(a.pred_off)[p3_node_of<ROLE>(lane, j)];
        const uint32_t rel[3] = {o.x, o.y, o.z};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int r = (int)(short)((rel[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
            // "never a node": the image's zero words. The second cell reads them 1 KiB further on, hence 1 KiB + of zeros behind the cells.
            L.addr[j][k] = r == 0x7FFF ? cells_lds + kP3ZeroOff : cells_lds + (uint32_t)(slot_a * kP3SlotBytes + r);
        }
    }
    L.opos[0] = 4u * a.pair_pos[ROLE ? 128 + 2 * lane : lane];
    L.opos[1] = 4u * a.pair_pos[ROLE ? 129 + 2 * lane : 64 + lane];
    {
        const uint32_t e = a.halo_list[tid];
        L.halo_ring = 4u * (e & 63u), L.halo_heap = 4u * ((e >> 8) & 511u), L.halo_lds = (e & 63u) * (uint32_t)kP3SlotBytes + (e >> 20);
    }
    float ga[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gb[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    i32x4 own_a[2] = {i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}}, own_b[2] = {i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}};

    const PredTileWalk walk(a.n_tiles);
    if (walk.first >= walk.end) return; // (a workgroup without a tile still takes part in the hand-over)
    const uint32_t last = walk.first + ((walk.end - 1 - walk.first) / walk.step) * walk.step; // this workgroup's last tile
    // Heap nodes 0 and 1 of every block cell of this workgroup's tiles (p3_lf_*), by the level-8 waves only (role 0 holds two parameter groups
    // in scalar registers and has none to spare: with this code in its prologue its tile loop spilled 68 scalars instead of 4): thread t of the
    // eight role-1 waves takes tile t >> 5 of the walk, 16 tiles per pass.
    constexpr uint32_t kLfTilesPerPass = kP3Threads / 2 / 32;
    const uint32_t my_tiles = (walk.end - walk.first + walk.step - 1u) / walk.step;
    const int lf_tid = (wave >> 1) * 64 + lane;
    P3LfItem lf_item;
    if (ROLE == 1) p3_lf_hop_a(a, walk.first + (uint32_t)(lf_tid >> 5) * walk.step, (uint32_t)(lf_tid >> 5) < my_tiles, lf_tid, lf_item);
    if (tid < kPredSlots) {
        s_ring[tid] = a.pred_slots[(size_t)walk.first * kPredSlots + tid];
        s_ring[kPredSlots + tid] = a.pred_slots[(size_t)min(walk.first + walk.step, last) * kPredSlots + tid];
    }
    __syncthreads();
    { // tile 0 straight into image 0: everything a wave stages is requested before the first value is converted
        uint32_t st_own_mask[2] = {0, 0};
        int raw_own[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            raw_own[c] = __builtin_amdgcn_readfirstlane(s_ring[slot_a + c]);
            const int cell = max(pred_slot_cell(raw_own[c]), 0);
            own_a[c] = p3_load_own<ROLE>(plane + (size_t)cell * kCell, (uint32_t)lane);
            if (ROLE == 1) st_own_mask[c] = (a.valid_mask + (size_t)cell * 16)[(uint32_t)lane & 15u];
        }
        P3Halo st_halo;
        p3_issue_halo(plane, s_ring, L.halo_ring, L.halo_heap, st_halo);
        P3LfValues lf_values;
        if (ROLE == 1) p3_lf_hop_b(a, plane, lf_item, lf_values); // behind the staging loads: one round trip for both
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            m = __builtin_fmaxf(m, p3_commit4<CHECK>(raw_own[c], own_a[c], s_cells + (slot_a + c) * kP3SlotBytes, L.opos[0], L.opos[1]));
            if (ROLE == 1 && lane < 16) s_masks[(slot_a + c) * 16 + lane] = st_own_mask[c];
        }
        m = __builtin_fmaxf(m, p3_commit_halo<CHECK>(st_halo, s_cells, L.halo_lds));
        if (CHECK) p3_check(m, lane, a.inexact);
        if (ROLE == 1 && !(ablate_flags(a.ablate) & 1)) p3_lf_finish<WORDS>(a, lds.hist, lf_item, lf_values);
    }
    if (ROLE == 1) {
        for (uint32_t base = kLfTilesPerPass; base < my_tiles; base += kLfTilesPerPass) { // more than 16 tiles per workgroup (large images): further passes, two round trips each
            const uint32_t k = base + (uint32_t)(lf_tid >> 5);
            P3LfItem it;
            P3LfValues x;
            p3_lf_hop_a(a, walk.first + k * walk.step, k < my_tiles, lf_tid, it);
            p3_lf_hop_b(a, plane, it, x);
            if (!(ablate_flags(a.ablate) & 1)) p3_lf_finish<WORDS>(a, lds.hist, it, x);
        }
    }
    // Only LDS is handed over here. __syncthreads() also waits for the LF pass's scattered global stores to be acknowledged: 1.8 us between "the first
    // tile's data has landed" and "prologue done" by the time stamps, most of it that wait.
    lds_barrier();
    trace_stamp(a.trace, blockIdx.x, 1, tid);

    int it = 0;
    for (uint32_t tile = walk.first; tile < walk.end;) { // unrolled by two: the LDS image a tile lives in is a compile-time constant
        p3_tile<0, ROLE, CHECK, WORDS>(a, plane, lds, it, tile + walk.step < walk.end, min(tile + 2 * walk.step, last), tid, lane, wave, slot_a, L, ga, gb, own_a, own_b);
        tile += walk.step, it++;
        if (tile >= walk.end) break;
        p3_tile<1, ROLE, CHECK, WORDS>(a, plane, lds, it, tile + walk.step < walk.end, min(tile + 2 * walk.step, last), tid, lane, wave, slot_a, L, ga, gb, own_b, own_a);
        tile += walk.step, it++;
    }
}

__device__ __forceinline__ PredArgs pred_plane_view(const PredArgs &a0, uint32_t plane) {
    PredArgs a = a0;
    a.coefs += plane * a0.coef_stride;
    if (a.bucket) a.bucket += plane * a0.out_stride;
    if (a.prediction) a.prediction += plane * a0.out_stride;
    if (a.words) a.words += plane * a0.out_stride;
    a.hist += (size_t)plane * kHistBins;
    a.n_oob += plane;
    a.acc += (size_t)plane * kPredShards * kPredAccWords;
    a.inexact = a.acc + kAccInexact;
    // two branches with their own loads (caller's array / argument segment): a select between the two sources would be a select between
    // address spaces, and the copy behind it would go through scratch memory
    // (and no dynamic index into the argument struct either: that alone keeps the whole struct in scratch memory)
    if (a0.params)
        a.pp = a0.params[plane];
    else if (plane == 0)
        a.pp = a0.pp3[0];
    else if (plane == 1)
        a.pp = a0.pp3[1];
    else
        a.pp = a0.pp3[2];
    return a;
}

// CHECK = false: the coefficients were written by this library's forward kernel in the same chain (fri_hip_encode_image*): every magnitude is
// <= 255 by construction and the staging does not look (18 max operations per lane and tile). CHECK = true: any int32 array; a value the LDS image
// cannot hold raises the plane's `inexact` flag (see PredArgs::trusted for what happens then).
// WORDS: see p3_half (instantiated for the library's own coefficients only).
template <bool CHECK, bool WORDS>
__global__ void __launch_bounds__(kP3Threads) predict_histogram_kernel3(const PredArgs a0) {
    const PredArgs a = pred_plane_view(a0, blockIdx.y);
    __shared__ __attribute__((aligned(16))) P3Lds lds;
    uint32_t *s_hist = lds.hist;
    uint8_t *s_cells = lds.cells[0];
    int32_t *s_ring = &lds.ring[0][0];
    uint16_t *s_bkt = lds.bkt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    trace_stamp(a.trace, blockIdx.x, 0, tid);
    for (int i = tid; i < kHistBins + 4; i += kP3Threads) s_hist[i] = 0;
    if (tid < 32) s_bkt[tid] = (uint16_t)(bucket_of((uint32_t)tid) << 12);
    for (int i = tid; i < 2 * (kP3ImageBytes - kP3ZeroOff) / 4; i += kP3Threads) // the zero words behind the cells of both images
        reinterpret_cast<uint32_t *>(s_cells + (i / ((kP3ImageBytes - kP3ZeroOff) / 4)) * kP3ImageBytes + kP3ZeroOff)[i % ((kP3ImageBytes - kP3ZeroOff) / 4)] = 0;
    if (wave & 1)
        p3_run<1, CHECK, WORDS>(a, lds, tid, lane, wave);
    else
        p3_run<0, CHECK, WORDS>(a, lds, tid, lane, wave);
    __syncthreads();
    trace_stamp(a.trace, blockIdx.x, 13, tid);
    pred_hand_over(a, s_hist, reinterpret_cast<uint32_t *>(s_ring), tid, kP3Threads);
    trace_exit(a.trace, blockIdx.x, tid);
}

// ---------------------------------------------------------------------------------------------------------------------
// K4, value pass, on THIS kernel's skeleton (round 3): fit_value_kernel3. The sums of optimize_value_prediction (context_modeling.rs:175-202) are
// the Gram matrix of u = [v0..v5, value] per layer group - the same six gathers per node as the predictor plus the node itself. fit_accumulate_kernel2
// (k4_fit.hip: 512 threads, int16 cell images, v_dot2 on packed pairs, gathers the compiler schedules) spends a fifth of its time on those products and
// the rest waiting; here the values arrive as f32 through the hand-pipelined ds_read_u16_d16_hi blocks of K2 (seven per node: the node's own value is a
// seventh gather instead of a register) and a product-sum is one v_fma_f32. Everything is an integer of magnitude <= 65 536 = 256^2 in f32, exact while a
// lane's sum stays below 2^24: a lane sees eight rows per tile, so the sums are flushed (as integers) every 16 tiles. One layer group per wave, and
// every wave works on ONE pair of a lane's nodes in the FOUR block cells of its block row (slots 1 KiB apart, like K2's two): role-0 waves take 2 lane,
// 2 lane + 1 (levels 0..6, group 2; even wave pairs) or 128 + 2 lane, + 1 (level 7, group 1; odd pairs), role-1 waves 256 + 4 lane + {0, 1} or + {2, 3}
// (level 8, group 0) - fourteen addresses per lane where K2's four nodes need twenty-four. The roles differ in what they STAGE of the wave pair's two cells: K2's split.
// Staging, images, slot ring and tile walk are K2's; hand-over, accumulator and the solve in the tail are fit_accumulate_kernel2's.
// STATUS: an experiment, NOT the product path (DevicePlan::k4_value3, FRI_HIP_K4_VALUE3=1 under FRI_HIP_TUNING=1; tests/test_gpu_fit.py checks it bit for bit
// against the product kernel). It is exact and it is slower: 41.7 us against 37.5 us for a 4096^2 plane. What the time stamps say: a tile takes 3.2-3.4 us here
// too - K2's figure, with a third of K2's arithmetic, with one or two gather sets, with the staging loads one or two tiles ahead - and the wave sums of sixteen
// waves cost ~4 us at the end. One 1024-thread workgroup per CU with a barrier per tile does not keep the CU busy; fit_accumulate_kernel2's two independent
// 512-thread workgroups per CU cover each other's stalls (28 us for its tiles against 33.5 us here). The skeleton to carry over is that one, not K2's.
// ---------------------------------------------------------------------------------------------------------------------
struct Fit3Args {
    const int32_t *coefs;
    size_t coef_stride;
    const int32_t *pred_slots;
    const uint32_t *gather_off; // [512][4] byte offsets of the six neighbours in the permuted 1 KiB layout (build_gather_tables)
    const uint16_t *pair_pos;
    const uint32_t *halo_list;
    const uint32_t *valid_mask;
    uint32_t n_tiles;
    unsigned long long *acc;       // per plane: kFitShards copies of kFitAccWords words, zero between launches (the layout of k4_fit.hip)
    unsigned long long *gram;      // [n_planes][3][28]
    unsigned long long *out_range; // [n_planes] or NULL
    float *solve_params;           // as FitArgs (k4_fit.hip): NULL = sums only
    float *host_params;
    unsigned long long *host_range;
    unsigned long long *trace;
};
constexpr int kF3AccInt = 3 * 28, kF3AccTicket = kF3AccInt + 18, kF3AccRange = kF3AccTicket + 1; // = kFitAcc* of k4_fit.hip
static_assert(kF3AccRange + 1 == (int)kFitAccWords, "fit accumulator layout");

// "never a node" entries read the zero words behind an image's cells at the cell's distance from the wave's first cell: up to 3 KiB for role 0's four cells
// (K2's two cells: 1 KiB), so this kernel's images carry 4 KiB + 64 bytes of zeros
constexpr int kF3ImageBytes = kP3ZeroOff + 4 * kP3SlotBytes + 64;
struct F3Lds {
    uint8_t cells[2][kF3ImageBytes];
    int32_t ring[4][kPredSlots]; // slot lists of tiles i .. i + 3: the loads of tile i + 2 are issued during tile i
    uint32_t masks[2][kPredSlots][16];
    unsigned long long s_int[3][28];
    uint32_t flag, range;
    Solve6Work work[3];
};
static_assert(sizeof(F3Lds) <= 160 * 1024 && kF3ImageBytes + 3 * kP3SlotBytes < 65536, "LDS budget / image + cell offset must fit a DS instruction's 16-bit offset field");

template <int OFFSET>
__device__ __forceinline__ void f3_issue(float (&g)[7], const uint32_t (&a)[7]) {
    asm volatile("ds_read_u16_d16_hi %0, %7 offset:%14\n\t"
                 "ds_read_u16_d16_hi %1, %8 offset:%14\n\t"
                 "ds_read_u16_d16_hi %2, %9 offset:%14\n\t"
                 "ds_read_u16_d16_hi %3, %10 offset:%14\n\t"
                 "ds_read_u16_d16_hi %4, %11 offset:%14\n\t"
                 "ds_read_u16_d16_hi %5, %12 offset:%14\n\t"
                 "ds_read_u16_d16_hi %6, %13 offset:%14"
                 : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "n"(OFFSET));
}
template <int OFFSET>
__device__ __forceinline__ void f3_issue_wait(float (&nxt)[7], const uint32_t (&a)[7], float (&cur)[7]) {
    asm volatile("ds_read_u16_d16_hi %0, %14 offset:%21\n\t"
                 "ds_read_u16_d16_hi %1, %15 offset:%21\n\t"
                 "ds_read_u16_d16_hi %2, %16 offset:%21\n\t"
                 "ds_read_u16_d16_hi %3, %17 offset:%21\n\t"
                 "ds_read_u16_d16_hi %4, %18 offset:%21\n\t"
                 "ds_read_u16_d16_hi %5, %19 offset:%21\n\t"
                 "ds_read_u16_d16_hi %6, %20 offset:%21\n\t"
                 "s_waitcnt lgkmcnt(7)"
                 : "+v"(nxt[0]), "+v"(nxt[1]), "+v"(nxt[2]), "+v"(nxt[3]), "+v"(nxt[4]), "+v"(nxt[5]), "+v"(nxt[6]), "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]),
                   "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "n"(OFFSET));
}
__device__ __forceinline__ void f3_wait(float (&cur)[7]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]));
}
// one row: acc += upper triangle of u u^T, u = [v0..v5, value]; MASKED: the row counts only when m = 1 (a None node / an LF node is no row of the fit)
template <bool MASKED>
__device__ __forceinline__ void f3_row(const float (&u)[7], float m, float (&acc)[28]) {
    float mu[7];
#pragma unroll
    for (int i = 0; i < 7; i++) mu[i] = MASKED ? __fmul_rn(m, u[i]) : u[i];
    int n = 0;
#pragma unroll
    for (int i = 0; i < 7; i++)
#pragma unroll
        for (int j = i; j < 7; j++, n++) acc[n] = __builtin_fmaf(mu[i], u[j], acc[n]);
}
// one pair of a lane's nodes in the four block cells of the wave's block row: eight rows in one pipeline. MASKED: m says which of them are rows
// (the staged Some/None masks; heap nodes 0 and 1); without it all four cells are interior cells and every node is a row.
template <int IMG, bool MASKED>
__device__ __forceinline__ void f3_row_of_cells(const uint32_t (&addr)[2][7], uint32_t rows, float (&acc)[28]) {
    constexpr int kOff = IMG * kF3ImageBytes;
    // the two gather sets live inside this function (low halves zero: a d16_hi load writes bits 31:16): carried from tile to tile like K2's, they crossed the
    // interior / boundary branch of the caller and cost fourteen registers there plus copies at its merge
    float g[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ONE gather set: issue, wait, 28 multiply-adds - the other three waves of the SIMD cover the wait. (Two alternating sets, as in K2, are fourteen registers
    // this kernel spends on the second staging set instead: it spilled with both.)
#pragma unroll
    for (int t = 0; t < 8; t++) {
        if ((t >> 1) == 0) f3_issue<kOff>(g, addr[t & 1]);
        if ((t >> 1) == 1) f3_issue<kOff + kP3SlotBytes>(g, addr[t & 1]);
        if ((t >> 1) == 2) f3_issue<kOff + 2 * kP3SlotBytes>(g, addr[t & 1]);
        if ((t >> 1) == 3) f3_issue<kOff + 3 * kP3SlotBytes>(g, addr[t & 1]);
        f3_wait(g);
        f3_row<MASKED>(g, (rows >> t) & 1u ? 1.f : 0.f, acc);
    }
}
// A wave's 28 sums, exact integers in f32, into the workgroup's table. A lane's sum is below 2^24, so the wave's is below 2^30: the whole reduction stays in
// 32 bits - four butterfly steps inside the rows of 16, then row_bcast:15 / row_bcast:31 carry the row totals into lane 63, which adds the total to LDS
// (fit2_wave_sums of k4_fit.hip, whose 32-bit bound holds per row only, reads four row totals through readlane into 64-bit scalar adds: four times the instructions).
__device__ __forceinline__ void f3_wave_sums(float (&acc)[28], int group, int lane, F3Lds &lds) {
#pragma unroll
    for (int k = 0; k < 28; k++) {
        int v = (int)acc[k];
        acc[k] = 0.f;
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast:15 into rows 1 and 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast:31 into rows 2 and 3
        if (lane == 63) atomicAdd(&lds.s_int[group][k], (unsigned long long)(long long)v);
    }
}
// staging of one of the wave's own block cells, K2's (p3_commit4) plus the fit's range rule: a Some coefficient outside [-256, 255] is reported
template <bool CHECK>
__device__ __forceinline__ void f3_commit4(int raw, i32x4 v, uint8_t *dst, uint32_t pos0, uint32_t pos1, int lane, uint32_t *range_counter) {
    if (CHECK && raw >= 0) { // a None is 0x80000000: it is not an outlier
        const int w[4] = {v.x, v.y, v.z, v.w};
        uint32_t m = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) m |= w[j] == kNone ? 0u : ((uint32_t)w[j] + 256u) & 0xFFFFFE00u;
        if (__any(m != 0) && lane == 0) atomicAdd(range_counter, 1u);
    }
    (void)p3_commit4<false>(raw, v, dst, pos0, pos1);
}
template <bool CHECK>
__device__ __forceinline__ void f3_commit_halo(const P3Halo &h, uint8_t *image, uint32_t lds_off, int lane, uint32_t *range_counter) {
    if (CHECK) {
        const uint32_t m = (h.raw < 0 || h.v == kNone) ? 0u : ((uint32_t)h.v + 256u) & 0xFFFFFE00u;
        if (__any(m != 0) && lane == 0) atomicAdd(range_counter, 1u);
    }
    (void)p3_commit_halo<false>(h, image, lds_off);
}

struct F3Lane {
    uint32_t addr[2][7]; // the six neighbours and the node itself, for the lane's two nodes (first cell of the block row)
    uint32_t opos[2], halo_ring, halo_heap, halo_lds;
};

// What a wave stages of one tile: its half of its two block cells, their Some/None masks (role 1), one halo value per lane.
struct F3Stage {
    i32x4 own[2];
    uint32_t mask[2];
    P3Halo halo;
    int raw[2];
};
template <int ROLE>
__device__ __forceinline__ void f3_stage_issue(const Fit3Args &a, const int32_t *plane, const int32_t *slots, int slot_a, int lane, const F3Lane &L, F3Stage &st) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
        st.raw[c] = __builtin_amdgcn_readfirstlane(slots[slot_a + c]);
        const int cell = max(pred_slot_cell(st.raw[c]), 0);
        st.own[c] = p3_load_own<ROLE>(plane + (size_t)cell * kCell, (uint32_t)lane);
        if (ROLE == 1) st.mask[c] = (a.valid_mask + (size_t)cell * 16)[(uint32_t)lane & 15u];
    }
    p3_issue_halo(plane, slots, L.halo_ring, L.halo_heap, st.halo);
}
template <int ROLE, bool CHECK>
__device__ __forceinline__ void f3_stage_commit(F3Lds &lds, uint32_t img, int slot_a, int lane, const F3Lane &L, const F3Stage &st) {
    asm volatile("" : "+s"(img)); // (as in p3_tile: keeps the write addresses from being hoisted out of the loop into registers it does not have)
    uint8_t *image = lds.cells[0] + img * kF3ImageBytes;
    uint32_t *masks = &lds.masks[0][0][0] + img * (kPredSlots * 16);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        f3_commit4<CHECK>(st.raw[c], st.own[c], image + (slot_a + c) * kP3SlotBytes, L.opos[0], L.opos[1], lane, &lds.range);
        if (ROLE == 1 && lane < 16) masks[(slot_a + c) * 16 + lane] = st.mask[c];
    }
    f3_commit_halo<CHECK>(st.halo, image, L.halo_lds, lane, &lds.range);
}

// One tile out of image IMG. The loads of tile i + 2 are issued here, into `issue`; `commit` holds tile i + 1, requested a tile ago, and goes into the other
// image behind this tile's sums: a staging load has two tile times to arrive. (With K2's one-tile window the tile time of this kernel was the latency of
// those loads - 3.2 us, K2's figure, with a third of K2's arithmetic.)
template <int IMG, int ROLE, bool CHECK>
__device__ __forceinline__ void f3_tile(const Fit3Args &a, const int32_t *plane, F3Lds &lds, int it, bool more1, bool more2, uint32_t next3_tile, int tid, int lane, int slot_a,
                                        int slot_row, uint32_t mask_word, uint32_t mask_shift, bool lf_lane, bool lf_wave, const F3Lane &L, F3Stage &issue, const F3Stage &commit,
                                        float (&acc)[28]) {
    const int32_t *cur_slots = lds.ring[it & 3];
    const int32_t slot_pre = a.pred_slots[(size_t)next3_tile * kPredSlots + tid % kPredSlots];
    if (more2) f3_stage_issue<ROLE>(a, plane, lds.ring[(it + 2) & 3], slot_a, lane, L, issue);
    {
        // all four cells interior and no LF lane in the wave: every node is a row (wave-uniform; decided before the first gather is issued)
        int all_interior = lf_wave ? 0 : 1;
#pragma unroll
        for (int c = 0; c < 4; c++) all_interior &= pred_slot_interior(__builtin_amdgcn_readfirstlane(cur_slots[slot_row + c])) ? 1 : 0;
        if (all_interior) {
            f3_row_of_cells<IMG, false>(L.addr, 255u, acc);
        } else {
            // the pair's Some bits in the four cells, read before the first gather is issued. A slot without a cell has no rows at all (its own values are
            // zeros, but its gathers would see the neighbouring cells).
            uint32_t rows = 0; // bit 2 c + n: node n of the pair in cell c is a row
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const bool absent = __builtin_amdgcn_readfirstlane(cur_slots[slot_row + c]) < 0;
                const uint32_t bits = (lf_lane || absent) ? 0u : lds.masks[IMG][slot_row + c][mask_word] >> mask_shift;
                rows |= (bits & 3u) << (2 * c);
            }
            asm volatile("" : "+v"(rows)); // (complete now: no LDS wait of the compiler's between the blocks below)
            f3_row_of_cells<IMG, true>(L.addr, rows, acc);
        }
    }
    if (more1) f3_stage_commit<ROLE, CHECK>(lds, IMG ^ 1, slot_a, lane, L, commit);
    if (tid < kPredSlots) lds.ring[(it + 3) & 3][tid] = slot_pre;
    lds_barrier();
    trace_stamp(a.trace, blockIdx.x, 2 + it, tid);
}

template <int ROLE, bool CHECK>
__device__ __forceinline__ void f3_run(const Fit3Args &a, const int32_t *plane, F3Lds &lds, int tid, int lane, int wave) {
    uint8_t *s_cells = lds.cells[0];
    int32_t *s_ring = &lds.ring[0][0];
    const int pair = wave >> 1;
    const int slot_a = (1 + (pair >> 1)) * kPredSide + 1 + 2 * (pair & 1); // the wave's two block cells (staging; role 1 also works on them)
    const int slot_row = (1 + (pair >> 1)) * kPredSide + 1;               // role 0 works on the four cells of its block row
    const int group = ROLE ? 0 : (pair & 1) ? 1 : 2;
    const int node0 = ROLE ? 256 + 4 * lane + 2 * (pair & 1) : (pair & 1) ? 128 + 2 * lane : 2 * lane;
    const bool lf_wave = ROLE == 0 && !(pair & 1), lf_lane = lf_wave && lane == 0; // heap nodes 0 and 1 are the LF predictor's: no rows of the fit
    const uint32_t mask_word = (uint32_t)node0 >> 5, mask_shift = (uint32_t)node0 & 31u;

    F3Lane L;
    const uint32_t cells_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)s_cells;
    const int base_slot = slot_row;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int node = node0 + j;
        const u32x4 o = reinterpret_cast<const u32x4 *>(a.gather_off)[node];
        const uint32_t rel[3] = {o.x, o.y, o.z};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int r = (int)(short)((rel[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
            // "never a node": the image's zero words; the further cells of the wave read them up to 3 KiB further on - zeros too (see the kernel)
            L.addr[j][k] = r == 0x7FFF ? cells_lds + kP3ZeroOff : cells_lds + (uint32_t)(base_slot * kP3SlotBytes + r);
        }
        L.addr[j][6] = cells_lds + (uint32_t)(base_slot * kP3SlotBytes) + 2u * (2u * a.pair_pos[node >> 1] + (uint32_t)(node & 1)); // the node itself
    }
    L.opos[0] = 4u * a.pair_pos[ROLE ? 128 + 2 * lane : lane];
    L.opos[1] = 4u * a.pair_pos[ROLE ? 129 + 2 * lane : 64 + lane];
    {
        const uint32_t e = a.halo_list[tid];
        L.halo_ring = 4u * (e & 63u), L.halo_heap = 4u * ((e >> 8) & 511u), L.halo_lds = (e & 63u) * (uint32_t)kP3SlotBytes + (e >> 20);
    }
    float acc[28];
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0.f;

    const PredTileWalk walk(a.n_tiles);
    if (walk.first >= walk.end) return;
    const uint32_t last = walk.first + ((walk.end - 1 - walk.first) / walk.step) * walk.step;
    if (tid < kPredSlots) {
        s_ring[tid] = a.pred_slots[(size_t)walk.first * kPredSlots + tid];
        s_ring[kPredSlots + tid] = a.pred_slots[(size_t)min(walk.first + walk.step, last) * kPredSlots + tid];
        s_ring[2 * kPredSlots + tid] = a.pred_slots[(size_t)min(walk.first + 2 * walk.step, last) * kPredSlots + tid];
    }
    __syncthreads();
    F3Stage st_a, st_b; // tiles of even / odd index
    f3_stage_issue<ROLE>(a, plane, s_ring, slot_a, lane, L, st_a);
    if (walk.first + walk.step < walk.end) f3_stage_issue<ROLE>(a, plane, s_ring + kPredSlots, slot_a, lane, L, st_b); // tile 1: in flight through tile 0
    f3_stage_commit<ROLE, CHECK>(lds, 0u, slot_a, lane, L, st_a);                                                       // tile 0 straight into image 0
    lds_barrier();
    trace_stamp(a.trace, blockIdx.x, 1, tid);

    int it = 0, since_flush = 0;
    for (uint32_t tile = walk.first; tile < walk.end;) { // unrolled by two: the LDS image a tile lives in is a compile-time constant
        f3_tile<0, ROLE, CHECK>(a, plane, lds, it, tile + walk.step < walk.end, tile + 2 * walk.step < walk.end, min(tile + 3 * walk.step, last), tid, lane, slot_a, slot_row, mask_word,
                                mask_shift, lf_lane, lf_wave, L, st_a, st_b, acc);
        tile += walk.step, it++;
        if (tile < walk.end) {
            f3_tile<1, ROLE, CHECK>(a, plane, lds, it, tile + walk.step < walk.end, tile + 2 * walk.step < walk.end, min(tile + 3 * walk.step, last), tid, lane, slot_a, slot_row,
                                    mask_word, mask_shift, lf_lane, lf_wave, L, st_b, st_a, acc);
            tile += walk.step, it++;
        }
        since_flush += 2;
        if (since_flush >= 16) { // eight rows per tile and lane, each product <= 2^16: sixteen tiles stay below 2^24 - exact in f32
            f3_wave_sums(acc, group, lane, lds);
            since_flush = 0;
        }
    }
    f3_wave_sums(acc, group, lane, lds);
}

// the three solves at the end of the kernel (as fit2_tail_solve of k4_fit.hip: not inlined, so that its f64 temporaries stay out of the tile loop's budget)
__device__ __attribute__((noinline)) void f3_tail_solve(const long long *sums_int, Solve6Work *w, float *params, float *host_params) {
    float out[6];
    fit_value_group(sums_int, out, *w);
#pragma unroll
    for (int k = 0; k < 6; k++) params[k] = out[k];
    if (host_params) {
#pragma unroll
        for (int k = 0; k < 6; k++) host_params[k] = out[k];
    }
}

template <bool CHECK>
__global__ void __launch_bounds__(kP3Threads) fit_value_kernel3(const Fit3Args a) {
    const uint32_t plane_i = blockIdx.y;
    const int32_t *const plane = a.coefs + plane_i * a.coef_stride;
    unsigned long long *const accp = a.acc + (size_t)plane_i * kFitShards * kFitAccWords;
    unsigned long long *const accs = accp + (size_t)(blockIdx.x % kFitShards) * kFitAccWords;
    __shared__ __attribute__((aligned(16))) F3Lds lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    trace_stamp(a.trace, blockIdx.x, 0, tid);
    if (tid < 3 * 28) (&lds.s_int[0][0])[tid] = 0;
    if (tid == 0) lds.range = 0;
    for (int i = tid; i < 2 * (kF3ImageBytes - kP3ZeroOff) / 4; i += kP3Threads) // the zero words behind the cells of both images
        reinterpret_cast<uint32_t *>(lds.cells[0] + (i / ((kF3ImageBytes - kP3ZeroOff) / 4)) * kF3ImageBytes + kP3ZeroOff)[i % ((kF3ImageBytes - kP3ZeroOff) / 4)] = 0;
    if (wave & 1)
        f3_run<1, CHECK>(a, plane, lds, tid, lane, wave);
    else
        f3_run<0, CHECK>(a, plane, lds, tid, lane, wave);
    __syncthreads();
    trace_stamp(a.trace, blockIdx.x, 13, tid);
    // hand-over as in fit_accumulate_kernel2: adds into this workgroup's copy of the plane's accumulator, a ticket, the last workgroup sums the copies
    if (tid < kF3AccInt) __hip_atomic_fetch_add(accs + tid, (&lds.s_int[0][0])[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0 && lds.range) __hip_atomic_fetch_add(accp + kF3AccRange, (unsigned long long)lds.range, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wait_for_own_memory_ops_then_barrier();
    if (tid == 0) lds.flag = __hip_atomic_fetch_add(accp + kF3AccTicket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    trace_exit(a.trace, blockIdx.x, tid);
    if (lds.flag == 0 || tid >= 64) return;
    unsigned long long *const out_int = a.gram + (size_t)plane_i * kF3AccInt;
    for (int i = tid; i < kF3AccInt; i += 64) {
        unsigned long long part[kFitShards], sum = 0;
#pragma unroll
        for (uint32_t sh = 0; sh < kFitShards; sh++) part[sh] = __hip_atomic_load(accp + sh * kFitAccWords + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (uint32_t sh = 0; sh < kFitShards; sh++) {
            sum += part[sh];
            __hip_atomic_store(accp + sh * kFitAccWords + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        out_int[i] = sum;
        (&lds.s_int[0][0])[i] = sum;
    }
    if (tid == 0) {
        const unsigned long long r = __hip_atomic_exchange(accp + kF3AccRange, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.out_range) a.out_range[plane_i] = r;
        if (a.host_range) a.host_range[plane_i] = r;
        __hip_atomic_store(accp + kF3AccTicket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!a.solve_params) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tid < 3) {
        const size_t at = (size_t)plane_i * (sizeof(PredictParams) / sizeof(float)) + tid * 6;
        f3_tail_solve(reinterpret_cast<const long long *>(lds.s_int[tid]), &lds.work[tid], a.solve_params + at, a.host_params ? a.host_params + at : nullptr);
    }
}

} // namespace

hipError_t launch_fit_value3(const DevicePlan &p, unsigned long long *acc, const PredBatch &b, unsigned long long *sums_int, unsigned long long *out_of_range, hipStream_t stream,
                             const FitSolve *solve) {
    if (!acc || !sums_int || !b.n_planes || b.n_planes > 65535u) return hipErrorInvalidValue;
    Fit3Args a{};
    a.coefs = b.coefs;
    a.coef_stride = b.coef_stride;
    a.pred_slots = p.pred_slots;
    a.gather_off = p.gather_off;
    a.pair_pos = p.pair_pos;
    a.halo_list = p.halo_list;
    a.valid_mask = p.valid_mask;
    a.n_tiles = p.n_pred_tiles;
    a.acc = acc;
    a.gram = sums_int;
    a.out_range = out_of_range;
    a.trace = p.trace;
    if (solve) {
        if (!solve->params) return hipErrorInvalidValue;
        a.solve_params = solve->params, a.host_params = solve->host_params, a.host_range = solve->host_range;
    }
    // the grid of launch_predict_histogram: one plane = a workgroup per CU; many planes = an eighth of the machine each, eight side by side
    uint32_t blocks = p.n_pred_tiles < p.pred_blocks ? p.n_pred_tiles : p.pred_blocks;
    if (b.n_planes > 1) {
        const uint32_t share = (p.n_pred_tiles + 7) / 8, eighth = p.pred_blocks / 8 ? p.pred_blocks / 8 : 1;
        blocks = share < eighth ? eighth : share;
        if (blocks > p.pred_blocks) blocks = p.pred_blocks;
        if (blocks > p.n_pred_tiles) blocks = p.n_pred_tiles;
    }
    if (!blocks) blocks = 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL((fit_value_kernel3<true>), dim3(blocks, b.n_planes), dim3(kP3Threads), 0, stream, a);
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

// The permuted 1 KiB cell layout of kernel3 (and of the fit kernels once they move to it): pair_pos from gather_layout.inc, its
// inverse, and per node the six neighbour offsets in bytes from the own slot (0x7FFF = "never a node": the image's zero word).
void build_gather_tables(const uint16_t *nbr_table, uint32_t *gather_off /* [512][4] */, uint16_t *pair_pos /* [256] */, uint16_t *heap_of_pos /* [512] */) {
    static const uint16_t kPairPos[256] = {
#include "gather_layout.inc"
    };
    for (int q = 0; q < 256; q++) {
        pair_pos[q] = kPairPos[q];
        heap_of_pos[2 * kPairPos[q]] = (uint16_t)(2 * q);
        heap_of_pos[2 * kPairPos[q] + 1] = (uint16_t)(2 * q + 1);
    }
    for (int p = 0; p < kCell; p++) {
        uint32_t h[6];
        for (int k = 0; k < 6; k++) {
            const uint32_t e = nbr_table[p * 6 + k];
            const int slot = (e >> 9) & 7; // index into {self, +V9[0..5]} = lattice deltas (0,0),(1,0),(1,-1),(0,-1),(-1,0),(-1,1),(0,1)
            const int da = (int)((0x0F14u >> (2 * slot)) & 3u), db = (int)((0x14F0u >> (2 * slot)) & 3u); // 2-bit fields: 0 -> 0, 1 -> +1, 3 -> -1
            const int sa = (da & 1) - (da & 2), sb = (db & 1) - (db & 2);
            const int heap = (int)(e & 511u);
            const int rel = (sa * kPredSide + sb) * 1024 + 2 * (2 * kPairPos[heap >> 1] + (heap & 1));
            h[k] = (e & 0x8000u) ? 0x7FFFu : ((uint32_t)rel & 0xFFFFu);
        }
        gather_off[4 * p] = h[0] | (h[1] << 16), gather_off[4 * p + 1] = h[2] | (h[3] << 16), gather_off[4 * p + 2] = h[4] | (h[5] << 16), gather_off[4 * p + 3] = 0;
    }
}

// Which values of the 20 halo cells of a tile are ever read. A gather that leaves the own cell goes to one of the six lattice neighbours, and
// the static neighbour table says where: per direction only 21..46 of the neighbour's 512 nodes (its rim towards this cell) are ever a
// neighbour position. For a 4 x 4 block that makes 902 (halo slot, heap node) pairs in 18 of the 20 halo slots - round 2 staged all
// 20 x 512 = 10 240 values per tile, 40 of a tile's 72 KB of loads and ~55 of its ~85 staging instructions per lane. One entry per thread of the
// 1024-thread workgroup: out[t] = slot | heap << 8 | (byte position inside the slot) << 20, sorted by (slot, heap) so that neighbouring lanes read
// neighbouring addresses. out[0] = 0xFFFFFFFF: the list does not fit (never with the reference's LITERALS).
void build_halo_list(const uint16_t *nbr_table, const uint16_t *pair_pos, uint32_t *out /* [kP3Threads] */) {
    std::vector<uint32_t> keys;
    for (int r = 1; r <= kPredBlock; r++)
        for (int c = 1; c <= kPredBlock; c++)
            for (int p = 2; p < kCell; p++) // heap nodes 0 and 1 are the LF predictor's, which reads global memory (p3_lf_*)
                for (int k = 0; k < 6; k++) {
                    const uint32_t e = nbr_table[p * 6 + k];
                    const int slot7 = (e >> 9) & 7;
                    if ((e & 0x8000u) || slot7 == 0) continue;
                    const int da = (int)((0x0F14u >> (2 * slot7)) & 3u), db = (int)((0x14F0u >> (2 * slot7)) & 3u); // as in build_gather_tables
                    const int rr = r + ((da & 1) - (da & 2)), cc = c + ((db & 1) - (db & 2));
                    if (rr >= 1 && rr <= kPredBlock && cc >= 1 && cc <= kPredBlock) continue; // another block cell: staged whole by its own waves
                    keys.push_back((uint32_t)(rr * kPredSide + cc) << 16 | (e & 511u));
                }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    out[0] = 0xFFFFFFFFu;
    if (keys.size() > (size_t)kP3Threads) return; // (cannot happen with the reference's LITERALS: 902 entries; the plan checks the table it uploads)
    // threads without an entry stage a node of slot 0 - a corner of the 6 x 6 window, which no gather ever reads - each a node (and a halfword) of its own
    for (size_t t = keys.size(); t < (size_t)kP3Threads; t++) out[t] = (uint32_t)((t - keys.size()) & 511u) << 8 | (uint32_t)(2 * (t - keys.size())) << 20;
    for (size_t i = 0; i < keys.size(); i++) {
        const uint32_t slot = keys[i] >> 16, heap = keys[i] & 511u;
        const uint32_t pos = 2u * (2u * pair_pos[heap >> 1] + (heap & 1u));
        out[i] = slot | heap << 8 | pos << 20;
    }
}

hipError_t launch_predict_histogram(const DevicePlan &p, uint32_t *acc, const PredBatch &b, uint8_t *bucket, int32_t *prediction, uint32_t *hist, unsigned long long *n_oob,
                                    int trust, hipStream_t stream) {
    if (!acc || !b.n_planes || b.n_planes > 65535u) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    PredArgs a{};
    a.acc = acc;
    a.coefs = b.coefs;
    a.coef_stride = b.coef_stride;
    a.out_stride = b.out_stride;
    a.params = b.params;
    for (int k = 0; k < 3; k++) a.pp3[k] = b.pp[k];
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
    a.trusted = trust != kPredAnyInt32 ? 1 : 0;
    a.words = b.words;
    if (b.words && (trust != kPredForwardOutput || p.k2_previous)) return hipErrorInvalidValue; // the halfword form exists for the chain's own coefficients only
    // One plane: a workgroup per CU. Many planes: a plane keeps an eighth of the machine busy (at least ~8 tiles per workgroup, so that
    // the start-up and the hand-over are paid once per 8 tiles) and eight planes run side by side.
    uint32_t blocks = p.n_pred_tiles < p.pred_blocks ? p.n_pred_tiles : p.pred_blocks;
    if (b.n_planes > 1) {
        const uint32_t share = (p.n_pred_tiles + 7) / 8, eighth = p.pred_blocks / 8 ? p.pred_blocks / 8 : 1;
        blocks = share < eighth ? eighth : share;
        if (blocks > p.pred_blocks) blocks = p.pred_blocks;
        if (blocks > p.n_pred_tiles) blocks = p.n_pred_tiles;
    }
    if (!blocks) blocks = 1;
    a.junk = p.junk;
    a.trace = p.trace;
    if (p.k2_previous) { // FRI_HIP_TUNING=1 FRI_HIP_K2_PREVIOUS=1: the pipelined kernel of round 1 (A/B on one box); one plane, both outputs
        if (!bucket || !prediction || b.n_planes != 1) return hipErrorInvalidValue;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(predict_histogram_kernel2), hipFuncAttributeMaxDynamicSharedMemorySize, kPred2LdsBytes);
        if (e != hipSuccess) return e;
        (void)hipGetLastError(); // the check behind the launch must not pick up an error an earlier, unrelated call left behind
        hipLaunchKernelGGL(predict_histogram_kernel2, dim3(blocks), dim3(kPred2Threads), kPred2LdsBytes, stream, a);
        return hipGetLastError();
    }
    a.pred_off = p.gather_off;
    a.pair_pos = p.pair_pos;
    a.heap_of_pos = p.heap_of_pos;
    a.halo_list = p.halo_list;
    a.ablate = p.k2_ablate;
    (void)hipGetLastError(); // the check behind the launch must not pick up an error an earlier, unrelated call left behind
    if (b.words)
        hipLaunchKernelGGL((predict_histogram_kernel3<false, true>), dim3(blocks, b.n_planes), dim3(kP3Threads), 0, stream, a);
    else if (trust == kPredForwardOutput)
        hipLaunchKernelGGL((predict_histogram_kernel3<false, false>), dim3(blocks, b.n_planes), dim3(kP3Threads), 0, stream, a);
    else
        hipLaunchKernelGGL((predict_histogram_kernel3<true, false>), dim3(blocks, b.n_planes), dim3(kP3Threads), 0, stream, a);
    e = hipGetLastError();
    if (e != hipSuccess || trust != kPredAnyInt32) return e; // the forward kernel's coefficients are differences of 8-bit pixels divided by a quantiser: always representable
    ExactArgs x{};
    x.coefs = b.coefs;
    x.coef_stride = b.coef_stride;
    x.out_stride = b.out_stride;
    x.params = b.params;
    for (int k = 0; k < 3; k++) x.pp3[k] = b.pp[k];
    x.nbr_table = p.nbr_table;
    x.nbr_cells = p.nbr_cells;
    x.interior = p.interior;
    x.valid_mask = p.valid_mask;
    x.bucket = bucket;
    x.prediction = prediction;
    x.hist = hist;
    x.n_oob = n_oob;
    x.acc = acc;
    x.F = p.F;
    const uint32_t xb = p.F < blocks ? p.F : blocks;
    (void)hipGetLastError(); // the check behind the launch must not pick up an error an earlier, unrelated call left behind
    hipLaunchKernelGGL(exact_predict_kernel, dim3(xb ? xb : 1, b.n_planes), dim3(kCell), 0, stream, x);
    return hipGetLastError();
}

} // namespace fri
