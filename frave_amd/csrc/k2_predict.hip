// k2_predict.hip -- K2 predict_histogram: 6-neighbour gather + context bucket + prediction + ANS symbol histogram for one
// channel plane (context_modeling.rs:25-77; stages/prediction.rs:86-207, 237-298).
#include "gather_common.hpp"

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

constexpr int kLfNever = 127;
struct PredArgs {
    const int32_t *coefs;      // one channel plane [F][512]
    const int32_t *pred_slots; // [n_tiles][kPredSlots]
    const uint16_t *nbr_table; // [512][6]
    const uint32_t *pred_off;  // [512][4] packed neighbour offsets of every node: bytes from the own slot in the permuted 1 KiB layout (build_gather_tables)
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
    uint32_t *acc;             // plan scratch, kPredAccWords per plane: the clearing workgroups' flags (serial numbers), the CHECK kernels' ticket, the inexact flag
    int8_t lf_delta[8];        // slot-list offsets of the neighbour cells (left, up-left, up-right) of heap nodes 0 and 1, kLfNever = the position is never a node
    uint32_t serial;           // this launch's number on this accumulator (never 0): what the clearing workgroups publish and everybody polls for
    uint32_t n_tiles;
    uint16_t *words;      // predict_histogram_kernel3<., true>: bucket << 10 | symbol per node, [n_planes] planes out_stride apart, INSTEAD of bucket / prediction
    const uint32_t *stream_pos; // STREAM form (round 5): [F][512] position of every node in a channel's symbol stream (the inverse of the stream order; None nodes: unused) -
                          // `words` is then the plane's STREAM (out_stride = distance of two planes' streams) and a node's halfword goes straight to its place in it
    int32_t trusted;      // the caller vouches for |coefficient| <= 256 (this library's forward kernel wrote them) and enqueues no exact kernel: a plane that
                          // raises `inexact` all the same reports n_oob = ~0 (an error the host maps to FRI_HIP_ERR_OUT_OF_RANGE) and lowers the flag itself
    PredictParams pp;     // this plane's parameters (filled per plane inside the kernel)
    PredictParams pp3[3]; // plane k < 3 of a launch without a params array
};

// Histogram hand-over (round 4). No memset in front of the kernel, no plan-side copy of the table, no copy-out behind it:
//  * the first min(gridDim.x, 10) workgroups of a plane clear the caller's hist[10][1024] (and workgroup 0 *n_oob) in their prologue - atomic exchanges,
//    i.e. read-modify-writes performed at the same coherence point as the adds that follow - and each then publishes the launch's serial number in its
//    own flag word of the plane's accumulator (pred_clear_outputs);
//  * at its end every workgroup polls those flags (they were raised tens of microseconds earlier: one round trip, hidden behind the last tile's stores) and
//    then adds its LDS table straight into hist / *n_oob with fire-and-forget device-scope atomics. A workgroup only ever waits for workgroups with a LOWER
//    block index of its own plane, which the dispatcher started before it and which wait for nobody in their prologue: the poll terminates whatever the grid,
//    whatever else shares the device. The serial number (one per launch on this accumulator, host-side counter) makes a reset of the flags unnecessary.
//  * CHECK = false (the chain's own coefficients): that is all - the end of the kernel completes the adds. Rounds 1-3 added into a plan accumulator and let the
//    workgroup with the last ticket copy the totals out and re-zero: adds -> their acknowledgement -> ticket -> 10 loads -> 20 stores per thread, 6-7 us on the
//    critical path of the workgroup that was last anyway.
//  * CHECK = true keeps a ticket: a plane whose values the LDS image cannot hold (`inexact`) must end with an all-zero histogram (the exact kernel behind it
//    adds the real one) or, for a caller who vouched for the coefficients, with n_oob = ~0: the workgroup with the last ticket does that, after every add.
constexpr int kAccZeroFlag = 0, kAccTicket = 12, kAccInexact = 14; // + the exact kernel's ticket at kAccInexact + 1
constexpr uint32_t kPredZeroBlocks = 10;                           // hist chunks of 1024 counters, one per clearing workgroup
static_assert(kAccInexact + 2 == (int)kPredAccWords && kAccZeroFlag + (int)kPredZeroBlocks <= kAccTicket, "accumulator layout");
// (issue only: the exchanges fly while the prologue makes its own round trips; pred_clear_publish raises the flag behind the prologue's first barrier, whose
// wait covers them - the clearing workgroups walk the longest tile sequences of the launch, a round trip of their own in front would be the launch's)
__device__ __forceinline__ void pred_clear_issue(const PredArgs &a, int tid, int n_threads) {
    const uint32_t nz = min(gridDim.x, kPredZeroBlocks);
    if (blockIdx.x >= nz) return; // (uniform)
    for (uint32_t c = blockIdx.x; c < 10u; c += nz)
        for (uint32_t j = (uint32_t)tid; j < 1024u; j += (uint32_t)n_threads) (void)__hip_atomic_exchange(a.hist + c * 1024u + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0 && tid == 0) (void)__hip_atomic_exchange(a.n_oob, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Behind a wait_for_own_memory_ops_then_barrier() of the whole workgroup: every clearing exchange has been performed, the flag may go up.
__device__ __forceinline__ void pred_clear_publish(const PredArgs &a, int tid) {
    if (blockIdx.x < min(gridDim.x, kPredZeroBlocks) && tid == 0)
        (void)__hip_atomic_exchange(a.acc + kAccZeroFlag + blockIdx.x, a.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One poll of the plane's flags (lane z < number of clearing workgroups reads flag z; the other lanes read as "cleared")
__device__ __forceinline__ uint32_t pred_clear_poll(const PredArgs &a, int lane) {
    return (uint32_t)lane < min(gridDim.x, kPredZeroBlocks) ? __hip_atomic_load(a.acc + kAccZeroFlag + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.serial;
}
template <bool CHECK>
__device__ __forceinline__ void pred_hand_over(const PredArgs &a, const uint32_t *s_hist, uint32_t *s_flag, int tid, int n_threads, uint32_t early_poll) {
    // (the caller's LDS barrier behind the tile loop has made s_hist complete; the output stores of the last tile are still in flight - nothing here waits for them)
    // early_poll: this wave's poll from the end of its prologue (the flags go up ~3 us into the launch): almost always the answer, and its round trip is long over
    if (!__all(early_poll == a.serial)) {
        const int lane = tid & 63;
        while (!__all(pred_clear_poll(a, lane) == a.serial)) __builtin_amdgcn_s_sleep(2); // every wave for itself: no barrier between the poll and the adds
    }
    asm volatile("" ::: "memory"); // the adds below stay behind the poll
    for (int i = tid; i < kHistBins; i += n_threads) {
        const uint32_t c = s_hist[i];
        if (c) (void)__hip_atomic_fetch_add(a.hist + i, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0 && s_hist[kHistBins]) (void)__hip_atomic_fetch_add(a.n_oob, (unsigned long long)s_hist[kHistBins], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!CHECK) return;
    // Order without fences (an agent-scope fence on this multi-XCD part writes back and invalidates the whole L2: +80 us per launch): the adds are device-scope
    // atomics, executed at the coherence point and acknowledged through vmcnt; every wave waits for its own vmcnt(0) - explicitly, see
    // wait_for_own_memory_ops_then_barrier - so all of this workgroup's adds are performed before thread 0 draws the ticket.
    wait_for_own_memory_ops_then_barrier();
    if (tid == 0) *s_flag = __hip_atomic_fetch_add(a.acc + kAccTicket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (*s_flag == 0) return;
    // all other workgroups of the plane have finished (their adds precede their tickets)
    const bool inexact = __hip_atomic_load(a.inexact, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    if (inexact) {
        for (int i = tid; i < kHistBins; i += n_threads) (void)__hip_atomic_exchange(a.hist + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            (void)__hip_atomic_exchange(a.n_oob, a.trusted ? ~0ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.trusted) __hip_atomic_store(a.inexact, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // no exact kernel follows to lower it
        }
    }
    if (tid == 0) __hip_atomic_store(a.acc + kAccTicket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
// STREAM (with WORDS, round 5): wd is the plane's symbol stream and pos[j] the place of the lane's j-th node in it (PredArgs::stream_pos): the halfwords of the Some nodes
// go there directly - four two-byte stores under the nodes' masks instead of one or two wide ones - and neither node-word plane nor gather kernel exist.
template <int IMG, int ROLE, int CELL, bool WORDS, bool STREAM = false>
__device__ __forceinline__ void p3_half(const uint32_t (&addr)[4][6], float (&ga)[6], float (&gb)[6], const int (&own)[4], const PredictParams &pp, bool interior, uint32_t some4,
                                        int lane, uint32_t *s_hist, const uint16_t *s_bkt, uint8_t *bd, int32_t *pd, uint16_t *wd, uint8_t *junk, int ablate,
                                        const uint32_t (&pos)[4] = {0, 0, 0, 0}, bool has = true) {
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
        if (WORDS && STREAM) {
            if (!has) return; // (a block slot without a retained cell: nothing belongs to the stream)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool lf = ROLE == 0 && j < 2 && lane == 0;                 // heap nodes 0 and 1: the LF pass writes them
                const bool some = interior ? !lf : ((some4 >> j) & 1u) != 0;      // (the LF nodes' bits of some4 are already cleared)
                if (some && !(ablate & 32)) wd[pos[j]] = (uint16_t)(bin[j] >> 2);  // bin = 4 x (bucket << 10 | symbol), out of alphabet: 4 x kHistBins
            }
            return;
        }
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
    // where the three neighbour CELLS of heap nodes 0 / 1 sit in the slot list is geometry (rows 0 and 1 of the static neighbour table): kernel arguments
    // (lf_delta, launch_predict_histogram) since round 4 - as a load of the table it was a dependent round trip in front of the slot-list loads, in the prologue
    // of every workgroup
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int d0 = a.lf_delta[k], d1 = a.lf_delta[3 + k]; // (static indices; selected per lane)
        const int delta = it.node ? d1 : d0;
        const int r = row[slot + (delta == kLfNever ? 0 : delta)];
        it.nb[k] = delta == kLfNever ? -1 : r;
    }
    if (!active) it.raw = -1;
}
struct P3LfValues {
    int v[3], value;
    uint32_t mask0;
};
template <bool C16 = false>
__device__ __forceinline__ void p3_lf_hop_b(const PredArgs &a, const int32_t *plane, const P3LfItem &it, P3LfValues &x) {
    const uint32_t n = (uint32_t)it.node;
    auto at = [&](int cell) -> int { return C16 ? (int)(reinterpret_cast<const int16_t *>(plane) + (size_t)cell * kCell)[n] : (plane + (size_t)cell * kCell)[n]; };
#pragma unroll
    for (int k = 0; k < 3; k++) x.v[k] = at(max(pred_slot_cell(it.nb[k]), 0)); // unconditional loads: an absent neighbour reads cell 0 and is zeroed below
    const int cell = max(pred_slot_cell(it.raw), 0);
    x.value = at(cell);
    x.mask0 = a.valid_mask[(size_t)cell * 16];
}
template <bool WORDS, bool STREAM = false>
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
    if (WORDS && STREAM) {
        if (some) a.words[a.stream_pos[(size_t)cell * kCell + it.node]] = (uint16_t)counter;
        return;
    }
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
        a.acc += (size_t)plane * kPredAccWords;
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
// C16 (round 5, here and in p3_load_own / p3_lf_hop_b): `plane` is a chain's compact plane - int16, None as 0 (k1_forward.hip, store_item<.., C16>); the None tests
// further down never fire on such values and need no second form.
template <bool C16 = false>
__device__ __forceinline__ void p3_issue_halo(const int32_t *plane, const int32_t *slots, uint32_t ring_off, uint32_t heap_off, P3Halo &h) {
    h.raw = *reinterpret_cast<const int32_t *>(reinterpret_cast<const uint8_t *>(slots) + ring_off);
    const uint32_t cell = (uint32_t)max(h.raw, 0) & (uint32_t)(kPredSlotInterior - 1); // a slot without a cell reads cell 0 and is zeroed at the commit
    if (C16)
        h.v = *reinterpret_cast<const int16_t *>(reinterpret_cast<const uint8_t *>(plane) + ((cell << 10) + (heap_off >> 1)));
    else
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
// C16: the four values arrive as two dwords of two int16 each (.x, .y; p3_unpack_own turns them into the four int32 once the loads have landed).
template <int ROLE, bool C16 = false>
__device__ __forceinline__ i32x4 p3_load_own(const int32_t *plane, size_t cell, uint32_t lane) {
    if (C16) {
        const int16_t *cell_base = reinterpret_cast<const int16_t *>(plane) + cell * kCell;
        if (ROLE) {
            const i32x2 d = reinterpret_cast<const i32x2 *>(cell_base + 256)[lane];
            return i32x4{d.x, d.y, 0, 0};
        }
        return i32x4{reinterpret_cast<const int32_t *>(cell_base)[lane], reinterpret_cast<const int32_t *>(cell_base + 128)[lane], 0, 0};
    }
    const int32_t *cell_base = plane + cell * kCell;
    if (ROLE) return reinterpret_cast<const i32x4 *>(cell_base + 256)[lane];
    const i32x2 a = reinterpret_cast<const i32x2 *>(cell_base)[lane], b = reinterpret_cast<const i32x2 *>(cell_base + 128)[lane];
    return i32x4{a.x, a.y, b.x, b.y};
}
template <bool C16>
__device__ __forceinline__ void p3_unpack_own(i32x4 &v) {
    if (C16) v = i32x4{(int)(short)v.x, v.x >> 16, (int)(short)v.y, v.y >> 16};
}
// the stream positions of the lane's four nodes of a cell (PredArgs::stream_pos), laid out like its own values
template <int ROLE>
__device__ __forceinline__ u32x4 p3_load_pos(const uint32_t *stream_pos, size_t cell, uint32_t lane) {
    const uint32_t *base = stream_pos + cell * kCell;
    if (ROLE) return reinterpret_cast<const u32x4 *>(base + 256)[lane];
    const i32x2 a = reinterpret_cast<const i32x2 *>(base)[lane], b = reinterpret_cast<const i32x2 *>(base + 128)[lane];
    return u32x4{(uint32_t)a.x, (uint32_t)a.y, (uint32_t)b.x, (uint32_t)b.y};
}

// OWN_CUR / OWN_NXT: the lane's own values (exact int32, for the residuals) of this tile's two cells and, loaded here, of the next
// tile's; the two register sets swap roles from tile to tile (IMG), so nothing is copied.
// STREAM: pos_cur / pos_nxt - the stream positions of the lane's nodes in this tile's two cells and, loaded here next to the own values, in the next tile's.
template <int IMG, int ROLE, bool CHECK, bool WORDS, bool C16 = false, bool STREAM = false>
__device__ __forceinline__ void p3_tile(const PredArgs &a, const int32_t *plane, P3Lds &lds, int it, bool more, uint32_t next2_tile, int tid, int lane, int wave, int slot_a,
                                        const P3Lane &L, float (&ga)[6], float (&gb)[6], const i32x4 (&own_cur)[2], i32x4 (&own_nxt)[2], const u32x4 (&pos_cur)[2],
                                        u32x4 (&pos_nxt)[2]) {
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
            own_nxt[c] = p3_load_own<ROLE, C16>(plane, (size_t)cell, (uint32_t)lane);
            if (STREAM) pos_nxt[c] = p3_load_pos<ROLE>(a.stream_pos, (size_t)cell, (uint32_t)lane);
            if (ROLE == 1) st_own_mask[c] = (a.valid_mask + (size_t)cell * 16)[(uint32_t)lane & 15u]; // the level-8 waves have registers to spare
        }
        p3_issue_halo<C16>(plane, nxt_slots, L.halo_ring, L.halo_heap, st_halo);
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
        uint16_t *wd = WORDS && STREAM ? a.words : WORDS && has ? a.words + (size_t)cell * kCell : reinterpret_cast<uint16_t *>(a.junk + junk + 512);
        const uint32_t pos4[4] = {pos_cur[c].x, pos_cur[c].y, pos_cur[c].z, pos_cur[c].w};
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
            p3_half<IMG, ROLE, 0, WORDS, STREAM>(L.addr, ga, gb, own4, a.pp, interior, some4, lane, s_hist, s_bkt, bd, pd, wd, a.junk + junk, ablate_flags(a.ablate), pos4, has);
        } else {
            p3_half<IMG, ROLE, 1, WORDS, STREAM>(L.addr, ga, gb, own4, a.pp, interior, some4, lane, s_hist, s_bkt, bd, pd, wd, a.junk + junk, ablate_flags(a.ablate), pos4, has);
        }
    }

    if (more) {
        // The staged registers are consumed from here on, not earlier: left alone, the compiler hoists uses of the next tile's own
        // values to the top of the iteration and waits for the loads there - in front of the arithmetic they are meant to hide behind.
        asm volatile("" : "+v"(own_nxt[0]), "+v"(own_nxt[1]), "+v"(st_halo.v));
        if (STREAM) asm volatile("" : "+v"(pos_nxt[0]), "+v"(pos_nxt[1]));
        p3_unpack_own<C16>(own_nxt[0]), p3_unpack_own<C16>(own_nxt[1]);
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

template <int ROLE, bool CHECK, bool WORDS, bool C16 = false, bool STREAM = false>
__device__ __forceinline__ void p3_run(const PredArgs &a, P3Lds &lds, int tid, int lane, int wave, uint32_t &early_poll) {
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
    u32x4 pos_a[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}, pos_b[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}; // STREAM: see p3_tile

    const PredTileWalk walk(a.n_tiles);
    if (walk.first >= walk.end) { // (a workgroup without a tile still takes part in the hand-over - and clears its part of the histogram if it is one of the first ten)
        wait_for_own_memory_ops_then_barrier();
        pred_clear_publish(a, tid);
        return;
    }
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
    wait_for_own_memory_ops_then_barrier(); // (also the clearing workgroups' exchanges: pred_clear_issue)
    pred_clear_publish(a, tid);
    { // tile 0 straight into image 0: everything a wave stages is requested before the first value is converted
        uint32_t st_own_mask[2] = {0, 0};
        int raw_own[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            raw_own[c] = __builtin_amdgcn_readfirstlane(s_ring[slot_a + c]);
            const int cell = max(pred_slot_cell(raw_own[c]), 0);
            own_a[c] = p3_load_own<ROLE, C16>(plane, (size_t)cell, (uint32_t)lane);
            if (STREAM) pos_a[c] = p3_load_pos<ROLE>(a.stream_pos, (size_t)cell, (uint32_t)lane);
            if (ROLE == 1) st_own_mask[c] = (a.valid_mask + (size_t)cell * 16)[(uint32_t)lane & 15u];
        }
        P3Halo st_halo;
        p3_issue_halo<C16>(plane, s_ring, L.halo_ring, L.halo_heap, st_halo);
        P3LfValues lf_values;
        if (ROLE == 1) p3_lf_hop_b<C16>(a, plane, lf_item, lf_values); // behind the staging loads: one round trip for both
        p3_unpack_own<C16>(own_a[0]), p3_unpack_own<C16>(own_a[1]);
        float m = 0.f;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            m = __builtin_fmaxf(m, p3_commit4<CHECK>(raw_own[c], own_a[c], s_cells + (slot_a + c) * kP3SlotBytes, L.opos[0], L.opos[1]));
            if (ROLE == 1 && lane < 16) s_masks[(slot_a + c) * 16 + lane] = st_own_mask[c];
        }
        m = __builtin_fmaxf(m, p3_commit_halo<CHECK>(st_halo, s_cells, L.halo_lds));
        if (CHECK) p3_check(m, lane, a.inexact);
        if (ROLE == 1 && !(ablate_flags(a.ablate) & 1)) p3_lf_finish<WORDS, STREAM>(a, lds.hist, lf_item, lf_values);
    }
    if (ROLE == 1) {
        for (uint32_t base = kLfTilesPerPass; base < my_tiles; base += kLfTilesPerPass) { // more than 16 tiles per workgroup (large images): further passes, two round trips each
            const uint32_t k = base + (uint32_t)(lf_tid >> 5);
            P3LfItem it;
            P3LfValues x;
            p3_lf_hop_a(a, walk.first + k * walk.step, k < my_tiles, lf_tid, it);
            p3_lf_hop_b<C16>(a, plane, it, x);
            if (!(ablate_flags(a.ablate) & 1)) p3_lf_finish<WORDS, STREAM>(a, lds.hist, it, x);
        }
    }
    // Only LDS is handed over here. __syncthreads() also waits for the LF pass's scattered global stores to be acknowledged: 1.8 us between "the first
    // tile's data has landed" and "prologue done" by the time stamps, most of it that wait.
    lds_barrier();
    trace_stamp(a.trace, blockIdx.x, 1, tid);
    early_poll = pred_clear_poll(a, lane); // consumed by pred_hand_over, a whole tile loop later

    int it = 0;
    for (uint32_t tile = walk.first; tile < walk.end;) { // unrolled by two: the LDS image a tile lives in is a compile-time constant
        p3_tile<0, ROLE, CHECK, WORDS, C16, STREAM>(a, plane, lds, it, tile + walk.step < walk.end, min(tile + 2 * walk.step, last), tid, lane, wave, slot_a, L, ga, gb, own_a, own_b, pos_a, pos_b);
        tile += walk.step, it++;
        if (tile >= walk.end) break;
        p3_tile<1, ROLE, CHECK, WORDS, C16, STREAM>(a, plane, lds, it, tile + walk.step < walk.end, min(tile + 2 * walk.step, last), tid, lane, wave, slot_a, L, ga, gb, own_b, own_a, pos_b, pos_a);
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
    a.acc += (size_t)plane * kPredAccWords;
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
template <bool CHECK, bool WORDS, bool C16 = false, bool STREAM = false>
__global__ void __launch_bounds__(kP3Threads) predict_histogram_kernel3(const PredArgs a0) {
    PredArgs a = pred_plane_view(a0, blockIdx.y);
    if (C16) a.coefs = reinterpret_cast<const int32_t *>(reinterpret_cast<const int16_t *>(a0.coefs) + blockIdx.y * a0.coef_stride); // (a compact plane: halfwords)
    __shared__ __attribute__((aligned(16))) P3Lds lds;
    uint32_t *s_hist = lds.hist;
    uint8_t *s_cells = lds.cells[0];
    int32_t *s_ring = &lds.ring[0][0];
    uint16_t *s_bkt = lds.bkt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    trace_stamp(a.trace, blockIdx.x, 0, tid);
    pred_clear_issue(a, tid, kP3Threads); // the first ten workgroups of the plane: the caller's histogram starts at zero (see pred_hand_over)
    for (int i = tid; i < kHistBins + 4; i += kP3Threads) s_hist[i] = 0;
    if (tid < 32) s_bkt[tid] = (uint16_t)(bucket_of((uint32_t)tid) << 12);
    for (int i = tid; i < 2 * (kP3ImageBytes - kP3ZeroOff) / 4; i += kP3Threads) // the zero words behind the cells of both images
        reinterpret_cast<uint32_t *>(s_cells + (i / ((kP3ImageBytes - kP3ZeroOff) / 4)) * kP3ImageBytes + kP3ZeroOff)[i % ((kP3ImageBytes - kP3ZeroOff) / 4)] = 0;
    uint32_t early_poll = 0;
    if (wave & 1)
        p3_run<1, CHECK, WORDS, C16, STREAM>(a, lds, tid, lane, wave, early_poll);
    else
        p3_run<0, CHECK, WORDS, C16, STREAM>(a, lds, tid, lane, wave, early_poll);
    lds_barrier(); // the table is complete; nobody waits here for the last tile's output stores
    trace_stamp(a.trace, blockIdx.x, 13, tid);
    pred_hand_over<CHECK>(a, s_hist, reinterpret_cast<uint32_t *>(s_ring), tid, kP3Threads, early_poll);
    trace_exit(a.trace, blockIdx.x, tid);
}

} // namespace


// Slot-list offsets (in a tile's 6 x 6 slot list) of the three neighbour cells the low-frequency predictor reads for heap nodes 0 and 1 (p3_lf_hop_a)
void build_lf_deltas(const uint16_t *nbr_table, int8_t *out /* [8] */) {
    for (int node = 0; node < 2; node++)
        for (int k = 0; k < 3; k++) {
            const uint32_t e = nbr_table[node * 6 + k];
            const int s7 = (e >> 9) & 7; // index into {self, +V9[0..5]}: lattice deltas as in build_gather_tables
            const int da = (int)((0x0F14u >> (2 * s7)) & 3u), db = (int)((0x14F0u >> (2 * s7)) & 3u);
            out[node * 3 + k] = (e & 0x8000u) ? (int8_t)kLfNever : (int8_t)(((da & 1) - (da & 2)) * kPredSide + ((db & 1) - (db & 2)));
        }
    out[6] = out[7] = 0;
}

// The permuted 1 KiB cell layout of kernel3 and - since round 4 - of the fit kernels: pair_pos from gather_layout.inc, its
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

hipError_t launch_predict_histogram(const DevicePlan &p, uint32_t *acc, uint32_t serial, const PredBatch &b, uint8_t *bucket, int32_t *prediction, uint32_t *hist,
                                    unsigned long long *n_oob, int trust, hipStream_t stream) {
    if (!acc || !serial || !b.n_planes || b.n_planes > 65535u) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    PredArgs a{};
    a.acc = acc;
    a.serial = serial;
    for (int k = 0; k < 8; k++) a.lf_delta[k] = p.lf_delta[k];
    if (b.coefs16 && !(b.words && trust == kPredForwardOutput)) return hipErrorInvalidValue; // compact planes: inside the symbol-stream chains only
    a.coefs = b.coefs16 ? reinterpret_cast<const int32_t *>(b.coefs16) : b.coefs;
    a.coef_stride = b.coef_stride;
    a.out_stride = b.out_stride;
    a.params = b.params;
    for (int k = 0; k < 3; k++) a.pp3[k] = b.pp[k];
    a.pred_slots = p.pred_slots;
    a.nbr_table = p.nbr_table;
    a.interior = p.interior;
    a.valid_mask = p.valid_mask;
    a.bucket = bucket;
    a.prediction = prediction;
    a.hist = hist;
    a.n_oob = n_oob;
    a.n_tiles = p.n_pred_tiles;
    a.trusted = trust != kPredAnyInt32 ? 1 : 0;
    a.words = b.words;
    if (b.words && trust != kPredForwardOutput) return hipErrorInvalidValue; // the halfword form exists for the chain's own coefficients only
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
    a.pred_off = p.gather_off;
    a.pair_pos = p.pair_pos;
    a.heap_of_pos = p.heap_of_pos;
    a.halo_list = p.halo_list;
    a.ablate = p.k2_ablate;
    (void)hipGetLastError(); // the check behind the launch must not pick up an error an earlier, unrelated call left behind
    a.stream_pos = b.stream_pos;
    if (b.stream_pos && !(b.words && b.coefs16)) return hipErrorInvalidValue; // the stream form: the compact symbol-stream chains only
    if (b.words && b.coefs16 && b.stream_pos)
        hipLaunchKernelGGL((predict_histogram_kernel3<false, true, true, true>), dim3(blocks, b.n_planes), dim3(kP3Threads), 0, stream, a);
    else if (b.words && b.coefs16)
        hipLaunchKernelGGL((predict_histogram_kernel3<false, true, true>), dim3(blocks, b.n_planes), dim3(kP3Threads), 0, stream, a);
    else if (b.words)
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
