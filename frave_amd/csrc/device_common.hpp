// device_common.hpp -- device helpers shared by the gfx950 kernels (included inside each kernel translation unit).
//
// Work decomposition shared by K1 and K3: one 64-lane wavefront owns one (cell, channel). Lane L owns the
// eight leaves 8L..8L+7 of the cell's digit tree; their pixel offsets from the lane base are the subset
// sums of LITERALS[0..2] (a fixed 4x3 footprint), the lane base is the subset sum of LITERALS[3..8]
// selected by the bits of L. Tree levels 8,7,6 are register arithmetic inside the lane, levels 5..0 are
// six cross-lane butterfly rounds (lane ^ 1, 2, 4, 8, 16, 32). The cell's 512 int32 coefficients leave
// as four fully coalesced store instructions (1 KiB + 512 B + 256 B + 256 B).
// Citations are relative to /root/reference/crates/libfri/src/.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>

#include "kernels.hpp"

// The hand-written waits of these kernels (s_waitcnt vmcnt(0) in front of a hand-over ticket, counted lgkmcnt(6) behind d16 LDS gathers) rely on gfx9-class
// counters - vmcnt covers stores and atomics without a return value, d16 loads write whole registers - and on gfx950's instruction set. Another target must
// not compile them silently (the Makefile's ARCH can be overridden).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "frave_amd's kernels are written for gfx950 (MI355X): the hand-counted s_waitcnt sequences and DPP / permlane forms do not carry over to other targets"
#endif

namespace fri {
namespace {

constexpr int kNone = INT32_MIN; // wire encoding of Option::None

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

// Native vector types on purpose: HIP's int4 / uint4 are structs whose copies become llvm.memcpy, which kept staging arrays in
// scratch memory (and put a vmcnt(0) behind every load).
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
// Timing-only ablations (skip staging / the cell loop / stores) exist in the `make tuning` build alone; in the product the
// flags are the constant 0 and every test on them folds away.
#ifndef FRI_HIP_ENABLE_ABLATE
#define FRI_HIP_ENABLE_ABLATE 0
#endif
constexpr bool kAblateBuild = FRI_HIP_ENABLE_ABLATE != 0;
__device__ __forceinline__ int ablate_flags(int flags) { return kAblateBuild ? flags : 0; }
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

// Hand-over through device-scope atomics (K2's histogram, K4's sums): a workgroup adds its partial results into a plan accumulator and then
// draws a ticket; the workgroup whose ticket is the last one reads the totals. That is only correct if every add of a workgroup has been
// PERFORMED before its ticket is drawn, i.e. if every wave has waited for its own vector-memory counter before the barrier in front of the
// ticket. __syncthreads() does NOT do that: for relaxed atomics hipcc emits a bare s_barrier (rounds 1 and 2 relied on a wait that was never
// there, and lost counts whenever the last workgroup's copy-out overtook another workgroup's adds - seen once many workgroups finish together).
// The wait is inline asm so that no compiler pass can drop it.
__device__ __forceinline__ void wait_for_own_memory_ops_then_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// wrapping integer arithmetic like a release build of the reference
__device__ __forceinline__ int iabs_w(int a) { return a < 0 ? (int)(0u - (unsigned)a) : a; }
__device__ __forceinline__ int sub_w(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int add_w(int a, int b) { return (int)((unsigned)a + (unsigned)b); }

} // namespace
} // namespace fri
