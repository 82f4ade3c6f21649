// k5_stream.hip -- K5 symbol_stream: the gather of (symbol, bucket) in the reference's stream order, on the device.
//
// The reference's emitter walks the ten scans of a channel (DC, root, levels 1..8, each in sort_lattice order, wavelet_transform.rs:657-705) and
// feeds pack_signed(value - prediction) with its context to the rANS coder (entropy_coding.rs:285-336). The walk is a permutation of the Some
// nodes that depends on the geometry only; round 2 did it on the host, which meant shipping 9 bytes per node (coefficient, prediction, bucket)
// over PCIe - 153 MB per 4096^2 plane - for a stream that needs 2 bytes per symbol. Here the plan holds the permutation (fri_hip_plan_set_stream_order:
// node index of the i-th symbol, None nodes already taken out) and one thread per symbol gathers its node and writes bucket << 10 | symbol.
// HBM-bound: 4 B of order + 9 B of gathers + 2 B out per symbol; the order is read coalesced, the gathers follow scan lines through the cells
// (runs of a few nodes per cell and level), the stream is written coalesced.
#include "device_common.hpp"

namespace fri {
namespace {

struct StreamArgs {
    const uint32_t *order; // [n_symbols] cell << 9 | heap index (= index into a plane)
    const int32_t *coefs;
    const int32_t *prediction;
    const uint8_t *bucket;
    uint16_t *out;
    uint64_t n_symbols;
    size_t coef_stride, out_stride, stream_stride; // planes of a batch (grid.y)
};

constexpr int kStreamThreads = 256, kStreamPerThread = 8;

__global__ void __launch_bounds__(kStreamThreads) symbol_stream_kernel(const StreamArgs a) {
    const uint32_t plane = blockIdx.y;
    const int32_t *coefs = a.coefs + plane * a.coef_stride;
    const int32_t *prediction = a.prediction + plane * a.out_stride;
    const uint8_t *bucket = a.bucket + plane * a.out_stride;
    uint16_t *out = a.out + plane * a.stream_stride;
    // a workgroup takes kStreamThreads * kStreamPerThread consecutive symbols; a thread's symbols are kStreamThreads apart (coalesced order reads and stream writes)
    // Consecutive stretches of the stream walk the same cells scan line after scan line, so the stretches an XCD's L2 sees must be neighbours: one contiguous
    // range of the stream per XCD (dealt round-robin, every one of the 8 L2s fetched every cell: 1.15 GB of fetches for a 151 MB input).
    const uint64_t base = (uint64_t)xcd_contiguous_share(blockIdx.x, gridDim.x) * (kStreamThreads * kStreamPerThread) + threadIdx.x;
    uint32_t node[kStreamPerThread];
#pragma unroll
    for (int k = 0; k < kStreamPerThread; k++) {
        const uint64_t i = base + (uint64_t)k * kStreamThreads;
        node[k] = i < a.n_symbols ? a.order[i] : 0u; // (plain loads: see symbol_gather_kernel)
    }
    int v[kStreamPerThread], p[kStreamPerThread];
    uint32_t b[kStreamPerThread];
#pragma unroll
    for (int k = 0; k < kStreamPerThread; k++) v[k] = coefs[node[k]], p[k] = prediction[node[k]], b[k] = bucket[node[k]];
#pragma unroll
    for (int k = 0; k < kStreamPerThread; k++) {
        const uint64_t i = base + (uint64_t)k * kStreamThreads;
        const int d = sub_w(v[k], p[k]);
        const uint32_t sym = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31); // pack_signed, utils.rs:34-40
        // (a symbol >= 1024 has no place in the alphabet: K2 counted it in n_out_of_alphabet and the caller must not emit; the low ten bits go out)
        if (i < a.n_symbols) __builtin_nontemporal_store((uint16_t)(b[k] << 10 | (sym & 1023u)), out + i);
    }
}

// The same stream from the halfword-per-node planes the scan kernel writes in its WORDS form (k2_predict.hip: bucket << 10 | symbol, the counter the node
// bumped): one 2-byte gather per symbol out of a 34 MB plane instead of three gathers (4 + 4 + 1 bytes) out of 151 MB, and the scan kernel stores 2 bytes per
// node instead of 5. A wave takes 512 consecutive symbols (2 KB of order in, 1 KB of stream out). The chunks are laid on the 16-byte grid of the OUTPUT
// address (a channel's stream starts wherever the previous one ended), so every store but a plane's first and last is one dwordx4.
struct GatherArgs {
    const uint32_t *order;
    const uint16_t *words;
    uint16_t *out;
    uint64_t n_symbols;
    size_t word_stride, stream_stride;
};
constexpr int kGatherThreads = 256;

#ifndef FRI_K5_CHUNKS
#define FRI_K5_CHUNKS 2
#endif
constexpr int kGatherChunks = FRI_K5_CHUNKS; // 512-symbol chunks per wave: all their order loads, then all their gathers, are in flight together
__global__ void __launch_bounds__(kGatherThreads) symbol_gather_kernel(const GatherArgs a) {
    __shared__ __attribute__((aligned(16))) uint16_t s_t[kGatherThreads * 8 * kGatherChunks];
    const uint32_t plane = blockIdx.y;
    const uint16_t *words = a.words + plane * a.word_stride;
    uint16_t *out = a.out + plane * a.stream_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t lead = (int64_t)((reinterpret_cast<uintptr_t>(out) >> 1) & 7u); // halfwords between the 16-byte grid and the stream's start
    // a wave takes kGatherChunks x 512 consecutive symbols; in gather k lane l fetches symbol 64 k + l, so the 64 gathers of one instruction are 64 NEIGHBOURS
    // of the stream - a scan-line stretch through three or four cells, a handful of 64-byte lines - where a lane-owns-eight-symbols split touches 64 lines
    // per instruction (55 us against 118 for the three-array kernel, the texture addresser the limit). The halfwords turn through LDS into the
    // lane-owns-eight layout of the stores.
    constexpr int kPerWave = 512 * kGatherChunks;
    const int64_t wave_first = ((int64_t)xcd_contiguous_share(blockIdx.x, gridDim.x) * (kGatherThreads / 64) + wave) * kPerWave - lead;
    const int64_t n = (int64_t)a.n_symbols;
    if (wave_first >= n) return;
    uint16_t *t = s_t + wave * kPerWave;
    if (wave_first >= 0 && wave_first + kPerWave <= n) {
        uint32_t o[8 * kGatherChunks];
#pragma unroll
        // plain loads: as nontemporal loads (through round 4) the order's 67 MB cost 4.3 us more - 41.2 against 36.9 us per 4096^2 plane, same box, same call
        // (gpurun_out/r5_k5; the floor without order and gather - words copied in stream order - is 15.7 us, with the order loads but no gather 25.2 us)
        for (int k = 0; k < 8 * kGatherChunks; k++) o[k] = a.order[wave_first + 64 * k + lane];
        uint16_t w[8 * kGatherChunks];
#pragma unroll
        for (int k = 0; k < 8 * kGatherChunks; k++) w[k] = words[o[k]];
#pragma unroll
        for (int k = 0; k < 8 * kGatherChunks; k++) t[64 * k + lane] = w[k];
        // (the wave reads what the wave wrote: no workgroup barrier; LDS operations of a wave complete in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#pragma unroll
        for (int c = 0; c < kGatherChunks; c++) {
            const u32x4 v = reinterpret_cast<const u32x4 *>(t + 512 * c)[lane];
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + wave_first + 512 * c) + lane);
        }
        return;
    }
    for (int64_t i = (wave_first < 0 ? 0 : wave_first) + lane; i < wave_first + kPerWave && i < n; i += 64) out[i] = words[a.order[i]]; // the two ragged ends of a plane
}

} // namespace

hipError_t launch_symbol_gather(const uint32_t *order, uint64_t n_symbols, uint32_t n_planes, const uint16_t *words, size_t word_stride, uint16_t *out, size_t stream_stride,
                                hipStream_t stream) {
    if (!order || !words || !out || !n_planes || n_planes > 65535u || (reinterpret_cast<uintptr_t>(order) & 15u)) return hipErrorInvalidValue;
    if (!n_symbols) return hipSuccess;
    GatherArgs a{};
    a.order = order, a.words = words, a.out = out, a.n_symbols = n_symbols, a.word_stride = word_stride, a.stream_stride = stream_stride;
    const uint64_t per_wg = (uint64_t)kGatherThreads * 8 * kGatherChunks;
    const uint64_t blocks = (n_symbols + 7 + per_wg - 1) / per_wg; // + 7: the lead of a stream that does not start on the 16-byte grid
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(symbol_gather_kernel, dim3((uint32_t)blocks, n_planes), dim3(kGatherThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_symbol_stream(const uint32_t *order, uint64_t n_symbols, uint32_t n_planes, const int32_t *coefs, size_t coef_stride, const uint8_t *bucket,
                                const int32_t *prediction, size_t out_stride, uint16_t *out, size_t stream_stride, hipStream_t stream) {
    if (!order || !coefs || !bucket || !prediction || !out || !n_planes || n_planes > 65535u) return hipErrorInvalidValue;
    if (!n_symbols) return hipSuccess;
    StreamArgs a{};
    a.order = order, a.coefs = coefs, a.prediction = prediction, a.bucket = bucket, a.out = out, a.n_symbols = n_symbols;
    a.coef_stride = coef_stride, a.out_stride = out_stride, a.stream_stride = stream_stride;
    const uint64_t per_wg = (uint64_t)kStreamThreads * kStreamPerThread;
    const uint64_t blocks = (n_symbols + per_wg - 1) / per_wg;
    if (blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(symbol_stream_kernel, dim3((uint32_t)blocks, n_planes), dim3(kStreamThreads), 0, stream, a);
    return hipGetLastError();
}

} // namespace fri
