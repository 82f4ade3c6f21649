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
        node[k] = i < a.n_symbols ? __builtin_nontemporal_load(a.order + i) : 0u;
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

} // namespace

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
