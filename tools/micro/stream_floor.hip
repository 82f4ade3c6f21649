// Microbenchmark: what the memory system allows a SINGLE launch with K1's byte mix - 16.8 MB read once, 68.2 MB written once with nontemporal
// stores (u8 in, i32 out: 1 B + 4 B per element) - when neither the source nor the destination can sit in the 256 MiB Infinity Cache: launches
// rotate over SLOTS buffer pairs (default 32: 2.7 GB). No arithmetic worth the name: a lane loads 16 bytes and stores 4 x 16 bytes. Variants:
//   mode 0: grid-stride over chunks, 1024 x 256 threads (K1's grid)        mode 1: one chunk per thread (65536 workgroups of 256)
//   mode 2: write-only (68.2 MB fill)                                      mode 3: read-only (16.8 MB, summed into one word per workgroup)
//   mode 4: as mode 0 with plain (cached) stores
//   mode 5: as mode 1, but the 2 KiB output pieces ("cells") leave in K1's order: inside every block of 128 cells (256 KB) a 16 x 8 transpose - the eight cells
//           a K1 tile writes are 16 KB apart (bands of 16 rows: a tile's cells sit in eight different centre rows)      mode 6: the same with 64 x 8 (bands of 64 rows)
// Prints microseconds per launch by HIP events around N back-to-back launches and the fraction of 8 TB/s for the 84.9 MB of algorithmic bytes.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_floor stream_floor.hip        run: ./stream_floor [slots] [launches]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, unsigned n_chunks, unsigned *sink) {
    unsigned acc = 0;
    const unsigned stride = (MODE == 1 || MODE >= 5) ? n_chunks : gridDim.x * blockDim.x;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_chunks; i += stride) {
        u32x4 v = u32x4{i, i, i, i};
        if (MODE != 2) v = in[i];
        if (MODE == 3) {
            acc += v.x ^ v.y ^ v.z ^ v.w;
            continue;
        }
        // 16 bytes in -> 64 bytes out (each byte widened to a dword), one wave writes 4 x 1 KiB contiguous pieces
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32x4 o = u32x4{w[q] & 0xFFu, (w[q] >> 8) & 0xFFu, (w[q] >> 16) & 0xFFu, w[q] >> 24};
            unsigned ii = i;
            if (MODE == 5 || MODE == 6) { // cell = 32 chunks of input = 2 KiB of output
                constexpr unsigned R = MODE == 5 ? 16 : 64;
                const unsigned cell = i >> 5, blk = cell / (R * 8), in = cell % (R * 8), w = in / R, row = in % R;
                ii = ((blk * (R * 8) + row * 8 + w) << 5) | (i & 31u);
            }
            u32x4 *dst = out + (size_t)(ii & ~63u) * 4 + (size_t)q * 64 + (ii & 63u);
            if (MODE == 4)
                *dst = o;
            else
                __builtin_nontemporal_store(o, dst);
        }
    }
    if (MODE == 3 && acc == 0x12345678u) sink[blockIdx.x] = acc;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));      \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    const int slots = argc > 1 ? std::atoi(argv[1]) : 32, launches = argc > 2 ? std::atoi(argv[2]) : 300;
    const size_t in_bytes = (size_t)4096 * 4096, out_bytes = 4 * in_bytes + 1400000; // K1 writes F * 2 KiB = 68.2 MB for 16.8 MB of pixels; the tail is not touched here
    const unsigned n_chunks = (unsigned)(in_bytes / 16);
    unsigned char *in = nullptr, *out = nullptr;
    unsigned *sink = nullptr;
    CK(hipMalloc((void **)&in, in_bytes * slots));
    CK(hipMalloc((void **)&out, out_bytes * slots));
    CK(hipMalloc((void **)&sink, 1 << 20));
    CK(hipMemset(in, 1, in_bytes * slots));
    CK(hipMemset(out, 0, out_bytes * slots));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto run = [&](int mode, int n) {
        for (int i = 0; i < n; i++) {
            const u32x4 *src = reinterpret_cast<const u32x4 *>(in + (size_t)(i % slots) * in_bytes);
            u32x4 *dst = reinterpret_cast<u32x4 *>(out + (size_t)(i % slots) * out_bytes);
            switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, s, src, dst, n_chunks, sink); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(n_chunks / 256), dim3(256), 0, s, src, dst, n_chunks, sink); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(1024), dim3(256), 0, s, src, dst, n_chunks, sink); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(1024), dim3(256), 0, s, src, dst, n_chunks, sink); break;
            case 4: hipLaunchKernelGGL(k<4>, dim3(1024), dim3(256), 0, s, src, dst, n_chunks, sink); break;
            case 5: hipLaunchKernelGGL(k<5>, dim3(n_chunks / 256), dim3(256), 0, s, src, dst, n_chunks, sink); break;
            default: hipLaunchKernelGGL(k<6>, dim3(n_chunks / 256), dim3(256), 0, s, src, dst, n_chunks, sink); break;
            }
        }
    };
    run(0, 3000); // spin-up, as bench.py
    CK(hipStreamSynchronize(s));
    const char *names[7] = {"copy 1B->4B, nt stores, 1024x256 grid-stride", "copy 1B->4B, nt stores, one chunk per thread", "write only (68.2 MB, nt)", "read only (16.8 MB)",
                            "copy 1B->4B, plain stores, 1024x256 grid-stride", "as mode 1, 2 KiB pieces in K1's order (16-row bands)", "as mode 1, 2 KiB pieces in K1's order (64-row bands)"};
    const double bytes[7] = {5.0 * in_bytes, 5.0 * in_bytes, 4.0 * in_bytes, 1.0 * in_bytes, 5.0 * in_bytes, 5.0 * in_bytes, 5.0 * in_bytes};
    for (int round = 0; round < 2; round++)
        for (int mode = 0; mode < 7; mode++) {
            run(mode, slots);
            CK(hipEventRecord(e0, s));
            run(mode, launches);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / launches;
            std::printf("slots=%d mode %d (%s): %.2f us per launch = %.2f TB/s = %.3f of 8 TB/s\n", slots, mode, names[mode], us, bytes[mode] / us / 1e6, bytes[mode] / us / 1e6 / 8.0);
        }
    return 0;
}
