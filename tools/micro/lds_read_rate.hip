// Microbenchmark: LDS read throughput per wave-instruction for u8 / u16 / b32 / b64 gathers (random-ish addresses).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(1024) k(const int *idx, int *out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((int *)lds)[i] = i * 2654435761u;
    __syncthreads();
    int a[8];
    for (int j = 0; j < 8; j++) a[j] = idx[threadIdx.x * 8 + j];
    int acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int ad = (a[j] + it * 68) & 65535;
            if (MODE == 0) acc += lds[ad];
            if (MODE == 1) acc += *(const unsigned short *)(lds + (ad & ~1));
            if (MODE == 2) acc += *(const int *)(lds + (ad & ~3));
            if (MODE == 3) { int2 v = *(const int2 *)(lds + (ad & ~7)); acc += v.x ^ v.y; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const int blocks = 256, threads = 1024, iters = 2000;
    std::vector<int> h(threads * 8);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) & 65535; }
    int *d_idx, *d_out;
    hipMalloc(&d_idx, h.size() * 4); hipMalloc(&d_out, blocks * threads * 4);
    hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = {"ds_read_u8", "ds_read_u16", "ds_read_b32", "ds_read_b64"};
    for (int mode = 0; mode < 4; mode++) {
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d_idx, d_out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d_idx, d_out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, d_idx, d_out, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, d_idx, d_out, iters);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double wave_instr_per_cu = 16.0 * iters * 8; // 16 waves per CU (1 block per CU)
        printf("%-12s %8.3f ms  -> %6.1f ns per wave-instruction per CU (~%5.1f cycles at 2.4 GHz), random addresses\n", names[mode], ms,
               ms * 1e6 / wave_instr_per_cu, ms * 1e6 / wave_instr_per_cu * 2.4);
    }
    return 0;
}
