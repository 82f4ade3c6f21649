// Microbenchmark: issue cost of the instructions the K2 redesign leans on, one workgroup of 1024 threads per CU (4 waves per SIMD).
//   mode 0: v_mul_f32 + v_add_f32 pairs            mode 1: v_pk_mul_f32 + v_pk_add_f32 (two floats per lane and instruction)
//   mode 2: v_fma_f32 with |src| modifier          mode 3: v_cvt_f32_i32
//   mode 4: ds_read_u16_d16_hi gathers (random halfwords of a 37 KB image)   mode 5: ds_read_i16 + v_cvt_f32_i32
//   mode 6: ds_add_u32 to random bins of a 40 KB table
// Prints cycles per wave-instruction per SIMD (4 resident waves) at the clock the run reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ unsigned long long g_clk[4];
template <int MODE, int REP>
__global__ void __launch_bounds__(1024) k(const int *idx, float *out, int iters, float p, float q) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    __shared__ __attribute__((aligned(16))) unsigned char lds[81920];
    for (int i = threadIdx.x; i < 81920 / 4; i += blockDim.x) ((unsigned *)lds)[i] = (unsigned)i * 2654435761u & 0x43004300u;
    __syncthreads();
    unsigned a[8];
    for (int j = 0; j < 8; j++) a[j] = (unsigned)idx[threadIdx.x * 8 + j];
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = (float)(threadIdx.x + j) * 0.001f;
    f2 y[4];
    for (int j = 0; j < 4; j++) y[j] = f2{x[2 * j], x[2 * j + 1]};
    float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int rep = 0; rep < REP; rep++) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                asm volatile("v_mul_f32 %0, %1, %0\n\tv_add_f32 %0, %2, %0" : "+v"(x[j]) : "v"(p), "v"(q));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f2 pp = {p, p}, qq = {q, q};
                asm volatile("v_pk_mul_f32 %0, %1, %0\n\tv_pk_add_f32 %0, %2, %0" : "+v"(y[j]) : "v"(pp), "v"(qq));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_fma_f32 %0, |%1|, %2, %0" : "+v"(x[j]) : "v"(p), "v"(q));
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; j++) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(x[j]) : "v"(a[j]));
        } else if (MODE == 4) {
            asm volatile("ds_read_u16_d16_hi %0, %8\n\tds_read_u16_d16_hi %1, %9\n\tds_read_u16_d16_hi %2, %10\n\tds_read_u16_d16_hi %3, %11\n\t"
                         "ds_read_u16_d16_hi %4, %12\n\tds_read_u16_d16_hi %5, %13\n\tds_read_u16_d16_hi %6, %14\n\tds_read_u16_d16_hi %7, %15\n\ts_waitcnt lgkmcnt(0)"
                         : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
            acc += g[0] + g[7];
        } else if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 8; j++) acc += (float)*(const short *)(lds + a[j]);
        } else if (MODE == 6) {
#pragma unroll
            for (int j = 0; j < 8; j++) atomicAdd((unsigned *)(lds + ((a[j] * 2u) & 40956u)), 1u);
        }
        if (MODE >= 4) {
#pragma unroll
            for (int j = 0; j < 8; j++) a[j] = (a[j] + 2 * 523u) & 32766u;
        }
      }
    }
    for (int j = 0; j < 8; j++) acc += x[j] + g[j];
    for (int j = 0; j < 4; j++) acc += y[j].x + y[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (float)((unsigned *)lds)[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) g_clk[0] = clock64() - c0, g_clk[1] = wall_clock64() - w0; // shader clock ticks, 100 MHz ticks
}

int main() {
    const int blocks = 256, threads = 1024, iters = 20000;
    std::vector<int> h(threads * 8);
    unsigned s = 12345;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) & 32766u; }
    int *d_idx; float *d_out;
    hipMalloc(&d_idx, h.size() * 4); hipMalloc(&d_out, blocks * threads * 4);
    hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const char *names[7] = {"v_mul_f32+v_add_f32 (16 instr)", "v_pk_mul_f32+v_pk_add_f32 (8 instr = 16 flops x2)", "v_fma_f32 |abs| (8)", "v_cvt_f32_i32 (8)",
                            "ds_read_u16_d16_hi x8 + wait (+16 addr VALU)", "ds_read_i16 + cvt + add x8 (+16 addr VALU)", "ds_add_u32 x8 (+24 addr VALU)"};
    const int per_iter[7] = {16, 8, 8, 8, 8, 8, 8};
    for (int mode = 0; mode < 7; mode++) {
        double t[2];
        for (int r = 0; r < 2; r++) {
            auto launch = [&]() {
#define L(M) if (r == 0) hipLaunchKernelGGL((k<M, 1>), dim3(blocks), dim3(threads), 0, 0, d_idx, d_out, iters, 1.0001f, 0.5f); else hipLaunchKernelGGL((k<M, 4>), dim3(blocks), dim3(threads), 0, 0, d_idx, d_out, iters, 1.0001f, 0.5f);
                switch (mode) {
                case 0: L(0) break;
                case 1: L(1) break;
                case 2: L(2) break;
                case 3: L(3) break;
                case 4: L(4) break;
                case 5: L(5) break;
                default: L(6) break;
                }
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            t[r] = ms;
        }
        unsigned long long hclk[4];
        hipMemcpyFromSymbol(hclk, HIP_SYMBOL(g_clk), sizeof hclk);
        const double mhz = (double)hclk[0] / ((double)hclk[1] / 100.0);
        // per SIMD: 4 waves x iters x per_iter x REP wave-instructions; the difference between REP = 4 and REP = 1 removes the loop overhead
        const double cyc = (t[1] - t[0]) * 1e-3 * clk_khz * 1e3;
        std::printf("%-52s %8.3f / %8.3f ms  %6.2f cycles per wave-instruction per SIMD, 4 resident waves (nominal %d MHz; measured %.0f MHz -> %.2f cycles)\n", names[mode], t[0], t[1], cyc / (4.0 * iters * per_iter[mode] * 3), clk_khz / 1000, mhz, cyc / (4.0 * iters * per_iter[mode] * 3) * mhz / (clk_khz / 1000.0));
    }
    return 0;
}
