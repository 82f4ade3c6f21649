// Is v_cvt_pk_u8_f32 the saturating TRUNCATION min(v_cvt_u32_f32(x), 255) for every float? (round 5: K2's bucket-table address is cvt_u32 + min(., 31) + shift-add today;
// a byte conversion would address a 256-entry table through the instruction's offset field.) Exhaustive over all 2^32 bit patterns.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void check(unsigned long long *bad, uint32_t *first) {
    const uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096ull;
    unsigned long long n = 0;
    for (uint64_t k = 0; k < 4096; k++) {
        const uint32_t bits = (uint32_t)(base + k);
        const float x = __builtin_bit_cast(float, bits);
        uint32_t a, b = 0;
        asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(a) : "v"(x));
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(b) : "v"(x), "v"(0u));
        a = a < 255u ? a : 255u;
        if (a != (b & 255u)) {
            if (n == 0) atomicMin(first, bits);
            n++;
        }
    }
    if (n) atomicAdd(bad, n);
}

int main() {
    unsigned long long *d_bad, h_bad = 0;
    uint32_t *d_first, h_first = 0xFFFFFFFFu;
    hipMalloc(&d_bad, 8), hipMalloc(&d_first, 4);
    hipMemcpy(d_bad, &h_bad, 8, hipMemcpyHostToDevice), hipMemcpy(d_first, &h_first, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d_bad, d_first); // 4096 x 256 x 4096 = 2^32
    hipDeviceSynchronize();
    hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost), hipMemcpy(&h_first, d_first, 4, hipMemcpyDeviceToHost);
    float f = __builtin_bit_cast(float, h_first);
    std::printf("v_cvt_pk_u8_f32 against min(v_cvt_u32_f32, 255) over all 2^32 floats: %llu differ", h_bad);
    if (h_bad) std::printf(" (smallest differing pattern 0x%08x = %g)", h_first, (double)f);
    std::printf("\n");
    return 0;
}
