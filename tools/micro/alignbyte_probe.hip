// What v_alignbyte_b32 does with shift amounts above 3 on this GPU (the ISA manuals differ between families).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *out) {
    const uint32_t s = threadIdx.x;
    out[s] = __builtin_amdgcn_alignbyte(0x77665544u, 0x33221100u, s);
}
int main() {
    uint32_t *d, h[64];
    hipMalloc(&d, sizeof h);
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; i++) std::printf("shift %2d -> %08x\n", i, h[i]);
    std::printf("shift 35 -> %08x, shift 0xFFFFFFFF&63=63 -> %08x\n", h[35], h[63]);
    return 0;
}
