// Check of v_bitop3_b32 as this toolchain emits it: D = TTBL[{S0[i], S1[i], S2[i]}] per bit, for the tables the library's kernels contain
// (0xe0 = a & (b | c), 0xc8 = a & b | ... etc.), on random operands, in three forms:
//   form 0: operands straight from registers
//   form 1: the sequence the fit kernel's packed gather compiled to - S2 = v_and(0xffff, lo), S1 = v_lshlrev(16, hi), S0 = mask
//   form 2: as form 1 with lo and hi freshly returned by ds_read_u16 (the compiler's own waits)
// Prints the number of mismatches per form and table. Build: hipcc --offload-arch=gfx950 -O2 -o bitop3_check bitop3_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int T>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(d) : "v"(a), "v"(b), "v"(c), "n"(T));
    return d;
}
__host__ __device__ inline uint32_t table(uint32_t t, uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d = 0;
    for (int i = 0; i < 32; i++) {
        const uint32_t idx = ((a >> i) & 1) << 2 | ((b >> i) & 1) << 1 | ((c >> i) & 1);
        d |= ((t >> idx) & 1u) << i;
    }
    return d;
}
template <int T>
__global__ void k(const uint32_t *in, uint32_t *out, int n) {
    __shared__ uint16_t lds[4096];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int j = threadIdx.x; j < 4096; j += blockDim.x) lds[j] = (uint16_t)(in[j % n] >> 7);
    __syncthreads();
    if (i >= n) return;
    const uint32_t a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
    out[4 * i + 0] = bitop3<T>(a, b, c);
    uint32_t lo, hi;
    asm volatile("v_and_b32 %0, 0xffff, %2\n\tv_lshlrev_b32 %1, 16, %3" : "=&v"(lo), "=&v"(hi) : "v"(c), "v"(b));
    out[4 * i + 1] = bitop3<T>(a, hi, lo);
    const uint32_t l2 = lds[c & 4095], h2 = lds[b & 4095];
    out[4 * i + 2] = ((l2 & 0xffffu) | (h2 << 16)) & a; // what the compiler makes of it (v_bitop3 0xe0 on this toolchain)
    out[4 * i + 3] = __builtin_amdgcn_perm(h2, l2, 0x05040100u) & a;
}
template <int T>
int run(const std::vector<uint32_t> &h, uint32_t *d_in, uint32_t *d_out, int n) {
    hipLaunchKernelGGL(k<T>, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, d_out, n);
    std::vector<uint32_t> o(4 * n);
    hipMemcpy(o.data(), d_out, o.size() * 4, hipMemcpyDeviceToHost);
    int bad[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const uint32_t a = h[3 * i], b = h[3 * i + 1], c = h[3 * i + 2];
        bad[0] += o[4 * i] != table(T, a, b, c);
        bad[1] += o[4 * i + 1] != table(T, a, b << 16, c & 0xffffu);
        const uint32_t l2 = (uint16_t)(h[(c & 4095) % n] >> 7), h2 = (uint16_t)(h[(b & 4095) % n] >> 7);
        bad[2] += o[4 * i + 2] != ((l2 | h2 << 16) & a);
        bad[3] += o[4 * i + 3] != ((l2 | h2 << 16) & a);
    }
    printf("table 0x%02x: mismatches registers %d, and/shift operands %d, compiled (lo | hi << 16) & mask %d, perm & mask %d of %d\n", T, bad[0], bad[1], bad[2], bad[3], n);
    return bad[0] + bad[1] + bad[2] + bad[3];
}
int main() {
    const int n = 1 << 16;
    std::vector<uint32_t> h(3 * n);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 16); }
    for (int i = 0; i < n; i += 3) h[3 * i] = 0xFFFFFFFFu; // all-ones masks as in interior cells
    uint32_t *d_in, *d_out;
    hipMalloc(&d_in, h.size() * 4);
    hipMalloc(&d_out, 4 * n * 4);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int bad = run<0xe0>(h, d_in, d_out, n) + run<0xc8>(h, d_in, d_out, n) + run<0x80>(h, d_in, d_out, n) + run<0xde>(h, d_in, d_out, n);
    return bad ? 1 : 0;
}
