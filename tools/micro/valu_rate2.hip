// Microbenchmark: issue cost of single vector instructions the fit and scan kernels lean on (one workgroup of 1024 threads per CU: 4 waves per SIMD, eight
// independent destination registers per instruction kind). Prints cycles per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rate2 tools/micro/valu_rate2.hip && tools/micro/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ unsigned long long g_clk[2];
#define KERNEL(NAME, TEXT)                                                                                                              \
    template <int REP>                                                                                                                  \
    __global__ void __launch_bounds__(1024) NAME(unsigned *out, int iters, unsigned p, unsigned q) {                                    \
        const unsigned long long c0 = clock64(), w0 = wall_clock64();                                                                   \
        unsigned x[8];                                                                                                                  \
        for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 7u + j;                                                                        \
        for (int it = 0; it < iters; it++) {                                                                                            \
            _Pragma("unroll") for (int rep = 0; rep < REP; rep++) {                                                                     \
                _Pragma("unroll") for (int j = 0; j < 8; j++) asm volatile(TEXT : "+v"(x[j]) : "v"(p), "v"(q));                       \
            }                                                                                                                           \
        }                                                                                                                               \
        unsigned acc = 0;                                                                                                               \
        for (int j = 0; j < 8; j++) acc += x[j];                                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = acc;                                                                               \
        if (blockIdx.x == 0 && threadIdx.x == 0) g_clk[0] = clock64() - c0, g_clk[1] = wall_clock64() - w0;                            \
    }
KERNEL(k_mul, "v_mul_f32_e32 %0, %1, %0")
KERNEL(k_fmac, "v_fmac_f32_e32 %0, %1, %2")
KERNEL(k_fma3, "v_fma_f32 %0, %1, |%2|, %0")
KERNEL(k_and, "v_and_b32_e32 %0, %1, %0")
KERNEL(k_addu, "v_add_u32_e32 %0, %1, %0")
KERNEL(k_dot2c, "v_dot2c_i32_i16_e32 %0, %1, %2")
KERNEL(k_dot2, "v_dot2_i32_i16 %0, %1, %2, %0")
KERNEL(k_cvt, "v_cvt_f32_i32_e32 %0, %0")
KERNEL(k_cvt_sdwa, "v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
KERNEL(k_pksub, "v_pk_sub_i16 %0, %0, %1")
KERNEL(k_pkmax, "v_pk_max_i16 %0, %0, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0xc8")
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 16, %1")
KERNEL(k_sad, "v_sad_u16 %0, %0, %1, %2")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(k_pkmul, "v_pk_mul_lo_u16 %0, %0, %1")
KERNEL(k_pkmad, "v_pk_mad_i16 %0, %0, %1, %2")
KERNEL(k_dpp, "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_cvtub, "v_cvt_f32_ubyte1_e32 %0, %0")
KERNEL(k_pkfma, "v_pk_fma_f16 %0, %0, %1, %2")
KERNEL(k_maxf, "v_max_f32_e32 %0, %1, %0")
// round 5: what K1's butterflies and routing are made of
KERNEL(k_lerp, "v_lerp_u8 %0, %0, %1, %2")
KERNEL(k_subu, "v_sub_u32_e32 %0, %1, %0")
KERNEL(k_sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1")
KERNEL(k_sub_sdwa0, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0")
KERNEL(k_cnd_vcc, "v_cndmask_b32_e32 %0, %1, %0, vcc")
KERNEL(k_cnd_sgpr, "v_cndmask_b32_e64 %0, %1, %0, s[10:11]")
KERNEL(k_movdpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_movdpp_rm, "v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_bfei, "v_bfe_i32 %0, %0, 0, 16")
KERNEL(k_ashr, "v_ashrrev_i32_e32 %0, 16, %0")
KERNEL(k_lshr, "v_lshrrev_b32_e32 %0, 15, %0")
KERNEL(k_xor, "v_xor_b32_e32 %0, %1, %0")
KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, %2")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_pkadd, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pklshr, "v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]")
KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
KERNEL(k_cmp, "v_cmp_lt_i32_e64 s[12:13], %0, %1")
KERNEL(k_pl16, "v_permlane16_swap_b32_e32 %0, %1")
KERNEL(k_pl32, "v_permlane32_swap_b32_e32 %0, %1")
KERNEL(k_mov, "v_mov_b32_e32 %0, %1")
KERNEL(k_bfeu, "v_bfe_u32 %0, %0, 8, 8")
KERNEL(k_sext_sdwa, "v_mov_b32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0")
KERNEL(k_sext_sdwa1, "v_mov_b32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
KERNEL(k_subdpp, "v_sub_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")

int main() {
    const int blocks = 256, threads = 1024, iters = 20000;
    unsigned *d_out;
    hipMalloc(&d_out, blocks * threads * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
#define RUN(NAME, LABEL)                                                                                                                \
    {                                                                                                                                   \
        float t[2];                                                                                                                     \
        for (int r = 0; r < 2; r++) {                                                                                                   \
            for (int w = 0; w < 2; w++) {                                                                                               \
                hipEventRecord(e0);                                                                                                     \
                if (r == 0) hipLaunchKernelGGL((NAME<1>), dim3(blocks), dim3(threads), 0, 0, d_out, iters, 0x00030001u, 0x3f800000u);   \
                else hipLaunchKernelGGL((NAME<4>), dim3(blocks), dim3(threads), 0, 0, d_out, iters, 0x00030001u, 0x3f800000u);          \
                hipEventRecord(e1);                                                                                                     \
                hipEventSynchronize(e1);                                                                                                \
                hipEventElapsedTime(&t[r], e0, e1);                                                                                     \
            }                                                                                                                           \
        }                                                                                                                               \
        unsigned long long hclk[2];                                                                                                     \
        hipMemcpyFromSymbol(hclk, HIP_SYMBOL(g_clk), sizeof hclk);                                                                      \
        const double mhz = (double)hclk[0] / ((double)hclk[1] / 100.0);                                                                 \
        const double cyc = (t[1] - t[0]) * 1e-3 * mhz * 1e6 / (4.0 * iters * 8 * 3);                                                    \
        std::printf("%-34s %7.3f / %7.3f ms  %5.2f cycles per wave-instruction per SIMD (4 resident waves, %.0f MHz)\n", LABEL, t[0], t[1], cyc, mhz); \
    }
    RUN(k_mul, "v_mul_f32_e32")
    RUN(k_fmac, "v_fmac_f32_e32")
    RUN(k_fma3, "v_fma_f32 (VOP3, |abs|)")
    RUN(k_maxf, "v_max_f32_e32")
    RUN(k_and, "v_and_b32_e32")
    RUN(k_addu, "v_add_u32_e32")
    RUN(k_dot2c, "v_dot2c_i32_i16_e32")
    RUN(k_dot2, "v_dot2_i32_i16 (VOP3P)")
    RUN(k_cvt, "v_cvt_f32_i32_e32")
    RUN(k_cvt_sdwa, "v_cvt_f32_i32_sdwa sext WORD_1")
    RUN(k_cvtub, "v_cvt_f32_ubyte1_e32")
    RUN(k_pksub, "v_pk_sub_i16")
    RUN(k_pkmax, "v_pk_max_i16")
    RUN(k_pkmul, "v_pk_mul_lo_u16")
    RUN(k_pkmad, "v_pk_mad_i16")
    RUN(k_pkfma, "v_pk_fma_f16")
    RUN(k_perm, "v_perm_b32")
    RUN(k_bitop3, "v_bitop3_b32")
    RUN(k_lshlor, "v_lshl_or_b32")
    RUN(k_sad, "v_sad_u16")
    RUN(k_mad24, "v_mad_u32_u24")
    RUN(k_dpp, "v_add_u32_dpp quad_perm")
    RUN(k_lerp, "v_lerp_u8")
    RUN(k_subu, "v_sub_u32_e32")
    RUN(k_sub_sdwa, "v_sub_u32_sdwa WORD_1 WORD_1")
    RUN(k_sub_sdwa0, "v_sub_u32_sdwa WORD_0 WORD_0")
    RUN(k_cnd_vcc, "v_cndmask_b32_e32 (vcc)")
    RUN(k_cnd_sgpr, "v_cndmask_b32_e64 (sgpr pair)")
    RUN(k_movdpp, "v_mov_b32_dpp quad_perm")
    RUN(k_movdpp_rm, "v_mov_b32_dpp row_mirror")
    RUN(k_subdpp, "v_sub_u32_dpp quad_perm")
    RUN(k_bfei, "v_bfe_i32")
    RUN(k_bfeu, "v_bfe_u32")
    RUN(k_sext_sdwa, "v_mov_b32_sdwa sext WORD_0")
    RUN(k_sext_sdwa1, "v_mov_b32_sdwa sext WORD_1")
    RUN(k_ashr, "v_ashrrev_i32_e32")
    RUN(k_lshr, "v_lshrrev_b32_e32")
    RUN(k_xor, "v_xor_b32_e32")
    RUN(k_mov, "v_mov_b32_e32")
    RUN(k_alignbyte, "v_alignbyte_b32")
    RUN(k_add3, "v_add3_u32")
    RUN(k_or3, "v_or3_b32")
    RUN(k_andor, "v_and_or_b32")
    RUN(k_lshladd, "v_lshl_add_u32")
    RUN(k_xad, "v_xad_u32")
    RUN(k_pkadd, "v_pk_add_u16")
    RUN(k_pklshr, "v_pk_lshrrev_b16")
    RUN(k_cmp, "v_cmp_lt_i32_e64 -> sgpr pair")
    RUN(k_pl16, "v_permlane16_swap_b32")
    RUN(k_pl32, "v_permlane32_swap_b32")
    return 0;
}
