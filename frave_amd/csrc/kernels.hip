// kernels.hip -- gfx950 (CDNA4, wave64) kernels of the libfri hot path.
//
//   K1 fwd_transform_quant   address-map gather + 9-level residue (S-)transform + per-layer quantiser
//                            (Fractal::extract_coefficients, stages/wavelet_transform.rs:179-225;
//                             quantization::encode, stages/quantization.rs:7-25)
//   K2 predict_histogram     6-neighbour gather + context bucket + prediction + ANS symbol histogram
//                            (context_modeling.rs:25-77; stages/prediction.rs:86-207, 237-298)
//   K3 inverse_transform     dequantisation + inverse residue transform + clamp
//                            (stages/quantization.rs:27-45; stages/wavelet_transform.rs:358-381; images.rs:103-111)
//
// Citations are relative to /root/reference/crates/libfri/src/. All three are byte/integer gather-scan
// kernels bounded by HBM traffic; there is no dense contraction here and no MFMA.
//
// Work decomposition shared by K1 and K3: one 64-lane wavefront owns one (cell, channel). Lane L owns the
// eight leaves 8L..8L+7 of the cell's digit tree; their pixel offsets from the lane base are the subset
// sums of LITERALS[0..2] (a fixed 4x3 footprint), the lane base is the subset sum of LITERALS[3..8]
// selected by the bits of L. Tree levels 8,7,6 are register arithmetic inside the lane, levels 5..0 are
// six cross-lane butterfly rounds (lane ^ 1, 2, 4, 8, 16, 32). The cell's 512 int32 coefficients leave
// as four fully coalesced store instructions (1 KiB + 512 B + 256 B + 256 B).
#include <hip/hip_runtime.h>

#include "kernels.hpp"

namespace fri {
namespace {

constexpr int kNone = INT32_MIN; // wire encoding of Option::None
constexpr int kFwdThreads = 256; // 4 waves per workgroup
constexpr int kFwdWaves = kFwdThreads / 64;

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

// Rust `/ 2` on i32: truncation toward zero.
__device__ __forceinline__ int half_trunc(int d) { return (d + (int)((unsigned)d >> 31)) >> 1; }

// One butterfly: d = l - r, s = r + d/2 with Option semantics (try_apply, wavelet_transform.rs:14-26, 211-218):
// a missing operand counts as 0 when the other exists; both missing -> None.
template <bool CHK>
__device__ __forceinline__ void pair_op(int l, int r, int &d, int &s) {
    if (!CHK) {
        d = l - r;
        s = r + half_trunc(d);
    } else {
        const bool ln = l == kNone, rn = r == kNone;
        const int lv = ln ? 0 : l, rv = rn ? 0 : r;
        const int dd = lv - rv;
        const bool none = ln && rn;
        d = none ? kNone : dd;
        s = none ? kNone : rv + half_trunc(dd);
    }
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

struct FwdArgs {
    const uint8_t *pixels;
    size_t pixel_stride;
    int32_t *coefs;
    size_t coef_stride;
    const Tile *tiles;
    const int32_t *tile_cells;
    const Int2 *centers;
    const uint8_t *interior;
    int32_t width, height;
    uint32_t F;
    int32_t pitch;
    uint32_t cpr, cpr_magic;
    int32_t q_identity;
    QMatrix q;
};

template <bool CHK>
__device__ __forceinline__ void fwd_wave(const int (&leaf)[8], int lane, int32_t *__restrict__ out, const FwdArgs &a) {
    int d8[4], s8[4], d7[2], s7[2], d6, s;
#pragma unroll
    for (int i = 0; i < 4; i++) pair_op<CHK>(leaf[2 * i], leaf[2 * i + 1], d8[i], s8[i]); // level 8: nodes 256 + 4L + i
#pragma unroll
    for (int i = 0; i < 2; i++) pair_op<CHK>(s8[2 * i], s8[2 * i + 1], d7[i], s7[i]); // level 7: nodes 128 + 2L + i
    pair_op<CHK>(s7[0], s7[1], d6, s);                                                 // level 6: node 64 + L
    const int tz = lane ? __builtin_ctz(lane) : 6;
    int vlow = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) { // levels 5..0
        const int other = __shfl_xor(s, 1 << j);
        const bool hi = (lane >> j) & 1; // bit clear = left child (the right child adds the LITERAL)
        const int l = hi ? other : s, r = hi ? s : other;
        int d;
        pair_op<CHK>(l, r, d, s);
        if (tz == j) vlow = d;
    }
    if (lane == 0) vlow = s; // coefficients[0] = low_pass_values[1] (wavelet_transform.rs:221)
    int low = __shfl(vlow, low_source_lane(lane));

    if (!a.q_identity) { // quantization.rs:13-17, truncating i32 division, None untouched
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (d8[i] != kNone) d8[i] /= a.q.q[quant_layer(256 + 4 * lane + i)];
#pragma unroll
        for (int i = 0; i < 2; i++)
            if (d7[i] != kNone) d7[i] /= a.q.q[quant_layer(128 + 2 * lane + i)];
        if (d6 != kNone) d6 /= a.q.q[quant_layer(64 + lane)];
        if (low != kNone) low /= a.q.q[quant_layer(lane)];
    }
    *reinterpret_cast<int4 *>(out + 256 + 4 * lane) = make_int4(d8[0], d8[1], d8[2], d8[3]);
    *reinterpret_cast<int2 *>(out + 128 + 2 * lane) = make_int2(d7[0], d7[1]);
    out[64 + lane] = d6;
    out[lane] = low;
}

// K1. grid = (tiles, images), block = 256. Dynamic LDS = pitch * lds_rows bytes: the tile's pixel rectangle,
// each row starting at the 16-byte boundary at or below its first byte so every global load is an aligned
// 16-byte vector load whatever the image width.
template <int C>
__global__ void __launch_bounds__(kFwdThreads) fwd_transform_quant_kernel(const FwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Tile t = a.tiles[blockIdx.x];
    const uint8_t *img = a.pixels + (size_t)blockIdx.y * a.pixel_stride;
    const uintptr_t addr0 = reinterpret_cast<uintptr_t>(img);
    const uintptr_t addr_end = addr0 + (size_t)a.width * a.height * C;
    const uint32_t row_bytes = (uint32_t)t.width_px * C;

    const uint32_t total = (uint32_t)t.n_rows * a.cpr;
    for (uint32_t i = tid; i < total; i += kFwdThreads) {
        const uint32_t r = __umulhi(i, a.cpr_magic); // i / cpr
        const uint32_t k = i - r * a.cpr;
        const uintptr_t g = addr0 + ((size_t)(t.y_lo + (int)r) * a.width + t.x_lo) * C;
        const uintptr_t ca = (g & ~(uintptr_t)15) + 16u * k;
        if (ca < g + row_bytes) {
            uint8_t *dst = lds + r * a.pitch + 16u * k;
            if (ca >= addr0 && ca + 16 <= addr_end) {
                *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(ca);
            } else { // first/last chunk of the buffer: stay inside the caller's allocation
                for (int b = 0; b < 16; b++) {
                    const uintptr_t p = ca + b;
                    dst[b] = (p >= addr0 && p < addr_end) ? *reinterpret_cast<const uint8_t *>(p) : 0;
                }
            }
        }
    }
    __syncthreads();

    const int ldx = lane_dx(lane), ldy = lane_dy(lane);
    const uint32_t a16 = (uint32_t)addr0;
    const int n_items = t.cell_count * C;
    for (int it = wave; it < n_items; it += kFwdWaves) {
        const int cl = it / C, ch = it - cl * C;
        const int cell = __builtin_amdgcn_readfirstlane(a.tile_cells[t.cell_begin + cl]);
        const Int2 cen = a.centers[cell];
        const bool interior = a.interior[cell] != 0;
        int32_t *out = a.coefs + (size_t)blockIdx.y * a.coef_stride + ((size_t)ch * a.F + (size_t)cell) * kCell;
        const int x0 = cen.x + ldx, y0 = cen.y + ldy;
        int rb[3]; // LDS byte address of (x0, y0 + dy, ch)
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int y = y0 + dy;
            const uint32_t sh = (a16 + ((uint32_t)y * (uint32_t)a.width + (uint32_t)t.x_lo) * C) & 15u;
            rb[dy] = (y - t.y_lo) * a.pitch + (int)sh + (x0 - t.x_lo) * C + ch;
        }
        int leaf[8];
        if (interior) {
#pragma unroll
            for (int j = 0; j < 8; j++) leaf[j] = lds[rb[leaf_dy(j)] + leaf_dx(j) * C];
            fwd_wave<false>(leaf, lane, out, a);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int x = x0 + leaf_dx(j), y = y0 + leaf_dy(j);
                const bool valid = x >= 0 && y >= 0 && x < a.width && y < a.height; // get_pixel, images.rs:90
                leaf[j] = valid ? (int)lds[rb[leaf_dy(j)] + leaf_dx(j) * C] : kNone;
            }
            fwd_wave<true>(leaf, lane, out, a);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K2: prediction + bucket + histogram for one channel plane.
// ------------------------------------------------------------------------------------------------
constexpr int kPredThreads = 1024;
constexpr int kHistBins = 10 * 1024;

// Rust `f32 as u32` / `f32 as i32`: saturating, NaN -> 0.
__device__ __forceinline__ uint32_t f32_as_u32(float x) {
    if (!(x == x) || x <= 0.0f) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
__device__ __forceinline__ int f32_as_i32(float x) {
    if (!(x == x)) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int)x;
}
// assign_bucket, prediction.rs:55-68
__device__ __forceinline__ uint32_t assign_bucket(float width) {
    const uint32_t w = f32_as_u32(width);
    return w < 3 ? 0 : w < 5 ? 1 : w < 6 ? 2 : w < 8 ? 3 : w < 12 ? 4 : w < 16 ? 5 : w < 20 ? 6 : w < 25 ? 7 : w < 30 ? 8 : 9;
}
__device__ __forceinline__ int iabs_w(int a) { return a < 0 ? (int)(0u - (unsigned)a) : a; }
__device__ __forceinline__ int sub_w(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int add_w(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
// pack_signed, utils.rs:34-40 (wrapping arithmetic like a release build)
__device__ __forceinline__ uint32_t pack_signed(int k) { return k >= 0 ? 2u * (uint32_t)k : (uint32_t)(-2ll * (long long)k - 1); }

struct PredArgs {
    const int32_t *coefs; // one channel plane [F][512]
    const int32_t *nbr_cells;
    const uint16_t *nbr_table;
    uint8_t *bucket;
    int32_t *prediction;
    uint32_t *hist;
    unsigned long long *n_oob;
    uint32_t F;
    PredictParams pp;
};

__global__ void __launch_bounds__(kPredThreads) predict_histogram_kernel(const PredArgs a) {
    __shared__ uint32_t s_hist[kHistBins];
    __shared__ uint16_t s_tab[kCell * 6];
    __shared__ int32_t s_nbr[kPredThreads / 64][kNbr];
    __shared__ unsigned int s_oob;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kHistBins; i += kPredThreads) s_hist[i] = 0;
    for (int i = tid; i < kCell * 6; i += kPredThreads) s_tab[i] = a.nbr_table[i];
    if (tid == 0) s_oob = 0;
    __syncthreads();

    const uint32_t waves_total = gridDim.x * (kPredThreads / 64);
    for (uint32_t cell = blockIdx.x * (kPredThreads / 64) + wave; cell < a.F; cell += waves_total) {
        if (lane < kNbr) s_nbr[wave][lane] = a.nbr_cells[(size_t)cell * kNbr + lane];
        __builtin_amdgcn_wave_barrier();
        const int32_t *own = a.coefs + (size_t)cell * kCell;
#pragma unroll 2
        for (int i = 0; i < kCell / 64; i++) {
            const int p = i * 64 + lane;
            const int value = own[p];
            uint32_t bucket = 0;
            int prediction = 0;
            if (value != kNone) {
                int v[6];
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const uint32_t e = s_tab[p * 6 + k];
                    int x = 0;
                    if (!(e & 0x8000u)) {
                        const int nb = s_nbr[wave][(e >> 9) & 7];
                        if (nb >= 0) {
                            x = a.coefs[(size_t)nb * kCell + (e & 511u)];
                            if (x == kNone) x = 0; // .unwrap_or(0)
                        }
                    }
                    v[k] = x;
                }
                if (p < 2) { // get_lf_context_bucket, prediction.rs:134-144
                    const uint32_t width = (uint32_t)iabs_w(sub_w(v[0], v[2]));
                    bucket = assign_bucket((float)width);
                    const int mx = max(v[0], v[2]), mn = min(v[0], v[2]);
                    prediction = v[1] >= mx ? mx : v[1] <= mn ? mn : sub_w(add_w(v[0], v[2]), v[1]);
                } else { // get_hf_context_bucket, prediction.rs:165-206: f32, left to right, one rounding per op
                    const int level = 31 - __clz(p);
                    const int g = level < 7 ? 2 : level == 7 ? 1 : 0;
                    const float *wp = a.pp.width[g], *vp = a.pp.value[g];
                    float width = wp[0];
                    width = __fadd_rn(width, __fmul_rn(wp[1], (float)iabs_w(sub_w(v[0], v[3]))));
                    width = __fadd_rn(width, __fmul_rn(wp[2], (float)iabs_w(sub_w(v[1], v[2]))));
                    width = __fadd_rn(width, __fmul_rn(wp[3], (float)iabs_w(sub_w(v[4], v[5]))));
                    width = __fadd_rn(width, __fmul_rn(wp[4], (float)iabs_w(sub_w(v[1], v[5]))));
                    width = __fadd_rn(width, __fmul_rn(wp[5], (float)iabs_w(sub_w(v[2], v[4]))));
                    bucket = assign_bucket(width);
                    float pr = __fmul_rn((float)v[0], vp[0]);
                    pr = __fadd_rn(pr, __fmul_rn((float)v[1], vp[1]));
                    pr = __fadd_rn(pr, __fmul_rn((float)v[2], vp[2]));
                    pr = __fadd_rn(pr, __fmul_rn((float)v[3], vp[3]));
                    pr = __fadd_rn(pr, __fmul_rn((float)v[4], vp[4]));
                    pr = __fadd_rn(pr, __fmul_rn((float)v[5], vp[5]));
                    prediction = f32_as_i32(pr);
                }
                const uint32_t sym = pack_signed(sub_w(value, prediction));
                if (sym < 1024u)
                    atomicAdd(&s_hist[bucket * 1024u + sym], 1u); // bump_freq, entropy_coding.rs:98-100
                else
                    atomicAdd(&s_oob, 1u); // the reference would panic (index out of bounds)
            }
            if (a.bucket) a.bucket[(size_t)cell * kCell + p] = (uint8_t)bucket;
            if (a.prediction) a.prediction[(size_t)cell * kCell + p] = prediction;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int i = tid; i < kHistBins; i += kPredThreads) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&a.hist[i], c);
    }
    if (tid == 0 && s_oob) atomicAdd(a.n_oob, (unsigned long long)s_oob);
}

// ------------------------------------------------------------------------------------------------
// K3: inverse transform. One wave per (cell, channel); mirror of fwd_wave. Pixels are written with
// byte stores straight from registers (each pixel has exactly one owning cell).
// ------------------------------------------------------------------------------------------------
struct InvArgs {
    const int32_t *coefs;
    uint8_t *pixels;
    const Int2 *centers;
    int32_t width, height, channels;
    uint32_t F;
    int32_t q_identity;
    QMatrix q;
};

__device__ __forceinline__ int dequant_ref(int v, int heap_index, const InvArgs &a) {
    // quantization::decode divides like encode (quantization.rs:37); reproduced bit for bit.
    if (a.q_identity || v == kNone) return v;
    return v / a.q.q[quant_layer(heap_index)];
}

// One inverse butterfly (extract_values, wavelet_transform.rs:365-376): children of a node whose difference is None stay 0.
__device__ __forceinline__ void unpair(int low, int d, int &left, int &right) {
    if (d == kNone) {
        left = 0;
        right = 0;
    } else {
        right = low - (d / 2);
        left = d + right;
    }
}

__global__ void __launch_bounds__(256) inverse_transform_kernel(const InvArgs a) {
    const int lane = threadIdx.x & 63;
    const uint32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= a.F * (uint32_t)a.channels) return;
    const uint32_t ch = item / a.F, cell = item - ch * a.F;
    const int32_t *in = a.coefs + ((size_t)ch * a.F + cell) * kCell;
    const int4 c8 = *reinterpret_cast<const int4 *>(in + 256 + 4 * lane);
    const int2 c7 = *reinterpret_cast<const int2 *>(in + 128 + 2 * lane);
    int d8[4] = {c8.x, c8.y, c8.z, c8.w}, d7[2] = {c7.x, c7.y};
    int d6 = in[64 + lane], low = in[lane];
#pragma unroll
    for (int i = 0; i < 4; i++) d8[i] = dequant_ref(d8[i], 256 + 4 * lane + i, a);
#pragma unroll
    for (int i = 0; i < 2; i++) d7[i] = dequant_ref(d7[i], 128 + 2 * lane + i, a);
    d6 = dequant_ref(d6, 64 + lane, a);
    low = dequant_ref(low, lane, a);

    int s = __shfl(low, 0); // low_pass_values[1] = coefficients[0].unwrap()  (:361)
#pragma unroll
    for (int j = 5; j >= 0; j--) { // levels 0..5
        const int lv = 5 - j;
        const int d = __shfl(low, (1 << lv) + (lane >> (j + 1)));
        int l, r;
        unpair(s, d, l, r);
        s = ((lane >> j) & 1) ? r : l;
    }
    int s7[2], s8[4], leaf[8];
    unpair(s, d6, s7[0], s7[1]);
#pragma unroll
    for (int i = 0; i < 2; i++) unpair(s7[i], d7[i], s8[2 * i], s8[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 4; i++) unpair(s8[i], d8[i], leaf[2 * i], leaf[2 * i + 1]);

    const Int2 cen = a.centers[cell];
    const int x0 = cen.x + lane_dx(lane), y0 = cen.y + lane_dy(lane);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (d8[j >> 1] == kNone) continue; // `if let Some(dif)` at the last level (:365, :368-372)
        const int x = x0 + leaf_dx(j), y = y0 + leaf_dy(j);
        if (x >= 0 && y >= 0 && x < a.width && y < a.height) // set_pixel, images.rs:104
            a.pixels[((size_t)y * a.width + x) * a.channels + ch] = (uint8_t)min(max(leaf[j], 0), 255);
    }
}

} // namespace

size_t fwd_lds_bytes(const DevicePlan &p) { return (size_t)p.lds_pitch * p.lds_rows; }

bool device_footprint_matches(const StaticTables &st) {
    for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++) {
            const Int2 o = st.leaf_off[8 * l + j];
            if (o.x != lane_dx(l) + leaf_dx(j) || o.y != lane_dy(l) + leaf_dy(j)) return false;
        }
    return true;
}

hipError_t launch_fwd_transform_quant(const DevicePlan &p, uint32_t n_images, const uint8_t *pixels, size_t pixel_stride, int32_t *coefs,
                                      size_t coef_stride, const QMatrix &q, hipStream_t stream) {
    FwdArgs a{};
    a.pixels = pixels;
    a.pixel_stride = pixel_stride;
    a.coefs = coefs;
    a.coef_stride = coef_stride;
    a.tiles = p.tiles;
    a.tile_cells = p.tile_cells;
    a.centers = p.centers;
    a.interior = p.interior;
    a.width = p.width;
    a.height = p.height;
    a.F = p.F;
    a.pitch = p.lds_pitch;
    a.cpr = (uint32_t)p.lds_pitch / 16u;
    a.cpr_magic = (uint32_t)(((1ull << 32) + a.cpr - 1) / a.cpr);
    a.q = q;
    a.q_identity = 1;
    for (int i = 0; i <= 9; i++) a.q_identity &= (q.q[i] == 1); // layers 0..9 are the only ones a 512-node cell uses
    const size_t lds = fwd_lds_bytes(p);
    const dim3 grid(p.n_tiles, n_images), block(kFwdThreads);
    auto kern = p.channels == 1 ? fwd_transform_quant_kernel<1> : fwd_transform_quant_kernel<3>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_predict_histogram(const DevicePlan &p, const int32_t *coefs_channel, const PredictParams &pp, uint8_t *bucket,
                                    int32_t *prediction, uint32_t *hist, unsigned long long *n_oob, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(hist, 0, kHistBins * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(n_oob, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    PredArgs a{};
    a.coefs = coefs_channel;
    a.nbr_cells = p.nbr_cells;
    a.nbr_table = p.nbr_table;
    a.bucket = bucket;
    a.prediction = prediction;
    a.hist = hist;
    a.n_oob = n_oob;
    a.F = p.F;
    a.pp = pp;
    const uint32_t per_block = kPredThreads / 64;
    uint32_t blocks = (p.F + per_block - 1) / per_block;
    if (blocks > p.hist_blocks) blocks = p.hist_blocks;
    hipLaunchKernelGGL(predict_histogram_kernel, dim3(blocks), dim3(kPredThreads), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_inverse_transform(const DevicePlan &p, const int32_t *coefs, const QMatrix &q, uint8_t *pixels, hipStream_t stream) {
    // RasterImage::from_wavelet starts from an all-zero raster (wavelet_transform.rs:309-317)
    hipError_t e = hipMemsetAsync(pixels, 0, (size_t)p.width * p.height * p.channels, stream);
    if (e != hipSuccess) return e;
    InvArgs a{};
    a.coefs = coefs;
    a.pixels = pixels;
    a.centers = p.centers;
    a.width = p.width;
    a.height = p.height;
    a.channels = p.channels;
    a.F = p.F;
    a.q = q;
    a.q_identity = 1;
    for (int i = 0; i <= 9; i++) a.q_identity &= (q.q[i] == 1);
    const uint32_t items = p.F * (uint32_t)p.channels;
    hipLaunchKernelGGL(inverse_transform_kernel, dim3((items + 3) / 4), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace fri
