// k3_inverse.hip -- K3 inverse_transform: dequantisation + inverse residue transform + clamp
// (stages/quantization.rs:27-45; stages/wavelet_transform.rs:358-381; images.rs:103-111).
#include "device_common.hpp"

namespace fri {
namespace {

// ------------------------------------------------------------------------------------------------
// K3: inverse transform. A workgroup walks the forward kernel's tiles of its share (<= cells_per_tile cells of one
// band each), one wave per (cell, channel) item; mirror of fwd_wave. A cell's 512 pixels are scattered over ~50
// rows, so byte stores straight from registers cost one L2 write request per pixel (measured: 41 us). Instead the
// waves scatter into an LDS image of the tile's pixel rectangle -- 16 bit per byte: value | 0x100, the ninth bit
// says "a cell of this tile owns the byte" -- and the workgroup then writes the rectangle out row by row: one
// aligned dword store where all four bytes are owned (90 % of the bytes: the interior of the tile's footprint);
// the dwords on its fractal rim are queued and written with byte stores from densely packed lanes afterwards
// (a store instruction costs the same with 2 or 64 active lanes). Every pixel has exactly one owning cell, so no
// byte is written twice and none is skipped. The coefficients of tile i + 1 are loaded while tile i is processed.
// ------------------------------------------------------------------------------------------------
#ifndef FRI_K3_THREADS
#define FRI_K3_THREADS 256
#endif
constexpr int kInvThreads = FRI_K3_THREADS;
constexpr int kInvWaves = kInvThreads / 64;
constexpr int kInvMaxItemsPerWave = 4; // (cell, channel) items of one tile per transform wave

struct InvArgs {
    const int32_t *coefs;
    uint8_t *pixels;
    size_t coef_stride, pixel_stride; // images of a batch (grid.y): image k at coefs + k * coef_stride (elements), pixels + k * pixel_stride (bytes)
    const Tile *tiles;
    const TileCell *tile_meta; // in tile order
    const int32_t *wg_tiles;   // [n_wg + 1]
    int32_t width, height, channels;
    uint32_t F, n_wg;          // n_wg: workgroups = groups of `group` consecutive shares
    uint32_t group;
    int32_t buf_bytes;   // LDS pixel rectangle (16 bit per byte), multiple of 16
    int32_t queue_bytes; // LDS rim queue per wave
    int32_t max_wg_tiles;
    int32_t q_identity;
    int32_t q_multiply; // 0: the reference's quantization::decode, which divides (quantization.rs:37); 1: the inverse of the quantiser (wrapping multiply)
    int32_t ablate; // timing experiments only (FRI_HIP_K3_ABLATE): 1 = no global stores, 2 = no LDS scatter, 4 = no transform, 8 = coefficient loads for the first tile only (lists kernel)
    unsigned long long *trace; // diagnostic timeline, null in production
    // static write-out lists (geometry.hpp: InvTileLists), used by inverse_transform_lists_kernel
    const InvTileLists *lists;
    const uint16_t *quads, *dwords;
    const uint32_t *parts;
    int32_t rect_bytes; // LDS bytes of the largest tile rectangle (1 byte per byte)
    QMatrix q;
};

__device__ __forceinline__ int dequant_ref(int v, int heap_index, const InvArgs &a) {
    // quantization::decode divides like encode (quantization.rs:37); reproduced bit for bit.
    if (a.q_identity || v == kNone) return v;
    if (a.q_multiply) return (int)((unsigned)v * (unsigned)a.q.q[quant_layer(heap_index)]); // fri_hip_plan_set_dequantiser(plan, FRI_HIP_DEQUANT_MULTIPLY)
    return v / a.q.q[quant_layer(heap_index)];
}

// The eight coefficient dwords a lane holds of one (cell, channel) item: heap nodes lane (levels 0-5 and the DC), 64 + lane,
// 128 + 2 lane + {0, 1}, 256 + 4 lane + {0..3}.
struct InvRegs {
    int32_t d8[4], d7[2], d6, low;
};

__device__ __forceinline__ InvRegs inv_load(const int32_t *in, int lane) {
    // read once: streaming loads keep the eight L2s for the pixel lines that neighbouring tiles complete
    const i32x4 c8 = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(in + 256 + 4 * lane));
    const i32x2 c7 = __builtin_nontemporal_load(reinterpret_cast<const i32x2 *>(in + 128 + 2 * lane));
    InvRegs r;
    r.d8[0] = c8.x, r.d8[1] = c8.y, r.d8[2] = c8.z, r.d8[3] = c8.w;
    r.d7[0] = c7.x, r.d7[1] = c7.y;
    r.d6 = __builtin_nontemporal_load(in + 64 + lane);
    r.low = __builtin_nontemporal_load(in + lane);
    return r;
}

// extract_values for one item (wavelet_transform.rs:358-380) on the lane-distributed tree: six cross-lane levels top-down,
// then levels 6-8 in registers. leaf[j] = value of leaf 8 lane + j (0 where the last difference is None: `if let Some(dif)`
// at :365 leaves those pixels at the raster's initial 0).
// SOME = the wave has checked that none of the item's 512 coefficients is None (every interior cell of an encoder's
// output): the butterflies then need no None test. Wrapping arithmetic like a release build of the reference.
template <bool SOME>
__device__ __forceinline__ void unpair_t(int low, int d, int &left, int &right) {
    if (!SOME && d == kNone) {
        left = 0;
        right = 0;
    } else {
        right = sub_w(low, d / 2);
        left = add_w(d, right);
    }
}
template <bool SOME>
__device__ __forceinline__ void inv_wave(InvRegs c, int lane, const InvArgs &a, int (&leaf)[8]) {
    if (!a.q_identity) {
#pragma unroll
        for (int i = 0; i < 4; i++) c.d8[i] = dequant_ref(c.d8[i], 256 + 4 * lane + i, a);
#pragma unroll
        for (int i = 0; i < 2; i++) c.d7[i] = dequant_ref(c.d7[i], 128 + 2 * lane + i, a);
        c.d6 = dequant_ref(c.d6, 64 + lane, a);
        c.low = dequant_ref(c.low, lane, a);
    }
    int s = __shfl(c.low, 0); // low_pass_values[1] = coefficients[0].unwrap()  (:361)
#pragma unroll
    for (int j = 5; j >= 0; j--) { // levels 0..5
        const int lv = 5 - j;
        const int d = __shfl(c.low, (1 << lv) + (lane >> (j + 1)));
        int l, r;
        unpair_t<SOME>(s, d, l, r);
        s = ((lane >> j) & 1) ? r : l;
    }
    int s7[2], s8[4];
    unpair_t<SOME>(s, c.d6, s7[0], s7[1]);
#pragma unroll
    for (int i = 0; i < 2; i++) unpair_t<SOME>(s7[i], c.d7[i], s8[2 * i], s8[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 4; i++) unpair_t<SOME>(s8[i], c.d8[i], leaf[2 * i], leaf[2 * i + 1]);
}
// item = cell of the tile x channels + channel. libfri's images have one or three channels (images.rs: Luma, RGB, YCbCr): a division by a run-time C is ~25 scalar
// instructions per item in front of the item's loads.
__device__ __forceinline__ void inv_item_split(int item, int C, int &cl, int &ch) {
    if (C == 1) {
        cl = item, ch = 0;
    } else if (C == 3) {
        cl = item / 3, ch = item - 3 * cl;
    } else {
        cl = item / C, ch = item - cl * C;
    }
}
// Touch the registers of prefetched items: the compiler places its wait for their loads in front of this, i.e. where the caller wants it.
template <int NI>
__device__ __forceinline__ void inv_land(InvRegs (&r)[NI]) {
#pragma unroll
    for (int s = 0; s < NI; s++)
        asm volatile("" : "+v"(r[s].d8[0]), "+v"(r[s].d8[1]), "+v"(r[s].d8[2]), "+v"(r[s].d8[3]), "+v"(r[s].d7[0]), "+v"(r[s].d7[1]), "+v"(r[s].d6), "+v"(r[s].low));
}
__device__ __forceinline__ bool inv_has_none(const InvRegs &c) {
    const bool n = c.d8[0] == kNone || c.d8[1] == kNone || c.d8[2] == kNone || c.d8[3] == kNone || c.d7[0] == kNone || c.d7[1] == kNone ||
                   c.d6 == kNone || c.low == kNone;
    return __any(n);
}

// The loads of one tile's items for this wave. Slots past the tile's items re-load its last item (no conditional loads:
// a load the compiler cannot prove executed costs an immediate wait).
template <int NI>
__device__ __forceinline__ void inv_prefetch(const InvArgs &a, const int32_t *img_coefs, const Tile &t, const TileCell *cells, int wave, int lane, InvRegs (&r)[NI]) {
    const int C = a.channels, n_items = t.cell_count * C;
#pragma unroll
    for (int s = 0; s < NI; s++) {
        const int item = min(wave + kInvWaves * s, n_items - 1);
        int cl, ch;
        inv_item_split(item, C, cl, ch);
        r[s] = inv_load(img_coefs + ((size_t)ch * a.F + (uint32_t)cells[cl].cell) * kCell, lane);
    }
}

// Descriptors come out of LDS into vector registers; every lane reads the same entry, so they are moved to scalar registers
// (uniform branches, scalar address arithmetic) - the compiler cannot prove the uniformity of an LDS load by itself.
__device__ __forceinline__ Tile scalar_tile(const Tile &l) {
    Tile t;
    t.x_lo = __builtin_amdgcn_readfirstlane(l.x_lo);
    t.y_lo = __builtin_amdgcn_readfirstlane(l.y_lo);
    t.width_px = __builtin_amdgcn_readfirstlane(l.width_px);
    t.n_rows = __builtin_amdgcn_readfirstlane(l.n_rows);
    t.cell_begin = __builtin_amdgcn_readfirstlane(l.cell_begin);
    t.cell_count = __builtin_amdgcn_readfirstlane(l.cell_count);
    return t;
}
__device__ __forceinline__ InvTileLists scalar_lists(const InvTileLists &l) {
    InvTileLists r;
    r.quad_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.quad_begin);
    r.quad_count = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.quad_count);
    r.dword_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.dword_begin);
    r.dword_count = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.dword_count);
    r.part_begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.part_begin);
    r.part_count = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.part_count);
    return r;
}

template <int NI>
__global__ void __launch_bounds__(kInvThreads) inverse_transform_kernel(const InvArgs a) {
    // this image of the batch (grid.y). Scalars, not a modified copy of the argument struct: a copy would live in scratch memory
    // (the quantiser array inside is indexed dynamically) and every argument access with it.
    const int32_t *const img_coefs = a.coefs + blockIdx.y * a.coef_stride;
    uint8_t *const img_pixels = a.pixels + blockIdx.y * a.pixel_stride;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint16_t *img16 = reinterpret_cast<uint16_t *>(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t *queue = reinterpret_cast<uint16_t *>(lds + a.buf_bytes + wave * a.queue_bytes);
    Tile *lds_tiles = reinterpret_cast<Tile *>(lds + a.buf_bytes + kInvWaves * a.queue_bytes);
    TileCell *lds_cells = reinterpret_cast<TileCell *>(lds_tiles + a.max_wg_tiles);
    // Blocks of one XCD take a contiguous range of shares: the rim bytes of neighbouring tiles complete their lines in ONE L2.
    const uint32_t wg = xcd_contiguous_share(blockIdx.x, a.n_wg);
    const int tb = a.wg_tiles[wg * a.group], te = a.wg_tiles[wg * a.group + a.group];
    const int C = a.channels;
    const uint32_t base_lo = (uint32_t)reinterpret_cast<uintptr_t>(img_pixels);
    const uint32_t wc = (uint32_t)a.width * (uint32_t)C;
    trace_stamp(a.trace, wg, 0, tid);
    {
        const Tile first = a.tiles[tb], last = a.tiles[te - 1];
        const int n_cells = last.cell_begin + last.cell_count - first.cell_begin;
        if (tid < te - tb) lds_tiles[tid] = a.tiles[tb + tid];
        for (int i = tid; i < n_cells; i += kInvThreads) {
            TileCell tc = a.tile_meta[first.cell_begin + i];
            lds_cells[i] = tc;
        }
        u32x4 *z = reinterpret_cast<u32x4 *>(lds);
        for (int i = tid; i < a.buf_bytes / 16; i += kInvThreads) z[i] = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    const int cell0 = __builtin_amdgcn_readfirstlane(lds_tiles[0].cell_begin);
    trace_stamp(a.trace, wg, 1, tid);

    InvRegs pre[NI];
    inv_prefetch<NI>(a, img_coefs, scalar_tile(lds_tiles[0]), lds_cells, wave, lane, pre);
    for (int ti = tb; ti < te; ti++) {
        const Tile t = scalar_tile(lds_tiles[ti - tb]);
        InvRegs cur[NI];
#pragma unroll
        for (int s = 0; s < NI; s++) cur[s] = pre[s];
        {
            const Tile tn = scalar_tile(lds_tiles[min(ti + 1, te - 1) - tb]);
            inv_prefetch<NI>(a, img_coefs, tn, lds_cells + (tn.cell_begin - cell0), wave, lane, pre);
        }
        const int n_items = t.cell_count * C;
        // Staged rows start at the 16-byte boundary at or below their first byte (lead-in 0..15), so every global quad is an
        // aligned 16-byte store; rq quads (16 output bytes = 16 LDS halfwords each) per row.
        const int rq = (t.width_px * C + 30) >> 4;
        const int pitch16 = rq * 16; // LDS halfwords per staged row
        int leaf[NI][8];
#pragma unroll
        for (int s = 0; s < NI; s++) {
            if (inv_has_none(cur[s])) inv_wave<false>(cur[s], lane, a, leaf[s]);
            else inv_wave<true>(cur[s], lane, a, leaf[s]);
        }
#pragma unroll
        for (int s = 0; s < NI; s++) {
            const int item = wave + kInvWaves * s;
            if (item < n_items && !(ablate_flags(a.ablate) & 2)) {
                const int cl = item / C, ch = item - cl * C;
                const TileCell tc = lds_cells[t.cell_begin - cell0 + cl];
                const int x0 = tc.cx + lane_dx(lane), y0 = tc.cy + lane_dy(lane);
                int rowbase[3];
#pragma unroll
                for (int dy = 0; dy < 3; dy++) {
                    const int y = y0 + dy;
                    const uint32_t g = (uint32_t)y * wc + (uint32_t)(t.x_lo * C); // byte offset of the staged row (only bits 0-3 matter)
                    rowbase[dy] = (y - t.y_lo) * pitch16 + (int)((base_lo + g) & 15u) + (x0 - t.x_lo) * C + ch;
                }
                if (__builtin_amdgcn_readfirstlane(tc.interior)) {
#pragma unroll
                    for (int j = 0; j < 8; j++) img16[rowbase[leaf_dy(j)] + leaf_dx(j) * C] = (uint16_t)(0x100 | min(max(leaf[s][j], 0), 255));
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int x = x0 + leaf_dx(j), y = y0 + leaf_dy(j);
                        if (x >= 0 && y >= 0 && x < a.width && y < a.height) // set_pixel, images.rs:104
                            img16[rowbase[leaf_dy(j)] + leaf_dx(j) * C] = (uint16_t)(0x100 | min(max(leaf[s][j], 0), 255));
                    }
                }
            }
        }
        lds_barrier(); // the rectangle is complete

        // Quad pass over the flattened rectangle: a quad whose 16 bytes are all owned goes out as one store, a partly owned
        // one is queued as (row << 8 | quad).
        int qn = 0;
        const int n_quads = t.n_rows * rq;
        const float inv_rq = 1.0f / (float)rq;
        for (int q0 = 0; q0 < n_quads; q0 += kInvThreads) {
            const int qi = q0 + tid;
            u32x4 lo = u32x4{0u, 0u, 0u, 0u}, hi = lo;
            u32x4 *src = reinterpret_cast<u32x4 *>(img16) + 2 * qi;
            int r = 0, k = 0;
            if (qi < n_quads) {
                lo = src[0], hi = src[1];
                r = (int)(((float)qi + 0.5f) * inv_rq); // exact: qi < 2^16
                k = qi - r * rq;
            }
            const uint32_t all = lo.x & lo.y & lo.z & lo.w & hi.x & hi.y & hi.z & hi.w & 0x01000100u;
            const uint32_t any = (lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) & 0x01000100u;
            const bool full = all == 0x01000100u, rim = any != 0 && !full;
            if (full) {
                src[0] = u32x4{0u, 0u, 0u, 0u};
                src[1] = u32x4{0u, 0u, 0u, 0u};
                const size_t g = ((size_t)(t.y_lo + r) * (size_t)a.width + (size_t)t.x_lo) * (size_t)C;
                uint8_t *p = img_pixels + g - (int)((base_lo + (uint32_t)g) & 15u) + 16 * k; // 16-byte aligned
                const u32x4 out{__builtin_amdgcn_perm(lo.y, lo.x, 0x06040200u), __builtin_amdgcn_perm(lo.w, lo.z, 0x06040200u),
                                __builtin_amdgcn_perm(hi.y, hi.x, 0x06040200u), __builtin_amdgcn_perm(hi.w, hi.z, 0x06040200u)};
                if (!(ablate_flags(a.ablate) & 1)) *reinterpret_cast<u32x4 *>(p) = out;
            }
            const unsigned long long m = __ballot(rim);
            if (rim) queue[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(r << 8 | k);
            qn += __popcll(m);
        }
        // Rim pass: one queued quad per lane; owned dwords as dword stores, the rest byte by byte.
        for (int e0 = 0; e0 < qn; e0 += 64) {
            const int e = e0 + lane;
            if (e < qn) {
                const int rk = queue[e], r = rk >> 8, k = rk & 255;
                const size_t g = ((size_t)(t.y_lo + r) * (size_t)a.width + (size_t)t.x_lo) * (size_t)C;
                uint8_t *p = img_pixels + g - (int)((base_lo + (uint32_t)g) & 15u) + 16 * k;
                u32x4 *src = reinterpret_cast<u32x4 *>(img16 + r * pitch16) + 2 * k;
                const u32x4 lo = src[0], hi = src[1];
                src[0] = u32x4{0u, 0u, 0u, 0u};
                src[1] = u32x4{0u, 0u, 0u, 0u};
                const uint32_t u[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (!(ablate_flags(a.ablate) & 1)) {
#pragma unroll
                    for (int d = 0; d < 4; d++) {
                        const uint32_t v0 = u[2 * d], v1 = u[2 * d + 1];
                        const uint32_t own = (v0 & 0x01000100u) | ((v1 & 0x01000100u) << 1);
                        if (own == 0x03000300u) {
                            *reinterpret_cast<uint32_t *>(p + 4 * d) = __builtin_amdgcn_perm(v1, v0, 0x06040200u);
                        } else if (own != 0) {
                            if (v0 & 0x00000100u) p[4 * d + 0] = (uint8_t)v0;
                            if (v0 & 0x01000000u) p[4 * d + 1] = (uint8_t)(v0 >> 16);
                            if (v1 & 0x00000100u) p[4 * d + 2] = (uint8_t)v1;
                            if (v1 & 0x01000000u) p[4 * d + 3] = (uint8_t)(v1 >> 16);
                        }
                    }
                }
            }
        }
        lds_barrier(); // the rectangle is all zero again
        trace_stamp(a.trace, wg, 2 + ti - tb, tid);
    }
    trace_exit(a.trace, wg, tid);
}

// K3 with static write-out lists. Which bytes of a tile's rectangle its own cells write is geometry, so the plan holds, per
// tile, the quads that are written whole, the whole dwords inside partly owned quads and the partly owned dwords with their byte
// masks (build_inverse_lists, geometry.hpp). The kernel then only scatters plain bytes into LDS and walks the two lists: no ownership bits, no scan
// of the (55 % empty) rectangle, no zeroing, no queue. Needs every image row to start 16-byte aligned (base pointer and
// width * channels multiples of 16): the launcher falls back to inverse_transform_kernel otherwise.
constexpr int kInvListPre = 3; // list entries a thread holds in flight per list and tile (more are loaded on demand)
static_assert((size_t)kInvListPre * kInvThreads <= kInvListPad, "the lists' pad covers a thread's unconditional loads");
template <int NI>
__global__ void __launch_bounds__(kInvThreads) inverse_transform_lists_kernel(const InvArgs a) {
    const int32_t *const img_coefs = a.coefs + blockIdx.y * a.coef_stride; // this image of the batch, see inverse_transform_kernel
    uint8_t *const img_pixels = a.pixels + blockIdx.y * a.pixel_stride;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t *img = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Tile *lds_tiles = reinterpret_cast<Tile *>(lds + a.rect_bytes);
    InvTileLists *lds_lists = reinterpret_cast<InvTileLists *>(lds_tiles + a.max_wg_tiles);
    TileCell *lds_cells = reinterpret_cast<TileCell *>(lds_lists + a.max_wg_tiles);
    const uint32_t wg = xcd_contiguous_share(blockIdx.x, a.n_wg);
    const int tb = a.wg_tiles[wg * a.group], te = a.wg_tiles[wg * a.group + a.group];
    const int C = a.channels;
    const size_t wc = (size_t)a.width * (size_t)C;
    trace_stamp(a.trace, wg, 0, tid);
    {
        const Tile first = a.tiles[tb], last = a.tiles[te - 1];
        const int n_cells = last.cell_begin + last.cell_count - first.cell_begin;
        if (tid < te - tb) {
            lds_tiles[tid] = a.tiles[tb + tid];
            lds_lists[tid] = a.lists[tb + tid];
        }
        for (int i = tid; i < n_cells; i += kInvThreads) {
            TileCell tc = a.tile_meta[first.cell_begin + i];
            lds_cells[i] = tc;
        }
    }
    __syncthreads();
    const int cell0 = __builtin_amdgcn_readfirstlane(lds_tiles[0].cell_begin);
    trace_stamp(a.trace, wg, 1, tid);

    InvRegs pre[NI];
    inv_prefetch<NI>(a, img_coefs, scalar_tile(lds_tiles[0]), lds_cells, wave, lane, pre);
    for (int ti = tb; ti < te; ti++) {
        const Tile t = scalar_tile(lds_tiles[ti - tb]);
        const InvTileLists L = scalar_lists(lds_lists[ti - tb]);
        InvRegs cur[NI];
#pragma unroll
        for (int s = 0; s < NI; s++) cur[s] = pre[s];
        {
            const Tile tn = scalar_tile(lds_tiles[min(ti + 1, te - 1) - tb]);
            if (!(ablate_flags(a.ablate) & 8)) inv_prefetch<NI>(a, img_coefs, tn, lds_cells + (tn.cell_begin - cell0), wave, lane, pre); // (8: timing only, the first tile's coefficients for every tile)
        }
        // this thread's first entries of the three lists, in flight across the transform (indices clamped: the loads must be
        // unconditional; a tile without entries of one kind re-reads entry 0 of the array, which always exists)
        // (round 5: no clamping - the arrays are followed by kInvListPad unused entries, a thread past its tile's count loads the next tile's or the pad's and never
        // looks at them; uniform base + the thread's 32-bit offset, the distance between a thread's entries in the instructions' offset field)
        uint16_t qe[kInvListPre], de[kInvListPre];
        uint32_t pe[kInvListPre];
        {
            const uint16_t *qb = a.quads + L.quad_begin, *db = a.dwords + L.dword_begin;
            const uint32_t *pb = a.parts + L.part_begin;
#pragma unroll
            for (int m = 0; m < kInvListPre; m++) qe[m] = qb[(uint32_t)tid + kInvThreads * m], de[m] = db[(uint32_t)tid + kInvThreads * m], pe[m] = pb[(uint32_t)tid + kInvThreads * m];
        }
        const int n_items = t.cell_count * C;
        const int a0 = (t.x_lo * C) & ~15;                         // first image byte column of the staged rows
        const int pitch = (((t.x_lo + t.width_px) * C - 1 - a0) / 16 + 1) * 16; // LDS bytes per staged row
        int leaf[NI][8];
        { // one test for the wave's items: the common case (no None anywhere) is ONE basic block in which the items' dependent chains - six LDS round trips each - interleave
            // (up to two items: four interleaved chains need 150 vector registers, a fourth workgroup per CU no longer fits)
            if (ablate_flags(a.ablate) & 4) { // (timing only: no transform)
#pragma unroll
                for (int s = 0; s < NI; s++) {
                    leaf[s][0] = cur[s].d8[0], leaf[s][1] = cur[s].d8[1], leaf[s][2] = cur[s].d8[2], leaf[s][3] = cur[s].d8[3];
                    leaf[s][4] = cur[s].d7[0], leaf[s][5] = cur[s].d7[1], leaf[s][6] = cur[s].d6, leaf[s][7] = cur[s].low;
                }
            } else if (NI <= 2) {
                bool none = false;
#pragma unroll
                for (int s = 0; s < NI; s++) none |= inv_has_none(cur[s]);
                if (none) {
#pragma unroll
                    for (int s = 0; s < NI; s++) inv_wave<false>(cur[s], lane, a, leaf[s]);
                } else {
#pragma unroll
                    for (int s = 0; s < NI; s++) inv_wave<true>(cur[s], lane, a, leaf[s]);
                }
            } else {
#pragma unroll
                for (int s = 0; s < NI; s++) {
                    if (inv_has_none(cur[s])) inv_wave<false>(cur[s], lane, a, leaf[s]);
                    else inv_wave<true>(cur[s], lane, a, leaf[s]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NI; s++) {
            const int item = wave + kInvWaves * s;
            if (item < n_items && !(ablate_flags(a.ablate) & 2)) {
                int cl, ch;
                inv_item_split(item, C, cl, ch);
                const TileCell tc = lds_cells[t.cell_begin - cell0 + cl];
                const int x0 = tc.cx + lane_dx(lane), y0 = tc.cy + lane_dy(lane);
                // (cell part: scalar) + (lane part): one vector multiply-add per item
                const int at = (__builtin_amdgcn_readfirstlane(tc.cy) - t.y_lo) * pitch + __builtin_amdgcn_readfirstlane(tc.cx) * C + ch - a0 +
                               lane_dy(lane) * pitch + lane_dx(lane) * C;
                if (__builtin_amdgcn_readfirstlane(tc.interior)) {
#pragma unroll
                    for (int j = 0; j < 8; j++) img[at + leaf_dy(j) * pitch + leaf_dx(j) * C] = (uint8_t)min(max(leaf[s][j], 0), 255);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int x = x0 + leaf_dx(j), y = y0 + leaf_dy(j);
                        if (x >= 0 && y >= 0 && x < a.width && y < a.height) // set_pixel, images.rs:104
                            img[at + leaf_dy(j) * pitch + leaf_dx(j) * C] = (uint8_t)min(max(leaf[s][j], 0), 255);
                    }
                }
            }
        }
        // The next tile's coefficients have had the transform to arrive: they land HERE, in front of this tile's stores. Loads and stores count in one in-order
        // counter and the write-out's store count is a run-time number: at the top of the next iteration the compiler's wait for them was a wait for zero, this
        // tile's stores included. (Measured, round 5: the same 27.3-28.8 us either way; requesting the tile AFTER next here, so that a load has a whole iteration
        // to arrive, is slower - 30.2-31.5 us: gpurun_out/r5_k3f.)
        inv_land<NI>(pre);
        lds_barrier(); // the rectangle holds every byte this tile owns
        uint8_t *out0 = img_pixels + (size_t)t.y_lo * wc + (size_t)a0; // quad (r, k) -> out0 + r * wc + 16 k, 16-byte aligned
        // (a tile's rows x the image's row bytes stay far below 2^32, r < 256 and the row bytes below 2^24: 24-bit multiplies, a 32-bit offset on a uniform base)
        const uint32_t wc24 = (uint32_t)wc, pitch24 = (uint32_t)pitch;
        for (uint32_t e = tid, m = 0; e < L.quad_count; e += kInvThreads, m++) { // whole quads
            const uint32_t rk = m < kInvListPre ? (m == 0 ? qe[0] : m == 1 ? qe[1] : qe[2]) : a.quads[L.quad_begin + e];
            const uint32_t r = rk >> 8, k = rk & 255u;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(img + __umul24(r, pitch24) + 16 * k);
            if (!(ablate_flags(a.ablate) & 1)) *reinterpret_cast<u32x4 *>(out0 + (__umul24(r, wc24) + 16 * k)) = v;
        }
        for (uint32_t e = tid, m = 0; e < L.dword_count; e += kInvThreads, m++) { // whole dwords of partly owned quads
            const uint32_t rd = m < kInvListPre ? (m == 0 ? de[0] : m == 1 ? de[1] : de[2]) : a.dwords[L.dword_begin + e];
            const uint32_t r = rd >> 8, d = rd & 255u;
            const uint32_t v = *reinterpret_cast<const uint32_t *>(img + __umul24(r, pitch24) + 4 * d);
            if (!(ablate_flags(a.ablate) & 1)) *reinterpret_cast<uint32_t *>(out0 + (__umul24(r, wc24) + 4 * d)) = v;
        }
        for (uint32_t e = tid, m = 0; e < L.part_count; e += kInvThreads, m++) { // the fractal rim proper: byte stores
            const uint32_t ent = m < kInvListPre ? (m == 0 ? pe[0] : m == 1 ? pe[1] : pe[2]) : a.parts[L.part_begin + e];
            const uint32_t r = ent >> 12, d = (ent >> 4) & 255u, nib = ent & 15u;
            const uint32_t v = *reinterpret_cast<const uint32_t *>(img + __umul24(r, pitch24) + 4 * d);
            uint8_t *p = out0 + (__umul24(r, wc24) + 4 * d);
            if (!(ablate_flags(a.ablate) & 1)) {
                if (nib & 1u) p[0] = (uint8_t)v;
                if (nib & 2u) p[1] = (uint8_t)(v >> 8);
                if (nib & 4u) p[2] = (uint8_t)(v >> 16);
                if (nib & 8u) p[3] = (uint8_t)(v >> 24);
            }
        }
        if (!(ablate_flags(a.ablate) & 16)) lds_barrier(); // everyone is done reading before the next tile scatters (16: timing only - what a second rectangle would save)
        trace_stamp(a.trace, wg, 2 + ti - tb, tid);
    }
    trace_exit(a.trace, wg, tid);
}


} // namespace

// 16 bit per staged byte; a staged row holds <= lds_pitch bytes including its lead-in (lds_pitch >= widest row + 15).
static size_t inv_buf_bytes(const DevicePlan &p) { return (size_t)p.lds_rows * (size_t)p.lds_pitch * 2; }
// worst case: every quad a wave looks at is a rim quad (2 bytes per entry), rounded to 16
static size_t inv_queue_bytes(const DevicePlan &p) {
    const size_t quads = (size_t)p.lds_rows * (size_t)(p.lds_pitch / 16);
    return (((quads + kInvThreads - 1) / kInvThreads) * 64 * 2 + 15) & ~(size_t)15;
}
size_t inv_lds_bytes(const DevicePlan &p) { return inv_buf_bytes(p) + kInvWaves * inv_queue_bytes(p) + (size_t)p.inv_max_wg_tiles * sizeof(Tile) + (size_t)p.inv_max_wg_cells * sizeof(TileCell); }
// Does a tiling fit the inverse kernels (what launch_inverse_transform would otherwise refuse with hipErrorInvalidConfiguration)? The static limits of both
// kernels and the LDS of the scanning kernel (always available); with_lists: also the lists kernel's LDS (inv_rect_bytes is known once the lists are built).
bool inv_plan_fits(const DevicePlan &p, bool with_lists) {
    const int items_per_wave = (p.max_tile_cells * p.channels + kInvWaves - 1) / kInvWaves;
    if (items_per_wave > kInvMaxItemsPerWave || p.inv_max_wg_tiles > kInvThreads || p.lds_rows > 256 || p.lds_pitch / 16 > 256) return false;
    if (inv_lds_bytes(p) > 160 * 1024) return false;
    if (with_lists && (size_t)p.inv_rect_bytes + (size_t)p.inv_max_wg_tiles * (sizeof(Tile) + sizeof(InvTileLists)) + (size_t)p.inv_max_wg_cells * sizeof(TileCell) > 160 * 1024) return false;
    return true;
}

hipError_t launch_inverse_transform(const DevicePlan &p, uint32_t n_images, const int32_t *coefs, size_t coef_stride, const QMatrix &q, uint8_t *pixels, size_t pixel_stride,
                                    hipStream_t stream) {
    if (!n_images || n_images > 65535u) return hipErrorInvalidValue;
    // RasterImage::from_wavelet starts from an all-zero raster (wavelet_transform.rs:309-317). When every pixel belongs to a
    // retained cell the kernel writes all of them (zeros included); only a lattice with holes (very thin images) needs the fill.
    if (!p.covers_image) {
        for (uint32_t k = 0; k < n_images; k++) {
            hipError_t e = hipMemsetAsync(pixels + k * pixel_stride, 0, (size_t)p.width * p.height * p.channels, stream);
            if (e != hipSuccess) return e;
        }
    }
    InvArgs a{};
    a.coefs = coefs;
    a.pixels = pixels;
    a.coef_stride = coef_stride;
    a.pixel_stride = pixel_stride;
    a.tiles = p.tiles;
    a.tile_meta = p.tile_meta;
    a.wg_tiles = p.wg_tiles;
    a.width = p.width;
    a.height = p.height;
    a.channels = p.channels;
    a.F = p.F;
    a.group = (uint32_t)p.inv_group;
    a.n_wg = p.n_wg / a.group;
    a.buf_bytes = (int32_t)inv_buf_bytes(p);
    a.max_wg_tiles = p.inv_max_wg_tiles;
    a.ablate = p.k3_ablate;
    a.trace = p.trace;
    a.q = q;
    a.q_identity = 1;
    for (int i = 0; i <= 9; i++) a.q_identity &= (q.q[i] == 1);
    a.q_multiply = p.k3_multiply ? 1 : 0;
    a.queue_bytes = (int32_t)inv_queue_bytes(p);
    const int items_per_wave = (p.max_tile_cells * p.channels + kInvWaves - 1) / kInvWaves;
    if (!inv_plan_fits(p, false)) return hipErrorInvalidConfiguration;
    // static write-out lists when every image row starts 16-byte aligned (and their rectangle fits: fri_hip_plan_create checks that for the tilings it builds)
    const bool lists = p.inv_lists && !p.k3_scan && inv_plan_fits(p, true) && (reinterpret_cast<uintptr_t>(pixels) & 15) == 0 && (((size_t)p.width * p.channels) & 15) == 0 &&
                       (n_images == 1 || (pixel_stride & 15) == 0) && (size_t)p.width * p.channels < ((size_t)1 << 24); // (the write-out's 24-bit row-bytes multiply)
    if (lists) {
        a.lists = p.inv_lists;
        a.quads = p.inv_quads;
        a.dwords = p.inv_dwords;
        a.parts = p.inv_parts;
        a.rect_bytes = p.inv_rect_bytes;
        const size_t lds2 = (size_t)p.inv_rect_bytes + (size_t)p.inv_max_wg_tiles * (sizeof(Tile) + sizeof(InvTileLists)) + (size_t)p.inv_max_wg_cells * sizeof(TileCell);
        void (*k2)(const InvArgs) = items_per_wave <= 1 ? inverse_transform_lists_kernel<1> : items_per_wave == 2 ? inverse_transform_lists_kernel<2> : inverse_transform_lists_kernel<4>;
        if (lds2 > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            if (e != hipSuccess) return e;
        }
        (void)hipGetLastError(); // the check behind the launch must not pick up an error an earlier, unrelated call left behind
        hipLaunchKernelGGL(k2, dim3(a.n_wg, n_images), dim3(kInvThreads), lds2, stream, a);
        return hipGetLastError();
    }
    const size_t lds = inv_lds_bytes(p);
    void (*kern)(const InvArgs) = items_per_wave <= 1 ? inverse_transform_kernel<1> : items_per_wave == 2 ? inverse_transform_kernel<2> : inverse_transform_kernel<4>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    (void)hipGetLastError(); // the check behind the launch must not pick up an error an earlier, unrelated call left behind
    hipLaunchKernelGGL(kern, dim3(a.n_wg, n_images), dim3(kInvThreads), lds, stream, a);
    return hipGetLastError();
}

} // namespace fri
