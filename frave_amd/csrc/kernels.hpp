// kernels.hpp -- launch interface between the C ABI (fri_hip.cpp) and the gfx950 kernels (k1_forward.hip, k2_predict.hip, k3_inverse.hip, k4_fit.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

#include "geometry.hpp"

namespace fri {

// Device-resident plan tables (all pointers are device pointers owned by the plan).
struct DevicePlan {
    const Tile *tiles = nullptr;
    const int32_t *tile_cells = nullptr;
    const TileCell *tile_meta = nullptr;  // [F] in tile order
    const int32_t *wg_tiles = nullptr;    // [n_wg + 1] tile range per workgroup share
    const int32_t *wg_tiles_batch = nullptr; // [n_wg_batch + 1] merged shares for launches over many images
    uint32_t n_wg_batch = 0;
    uint32_t n_wg = 0;
    int32_t max_tile_cells = 0;
    bool covers_image = false;            // every pixel of the image is a leaf of a retained cell
    int32_t max_wg_tiles = 0;
    int32_t max_wg_cells = 0;             // most cells in one workgroup share
    int32_t inv_group = 1;                // the inverse kernel's workgroups walk this many consecutive shares each (large images: K1 wants many short shares, K3 one per resident workgroup)
    int32_t inv_max_wg_tiles = 0, inv_max_wg_cells = 0; // the same two maxima per group of inv_group shares
    const Int2 *centers = nullptr;
    const uint8_t *interior = nullptr;
    const uint32_t *valid_mask = nullptr; // [F][16]
    const int32_t *nbr_cells = nullptr;   // [F][kNbr]
    const uint16_t *nbr_table = nullptr;  // [512][6]
    const uint32_t *gather_off = nullptr; // [512][4] the same in bytes for the permuted 1 KiB cell layout (build_gather_tables)
    const uint16_t *pair_pos = nullptr;   // [256] gather_layout.inc
    const uint16_t *heap_of_pos = nullptr; // [512]
    const uint32_t *halo_list = nullptr;   // [1024] K2's sparse halo staging list (build_halo_list)
    int8_t lf_delta[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // K2's low-frequency predictor: slot-list offsets of its neighbour cells (build_lf_deltas), kernel arguments
    const int32_t *pred_slots = nullptr;  // [n_pred_tiles][kPredSlots]
    uint32_t n_pred_tiles = 0;
    uint32_t *hist_partial = nullptr;     // [hist_blocks][10*1024] scratch for the histogram reduction
    unsigned long long *oob_partial = nullptr; // [hist_blocks]
    uint32_t hist_blocks = 0;
    uint8_t *junk = nullptr;              // [pred_blocks][kPredJunkWaves][kPredJunkBytes]: output lines of block slots without a cell (pipelined K2)
    uint32_t pred_blocks = 0;             // workgroups of the pipelined K2: one 1024-thread workgroup per CU
    uint32_t n_tiles = 0;
    uint32_t F = 0;
    int32_t width = 0, height = 0, channels = 0;
    int32_t lds_pitch = 0, lds_rows = 0, cells_per_tile = 0;
    bool k1_batch_shares = true; // FRI_HIP_K1_BATCH_SHARES=0 disables the merged shares (A/B)
    int k1_cached_stores = -1; // tuning (FRI_HIP_K1_CACHED_STORES=0 / 1): force nontemporal / plain coefficient stores; -1: the caller of the launch decides
    int32_t k1_ablate = 0; // timing-only ablation flags, see FwdArgs::ablate
    int32_t k2_ablate = 0; // the same for K2, see PredArgs::ablate
    bool k3_multiply = false; // fri_hip_plan_set_dequantiser: the inverse kernel multiplies by the quantiser instead of reproducing the reference's division
    unsigned long long *trace = nullptr; // [n_wg][16] diagnostic timeline (FRI_HIP_TRACE=1), else null
    bool k1_measuring = false; // fri_hip_plan_tune_forward's measuring copies: their forward launches run the kernel's MEASURE instance (a name of its own in traces)
    unsigned long long *k1_xcd_stat = nullptr; // [8][2] per-XCD workgroup lifetimes of the forward kernel: set by fri_hip_plan_tune_forward on its measuring copies only
    // K3's static write-out lists (null = not built: the kernel scans the rectangle)
    const InvTileLists *inv_lists = nullptr;
    const uint16_t *inv_quads = nullptr, *inv_dwords = nullptr; // each array is followed by kInvListPad entries a thread may load and never uses (the lists kernel
    const uint32_t *inv_parts = nullptr;                        // requests its first entries of a tile without looking at the tile's counts)
    int32_t inv_rect_bytes = 0;
    bool k3_scan = false; // FRI_HIP_K3_SCAN=1: always use the scanning kernel (A/B)
    int32_t k3_ablate = 0; // same for the inverse kernel, see InvArgs::ablate
    int32_t k4_ablate = 0; // same for the fit kernel, see FitArgs::ablate
    int32_t k4_older_eighths = 5; // fit kernel: share of a CU's tiles that goes to its first-dispatched workgroup, in eighths (FRI_HIP_K4_OLDER_EIGHTHS; 0 = equal shares)
};

constexpr uint32_t kPredJunkWaves = 16, kPredJunkBytes = 2560; // per wave: 512 B of bucket + 2 KiB of prediction
constexpr uint32_t kFitAccWords = 3 * 28 + 18 + 2; // integer sums, fixed-point sums (W^T r), ticket, out-of-range count
// The fit kernel's 512 workgroups all add into the same ~100 words: at the memory side same-address atomics take ~12 ns each, so 512 arrivals per
// word were 5-6 us of every launch (per-workgroup time stamps: "loop done" -> "ticket drawn"). The accumulator of a plane is therefore kept in
// kFitShards copies, workgroup b adds into copy b % kFitShards, and the workgroup that draws the last ticket (copy 0 holds it) sums the copies.
constexpr uint32_t kFitShards = 16;
// K2 adds its counts straight into the caller's histogram (k2_predict.hip, pred_hand_over): its accumulator is only the bookkeeping of that hand-over
constexpr uint32_t kPredAccRing = 8, kPredAccWords = 16; // per plane: ten "cleared" flags, pad, ticket, pad, "inexact" flag, the exact kernel's ticket

struct QMatrix {
    int32_t q[32];
};

struct PredictParams {
    float value[3][6];
    float width[3][6];
};

// K1: address-map gather + 9-level residue transform + per-layer quantisation.
// cached_stores: plain coefficient stores (the next kernel of a chain reads them straight away) instead of nontemporal ones (the default: written once, read later or never)
hipError_t launch_fwd_transform_quant(const DevicePlan &p, uint32_t n_images, const uint8_t *pixels, size_t pixel_stride, int32_t *coefs,
                                      size_t coef_stride, const QMatrix &q, hipStream_t stream, bool cached_stores = false,
                                      int16_t *coefs16 = nullptr /* the chains' compact planes instead of `coefs`: int16, None as 0, coef_stride in halfwords */);
// K2: neighbour gather + bucket/prediction + LDS histogram, then the partial-histogram reduction.
// acc_slot < kPredAccRing selects the plan accumulator the launch hands its sums over through (one per stream, fri_hip.cpp).
// The planes (image x channel) one K2 / K4 launch works on: plane k reads coefs + k * coef_stride (int32 elements) and writes its
// per-node outputs at + k * out_stride (elements); params is a DEVICE array of per-plane parameters or NULL (then pp for all).
struct PredBatch {
    uint32_t n_planes = 1;
    const int32_t *coefs = nullptr;
    const int16_t *coefs16 = nullptr; // instead of coefs: a chain's compact planes (int16, None as 0; launch_fwd_transform_quant wrote them), coef_stride in halfwords; kPredForwardOutput only
    size_t coef_stride = 0, out_stride = 0;
    const PredictParams *params = nullptr;
    PredictParams pp[3] = {}; // used when params is NULL: plane k takes pp[min(k, 2)] (one image's channels travel as kernel arguments)
    const uint32_t *stream_pos = nullptr; // K2 only, with words and coefs16: [F][512] stream position of every node - `words` is then the planes' STREAMS (out_stride apart) and the scan writes them directly
    uint16_t *words = nullptr; // K2 only, with kPredForwardOutput only: write bucket << 10 | symbol per node ([n_planes] planes, out_stride apart) and neither bucket nor prediction
};
// K2. acc: n_planes x kPredAccWords words of hand-over bookkeeping, zero when allocated; serial: the number of this launch on `acc` (1, 2, ...: the caller counts; never 0).
// hist [n_planes][10][1024] and n_oob [n_planes] are device memory the kernel clears itself and then adds into with device-scope atomics.
// trust: what is known about the coefficients. kPredAnyInt32: nothing - the fast kernel checks what it stages and the exact int32 kernel behind it
// redoes a plane whose values its LDS image cannot hold. kPredPromised: the caller promises the forward kernel's output (magnitudes <= 255 - the LDS image holds up to 256 -,
// fri_hip_plan_assume_forward_coefficients): still checked, no exact kernel, a broken promise comes back as n_oob = ~0. kPredForwardOutput: this
// library's forward kernel wrote them earlier in the same call: not checked.
constexpr int kPredAnyInt32 = 0, kPredPromised = 1, kPredForwardOutput = 2;
hipError_t launch_predict_histogram(const DevicePlan &p, uint32_t *acc, uint32_t serial, const PredBatch &b, uint8_t *bucket, int32_t *prediction, uint32_t *hist,
                                    unsigned long long *n_oob, int trust, hipStream_t stream);
// Fit accumulators: mode 0 = value fit (sums_int[n_planes][3][28]), mode 1 = width fit (sums_int[n_planes][3][21], sums_dbl[n_planes][3][6]).
// acc: n_planes accumulators of kFitShards x kFitAccWords words, all zero between launches.
// out_of_range (may be NULL): per plane, the number of waves that staged a Some coefficient outside [-256, 255] - the sums are then not to be trusted.
// solve (may be NULL): the workgroup that moves a plane's totals out also solves the plane's three 6 x 6 systems (solve6.hpp) and writes the parameters -
// mode 0: .value, mode 1: .width of params[plane] - in device memory, and, when given, into mapped host memory together with the out-of-range count.
struct FitSolve {
    float *params = nullptr;                 // PredictParams[n_planes], device memory
    float *host_params = nullptr;            // PredictParams[n_planes], mapped host memory (device-visible pointer), or NULL
    unsigned long long *host_range = nullptr; // [n_planes] mapped host memory, or NULL (written by the mode-0 launch, which counts)
    unsigned long long rows[3] = {0, 0, 0};  // mode 1: heights of the reference's matrices (F * {256, 128, 128})
};
hipError_t launch_fit_accumulate(const DevicePlan &p, unsigned long long *acc, int mode, const PredBatch &b, unsigned long long *sums_int, double *sums_dbl,
                                 unsigned long long *out_of_range, hipStream_t stream, const FitSolve *solve = nullptr, int trust = 0 /* kPredAnyInt32; kPredForwardOutput: not checked, see launch_predict_histogram */);
// The fit's 6 x 6 solves on the device: sums of a launch_fit_accumulate (mode 0: sums_int[n_planes][3][28]; mode 1: sums_int[n_planes][3][21],
// sums_dbl[n_planes][3][6], rows[3] = heights of the reference's matrices) -> params[n_planes] (PredictParams: mode 0 writes .value, mode 1 .width).
// host_params / host_range (device-visible pointers into mapped host memory, or NULL): the solving threads also leave the parameters - and the
// planes' out-of-range counts `range` - there, for callers that want them on the host without a copy command.
hipError_t launch_fit_solve(int mode, uint32_t n_planes, const unsigned long long *sums_int, const double *sums_dbl, const unsigned long long rows[3], float *params,
                            hipStream_t stream, float *host_params = nullptr, const unsigned long long *range = nullptr, unsigned long long *host_range = nullptr);
// K3: (reference-faithful) dequantisation + inverse transform + clamp.
// n_images images of the plan's shape: image k at coefs + k * coef_stride (int32 elements), pixels + k * pixel_stride (bytes)
hipError_t launch_inverse_transform(const DevicePlan &p, uint32_t n_images, const int32_t *coefs, size_t coef_stride, const QMatrix &q, uint8_t *pixels, size_t pixel_stride,
                                    hipStream_t stream);

// K5, gather form: out[i] = words[order[i]] for the halfword planes K2 writes with PredBatch::words (order: n_symbols entries, cell << 9 | heap index in
// the reference's stream order with the None nodes taken out; 16-byte aligned).
hipError_t launch_symbol_gather(const uint32_t *order, uint64_t n_symbols, uint32_t n_planes, const uint16_t *words, size_t word_stride, uint16_t *out, size_t stream_stride,
                                hipStream_t stream);
// K5 from the three arrays of the scan's array form: out[k][i] = bucket << 10 | pack_signed(coef - prediction) of node order[i].
hipError_t launch_symbol_stream(const uint32_t *order, uint64_t n_symbols, uint32_t n_planes, const int32_t *coefs, size_t coef_stride, const uint8_t *bucket,
                                const int32_t *prediction, size_t out_stride, uint16_t *out, size_t stream_stride, hipStream_t stream);

// K2's per-node neighbour offsets (LDS halfword offsets relative to the own slot, two per word) from the static neighbour table
void build_lf_deltas(const uint16_t *nbr_table, int8_t *out /* [8] */);
void build_gather_tables(const uint16_t *nbr_table, uint32_t *gather_off /* [512][4] */, uint16_t *pair_pos /* [256] */, uint16_t *heap_of_pos /* [512] */);
// K2's sparse halo staging: the (halo slot, heap node) pairs a 4 x 4 block ever gathers, one per thread of its 1024-thread workgroup (0xFFFFFFFF = none)
void build_halo_list(const uint16_t *nbr_table, const uint16_t *pair_pos, uint32_t *out /* [1024] */);
size_t fwd_lds_bytes(const DevicePlan &p);
size_t inv_lds_bytes(const DevicePlan &p);
// True iff the lane/leaf footprint hard-wired in the kernels equals the table derived from LITERALS.
bool device_footprint_matches(const StaticTables &st);
// True iff the plan's tiles fit the forward kernel's static register/LDS budget.
bool fwd_plan_fits(const DevicePlan &p);
constexpr size_t kInvListPad = 2048; // >= kInvListPre x kInvThreads of k3_inverse.hip
bool inv_plan_fits(const DevicePlan &p, bool with_lists); // k3_inverse.hip: the inverse kernels' own limits (its tiling is its own since round 4)

} // namespace fri
