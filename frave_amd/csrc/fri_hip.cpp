// fri_hip.cpp -- the C ABI of libfri_hip.so (include/fri_hip.h). Host-side glue only: plan construction,
// table upload, staging for the host-pointer entry points, launches. No CPU compute fallback.
#include "fri_hip.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "geometry.hpp"
#include "kernels.hpp"
#include "solve6.hpp"

using namespace fri;

struct fri_hip_ctx {
    int device = -1;
    std::string arch;
    int cu_count = 256;
    std::string last_error;
};

namespace {

constexpr int kBatchSlots = 3;

struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *h_pixels = nullptr; // pinned
    int32_t *h_coefs = nullptr;  // pinned
    uint8_t *d_pixels = nullptr;
    int32_t *d_coefs = nullptr;
    int pending = -1; // image index whose result sits in h_coefs once the stream drains
    // the rest of an image's encode outputs (fri_hip_encode_image_batch), allocated on first use
    uint8_t *h_bucket = nullptr, *d_bucket = nullptr;
    int32_t *h_prediction = nullptr, *d_prediction = nullptr;
    uint32_t *h_hist = nullptr, *d_hist = nullptr;             // [C][10][1024]
    unsigned long long *h_oob = nullptr, *d_oob = nullptr;     // [C] out of alphabet, then [C] the fit's out-of-range counts
    float *h_params = nullptr, *d_params = nullptr;            // [C][2][3][6]
};

} // namespace

struct fri_hip_plan {
    fri_hip_ctx *ctx = nullptr;
    Geometry geo;
    DevicePlan dev;
    // The inverse kernel walks tiles of its own (round 4): what suits the forward kernel's loads and stores (bands of 16 rows) is not what suits the inverse's
    // write-out (bands of 32 for planes). geo_inv holds only the tiling (tiles, cell records, shares, write-out lists); dev_inv is dev with those swapped in.
    // Host-only plans and plans created with FRI_HIP_INV_SHARED=1 (tuning) keep one tiling: inv_geo() == geo.
    Geometry geo_inv;
    bool own_inverse_tiling = false;
    DevicePlan dev_inv;
    const Geometry &inv_geo() const { return own_inverse_tiling ? geo_inv : geo; }
    std::vector<void *> owned; // device allocations backing dev.*
    // staging for the host-pointer entry points (lazy)
    uint8_t *d_pixels = nullptr;
    int32_t *d_coefs = nullptr;
    uint8_t *d_bucket = nullptr;
    int32_t *d_prediction = nullptr;
    uint32_t *d_hist = nullptr;
    unsigned long long *d_oob = nullptr;
    unsigned long long *d_fit_int = nullptr; // [3][28]
    double *d_fit_dbl = nullptr;             // [3][6]
    Slot slots[kBatchSlots];
    bool slots_ready = false;
    // K2 / K4 hand their sums over through plan-owned accumulators that are all zero between launches (kernels.hpp). One
    // accumulator per stream that launches on this plan: launches of one stream are ordered anyway; when more streams than
    // accumulators are in play a stream taking over an accumulator first waits (on the device) for its previous user.
    struct AccSlot {
        hipStream_t stream = nullptr;
        bool used = false;
        hipEvent_t handed_over = nullptr;
        uint32_t *pred_acc = nullptr;           // [planes][kPredAccWords]: K2's hand-over bookkeeping (kernels.hpp), grown on demand, zero when allocated
        uint32_t pred_serial = 0;               // launches of K2 on this accumulator so far: every launch publishes and polls for its own number
        unsigned long long *fit_acc = nullptr;  // [planes][kFitShards][kFitAccWords]
        // scratch of the device-side fit (fit_chain): the sums of a launch's planes on their way to the solve kernels, their out-of-range
        // counts, and parameter sets for callers that keep theirs on the host. Per stream like the accumulators: chains of several streams
        // (fri_hip_multi_encode_image's slots) run side by side on one plan.
        unsigned long long *sums_int = nullptr; // [planes][3][28]
        double *sums_dbl = nullptr;             // [planes][3][6]
        unsigned long long *range = nullptr;    // [planes]
        float *params = nullptr;                // [planes][2][3][6]
        uint32_t planes = 0;
    } acc_slots[kPredAccRing];
    std::vector<void *> retired_acc; // accumulators outgrown by a larger batch: freed with the plan (a launch may still be draining them)
    // fri_hip_encode_image (host form): all channels' outputs on the device
    uint8_t *d_bucket_all = nullptr;
    int32_t *d_prediction_all = nullptr;
    uint32_t *d_hist_all = nullptr;           // [C][10][1024]
    unsigned long long *d_oob_all = nullptr;  // [C]
    // host-facing fit (fri_hip_encode_image_dev / fri_hip_predict_image_dev with fit != 0): the fitted parameters and the range counts come
    // back through pinned memory behind an event, while the scan kernel that follows them is already queued
    void *h_fit = nullptr;     // pinned + mapped: [3] PredictParams + [3] u64
    void *d_h_fit = nullptr;   // the device's address of h_fit
    hipEvent_t ev_fit = nullptr;
    bool assume_forward = false; // fri_hip_plan_assume_forward_coefficients
    uint32_t *d_stream_order = nullptr; // fri_hip_plan_set_stream_order: node index of the i-th symbol of a channel, [geo.n_some]
    uint32_t *d_stream_pos = nullptr;   // ... and its inverse, [F][512]: the position of a node's symbol in a channel's stream (None nodes: ~0, never used) - the scan's STREAM form
    // The symbol-stream chains' compact coefficient planes (round 5): int16, None as 0, [planes][F][512] - between the forward kernel, the fit and the scan when the
    // caller does not ask for the coefficients (fri_hip_encode_image_symbols; fri_hip_encode_symbols_batch_dev with d_coefs == NULL). Half the bytes written and read three times.
    int16_t *d_coefs16 = nullptr;
    size_t coefs16_planes = 0;
    hipEvent_t ev_coefs16 = nullptr;    // behind the last chain that used them: a chain on ANOTHER stream waits for it before it overwrites the planes
    hipStream_t coefs16_stream = nullptr;
    bool coefs16_used = false;
    uint16_t *d_symbols = nullptr;      // fri_hip_encode_image_symbols: [C][geo.n_some]
    uint32_t acc_next = 0;
    bool acc_dirty = false; // a launch on this plan failed: the accumulators are re-zeroed before the next use
    hipEvent_t ev_begin = nullptr, ev_end = nullptr; // timing helper's events, created with the plan (creating an event is not work to be timed)
    // fri_hip_plan_tune_forward: the parameters the forward tiling was built with, and whether the plan may exchange it (a plan whose tiling the
    // environment pinned, whose inverse kernel shares the forward tiles, or that is cut into many short shares keeps what it has)
    TilingParams fwd_tp;
    bool fwd_tunable = false;
    std::string fwd_tiling_note = "default";
};

struct fri_hip_multi {
    std::vector<fri_hip_ctx *> ctxs;
    std::vector<fri_hip_plan *> plans;
};

namespace {

int fail_hip(fri_hip_ctx *ctx, hipError_t e, const char *what) {
    if (ctx) {
        ctx->last_error = std::string(what) + ": " + hipGetErrorString(e);
    }
    return e == hipErrorOutOfMemory ? FRI_HIP_ERR_OUT_OF_MEMORY : FRI_HIP_ERR_HIP;
}

#define HIP_TRY(ctx, expr)                                 \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #expr); \
    } while (0)

template <typename T>
int upload(fri_hip_plan *p, const std::vector<T> &v, const T *&out, size_t pad = 0) { // pad: zeroed elements behind the array
    void *d = nullptr;
    size_t bytes = (v.size() + pad) * sizeof(T);
    if (!bytes) bytes = sizeof(T);
    HIP_TRY(p->ctx, hipMalloc(&d, bytes));
    p->owned.push_back(d);
    if (pad) HIP_TRY(p->ctx, hipMemset(static_cast<T *>(d) + v.size(), 0, pad * sizeof(T)));
    if (!v.empty()) HIP_TRY(p->ctx, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    out = static_cast<const T *>(d);
    return FRI_HIP_OK;
}

int check_q(const int32_t q[32], QMatrix &out) {
    if (!q) return FRI_HIP_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 32; i++) out.q[i] = q[i];
    for (int i = 0; i <= 9; i++) // layers a depth-9 cell can reach (quantization.rs:13)
        if (q[i] == 0) return FRI_HIP_ERR_DIVIDE_BY_ZERO;
    return FRI_HIP_OK;
}

int need_device(const fri_hip_plan *p) {
    if (!p) return FRI_HIP_ERR_INVALID_ARGUMENT;
    if (!p->ctx) return FRI_HIP_ERR_NO_DEVICE;
    return FRI_HIP_OK;
}

// Tuning knobs (tile shapes, share weights, kernel A/B switches, trace) are read from the environment only when the caller opts
// in with FRI_HIP_TUNING=1; otherwise the library's behaviour does not depend on the process environment.
bool tuning_enabled() {
    const char *s = std::getenv("FRI_HIP_TUNING");
    return s && s[0] == '1' && s[1] == 0;
}
const char *env_str(const char *name) { return tuning_enabled() ? std::getenv(name) : nullptr; }
int env_int(const char *name) {
    const char *s = env_str(name);
    return s ? std::atoi(s) : 0;
}

int ensure_staging(fri_hip_plan *p) {
    fri_hip_ctx *c = p->ctx;
    const size_t F = p->geo.centers.size();
    if (!p->d_pixels) HIP_TRY(c, hipMalloc((void **)&p->d_pixels, fri_hip_plan_pixel_bytes(p)));
    if (!p->d_coefs) HIP_TRY(c, hipMalloc((void **)&p->d_coefs, fri_hip_plan_coef_count(p) * sizeof(int32_t)));
    if (!p->d_bucket) HIP_TRY(c, hipMalloc((void **)&p->d_bucket, F * kCell));
    if (!p->d_prediction) HIP_TRY(c, hipMalloc((void **)&p->d_prediction, F * kCell * sizeof(int32_t)));
    if (!p->d_hist) HIP_TRY(c, hipMalloc((void **)&p->d_hist, 10 * 1024 * sizeof(uint32_t)));
    if (!p->d_oob) HIP_TRY(c, hipMalloc((void **)&p->d_oob, sizeof(unsigned long long)));
    if (!p->d_fit_int) HIP_TRY(c, hipMalloc((void **)&p->d_fit_int, 3 * 28 * sizeof(unsigned long long)));
    if (!p->d_fit_dbl) HIP_TRY(c, hipMalloc((void **)&p->d_fit_dbl, 18 * sizeof(double)));
    return FRI_HIP_OK;
}

// Accumulators for a K2 / K4 launch over n_planes planes on `stream` (see fri_hip_plan::acc_slots). Returns the slot index, or a negative error code.
int acquire_acc(fri_hip_plan *p, hipStream_t stream, uint32_t n_planes = 1) {
    fri_hip_ctx *c = p->ctx;
    // The accumulators are allocated (and, on the rare paths, cleared / waited for) right here, on the launch path of the `_dev` entry
    // points, which take whatever device the calling thread has current: bind the plan's device first, or a plan of fri_hip_multi's
    // device d > 0 would get its accumulators on another GPU.
    HIP_TRY(c, hipSetDevice(c->device));
    if (p->acc_dirty) { // rare: a previous launch failed part-way; nothing may be in flight on the accumulators when they are cleared
        HIP_TRY(c, hipDeviceSynchronize());
        for (auto &a : p->acc_slots) {
            if (a.pred_acc) HIP_TRY(c, hipMemset(a.pred_acc, 0, (size_t)a.planes * kPredAccWords * sizeof(uint32_t)));
            if (a.fit_acc) HIP_TRY(c, hipMemset(a.fit_acc, 0, (size_t)a.planes * kFitShards * kFitAccWords * sizeof(unsigned long long)));
        }
        HIP_TRY(c, hipDeviceSynchronize()); // (the memsets ran on the null stream: done before any stream launches on the accumulators again)
        p->acc_dirty = false;
    }
    int idx = -1;
    for (uint32_t i = 0; i < kPredAccRing && idx < 0; i++)
        if (p->acc_slots[i].used && p->acc_slots[i].stream == stream) idx = (int)i;
    for (uint32_t i = 0; i < kPredAccRing && idx < 0; i++)
        if (!p->acc_slots[i].used) {
            p->acc_slots[i].used = true;
            p->acc_slots[i].stream = stream;
            idx = (int)i;
        }
    if (idx < 0) {
        auto &v = p->acc_slots[p->acc_next];
        idx = (int)p->acc_next;
        p->acc_next = (p->acc_next + 1) % kPredAccRing;
        if (!v.handed_over) HIP_TRY(c, hipEventCreateWithFlags(&v.handed_over, hipEventDisableTiming));
        // everything the previous owner has queued so far finishes before the new owner's kernel starts. The previous owner is known by
        // its handle only: the caller may have destroyed that stream since (the record then fails, or lands on a recycled handle). Either
        // way the slot must come out usable: without a valid event, wait for the whole device once - whatever was queued on a stream
        // that no longer exists has drained by then.
        hipError_t e = hipEventRecord(v.handed_over, v.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(stream, v.handed_over, 0);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            HIP_TRY(c, hipDeviceSynchronize());
        }
        v.stream = stream;
    }
    auto &a = p->acc_slots[idx];
    if (a.planes < n_planes) { // grow: the old buffers may still be in use by queued launches, so they are retired, not freed
        // (geometric: a caller whose batches grow by one plane per call retires at most 16 generations per slot - ADVICE r3)
        const uint32_t planes = std::min<uint32_t>(65535u, std::max<uint32_t>({4u, n_planes, 2u * a.planes}));
        void *pa = nullptr, *fa = nullptr, *si = nullptr, *sd = nullptr, *rg = nullptr, *pr = nullptr;
        const size_t pb = (size_t)planes * kPredAccWords * sizeof(uint32_t), fb = (size_t)planes * kFitShards * kFitAccWords * sizeof(unsigned long long);
        hipError_t e = hipMalloc(&pa, pb);
        if (e == hipSuccess) e = hipMalloc(&fa, fb);
        if (e == hipSuccess) e = hipMalloc(&si, (size_t)planes * 3 * 28 * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc(&sd, (size_t)planes * 18 * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&rg, (size_t)planes * sizeof(unsigned long long));
        if (e == hipSuccess) e = hipMalloc(&pr, (size_t)planes * sizeof(PredictParams));
        // zeroed ON THE LAUNCHING STREAM: a hipMemset on the null stream is not ordered against a non-blocking stream, and a memset that lands while
        // the first kernel is adding into the accumulator wipes its partial sums (seen with twelve torch streams: an all-zero Gram matrix)
        if (e == hipSuccess) e = hipMemsetAsync(pa, 0, pb, stream);
        if (e == hipSuccess) e = hipMemsetAsync(fa, 0, fb, stream);
        if (e != hipSuccess) {
            for (void *d : {pa, fa, si, sd, rg, pr})
                if (d) (void)hipFree(d);
            return fail_hip(c, e, "accumulator allocation");
        }
        for (void *d : {(void *)a.pred_acc, (void *)a.fit_acc, (void *)a.sums_int, (void *)a.sums_dbl, (void *)a.range, (void *)a.params})
            if (d) p->retired_acc.push_back(d);
        a.pred_acc = static_cast<uint32_t *>(pa);
        a.fit_acc = static_cast<unsigned long long *>(fa);
        a.sums_int = static_cast<unsigned long long *>(si);
        a.sums_dbl = static_cast<double *>(sd);
        a.range = static_cast<unsigned long long *>(rg);
        a.params = static_cast<float *>(pr);
        a.planes = planes;
    }
    return idx;
}

void free_slots(fri_hip_plan *p) {
    for (Slot &s : p->slots) {
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.h_pixels) (void)hipHostFree(s.h_pixels);
        if (s.h_coefs) (void)hipHostFree(s.h_coefs);
        for (void *d : {(void *)s.d_pixels, (void *)s.d_coefs, (void *)s.d_bucket, (void *)s.d_prediction, (void *)s.d_hist, (void *)s.d_oob, (void *)s.d_params})
            if (d) (void)hipFree(d);
        for (void *h : {(void *)s.h_bucket, (void *)s.h_prediction, (void *)s.h_hist, (void *)s.h_oob, (void *)s.h_params})
            if (h) (void)hipHostFree(h);
        s = Slot{};
    }
    p->slots_ready = false;
}

int ensure_slots(fri_hip_plan *p) {
    if (p->slots_ready) return FRI_HIP_OK;
    fri_hip_ctx *c = p->ctx;
    const size_t pb = fri_hip_plan_pixel_bytes(p), cb = fri_hip_plan_coef_count(p) * sizeof(int32_t);
    auto alloc_all = [&]() -> int {
        for (Slot &s : p->slots) {
            HIP_TRY(c, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
            HIP_TRY(c, hipHostMalloc((void **)&s.h_pixels, pb, hipHostMallocDefault));
            HIP_TRY(c, hipHostMalloc((void **)&s.h_coefs, cb, hipHostMallocDefault));
            HIP_TRY(c, hipMalloc((void **)&s.d_pixels, pb));
            HIP_TRY(c, hipMalloc((void **)&s.d_coefs, cb));
        }
        return FRI_HIP_OK;
    };
    if (int rc = alloc_all()) { // a partial set is of no use: give everything back, the next call starts over
        free_slots(p);
        return rc;
    }
    p->slots_ready = true;
    return FRI_HIP_OK;
}

// the encode outputs of the batch slots (bucket / prediction only when somebody wants them: 5 bytes per coefficient of pinned memory)
int ensure_encode_slots(fri_hip_plan *p, bool want_bucket, bool want_prediction) {
    if (int rc = ensure_slots(p)) return rc;
    fri_hip_ctx *c = p->ctx;
    const size_t C = p->geo.channels, n = fri_hip_plan_coef_count(p);
    for (Slot &s : p->slots) {
        if (want_bucket && !s.d_bucket) {
            HIP_TRY(c, hipMalloc((void **)&s.d_bucket, n));
            HIP_TRY(c, hipHostMalloc((void **)&s.h_bucket, n, hipHostMallocDefault));
        }
        if (want_prediction && !s.d_prediction) {
            HIP_TRY(c, hipMalloc((void **)&s.d_prediction, n * sizeof(int32_t)));
            HIP_TRY(c, hipHostMalloc((void **)&s.h_prediction, n * sizeof(int32_t), hipHostMallocDefault));
        }
        if (!s.d_hist) {
            HIP_TRY(c, hipMalloc((void **)&s.d_hist, C * 10 * 1024 * sizeof(uint32_t)));
            HIP_TRY(c, hipHostMalloc((void **)&s.h_hist, C * 10 * 1024 * sizeof(uint32_t), hipHostMallocDefault));
            HIP_TRY(c, hipMalloc((void **)&s.d_oob, 2 * C * sizeof(unsigned long long)));
            HIP_TRY(c, hipHostMalloc((void **)&s.h_oob, 2 * C * sizeof(unsigned long long), hipHostMallocDefault));
            HIP_TRY(c, hipMalloc((void **)&s.d_params, C * sizeof(PredictParams)));
            HIP_TRY(c, hipHostMalloc((void **)&s.h_params, C * sizeof(PredictParams), hipHostMallocDefault));
        }
    }
    return FRI_HIP_OK;
}


// ---- forward tiling chosen by measurement (fri_hip_plan_tune_forward) ---------------------------------------------------------------------------------
// Which tiling the forward kernel runs fastest on is not monotonic in any parameter and differs by image size (at 4096^2 contiguous
// shares with bands of 72 rows beat the default by 4 %, at 6000 x 4000 they lose 13 %; nine cells per tile win at 4096^2 and lose at 2048^2), because with
// four or five tiles per share the tile count quantises the one round of shares differently at every size. So the plan can MEASURE, like an FFT plan: a
// handful of candidate tilings are built, uploaded and timed on scratch buffers large enough that every byte comes from HBM, and the winner replaces the
// default tiling. Results never depend on the tiling (the parity tests run on tuned and untuned plans alike); only the forward kernel's speed does.
struct FwdTiling {  // one candidate: host geometry + its device tables
    TilingParams tp;
    Geometry geo;
    DevicePlan dev;            // p->dev with this candidate's tiling swapped in
    std::vector<void *> bufs;  // device tables of this candidate
    std::string label;
    double us = 0;
};

void fill_forward_tiling(DevicePlan &d, const Geometry &g) {
    d.n_tiles = (uint32_t)g.tiles.size();
    d.lds_pitch = g.lds_pitch;
    d.lds_rows = g.lds_rows;
    d.cells_per_tile = g.cells_per_tile;
    d.max_tile_cells = g.max_tile_cells;
    d.max_wg_tiles = std::max(g.max_wg_tiles, g.max_wg_tiles_batch);
    d.n_wg_batch = (uint32_t)g.wg_tiles_batch.size() - 1;
    d.max_wg_cells = g.max_wg_cells;
    d.n_wg = (uint32_t)g.wg_tiles.size() - 1;
}

template <typename T>
hipError_t upload_raw(const std::vector<T> &v, const T *&out, std::vector<void *> &bufs) {
    void *d = nullptr;
    const size_t bytes = std::max(v.size(), (size_t)1) * sizeof(T);
    if (hipError_t e = hipMalloc(&d, bytes)) return e;
    bufs.push_back(d);
    if (!v.empty())
        if (hipError_t e = hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)) return e;
    out = static_cast<const T *>(d);
    return hipSuccess;
}

// builds candidate `c.tp` for the plan's shape; false = it does not fit the forward kernel with the plan's resident workgroup count (not a candidate)
bool build_forward_candidate(const fri_hip_plan *p, FwdTiling &c) {
    const Geometry &g0 = p->geo;
    if (!build_geometry(g0.width, g0.height, g0.channels, c.tp, c.geo).empty()) return false;
    if (c.geo.centers.size() != g0.centers.size()) return false;
    DevicePlan probe;
    probe.channels = (int32_t)g0.channels;
    probe.lds_pitch = c.geo.lds_pitch;
    probe.lds_rows = c.geo.lds_rows;
    probe.max_tile_cells = c.geo.max_tile_cells;
    probe.max_wg_tiles = std::max(c.geo.max_wg_tiles, c.geo.max_wg_tiles_batch);
    if (!fwd_plan_fits(probe)) return false;
    if ((int)std::min<size_t>(4, (160 * 1024) / fwd_lds_bytes(probe)) < c.tp.ranks) return false; // the shares are sized for tp.ranks resident workgroups per CU
    return true;
}

hipError_t upload_forward_candidate(const fri_hip_plan *p, FwdTiling &c) {
    c.dev = p->dev;
    c.dev.k1_measuring = true; // (adopt_forward_tiling copies the tables into the plan's own DevicePlan, not this flag)
    hipError_t e;
    if ((e = upload_raw(c.geo.tiles, c.dev.tiles, c.bufs)) || (e = upload_raw(c.geo.tile_cells, c.dev.tile_cells, c.bufs)) || (e = upload_raw(c.geo.tile_meta, c.dev.tile_meta, c.bufs)) ||
        (e = upload_raw(c.geo.wg_tiles, c.dev.wg_tiles, c.bufs)) || (e = upload_raw(c.geo.wg_tiles_batch, c.dev.wg_tiles_batch, c.bufs)))
        return e;
    fill_forward_tiling(c.dev, c.geo);
    return hipSuccess;
}

// the plan takes the candidate's tiling: host vectors and device tables (the old tables stay allocated until the plan is destroyed - a launch may still read them)
void adopt_forward_tiling(fri_hip_plan *p, FwdTiling &c) {
    Geometry &g = p->geo;
    g.tiles.swap(c.geo.tiles), g.tile_cells.swap(c.geo.tile_cells), g.tile_meta.swap(c.geo.tile_meta), g.wg_tiles.swap(c.geo.wg_tiles), g.wg_tiles_batch.swap(c.geo.wg_tiles_batch);
    g.max_wg_tiles_batch = c.geo.max_wg_tiles_batch, g.lds_pitch = c.geo.lds_pitch, g.lds_rows = c.geo.lds_rows, g.band_rows = c.geo.band_rows, g.cells_per_tile = c.geo.cells_per_tile;
    g.cells_per_wg = c.geo.cells_per_wg, g.max_tile_cells = c.geo.max_tile_cells, g.max_wg_tiles = c.geo.max_wg_tiles, g.max_wg_cells = c.geo.max_wg_cells;
    DevicePlan &d = p->dev;
    d.tiles = c.dev.tiles, d.tile_cells = c.dev.tile_cells, d.tile_meta = c.dev.tile_meta, d.wg_tiles = c.dev.wg_tiles, d.wg_tiles_batch = c.dev.wg_tiles_batch;
    fill_forward_tiling(d, g);
    d.inv_max_wg_tiles = d.max_wg_tiles, d.inv_max_wg_cells = d.max_wg_cells; // (only read by an inverse kernel on the forward tiling; tunable plans have their own)
    for (void *b : c.bufs) p->owned.push_back(b);
    c.bufs.clear();
    p->fwd_tp = c.tp;
    p->fwd_tiling_note = c.label;
}

std::string tiling_label(const TilingParams &tp, const Geometry &g) {
    char b[96];
    std::snprintf(b, sizeof b, "%s/band%d/cells%d", tp.strided_shares ? "interleaved" : "contiguous", g.band_rows, g.cells_per_tile);
    std::string l = b;
    if (tp.rank_weight[0] > 0) {
        std::snprintf(b, sizeof b, "/w%.2f-%.2f", tp.rank_weight[0], tp.rank_weight[std::max(0, std::min(tp.ranks, 4) - 1)]);
        l += b;
    }
    return l;
}

// winners of this process, per (device, width, height, channels): later plans of the same shape start from them without measuring again
struct TunedKey {
    int device;
    uint32_t w, h, c;
    bool operator==(const TunedKey &o) const { return device == o.device && w == o.w && h == o.h && c == o.c; }
};
std::mutex g_tuned_mu;
std::vector<std::pair<TunedKey, TilingParams>> g_tuned;

void adopt_cached_forward_tiling(fri_hip_plan *p) {
    TilingParams tp;
    {
        std::lock_guard<std::mutex> lk(g_tuned_mu);
        const TunedKey k{p->ctx->device, p->geo.width, p->geo.height, p->geo.channels};
        auto it = std::find_if(g_tuned.begin(), g_tuned.end(), [&](const auto &e) { return e.first == k; });
        if (it == g_tuned.end()) return;
        tp = it->second;
    }
    FwdTiling c;
    c.tp = tp;
    if (!build_forward_candidate(p, c)) return;
    if (upload_forward_candidate(p, c) != hipSuccess) {
        for (void *b : c.bufs) (void)hipFree(b);
        (void)hipGetLastError();
        return;
    }
    c.label = tiling_label(c.tp, c.geo) + " (measured earlier in this process)";
    adopt_forward_tiling(p, c);
}

} // namespace

extern "C" {

const char *fri_hip_strerror(int code) {
    switch (code) {
    case FRI_HIP_OK: return "ok";
    case FRI_HIP_ERR_INVALID_ARGUMENT: return "invalid argument";
    case FRI_HIP_ERR_HIP: return "HIP runtime error (see fri_hip_last_hip_error)";
    case FRI_HIP_ERR_NO_DEVICE: return "no gfx950 device bound (there is no CPU fallback)";
    case FRI_HIP_ERR_OUT_OF_MEMORY: return "out of device or pinned memory";
    case FRI_HIP_ERR_DIVIDE_BY_ZERO: return "quantisation matrix has a zero divisor";
    case FRI_HIP_ERR_EMPTY_LATTICE: return "no cell of the lattice touches the image";
    case FRI_HIP_ERR_OUT_OF_RANGE: return "a Some coefficient lies outside [-256, 255]: the fit sums would overflow (the forward transform never produces one)";
    default: return "unknown error";
    }
}

const char *fri_hip_version(void) { return "frave_amd/libfri_hip 0.1 (gfx950)"; }

int fri_hip_ctx_create(int device, fri_hip_ctx **out) {
    if (!out) return FRI_HIP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FRI_HIP_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return FRI_HIP_ERR_INVALID_ARGUMENT;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FRI_HIP_ERR_HIP;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FRI_HIP_ERR_NO_DEVICE; // kernels are built for gfx950 only
    if (hipSetDevice(device) != hipSuccess) return FRI_HIP_ERR_HIP;
    fri_hip_ctx *c = new (std::nothrow) fri_hip_ctx;
    if (!c) return FRI_HIP_ERR_OUT_OF_MEMORY;
    c->device = device;
    c->arch = prop.gcnArchName;
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = c;
    return FRI_HIP_OK;
}

int fri_hip_ctx_destroy(fri_hip_ctx *ctx) {
    delete ctx;
    return FRI_HIP_OK;
}

const char *fri_hip_backend(const fri_hip_ctx *ctx) { return ctx ? "hip:gfx950" : "none"; }
const char *fri_hip_last_hip_error(const fri_hip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int fri_hip_plan_create(fri_hip_ctx *ctx, uint32_t width, uint32_t height, uint32_t channels, fri_hip_plan **out) {
    if (!out) return FRI_HIP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const StaticTables &st = static_tables();
    if (!st.error.empty() || !device_footprint_matches(st)) return FRI_HIP_ERR_INVALID_ARGUMENT;
    fri_hip_plan *p = new (std::nothrow) fri_hip_plan;
    if (!p) return FRI_HIP_ERR_OUT_OF_MEMORY;
    p->ctx = ctx;
    TilingParams tp;
    tp.band_rows = env_int("FRI_HIP_BAND_ROWS");
    tp.cells_per_tile = env_int("FRI_HIP_CELLS_PER_TILE");
    tp.cells_per_wg = env_int("FRI_HIP_CELLS_PER_WG");
    tp.batch_share_tiles = env_int("FRI_HIP_BATCH_SHARE_TILES");
    // Interleaved shares (geometry.cpp): the resident workgroups work on one window sliding over the image. Default since round 4 (4096^2 from HBM: planes
    // 20.3 -> 19.1 us, RGB 55.5 -> 51.9 us, the inverse 29.6 -> 27.8 / 86 -> 72 us); FRI_HIP_STRIDED_SHARES=0 (tuning) restores one contiguous run per share.
    // Up to ~200 000 cells, that is: at 12000^2 and 16384^2 (282 K / 526 K cells, many short shares dispatched in order: the resident set slides already) the
    // interleaving costs 6 % (150 -> 160 us, 270 -> 285 us), at 8192^2 (131 K) it still gains 7 % (80 -> 74 us).
    const bool strided_env = env_str("FRI_HIP_STRIDED_SHARES") != nullptr;
    tp.strided_shares = strided_env ? env_int("FRI_HIP_STRIDED_SHARES") > 0 : ctx != nullptr;
    constexpr size_t kStridedMaxCells = 200000, kOwnInverseMaxCells = 400000;
    if (!strided_env && (size_t)width * height / kCell >= kStridedMaxCells) tp.strided_shares = false; // (a cell per 512 pixels: spares the large plan a geometry pass)
    // Band height of the forward tiles: 16 rows for planes and RGB (planes were 32 through round 3 - tuned while the bench's pixels came out of the Infinity
    // Cache; from HBM 16 with interleaved shares is 3-5 % faster at 4096^2 and 15 % at 6000 x 4000). The inverse kernel keeps 32 for planes (below).
    if (tp.band_rows <= 0 && ctx) tp.band_rows = 16;
    // LDS budget of one forward tile buffer: 4 chunks of 16 bytes per thread for planes (the tuned variant), 6 for RGB
    tp.tile_buffer_bytes = (channels == 1 ? 4 : 6) * 256 * 16;
    if (env_int("FRI_HIP_TILE_BYTES") > 0) tp.tile_buffer_bytes = env_int("FRI_HIP_TILE_BYTES");
    // Shares by dispatch rank, measured on MI355X at 4096^2 (tools/sweep_rank_weights.sh): equal shares 22.3-23.4 us,
    // these weights 20.5-21.7 us; steeper is worse again. Host-only plans keep equal shares.
    // Resident K1 workgroups per CU: 4 by registers; fewer when the LDS of one workgroup (two tile buffers) does not fit four times:
    // RGB tiles need ~47 KB, so three. One share per resident workgroup, sized by dispatch rank.
    auto set_ranks = [&](int ranks) {
        tp.ranks = ranks;
        // (three ranks = RGB: 1.3 / 1.1 / 0.6 through round 3; with interleaved shares and all bytes from HBM 1.2 / 1.0 / 0.8 is 2 % faster: 51.9 -> 50.6 us)
        static const float w4[4] = {1.3f, 1.1f, 0.9f, 0.7f}, w3[4] = {1.2f, 1.0f, 0.8f, 0.f}, w2[4] = {1.15f, 0.85f, 0.f, 0.f}, w1[4] = {1.f, 0.f, 0.f, 0.f};
        const float *w = ranks >= 4 ? w4 : ranks == 3 ? w3 : ranks == 2 ? w2 : w1;
        for (int i = 0; i < 4; i++) tp.rank_weight[i] = ctx ? w[i] : 0.f;
        if (const char *e = env_str("FRI_HIP_RANK_WEIGHTS")) { // "w0,w1,w2,w3" (tuning; "1,1,1,1" = equal shares)
            float v[4];
            if (std::sscanf(e, "%f,%f,%f,%f", &v[0], &v[1], &v[2], &v[3]) == 4)
                for (int i = 0; i < 4; i++) tp.rank_weight[i] = v[i];
        }
        tp.target_wgs = env_int("FRI_HIP_TARGET_WGS");
        if (tp.target_wgs <= 0 && ctx) tp.target_wgs = ctx->cu_count * ranks;
    };
    set_ranks(env_int("FRI_HIP_RANKS") > 0 ? std::min(env_int("FRI_HIP_RANKS"), 4) : 4);
    bool many_shares = false;
    auto ctx_wgs = [](const fri_hip_ctx *c, int ranks) { return c ? c->cu_count * ranks : 1024; };
    // Shrink the tiles until they fit the forward kernel's static register / LDS budget (irregular centre spacing
    // makes a few tiles wider than the average; RGB triples the bytes per pixel).
    for (;;) {
        std::string err = build_geometry(width, height, channels, tp, p->geo);
        if (!err.empty()) {
            int rc = err == "empty lattice" ? FRI_HIP_ERR_EMPTY_LATTICE : FRI_HIP_ERR_INVALID_ARGUMENT;
            delete p;
            return rc;
        }
        DevicePlan probe;
        probe.channels = (int32_t)channels;
        probe.lds_pitch = p->geo.lds_pitch;
        probe.lds_rows = p->geo.lds_rows;
        probe.max_tile_cells = p->geo.max_tile_cells;
        probe.max_wg_tiles = std::max(p->geo.max_wg_tiles, p->geo.max_wg_tiles_batch);
        if (fwd_plan_fits(probe)) {
            if (ctx && !strided_env && tp.strided_shares && p->geo.centers.size() >= kStridedMaxCells) { // large image: contiguous shares (see above)
                tp.strided_shares = false;
                continue;
            }
            const int resident = (int)std::min<size_t>(4, (160 * 1024) / fwd_lds_bytes(probe));
            if (ctx && env_int("FRI_HIP_RANKS") <= 0 && resident >= 1 && resident < tp.ranks) { // e.g. RGB: three workgroups per CU
                set_ranks(resident);
                continue;
            }
            // Large images: with one share per resident workgroup every workgroup streams through its own 1/1024 of the image for the whole
            // launch - 1024 read and 1024 write fronts spread over all of it (1.4 GB at 16384^2). Many short shares (~32 cells), handed
            // out in dispatch order, let the resident set slide over the image instead: 8192^2 60.2 -> 56.5 us, 12000^2 120.7 -> 107.1 us,
            // 16384^2 297-311 -> 275-279 us (tools/sweep_16k_shares.sh; equal shares: the dispatch-rank weights are for a launch of one round).
            // Below ~128 cells per workgroup (8192^2: neutral) the one-round launch with weighted shares is kept.
            const size_t F = p->geo.centers.size(), one_round = (size_t)ctx_wgs(ctx, tp.ranks);
            if (ctx && !many_shares && env_int("FRI_HIP_TARGET_WGS") <= 0 && tp.cells_per_wg <= 0 && F >= 128 * one_round) {
                many_shares = true;
                tp.target_wgs = (int)(((F / 32 + 1023) / 1024) * 1024);
                for (int i = 0; i < 4; i++) tp.rank_weight[i] = i < tp.ranks ? 1.f : 0.f;
                continue;
            }
            break;
        }
        tp.band_rows = p->geo.band_rows;
        tp.cells_per_tile = p->geo.cells_per_tile - 1;
        if (tp.cells_per_tile < 1) {
            tp.cells_per_tile = 1;
            tp.band_rows = p->geo.band_rows / 2;
            if (tp.band_rows < 1) {
                delete p;
                return FRI_HIP_ERR_INVALID_ARGUMENT;
            }
        }
    }
    const TilingParams tp_forward = tp; // what the forward tiling was built with (fri_hip_plan_tune_forward starts from it)
    // (from ~400 000 cells on - 16384^2 - the inverse kernel is fastest on groups of the forward plan's short shares, as in round 3: 386 against 408 us)
    if (ctx && (env_str("FRI_HIP_INV_SHARED") ? env_int("FRI_HIP_INV_SHARED") <= 0 : p->geo.centers.size() < kOwnInverseMaxCells)) {
        // the inverse kernel's own tiling: one share per resident workgroup (it prefers that at every size), dispatch-rank weights, interleaved, bands of
        // 32 rows for planes / 16 for RGB (FRI_HIP_INV_BAND_ROWS); shrunk like the forward tiles until the (shared) LDS rectangle budget holds
        set_ranks(tp.ranks); // (the forward plan is built: tp is free; this restores the dispatch-rank weights a many-shares forward plan had flattened)
        TilingParams ti = tp;
        ti.band_rows = env_int("FRI_HIP_INV_BAND_ROWS") > 0 ? env_int("FRI_HIP_INV_BAND_ROWS") : (channels == 1 ? 32 : 16);
        ti.cells_per_tile = env_int("FRI_HIP_CELLS_PER_TILE");
        ti.cells_per_wg = 0;
        ti.target_wgs = ctx_wgs(ctx, tp.ranks);
        ti.strided_shares = env_str("FRI_HIP_INV_STRIDED_SHARES") ? env_int("FRI_HIP_INV_STRIDED_SHARES") > 0 : (strided_env ? tp.strided_shares : true); // interleaved at every size it is built for
        bool ok = false;
        for (int guard = 0; guard < 64; guard++) {
            if (!build_geometry(width, height, channels, ti, p->geo_inv).empty()) break;
            DevicePlan probe;
            probe.channels = (int32_t)channels;
            probe.lds_pitch = p->geo_inv.lds_pitch;
            probe.lds_rows = p->geo_inv.lds_rows;
            probe.max_tile_cells = p->geo_inv.max_tile_cells;
            probe.max_wg_tiles = std::max(p->geo_inv.max_wg_tiles, p->geo_inv.max_wg_tiles_batch);
            probe.inv_max_wg_tiles = p->geo_inv.max_wg_tiles, probe.inv_max_wg_cells = p->geo_inv.max_wg_cells;
            // the inverse kernel's OWN limits (the forward kernel's budget said nothing about them); the lists' rectangle is checked once they are built
            if (fwd_plan_fits(probe) && inv_plan_fits(probe, false)) { // (the forward budget too, as through round 4: it keeps the tiles at the size the kernel was tuned for)
                ok = true;
                break;
            }
            ti.band_rows = p->geo_inv.band_rows;
            ti.cells_per_tile = p->geo_inv.cells_per_tile - 1;
            if (ti.cells_per_tile < 1) {
                ti.cells_per_tile = 1;
                ti.band_rows = p->geo_inv.band_rows / 2;
                if (ti.band_rows < 1) break;
            }
        }
        p->own_inverse_tiling = ok;
        if (!ok) p->geo_inv = Geometry{};
    }
    build_inverse_lists(p->own_inverse_tiling ? p->geo_inv : p->geo, (size_t)256 << 20); // 6 MB at 4096^2; beyond 256 MB the inverse kernel scans the rectangle instead
    if (p->own_inverse_tiling) { // with the lists built their LDS rectangle is known: a tiling whose lists kernel would not fit falls back to the shared (forward) tiling
        DevicePlan probe;
        probe.channels = (int32_t)channels;
        probe.lds_pitch = p->geo_inv.lds_pitch, probe.lds_rows = p->geo_inv.lds_rows, probe.max_tile_cells = p->geo_inv.max_tile_cells;
        probe.inv_max_wg_tiles = p->geo_inv.max_wg_tiles, probe.inv_max_wg_cells = p->geo_inv.max_wg_cells, probe.inv_rect_bytes = p->geo_inv.inv_rect_bytes;
        if (!p->geo_inv.inv_lists.empty() && !inv_plan_fits(probe, true)) {
            p->own_inverse_tiling = false;
            p->geo_inv = Geometry{};
            build_inverse_lists(p->geo, (size_t)256 << 20);
        }
    }
    if (ctx) {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            delete p;
            return FRI_HIP_ERR_HIP;
        }
        const Geometry &g = p->geo;
        DevicePlan &d = p->dev;
        int rc = FRI_HIP_OK;
        std::vector<uint16_t> tab(&st.nbr_table[0][0], &st.nbr_table[0][0] + kCell * 6);
        std::vector<uint32_t> gather_off((size_t)kCell * 4);
        std::vector<uint16_t> pair_pos(256), heap_of_pos(kCell);
        build_gather_tables(tab.data(), gather_off.data(), pair_pos.data(), heap_of_pos.data());
        std::vector<uint32_t> halo_list(1024);
        build_halo_list(tab.data(), pair_pos.data(), halo_list.data());
        build_lf_deltas(tab.data(), d.lf_delta);
        bool node0_gathered = false; // the fit kernels stage heap node 0 (the DC value) as the zero their "never a node" entries read: no node with a row may gather it
        for (int n = 2; n < kCell; n++)
            for (int k = 0; k < 6; k++) node0_gathered |= !(tab[(size_t)n * 6 + k] & 0x8000u) && (tab[(size_t)n * 6 + k] & 511u) == 0;
        if (halo_list[0] == 0xFFFFFFFFu || node0_gathered || pair_pos[0] != 0) { // (pair 0 = heap nodes 0 and 1 at the head of a cell: where the fit kernels' zero lives) // more halo values than threads / ...: the neighbour table is not the one the kernels were laid out for
            fri_hip_plan_destroy(p);
            return FRI_HIP_ERR_INVALID_ARGUMENT;
        }
        if ((rc = upload(p, g.tiles, d.tiles)) || (rc = upload(p, g.tile_cells, d.tile_cells)) || (rc = upload(p, g.tile_meta, d.tile_meta)) || (rc = upload(p, g.wg_tiles, d.wg_tiles)) || (rc = upload(p, g.wg_tiles_batch, d.wg_tiles_batch)) || (rc = upload(p, g.centers, d.centers)) ||
            (rc = upload(p, g.interior, d.interior)) || (rc = upload(p, g.valid_mask, d.valid_mask)) ||
            (rc = upload(p, g.nbr_cells, d.nbr_cells)) || (rc = upload(p, g.pred_slots, d.pred_slots)) || (rc = upload(p, tab, d.nbr_table)) ||
            (rc = upload(p, gather_off, d.gather_off)) || (rc = upload(p, pair_pos, d.pair_pos)) || (rc = upload(p, heap_of_pos, d.heap_of_pos)) ||
            (rc = upload(p, halo_list, d.halo_list))) {
            fri_hip_plan_destroy(p);
            return rc;
        }
        d.n_pred_tiles = g.n_pred_tiles;
        d.n_tiles = (uint32_t)g.tiles.size();
        d.F = (uint32_t)g.centers.size();
        d.width = (int32_t)g.width;
        d.height = (int32_t)g.height;
        d.channels = (int32_t)g.channels;
        d.lds_pitch = g.lds_pitch;
        d.lds_rows = g.lds_rows;
            d.cells_per_tile = g.cells_per_tile;
        d.max_tile_cells = g.max_tile_cells;
        d.covers_image = g.n_valid_leaves == (uint64_t)g.width * g.height;
        d.max_wg_tiles = std::max(g.max_wg_tiles, g.max_wg_tiles_batch);
        d.n_wg_batch = (uint32_t)g.wg_tiles_batch.size() - 1;
        d.max_wg_cells = g.max_wg_cells;
        // the inverse kernel keeps one share per resident workgroup: with the forward kernel's many short shares it walks groups of them
        d.inv_group = many_shares && g.wg_tiles.size() > 1 && (g.wg_tiles.size() - 1) % 1024 == 0 ? (int32_t)((g.wg_tiles.size() - 1) / 1024) : 1;
        d.inv_max_wg_tiles = d.inv_group > 1 ? 0 : d.max_wg_tiles, d.inv_max_wg_cells = d.inv_group > 1 ? 0 : d.max_wg_cells;
        for (size_t sh = 0; d.inv_group > 1 && sh + d.inv_group < g.wg_tiles.size(); sh += d.inv_group) {
            const Tile &first = g.tiles[g.wg_tiles[sh]], &last = g.tiles[g.wg_tiles[sh + d.inv_group] - 1];
            d.inv_max_wg_cells = std::max(d.inv_max_wg_cells, last.cell_begin + last.cell_count - first.cell_begin);
            d.inv_max_wg_tiles = std::max(d.inv_max_wg_tiles, g.wg_tiles[sh + d.inv_group] - g.wg_tiles[sh]);
        }
        d.n_wg = (uint32_t)g.wg_tiles.size() - 1;
        d.pred_blocks = (uint32_t)ctx->cu_count;
        if (env_int("FRI_HIP_PRED_BLOCKS") > 0) d.pred_blocks = (uint32_t)env_int("FRI_HIP_PRED_BLOCKS");
        {
            void *j = nullptr;
            if (hipMalloc(&j, (size_t)d.pred_blocks * kPredJunkWaves * kPredJunkBytes) != hipSuccess) {
                fri_hip_plan_destroy(p);
                return FRI_HIP_ERR_HIP;
            }
            p->owned.push_back(j);
            d.junk = static_cast<uint8_t *>(j);
        }
        d.hist_blocks = 2u * (uint32_t)ctx->cu_count; // two resident 512-thread workgroups per CU (LDS: 2 x 78 KiB)
        d.k1_ablate = env_int("FRI_HIP_K1_ABLATE");
        if (env_str("FRI_HIP_K1_CACHED_STORES")) d.k1_cached_stores = env_int("FRI_HIP_K1_CACHED_STORES") > 0 ? 1 : 0; // tuning: force plain (1) / nontemporal (0) coefficient stores everywhere
        d.k2_ablate = env_int("FRI_HIP_K2_ABLATE");
        if (const char *e = env_str("FRI_HIP_K1_BATCH_SHARES")) d.k1_batch_shares = std::atoi(e) != 0;
        d.k3_ablate = env_int("FRI_HIP_K3_ABLATE");
        d.k4_ablate = env_int("FRI_HIP_K4_ABLATE");
        if (const char *e = env_str("FRI_HIP_K4_OLDER_EIGHTHS")) d.k4_older_eighths = std::atoi(e);
        d.k3_scan = env_int("FRI_HIP_K3_SCAN") > 0;
        {
            const Geometry &gi = p->inv_geo();
            if (!gi.inv_lists.empty()) {
                if ((rc = upload(p, gi.inv_lists, d.inv_lists)) || (rc = upload(p, gi.inv_quads, d.inv_quads, kInvListPad)) || (rc = upload(p, gi.inv_dwords, d.inv_dwords, kInvListPad)) ||
                    (rc = upload(p, gi.inv_parts, d.inv_parts, kInvListPad))) {
                    fri_hip_plan_destroy(p);
                    return rc;
                }
                d.inv_rect_bytes = gi.inv_rect_bytes;
            }
        }
        if (env_int("FRI_HIP_TRACE") > 0) {
            const size_t bytes = (size_t)std::max<size_t>(d.n_wg, p->own_inverse_tiling ? p->geo_inv.wg_tiles.size() : 0) * 16 * sizeof(unsigned long long);
            void *t = nullptr;
            if (hipMalloc(&t, bytes) != hipSuccess || hipMemset(t, 0, bytes) != hipSuccess) {
                fri_hip_plan_destroy(p);
                return FRI_HIP_ERR_HIP;
            }
            p->owned.push_back(t);
            d.trace = static_cast<unsigned long long *>(t);
        }
        int hb = env_int("FRI_HIP_HIST_BLOCKS");
        if (hb > 0) d.hist_blocks = (uint32_t)hb;
        if (hipEventCreate(&p->ev_begin) != hipSuccess || hipEventCreate(&p->ev_end) != hipSuccess) {
            fri_hip_plan_destroy(p);
            return FRI_HIP_ERR_HIP;
        }
        // the inverse kernel's view of the plan: everything of `dev`, with its own tiles, cell records and shares swapped in
        p->dev_inv = d;
        if (p->own_inverse_tiling) {
            Geometry &gi = p->geo_inv;
            DevicePlan &di = p->dev_inv;
            if ((rc = upload(p, gi.tiles, di.tiles)) || (rc = upload(p, gi.tile_cells, di.tile_cells)) || (rc = upload(p, gi.tile_meta, di.tile_meta)) || (rc = upload(p, gi.wg_tiles, di.wg_tiles))) {
                fri_hip_plan_destroy(p);
                return rc;
            }
            di.wg_tiles_batch = nullptr, di.n_wg_batch = 0;
            di.n_tiles = (uint32_t)gi.tiles.size();
            di.n_wg = (uint32_t)gi.wg_tiles.size() - 1;
            di.lds_pitch = gi.lds_pitch, di.lds_rows = gi.lds_rows, di.cells_per_tile = gi.cells_per_tile, di.max_tile_cells = gi.max_tile_cells;
            di.max_wg_tiles = gi.max_wg_tiles, di.max_wg_cells = gi.max_wg_cells;
            di.inv_group = 1, di.inv_max_wg_tiles = gi.max_wg_tiles, di.inv_max_wg_cells = gi.max_wg_cells;
            // what the plan keeps of geo_inv on the host: the tiling (fri_hip_plan_inverse_lists reads the list sizes); the lattice arrays are geo's
            for (auto *v : {&gi.valid_mask}) std::vector<uint32_t>().swap(*v);
            std::vector<int32_t>().swap(gi.nbr_cells), std::vector<int32_t>().swap(gi.pred_slots);
        }
    }
    p->fwd_tp = tp_forward;
    p->fwd_tunable = ctx && p->own_inverse_tiling && !many_shares && !p->dev.trace && !env_str("FRI_HIP_BAND_ROWS") && !env_str("FRI_HIP_CELLS_PER_TILE") &&
                     !env_str("FRI_HIP_CELLS_PER_WG") && !env_str("FRI_HIP_STRIDED_SHARES") && !env_str("FRI_HIP_TARGET_WGS") && !env_str("FRI_HIP_RANKS") &&
                     !env_str("FRI_HIP_TILE_BYTES") && !env_str("FRI_HIP_RANK_WEIGHTS");
    if (p->fwd_tunable) adopt_cached_forward_tiling(p); // a tiling measured for this device and shape earlier in the process (fri_hip_plan_tune_forward)
    *out = p;
    return FRI_HIP_OK;
}

int fri_hip_plan_destroy(fri_hip_plan *p) {
    if (!p) return FRI_HIP_OK;
    if (p->ctx) {
        (void)hipSetDevice(p->ctx->device);
        for (void *d : p->owned) (void)hipFree(d);
        for (void *d : {(void *)p->d_pixels, (void *)p->d_coefs, (void *)p->d_bucket, (void *)p->d_prediction, (void *)p->d_hist, (void *)p->d_oob, (void *)p->d_fit_int, (void *)p->d_fit_dbl})
            if (d) (void)hipFree(d);
        free_slots(p);
        for (auto &a : p->acc_slots) {
            if (a.handed_over) (void)hipEventDestroy(a.handed_over);
            for (void *d : {(void *)a.pred_acc, (void *)a.fit_acc, (void *)a.sums_int, (void *)a.sums_dbl, (void *)a.range, (void *)a.params})
                if (d) (void)hipFree(d);
        }
        for (void *d : p->retired_acc) (void)hipFree(d);
        for (void *d : {(void *)p->d_bucket_all, (void *)p->d_prediction_all, (void *)p->d_hist_all, (void *)p->d_oob_all})
            if (d) (void)hipFree(d);
        if (p->d_stream_order) (void)hipFree(p->d_stream_order);
        if (p->d_stream_pos) (void)hipFree(p->d_stream_pos);
        if (p->d_coefs16) (void)hipFree(p->d_coefs16);
        if (p->ev_coefs16) (void)hipEventDestroy(p->ev_coefs16);
        if (p->d_symbols) (void)hipFree(p->d_symbols);
        if (p->h_fit) (void)hipHostFree(p->h_fit);
        if (p->ev_fit) (void)hipEventDestroy(p->ev_fit);
        if (p->ev_begin) (void)hipEventDestroy(p->ev_begin);
        if (p->ev_end) (void)hipEventDestroy(p->ev_end);
    }
    delete p;
    return FRI_HIP_OK;
}

uint32_t fri_hip_plan_num_cells(const fri_hip_plan *p) { return p ? (uint32_t)p->geo.centers.size() : 0; }
uint32_t fri_hip_plan_num_bfs_cells(const fri_hip_plan *p) { return p ? p->geo.n_bfs_cells : 0; }
uint32_t fri_hip_plan_num_interior_cells(const fri_hip_plan *p) { return p ? p->geo.n_interior : 0; }
size_t fri_hip_plan_coef_count(const fri_hip_plan *p) { return p ? (size_t)p->geo.channels * p->geo.centers.size() * kCell : 0; }
size_t fri_hip_plan_pixel_bytes(const fri_hip_plan *p) { return p ? (size_t)p->geo.width * p->geo.height * p->geo.channels : 0; }
uint64_t fri_hip_plan_num_some(const fri_hip_plan *p) { return p ? p->geo.n_some : 0; }

int fri_hip_plan_assume_forward_coefficients(fri_hip_plan *p, int on) {
    if (!p) return FRI_HIP_ERR_INVALID_ARGUMENT;
    p->assume_forward = on != 0;
    return FRI_HIP_OK;
}

int fri_hip_plan_set_dequantiser(fri_hip_plan *p, int mode) {
    if (!p || (mode != FRI_HIP_DEQUANT_REFERENCE && mode != FRI_HIP_DEQUANT_MULTIPLY)) return FRI_HIP_ERR_INVALID_ARGUMENT;
    p->dev.k3_multiply = p->dev_inv.k3_multiply = mode == FRI_HIP_DEQUANT_MULTIPLY;
    return FRI_HIP_OK;
}

int fri_hip_plan_centers(const fri_hip_plan *p, int32_t *centers) {
    if (!p || !centers) return FRI_HIP_ERR_INVALID_ARGUMENT;
    std::memcpy(centers, p->geo.centers.data(), p->geo.centers.size() * sizeof(Int2));
    return FRI_HIP_OK;
}
int fri_hip_plan_valid_mask(const fri_hip_plan *p, uint32_t *mask) {
    if (!p || !mask) return FRI_HIP_ERR_INVALID_ARGUMENT;
    std::memcpy(mask, p->geo.valid_mask.data(), p->geo.valid_mask.size() * sizeof(uint32_t));
    return FRI_HIP_OK;
}
int fri_hip_plan_neighbour_cells(const fri_hip_plan *p, int32_t *ids) {
    if (!p || !ids) return FRI_HIP_ERR_INVALID_ARGUMENT;
    std::memcpy(ids, p->geo.nbr_cells.data(), p->geo.nbr_cells.size() * sizeof(int32_t));
    return FRI_HIP_OK;
}
int fri_hip_plan_tiling(const fri_hip_plan *p, int32_t out[8]) {
    if (!p || !out) return FRI_HIP_ERR_INVALID_ARGUMENT;
    const Geometry &g = p->geo;
    const int32_t v[8] = {(int32_t)g.wg_tiles.size() - 1, (int32_t)g.tiles.size(), g.lds_pitch, g.lds_rows, g.max_tile_cells, g.band_rows, g.cells_per_tile, g.cells_per_wg};
    std::memcpy(out, v, sizeof(v));
    return FRI_HIP_OK;
}
int fri_hip_plan_inverse_lists(const fri_hip_plan *p, uint64_t out[5]) {
    if (!p || !out) return FRI_HIP_ERR_INVALID_ARGUMENT;
    const Geometry &g = p->inv_geo();
    uint64_t bits = 0;
    for (uint32_t e : g.inv_parts) bits += (uint64_t)__builtin_popcount(e & 15u);
    out[0] = g.inv_lists.empty() ? 0 : 1;
    out[1] = g.inv_quads.size();
    out[2] = g.inv_dwords.size();
    out[3] = bits;
    out[4] = (uint64_t)g.inv_rect_bytes;
    return FRI_HIP_OK;
}
int fri_hip_plan_read_trace(fri_hip_plan *p, uint64_t *out) {
    if (!p || !out || !p->ctx || !p->dev.trace) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    HIP_TRY(p->ctx, hipDeviceSynchronize());
    HIP_TRY(p->ctx, hipMemcpy(out, p->dev.trace, (size_t)p->dev.n_wg * 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FRI_HIP_OK;
}
int fri_hip_plan_tile_table(const fri_hip_plan *p, int32_t *tiles, int32_t *tile_cells, int32_t *wg_tiles) {
    if (!p) return FRI_HIP_ERR_INVALID_ARGUMENT;
    const Geometry &g = p->geo;
    static_assert(sizeof(Tile) == 6 * sizeof(int32_t), "Tile layout");
    if (tiles) std::memcpy(tiles, g.tiles.data(), g.tiles.size() * sizeof(Tile));

    if (tile_cells) std::memcpy(tile_cells, g.tile_cells.data(), g.tile_cells.size() * sizeof(int32_t));
    if (wg_tiles) std::memcpy(wg_tiles, g.wg_tiles.data(), g.wg_tiles.size() * sizeof(int32_t));
    return FRI_HIP_OK;
}
int fri_hip_plan_neighbour_table(const fri_hip_plan *p, uint16_t *table) {
    if (!p || !table) return FRI_HIP_ERR_INVALID_ARGUMENT;
    std::memcpy(table, &static_tables().nbr_table[0][0], sizeof(uint16_t) * kCell * 6);
    return FRI_HIP_OK;
}

/* ---- forward -------------------------------------------------------------------------------- */
int fri_hip_transform_quant_batch_dev(fri_hip_plan *p, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride,
                                      const int32_t qmatrix[32], int32_t *d_coefs, size_t coef_stride, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_pixels || !d_coefs || !n_images || n_images > 65535u) return FRI_HIP_ERR_INVALID_ARGUMENT;
    if (n_images > 1 && (pixel_stride < fri_hip_plan_pixel_bytes(p) || coef_stride < fri_hip_plan_coef_count(p))) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    HIP_TRY(p->ctx, launch_fwd_transform_quant(p->dev, n_images, d_pixels, pixel_stride, d_coefs, coef_stride, q, (hipStream_t)stream));
    return FRI_HIP_OK;
}

int fri_hip_transform_quant_dev(fri_hip_plan *p, const uint8_t *d_pixels, const int32_t qmatrix[32], int32_t *d_coefs, void *stream) {
    return fri_hip_transform_quant_batch_dev(p, 1, d_pixels, 0, qmatrix, d_coefs, 0, stream);
}

int fri_hip_transform_quant(fri_hip_plan *p, const uint8_t *pixels, const int32_t qmatrix[32], int32_t *coefs) {
    if (int rc = need_device(p)) return rc;
    if (!pixels || !coefs) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    if (int rc = ensure_staging(p)) return rc;
    HIP_TRY(p->ctx, hipMemcpy(p->d_pixels, pixels, fri_hip_plan_pixel_bytes(p), hipMemcpyHostToDevice));
    HIP_TRY(p->ctx, launch_fwd_transform_quant(p->dev, 1, p->d_pixels, 0, p->d_coefs, 0, q, nullptr));
    HIP_TRY(p->ctx, hipMemcpy(coefs, p->d_coefs, fri_hip_plan_coef_count(p) * sizeof(int32_t), hipMemcpyDeviceToHost));
    return FRI_HIP_OK;
}

int fri_hip_transform_quant_batch(fri_hip_plan *p, uint32_t n_images, const uint8_t *const *pixels, const int32_t qmatrix[32],
                                  int32_t *const *coefs) {
    if (int rc = need_device(p)) return rc;
    if (!pixels || !coefs) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    if (int rc = ensure_slots(p)) return rc;
    const size_t pb = fri_hip_plan_pixel_bytes(p), cb = fri_hip_plan_coef_count(p) * sizeof(int32_t);
    auto drain = [&](Slot &s) -> int {
        if (s.pending < 0) return FRI_HIP_OK;
        HIP_TRY(p->ctx, hipStreamSynchronize(s.stream));
        std::memcpy(coefs[s.pending], s.h_coefs, cb);
        s.pending = -1;
        return FRI_HIP_OK;
    };
    for (Slot &s : p->slots) s.pending = -1;
    for (uint32_t i = 0; i < n_images; i++) {
        if (!pixels[i] || !coefs[i]) return FRI_HIP_ERR_INVALID_ARGUMENT;
        Slot &s = p->slots[i % kBatchSlots];
        if (int rc = drain(s)) return rc; // H2D / kernel / D2H of the other slots keep running meanwhile
        std::memcpy(s.h_pixels, pixels[i], pb);
        HIP_TRY(p->ctx, hipMemcpyAsync(s.d_pixels, s.h_pixels, pb, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(p->ctx, launch_fwd_transform_quant(p->dev, 1, s.d_pixels, 0, s.d_coefs, 0, q, s.stream));
        HIP_TRY(p->ctx, hipMemcpyAsync(s.h_coefs, s.d_coefs, cb, hipMemcpyDeviceToHost, s.stream));
        s.pending = (int)i;
    }
    for (Slot &s : p->slots)
        if (int rc = drain(s)) return rc;
    return FRI_HIP_OK;
}

/* ---- sharding over GPUs ---------------------------------------------------------------------- */
uint32_t fri_hip_shard_size(uint32_t n_images, uint32_t shard, uint32_t n_shards) {
    if (!n_shards || shard >= n_shards) return 0;
    return n_images / n_shards + (shard < n_images % n_shards ? 1u : 0u);
}
uint32_t fri_hip_shard_image(uint32_t k, uint32_t shard, uint32_t n_shards) { return k * n_shards + shard; }

int fri_hip_multi_create(const int *devices, uint32_t n_devices, uint32_t width, uint32_t height, uint32_t channels, fri_hip_multi **out) {
    if (!out) return FRI_HIP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (!devices || !n_devices || n_devices > 64) return FRI_HIP_ERR_INVALID_ARGUMENT;
    fri_hip_multi *m = new (std::nothrow) fri_hip_multi;
    if (!m) return FRI_HIP_ERR_OUT_OF_MEMORY;
    m->ctxs.assign(n_devices, nullptr);
    m->plans.assign(n_devices, nullptr);
    for (uint32_t d = 0; d < n_devices; d++) {
        int rc = fri_hip_ctx_create(devices[d], &m->ctxs[d]);
        if (rc == FRI_HIP_OK) rc = fri_hip_plan_create(m->ctxs[d], width, height, channels, &m->plans[d]);
        if (rc != FRI_HIP_OK) {
            fri_hip_multi_destroy(m);
            return rc;
        }
    }
    *out = m;
    return FRI_HIP_OK;
}

int fri_hip_multi_destroy(fri_hip_multi *m) {
    if (!m) return FRI_HIP_OK;
    for (fri_hip_plan *p : m->plans) fri_hip_plan_destroy(p);
    for (fri_hip_ctx *c : m->ctxs) fri_hip_ctx_destroy(c);
    delete m;
    return FRI_HIP_OK;
}

uint32_t fri_hip_multi_num_devices(const fri_hip_multi *m) { return m ? (uint32_t)m->plans.size() : 0; }
fri_hip_plan *fri_hip_multi_plan(fri_hip_multi *m, uint32_t d) { return m && d < m->plans.size() ? m->plans[d] : nullptr; }

int fri_hip_multi_transform_quant(fri_hip_multi *m, uint32_t n_images, const uint8_t *const *pixels, const int32_t qmatrix[32], int32_t *const *coefs) {
    if (!m || !pixels || !coefs || !qmatrix) return FRI_HIP_ERR_INVALID_ARGUMENT;
    const uint32_t n_devices = (uint32_t)m->plans.size();
    std::vector<int> rcs(n_devices, FRI_HIP_OK);
    auto work = [&](uint32_t d) {
        const uint32_t n = fri_hip_shard_size(n_images, d, n_devices);
        if (!n) return;
        std::vector<const uint8_t *> in(n);
        std::vector<int32_t *> out(n);
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t i = fri_hip_shard_image(k, d, n_devices);
            in[k] = pixels[i];
            out[k] = coefs[i];
        }
        rcs[d] = fri_hip_transform_quant_batch(m->plans[d], n, in.data(), qmatrix, out.data());
    };
    std::vector<std::thread> threads;
    for (uint32_t d = 1; d < n_devices; d++) threads.emplace_back(work, d);
    work(0); // the calling thread drives the first device
    for (auto &t : threads) t.join();
    for (int rc : rcs)
        if (rc != FRI_HIP_OK) return rc;
    return FRI_HIP_OK;
}

// In a chain that goes straight from the forward kernel to the scan (parameters given: no fit in between) the forward kernel writes its coefficients with
// plain stores instead of nontemporal ones when the launch's coefficients fit the Infinity Cache comfortably: the scan then reads them from the cache instead of
// HBM (4096^2: K1 +1.5 us, K2 -4 us: 63.5 -> 61.0 us per image). With the fit in between (two more passes over the plane) it does not pay (161 = 162 us), and
// a forward kernel on its own is faster with nontemporal stores (18.7 against 20.3 us).
static bool chain_wants_cached_coefficients(const fri_hip_plan *p, uint32_t n_images, int fit) {
    return !fit && (size_t)n_images * fri_hip_plan_coef_count(p) * sizeof(int32_t) <= ((size_t)128 << 20);
}

/* ---- prediction + histogram ----------------------------------------------------------------- */
static int predict_launch(fri_hip_plan *p, const PredBatch &b, uint8_t *d_bucket, int32_t *d_prediction, uint32_t *d_hist, uint64_t *d_oob, int trust,
                          hipStream_t stream) {
    // The scan's histogram hand-over numbers its launches on the HOST (pred_serial below, a kernel argument): a captured launch would freeze that number, and from the
    // second replay on the clearing workgroups' flags already hold it - nothing would order the clearing of the histogram before the other workgroups' adds any more
    // (counts wiped or doubled). So a scan launch refuses to be captured, loudly, instead of recording something that must not be replayed.
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
        if (p->ctx) p->ctx->last_error = "the predict + histogram kernel cannot be captured into a HIP graph (its hand-over counts launches on the host): launch it on a stream";
        return FRI_HIP_ERR_INVALID_ARGUMENT;
    }
    (void)hipGetLastError();
    const int slot = acquire_acc(p, stream, b.n_planes);
    if (slot < 0) return slot;
    auto &acc = p->acc_slots[slot];
    if (++acc.pred_serial == 0) acc.pred_serial = 1; // 0 is what a fresh accumulator holds
    if (hipError_t e = launch_predict_histogram(p->dev, acc.pred_acc, acc.pred_serial, b, d_bucket, d_prediction, d_hist, (unsigned long long *)d_oob, trust, stream)) {
        p->acc_dirty = true;
        return fail_hip(p->ctx, e, "launch_predict_histogram");
    }
    return FRI_HIP_OK;
}
// d_range (may be NULL): per plane, how many waves staged a Some coefficient outside [-256, 255] (include/fri_hip.h: the fit's precondition)
static int fit_launch(fri_hip_plan *p, int mode, const PredBatch &b, int64_t *d_int, double *d_dbl, unsigned long long *d_range, hipStream_t stream,
                      const FitSolve *solve = nullptr, int trust = kPredAnyInt32) {
    const int slot = acquire_acc(p, stream, b.n_planes);
    if (slot < 0) return slot;
    const hipError_t e = launch_fit_accumulate(p->dev, p->acc_slots[slot].fit_acc, mode, b, (unsigned long long *)d_int, d_dbl, d_range, stream, solve, trust);
    if (e != hipSuccess) {
        p->acc_dirty = true;
        return fail_hip(p->ctx, e, "launch_fit_accumulate");
    }
    return FRI_HIP_OK;
}

int fri_hip_predict_histogram_batch_dev(fri_hip_plan *p, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, const float *d_params, uint8_t *d_bucket,
                                        int32_t *d_prediction, size_t out_stride, uint32_t *d_hist, uint64_t *d_n_out_of_alphabet, void *stream) {
    if (int rc = need_device(p)) return rc;
    const size_t plane = p->geo.centers.size() * kCell;
    if (!d_coefs || !d_params || !d_hist || !d_n_out_of_alphabet || !n_planes || n_planes > 65535u) return FRI_HIP_ERR_INVALID_ARGUMENT;
    if (n_planes > 1 && (coef_stride < plane || ((d_bucket || d_prediction) && out_stride < plane))) return FRI_HIP_ERR_INVALID_ARGUMENT;
    PredBatch b;
    b.n_planes = n_planes;
    b.coefs = d_coefs;
    b.coef_stride = coef_stride;
    b.out_stride = out_stride;
    b.params = reinterpret_cast<const PredictParams *>(d_params);
    return predict_launch(p, b, d_bucket, d_prediction, d_hist, d_n_out_of_alphabet, p->assume_forward ? kPredPromised : kPredAnyInt32, (hipStream_t)stream);
}

int fri_hip_predict_histogram_dev(fri_hip_plan *p, const int32_t *d_coefs, uint32_t channel, const float value_params[3][6],
                                  const float width_params[3][6], uint8_t *d_bucket, int32_t *d_prediction, uint32_t *d_hist,
                                  uint64_t *d_n_out_of_alphabet, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_coefs || !value_params || !width_params || !d_hist || !d_n_out_of_alphabet || channel >= p->geo.channels) return FRI_HIP_ERR_INVALID_ARGUMENT;
    PredBatch b;
    std::memcpy(b.pp[0].value, value_params, sizeof(b.pp[0].value));
    std::memcpy(b.pp[0].width, width_params, sizeof(b.pp[0].width));
    b.coefs = d_coefs + (size_t)channel * p->geo.centers.size() * kCell;
    return predict_launch(p, b, d_bucket, d_prediction, d_hist, d_n_out_of_alphabet, p->assume_forward ? kPredPromised : kPredAnyInt32, (hipStream_t)stream);
}

int fri_hip_predict_histogram(fri_hip_plan *p, const int32_t *coefs, uint32_t channel, const float value_params[3][6],
                              const float width_params[3][6], uint8_t *bucket, int32_t *prediction, uint32_t *hist,
                              uint64_t *n_out_of_alphabet) {
    if (int rc = need_device(p)) return rc;
    if (!coefs || !hist || channel >= p->geo.channels) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    if (int rc = ensure_staging(p)) return rc;
    const size_t F = p->geo.centers.size();
    HIP_TRY(p->ctx, hipMemcpy(p->d_coefs, coefs, fri_hip_plan_coef_count(p) * sizeof(int32_t), hipMemcpyHostToDevice));
    if (int rc = fri_hip_predict_histogram_dev(p, p->d_coefs, channel, value_params, width_params, bucket ? p->d_bucket : nullptr,
                                               prediction ? p->d_prediction : nullptr, p->d_hist, (uint64_t *)p->d_oob, nullptr))
        return rc;
    HIP_TRY(p->ctx, hipMemcpy(hist, p->d_hist, 10 * 1024 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (bucket) HIP_TRY(p->ctx, hipMemcpy(bucket, p->d_bucket, F * kCell, hipMemcpyDeviceToHost));
    if (prediction) HIP_TRY(p->ctx, hipMemcpy(prediction, p->d_prediction, F * kCell * sizeof(int32_t), hipMemcpyDeviceToHost));
    unsigned long long oob = 0;
    HIP_TRY(p->ctx, hipMemcpy(&oob, p->d_oob, sizeof(oob), hipMemcpyDeviceToHost));
    if (n_out_of_alphabet) *n_out_of_alphabet = oob;
    return FRI_HIP_OK;
}

/* ---- context-model fit sums ----------------------------------------------------------------------- */
int fri_hip_fit_value_sums_batch_dev(fri_hip_plan *p, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, int64_t *d_gram, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_coefs || !d_gram || !n_planes || n_planes > 65535u || (n_planes > 1 && coef_stride < p->geo.centers.size() * kCell)) return FRI_HIP_ERR_INVALID_ARGUMENT;
    PredBatch b;
    b.n_planes = n_planes;
    b.coefs = d_coefs;
    b.coef_stride = coef_stride;
    return fit_launch(p, 0, b, d_gram, nullptr, nullptr, (hipStream_t)stream);
}

int fri_hip_fit_width_sums_batch_dev(fri_hip_plan *p, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, const float *d_params, int64_t *d_wtw, double *d_wtr,
                                     void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_coefs || !d_params || !d_wtw || !d_wtr || !n_planes || n_planes > 65535u || (n_planes > 1 && coef_stride < p->geo.centers.size() * kCell))
        return FRI_HIP_ERR_INVALID_ARGUMENT;
    PredBatch b;
    b.n_planes = n_planes;
    b.coefs = d_coefs;
    b.coef_stride = coef_stride;
    b.params = reinterpret_cast<const PredictParams *>(d_params);
    return fit_launch(p, 1, b, d_wtw, d_wtr, nullptr, (hipStream_t)stream);
}

int fri_hip_fit_value_sums_dev(fri_hip_plan *p, const int32_t *d_coefs, uint32_t channel, int64_t *d_gram, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_coefs || !d_gram || channel >= p->geo.channels) return FRI_HIP_ERR_INVALID_ARGUMENT;
    PredBatch b;
    b.coefs = d_coefs + (size_t)channel * p->geo.centers.size() * kCell;
    return fit_launch(p, 0, b, d_gram, nullptr, nullptr, (hipStream_t)stream);
}

int fri_hip_fit_width_sums_dev(fri_hip_plan *p, const int32_t *d_coefs, uint32_t channel, const float value_params[3][6], int64_t *d_wtw, double *d_wtr,
                               void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_coefs || !value_params || !d_wtw || !d_wtr || channel >= p->geo.channels) return FRI_HIP_ERR_INVALID_ARGUMENT;
    PredBatch b;
    std::memcpy(b.pp[0].value, value_params, sizeof(b.pp[0].value));
    b.coefs = d_coefs + (size_t)channel * p->geo.centers.size() * kCell;
    return fit_launch(p, 1, b, d_wtw, d_wtr, nullptr, (hipStream_t)stream);
}

int fri_hip_fit_value_sums(fri_hip_plan *p, const int32_t *coefs, uint32_t channel, int64_t gram[3][28]) {
    if (int rc = need_device(p)) return rc;
    if (!coefs || !gram || channel >= p->geo.channels) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    if (int rc = ensure_staging(p)) return rc;
    HIP_TRY(p->ctx, hipMemcpy(p->d_coefs, coefs, fri_hip_plan_coef_count(p) * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        PredBatch b;
        b.coefs = p->d_coefs + (size_t)channel * p->geo.centers.size() * kCell;
        if (int rc = fit_launch(p, 0, b, (int64_t *)p->d_fit_int, nullptr, p->d_oob, nullptr)) return rc;
    }
    HIP_TRY(p->ctx, hipMemcpy(gram, p->d_fit_int, 3 * 28 * sizeof(int64_t), hipMemcpyDeviceToHost));
    unsigned long long out_of_range = 0;
    HIP_TRY(p->ctx, hipMemcpy(&out_of_range, p->d_oob, sizeof(out_of_range), hipMemcpyDeviceToHost));
    return out_of_range ? FRI_HIP_ERR_OUT_OF_RANGE : FRI_HIP_OK;
}

int fri_hip_fit_width_sums(fri_hip_plan *p, const int32_t *coefs, uint32_t channel, const float value_params[3][6], int64_t wtw[3][21], double wtr[3][6],
                           uint64_t rows[3]) {
    if (int rc = need_device(p)) return rc;
    if (!coefs || !value_params || !wtw || !wtr || channel >= p->geo.channels) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    if (int rc = ensure_staging(p)) return rc;
    HIP_TRY(p->ctx, hipMemcpy(p->d_coefs, coefs, fri_hip_plan_coef_count(p) * sizeof(int32_t), hipMemcpyHostToDevice));
    {
        PredBatch b;
        std::memcpy(b.pp[0].value, value_params, sizeof(b.pp[0].value));
        b.coefs = p->d_coefs + (size_t)channel * p->geo.centers.size() * kCell;
        if (int rc = fit_launch(p, 1, b, (int64_t *)p->d_fit_int, p->d_fit_dbl, p->d_oob, nullptr)) return rc;
    }
    unsigned long long out_of_range = 0;
    HIP_TRY(p->ctx, hipMemcpy(&out_of_range, p->d_oob, sizeof(out_of_range), hipMemcpyDeviceToHost));
    if (out_of_range) return FRI_HIP_ERR_OUT_OF_RANGE;
    HIP_TRY(p->ctx, hipMemcpy(wtw, p->d_fit_int, 3 * 21 * sizeof(int64_t), hipMemcpyDeviceToHost));
    HIP_TRY(p->ctx, hipMemcpy(wtr, p->d_fit_dbl, 18 * sizeof(double), hipMemcpyDeviceToHost));
    if (rows) { // num_ctx_last_layer / num_ctx_middle_layer, context_modeling.rs:84-85
        const uint64_t F = p->geo.centers.size();
        rows[0] = F * 256;
        rows[1] = F * 128;
        rows[2] = F * 128;
    }
    return FRI_HIP_OK;
}

/* ---- the 6 x 6 solves behind the fit (solve6.hpp: one source for these host functions and for the device's solve kernel) ------ */
void fri_hip_solve6(const double m[6][6], const double y[6], double x[6]) {
    Solve6Work w;
    std::memcpy(w.m, m, sizeof(w.m));
    std::memcpy(w.y, y, sizeof(w.y));
    solve6(w);
    std::memcpy(x, w.x, sizeof(w.x));
}

void fri_hip_fit_value_params(const int64_t gram[3][28], float value_params[3][6]) {
    Solve6Work w;
    for (int g = 0; g < 3; g++) fit_value_group(reinterpret_cast<const long long *>(gram[g]), value_params[g], w); // optimize_value_prediction, context_modeling.rs:175-202
}

void fri_hip_fit_width_params(const int64_t wtw[3][21], const double wtr[3][6], const uint64_t rows[3], float width_params[3][6]) {
    Solve6Work w;
    for (int g = 0; g < 3; g++) fit_width_group(reinterpret_cast<const long long *>(wtw[g]), wtr[g], rows[g], width_params[g], w); // optimize_width_prediction, :144-173
}

/* ---- the device part of FRIEncoder::encode in one call ------------------------------------------------ */
// the plan's compact coefficient planes for `planes` planes (grown, never shrunk; a growing call waits for the device: hipFree synchronises)
static int ensure_coefs16(fri_hip_plan *p, size_t planes) {
    if (p->coefs16_planes >= planes) return FRI_HIP_OK;
    fri_hip_ctx *c = p->ctx;
    if (p->d_coefs16) {
        HIP_TRY(c, hipFree(p->d_coefs16));
        p->d_coefs16 = nullptr, p->coefs16_planes = 0;
    }
    HIP_TRY(c, hipMalloc((void **)&p->d_coefs16, planes * p->geo.centers.size() * kCell * sizeof(int16_t)));
    p->coefs16_planes = planes;
    return FRI_HIP_OK;
}
// The planes belong to one chain at a time. Chains on one stream are ordered by it; a chain on another stream first waits for the event the previous one left.
static int coefs16_begin(fri_hip_plan *p, hipStream_t s) {
    if (p->coefs16_used && p->coefs16_stream != s && p->ev_coefs16) HIP_TRY(p->ctx, hipStreamWaitEvent(s, p->ev_coefs16, 0));
    return FRI_HIP_OK;
}
static int coefs16_end(fri_hip_plan *p, hipStream_t s) {
    if (!p->ev_coefs16) HIP_TRY(p->ctx, hipEventCreateWithFlags(&p->ev_coefs16, hipEventDisableTiming));
    HIP_TRY(p->ctx, hipEventRecord(p->ev_coefs16, s));
    p->coefs16_stream = s, p->coefs16_used = true;
    return FRI_HIP_OK;
}
static int ensure_encode_staging(fri_hip_plan *p, bool node_arrays = true) {
    fri_hip_ctx *c = p->ctx;
    const size_t C = p->geo.channels, plane = p->geo.centers.size() * kCell;
    if (node_arrays && !p->d_bucket_all) HIP_TRY(c, hipMalloc((void **)&p->d_bucket_all, C * plane));
    if (node_arrays && !p->d_prediction_all) HIP_TRY(c, hipMalloc((void **)&p->d_prediction_all, C * plane * sizeof(int32_t)));
    if (!p->d_hist_all) HIP_TRY(c, hipMalloc((void **)&p->d_hist_all, C * 10 * 1024 * sizeof(uint32_t)));
    if (!p->d_oob_all) HIP_TRY(c, hipMalloc((void **)&p->d_oob_all, C * sizeof(unsigned long long)));
    return FRI_HIP_OK;
}

// ContextModeler::optimize_parameters for the planes of `b` (prediction.rs:232-235, context_modeling.rs:204-213), entirely on the device and
// without a word to the host: value sums -> 6 x 6 solves -> width sums (with the value parameters just written) -> solves. The parameters land
// in the device array b.params (PredictParams per plane), where the scan kernel reads them. d_range (may be NULL) receives, per plane, the
// number of waves that staged a Some coefficient outside [-256, 255] (the sums are then not to be trusted).
// trust: kPredForwardOutput when this library's forward kernel wrote the planes earlier in the same call (the staging then does not look, and the count is 0).
static int fit_chain(fri_hip_plan *p, const PredBatch &b, unsigned long long *d_range, hipStream_t s, float *host_params = nullptr, unsigned long long *host_range = nullptr,
                     int trust = kPredAnyInt32) {
    if (!b.params) return FRI_HIP_ERR_INVALID_ARGUMENT;
    const int slot = acquire_acc(p, s, b.n_planes); // the stream's accumulators and fit scratch (the launches below find the same slot)
    if (slot < 0) return slot;
    auto &k = p->acc_slots[slot];
    float *params = const_cast<float *>(reinterpret_cast<const float *>(b.params));
    const uint64_t F = p->geo.centers.size();
    const unsigned long long rows[3] = {F * 256, F * 128, F * 128}; // num_ctx_last_layer / num_ctx_middle_layer, context_modeling.rs:84-85
    unsigned long long *range = d_range ? d_range : k.range;
    // two launches: each sums kernel solves in its tail (the workgroup that moves a plane's totals out)
    FitSolve solve;
    solve.params = params, solve.host_params = host_params, solve.host_range = host_range;
    for (int g = 0; g < 3; g++) solve.rows[g] = rows[g];
    if (int rc = fit_launch(p, 0, b, (int64_t *)k.sums_int, nullptr, range, s, &solve, trust)) return rc;
    solve.host_range = nullptr; // (the width pass does not count)
    return fit_launch(p, 1, b, (int64_t *)k.sums_int, k.sums_dbl, nullptr, s, &solve, trust);
}

// prediction::encode for all channels of one image whose coefficients are in device memory (prediction.rs:224-323 minus the host's ANS
// model), parameters on the HOST side of the ABI: given (fit == 0: they travel as kernel arguments) or fitted and returned (fit != 0: the
// device-side fit above, then the parameters and the range counts come back through pinned memory behind an event the host waits for
// while the scan kernel, already queued behind them, runs).
static int predict_image_dev(fri_hip_plan *p, const int32_t *d_coefs, int fit, float *value_params, float *width_params, uint8_t *d_bucket, int32_t *d_prediction,
                             uint32_t *d_hist, uint64_t *d_oob, int trust, hipStream_t s, uint16_t *d_words = nullptr, const int16_t *d_coefs16 = nullptr,
                             uint16_t *d_streams = nullptr) {
    fri_hip_ctx *c = p->ctx;
    const uint32_t C = p->geo.channels;
    const size_t plane = p->geo.centers.size() * kCell;
    PredBatch b;
    b.n_planes = C;
    b.coefs = d_coefs;
    b.coefs16 = d_coefs16; // (instead of d_coefs: the plan's compact planes, see fri_hip_encode_image_symbols)
    b.coef_stride = plane;
    b.out_stride = plane;
    b.words = d_words; // (the halfword form of the scan: see fri_hip_encode_image_symbols)
    if (d_streams) b.words = d_streams, b.out_stride = p->geo.n_some, b.stream_pos = p->d_stream_pos; // (its stream form: the channels' streams one behind the other, written by the scan itself)
    if (!fit) {
        for (uint32_t ch = 0; ch < C; ch++) {
            std::memcpy(b.pp[ch].value, value_params + ch * 18, sizeof(b.pp[ch].value));
            std::memcpy(b.pp[ch].width, width_params + ch * 18, sizeof(b.pp[ch].width));
        }
        return predict_launch(p, b, d_bucket, d_prediction, d_hist, d_oob, trust, s); // the scan loop of prediction::encode (prediction.rs:237-298), every channel
    }
    const int slot = acquire_acc(p, s, C);
    if (slot < 0) return slot;
    auto &k = p->acc_slots[slot];
    constexpr size_t kFitBytes = 3 * sizeof(PredictParams) + 3 * sizeof(unsigned long long);
    if (!p->h_fit) { // mapped and fine-grained: the solve kernels write the parameters straight into it
        HIP_TRY(c, hipHostMalloc(&p->h_fit, kFitBytes, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(p->h_fit, 0, kFitBytes);
        HIP_TRY(c, hipHostGetDevicePointer(&p->d_h_fit, p->h_fit, 0));
    }
    if (!p->ev_fit) HIP_TRY(c, hipEventCreateWithFlags(&p->ev_fit, hipEventDisableTiming));
    b.params = reinterpret_cast<const PredictParams *>(k.params);
    char *const h = static_cast<char *>(p->h_fit), *const dh = static_cast<char *>(p->d_h_fit);
    if (int rc = fit_chain(p, b, k.range, s, reinterpret_cast<float *>(dh), reinterpret_cast<unsigned long long *>(dh + 3 * sizeof(PredictParams)), trust)) return rc;
    HIP_TRY(c, hipEventRecord(p->ev_fit, s));
    // the scan is queued before the host looks at the fit: it runs while the host picks the parameters up
    const int rc_scan = predict_launch(p, b, d_bucket, d_prediction, d_hist, d_oob, trust, s);
    // poll: a sleeping wait (hipEventSynchronize) wakes up 15-20 us late, a tenth of the chain
    for (uint64_t spins = 0;; spins++) {
        const hipError_t q = hipEventQuery(p->ev_fit);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(c, q);
        if (spins > (1ull << 22)) {
            HIP_TRY(c, hipEventSynchronize(p->ev_fit));
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const unsigned long long *h_range = reinterpret_cast<const unsigned long long *>(h + 3 * sizeof(PredictParams));
    const PredictParams *h_params = reinterpret_cast<const PredictParams *>(h);
    for (uint32_t ch = 0; ch < C; ch++) {
        std::memcpy(value_params + ch * 18, h_params[ch].value, sizeof(h_params[ch].value));
        std::memcpy(width_params + ch * 18, h_params[ch].width, sizeof(h_params[ch].width));
    }
    for (uint32_t ch = 0; ch < C; ch++)
        if (h_range[ch]) return FRI_HIP_ERR_OUT_OF_RANGE; // a Some coefficient outside [-256, 255]: the fit's 32-bit partial sums would overflow
    return rc_scan;
}

int fri_hip_encode_image_dev(fri_hip_plan *p, const uint8_t *d_pixels, const int32_t qmatrix[32], int fit, float *value_params, float *width_params, int32_t *d_coefs,
                             uint8_t *d_bucket, int32_t *d_prediction, uint32_t *d_hist, uint64_t *d_n_out_of_alphabet, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_pixels || !value_params || !width_params || !d_coefs || !d_hist || !d_n_out_of_alphabet) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    // wavelet_transform::encode + quantization::encode (encoder.rs:24-31): one kernel, all channels; the coefficients then stay where they are
    HIP_TRY(p->ctx, launch_fwd_transform_quant(p->dev, 1, d_pixels, 0, d_coefs, 0, q, (hipStream_t)stream, chain_wants_cached_coefficients(p, 1, fit)));
    // |coefficient| <= 255 for quantisers of magnitude >= 1, and the kernel above wrote every one of them: the scan need not look
    return predict_image_dev(p, d_coefs, fit, value_params, width_params, d_bucket, d_prediction, d_hist, d_n_out_of_alphabet, kPredForwardOutput, (hipStream_t)stream);
}

// n_images images in one asynchronous chain, parameters in DEVICE memory (no host round trip, no synchronisation):
// K1 (all images, all channels) -> [value sums -> solves -> width sums -> solves] -> K2 (all planes).
int fri_hip_encode_image_batch_dev(fri_hip_plan *p, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride, const int32_t qmatrix[32], int fit, float *d_params,
                                   int32_t *d_coefs, size_t coef_stride, uint8_t *d_bucket, int32_t *d_prediction, size_t out_stride, uint32_t *d_hist,
                                   uint64_t *d_n_out_of_alphabet, uint64_t *d_fit_out_of_range, void *stream) {
    if (int rc = need_device(p)) return rc;
    const uint32_t C = p->geo.channels;
    const size_t plane = p->geo.centers.size() * kCell, image = (size_t)C * plane;
    if (!d_pixels || !d_params || !d_coefs || !d_hist || !d_n_out_of_alphabet || !n_images || (uint64_t)n_images * C > 65535u) return FRI_HIP_ERR_INVALID_ARGUMENT;
    if (n_images > 1 && (pixel_stride < fri_hip_plan_pixel_bytes(p) || coef_stride < image || ((d_bucket || d_prediction) && out_stride < image))) return FRI_HIP_ERR_INVALID_ARGUMENT;
    // the planes of the batch must be evenly spaced for the one-launch-per-stage form: C == 1 (any stride) or images back to back
    if (C > 1 && n_images > 1 && (coef_stride != image || ((d_bucket || d_prediction) && out_stride != image))) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(p->ctx, launch_fwd_transform_quant(p->dev, n_images, d_pixels, pixel_stride, d_coefs, coef_stride, q, s, chain_wants_cached_coefficients(p, n_images, fit)));
    PredBatch b;
    b.n_planes = n_images * C;
    b.coefs = d_coefs;
    b.coef_stride = C > 1 ? plane : coef_stride;
    b.out_stride = C > 1 ? plane : out_stride;
    b.params = reinterpret_cast<const PredictParams *>(d_params);
    if (fit)
        if (int rc = fit_chain(p, b, (unsigned long long *)d_fit_out_of_range, s, nullptr, nullptr, kPredForwardOutput)) return rc;
    return predict_launch(p, b, d_bucket, d_prediction, d_hist, d_n_out_of_alphabet, kPredForwardOutput, s);
}

// The same chain all the way to the emitter's input: the scan kernel writes one halfword per node (bucket << 10 | symbol) instead of bucket and
// prediction arrays, and the gather kernel puts them in the reference's stream order. Asynchronous like the call above.
int fri_hip_encode_symbols_batch_dev(fri_hip_plan *p, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride, const int32_t qmatrix[32], int fit, float *d_params,
                                     int32_t *d_coefs, size_t coef_stride, uint16_t *d_node_words, size_t word_stride, uint16_t *d_symbols, size_t symbol_stride,
                                     uint32_t *d_hist, uint64_t *d_n_out_of_alphabet, uint64_t *d_fit_out_of_range, void *stream) {
    if (int rc = need_device(p)) return rc;
    const uint32_t C = p->geo.channels;
    const size_t plane = p->geo.centers.size() * kCell, image = (size_t)C * plane, n = p->geo.n_some;
    if (!d_pixels || !d_params || !d_symbols || !d_hist || !d_n_out_of_alphabet || !n_images || (uint64_t)n_images * C > 65535u || !p->d_stream_order)
        return FRI_HIP_ERR_INVALID_ARGUMENT;
    const bool compact = d_coefs == nullptr; // the caller does not want the coefficients: they travel between the kernels as the plan's int16 planes
    const bool direct = d_node_words == nullptr; // ... nor the node words: the scan writes every symbol straight to its place in the stream, no gather kernel
    if (direct && !compact) return FRI_HIP_ERR_INVALID_ARGUMENT; // (the stream form of the scan exists for the compact planes only)
    if (compact) coef_stride = image;
    if (direct) word_stride = image;
    if (n_images > 1 && (pixel_stride < fri_hip_plan_pixel_bytes(p) || coef_stride < image || word_stride < image || symbol_stride < (size_t)C * n)) return FRI_HIP_ERR_INVALID_ARGUMENT;
    // evenly spaced planes, as above; a channel's stream follows the previous channel's
    if (C > 1 && n_images > 1 && (coef_stride != image || word_stride != image || symbol_stride != (size_t)C * n)) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    if (compact) {
        if (int rc = ensure_coefs16(p, (size_t)n_images * C)) return rc;
        if (int rc = coefs16_begin(p, s)) return rc;
    }
    HIP_TRY(p->ctx, launch_fwd_transform_quant(p->dev, n_images, d_pixels, pixel_stride, d_coefs, coef_stride, q, s, chain_wants_cached_coefficients(p, n_images, fit),
                                               compact ? p->d_coefs16 : nullptr));
    PredBatch b;
    b.n_planes = n_images * C;
    b.coefs = d_coefs;
    b.coefs16 = compact ? p->d_coefs16 : nullptr;
    b.coef_stride = C > 1 || compact ? plane : coef_stride;
    b.out_stride = C > 1 ? plane : word_stride;
    b.params = reinterpret_cast<const PredictParams *>(d_params);
    b.words = d_node_words;
    if (direct) b.words = d_symbols, b.out_stride = C > 1 ? n : symbol_stride, b.stream_pos = p->d_stream_pos; // the planes' streams, one behind the other as the gather would lay them
    if (fit)
        if (int rc = fit_chain(p, b, (unsigned long long *)d_fit_out_of_range, s, nullptr, nullptr, kPredForwardOutput)) return rc;
    if (int rc = predict_launch(p, b, nullptr, nullptr, d_hist, d_n_out_of_alphabet, kPredForwardOutput, s)) return rc;
    if (!direct) HIP_TRY(p->ctx, launch_symbol_gather(p->d_stream_order, n, b.n_planes, d_node_words, b.out_stride, d_symbols, C > 1 ? n : symbol_stride, s));
    if (compact) // (the scan was the planes' last reader; the event sits behind the gather, which is later than it must be and costs nothing)
        if (int rc = coefs16_end(p, s)) return rc;
    return FRI_HIP_OK;
}

int fri_hip_fit_value_params_batch_dev(fri_hip_plan *p, uint32_t n_planes, const int64_t *d_gram, float *d_params, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_gram || !d_params || !n_planes || n_planes > 65535u) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    HIP_TRY(p->ctx, launch_fit_solve(0, n_planes, reinterpret_cast<const unsigned long long *>(d_gram), nullptr, nullptr, d_params, (hipStream_t)stream));
    return FRI_HIP_OK;
}

int fri_hip_fit_width_params_batch_dev(fri_hip_plan *p, uint32_t n_planes, const int64_t *d_wtw, const double *d_wtr, float *d_params, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_wtw || !d_wtr || !d_params || !n_planes || n_planes > 65535u) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    const uint64_t F = p->geo.centers.size();
    const unsigned long long rows[3] = {F * 256, F * 128, F * 128}; // num_ctx_last_layer / num_ctx_middle_layer, context_modeling.rs:84-85
    HIP_TRY(p->ctx, launch_fit_solve(1, n_planes, reinterpret_cast<const unsigned long long *>(d_wtw), d_wtr, rows, d_params, (hipStream_t)stream));
    return FRI_HIP_OK;
}

int fri_hip_fit_params_batch_dev(fri_hip_plan *p, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, float *d_params, uint64_t *d_fit_out_of_range, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_coefs || !d_params || !n_planes || n_planes > 65535u || (n_planes > 1 && coef_stride < p->geo.centers.size() * kCell)) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    PredBatch b;
    b.n_planes = n_planes;
    b.coefs = d_coefs;
    b.coef_stride = coef_stride;
    b.params = reinterpret_cast<const PredictParams *>(d_params);
    return fit_chain(p, b, (unsigned long long *)d_fit_out_of_range, (hipStream_t)stream);
}

int fri_hip_predict_image_dev(fri_hip_plan *p, const int32_t *d_coefs, int fit, float *value_params, float *width_params, uint8_t *d_bucket, int32_t *d_prediction,
                              uint32_t *d_hist, uint64_t *d_n_out_of_alphabet, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_coefs || !value_params || !width_params || !d_hist || !d_n_out_of_alphabet) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    return predict_image_dev(p, d_coefs, fit, value_params, width_params, d_bucket, d_prediction, d_hist, d_n_out_of_alphabet, kPredAnyInt32, (hipStream_t)stream);
}

int fri_hip_predict_image(fri_hip_plan *p, const int32_t *coefs, int fit, float *value_params, float *width_params, uint8_t *bucket, int32_t *prediction, uint32_t *hist,
                          uint64_t *n_out_of_alphabet) {
    if (int rc = need_device(p)) return rc;
    if (!coefs || !value_params || !width_params || !hist || !n_out_of_alphabet) return FRI_HIP_ERR_INVALID_ARGUMENT;
    fri_hip_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure_staging(p)) return rc;
    if (int rc = ensure_encode_staging(p)) return rc;
    const size_t C = p->geo.channels, plane = p->geo.centers.size() * kCell;
    HIP_TRY(c, hipMemcpy(p->d_coefs, coefs, C * plane * sizeof(int32_t), hipMemcpyHostToDevice)); // once, for the fit and the scan of every channel
    if (int rc = predict_image_dev(p, p->d_coefs, fit, value_params, width_params, bucket ? p->d_bucket_all : nullptr, prediction ? p->d_prediction_all : nullptr, p->d_hist_all,
                                   (uint64_t *)p->d_oob_all, kPredAnyInt32, nullptr))
        return rc;
    if (bucket) HIP_TRY(c, hipMemcpy(bucket, p->d_bucket_all, C * plane, hipMemcpyDeviceToHost));
    if (prediction) HIP_TRY(c, hipMemcpy(prediction, p->d_prediction_all, C * plane * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(hist, p->d_hist_all, C * 10 * 1024 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(n_out_of_alphabet, p->d_oob_all, C * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FRI_HIP_OK;
}

int fri_hip_encode_image(fri_hip_plan *p, const uint8_t *pixels, const int32_t qmatrix[32], int fit, float *value_params, float *width_params, int32_t *coefs, uint8_t *bucket,
                         int32_t *prediction, uint32_t *hist, uint64_t *n_out_of_alphabet) {
    if (int rc = need_device(p)) return rc;
    if (!pixels || !value_params || !width_params || !coefs || !hist || !n_out_of_alphabet) return FRI_HIP_ERR_INVALID_ARGUMENT;
    fri_hip_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure_staging(p)) return rc;
    if (int rc = ensure_encode_staging(p)) return rc;
    const size_t C = p->geo.channels, plane = p->geo.centers.size() * kCell;
    // one upload of the pixels, one download of each output; the coefficients stay on the device between the stages
    HIP_TRY(c, hipMemcpy(p->d_pixels, pixels, fri_hip_plan_pixel_bytes(p), hipMemcpyHostToDevice));
    if (int rc = fri_hip_encode_image_dev(p, p->d_pixels, qmatrix, fit, value_params, width_params, p->d_coefs, bucket ? p->d_bucket_all : nullptr,
                                          prediction ? p->d_prediction_all : nullptr, p->d_hist_all, (uint64_t *)p->d_oob_all, nullptr))
        return rc;
    HIP_TRY(c, hipMemcpy(coefs, p->d_coefs, C * plane * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (bucket) HIP_TRY(c, hipMemcpy(bucket, p->d_bucket_all, C * plane, hipMemcpyDeviceToHost));
    if (prediction) HIP_TRY(c, hipMemcpy(prediction, p->d_prediction_all, C * plane * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(hist, p->d_hist_all, C * 10 * 1024 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(n_out_of_alphabet, p->d_oob_all, C * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FRI_HIP_OK;
}

// The reference's per-image loop (crates/fri-cli/src/commands/bench.rs:15-120 around FRIEncoder::encode, encoder.rs:87-109) for host buffers on ONE
// device: each image runs the whole asynchronous chain (fri_hip_encode_image_batch_dev with one image) on one of three streams, so that image
// i + 1's upload and image i - 1's download overlap image i's kernels.
int fri_hip_encode_image_batch(fri_hip_plan *p, uint32_t n_images, const uint8_t *const *pixels, const int32_t qmatrix[32], int fit, float *const *params,
                               int32_t *const *coefs, uint8_t *const *bucket, int32_t *const *prediction, uint32_t *const *hist, uint64_t *const *n_out_of_alphabet) {
    if (int rc = need_device(p)) return rc;
    if (!pixels || !params || !coefs || !hist || !n_out_of_alphabet) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    fri_hip_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = ensure_encode_slots(p, bucket != nullptr, prediction != nullptr)) return rc;
    const size_t C = p->geo.channels, pb = fri_hip_plan_pixel_bytes(p), n = fri_hip_plan_coef_count(p);
    int first_error = FRI_HIP_OK;
    auto drain = [&](Slot &s) -> int {
        if (s.pending < 0) return FRI_HIP_OK;
        const int i = s.pending;
        s.pending = -1;
        HIP_TRY(c, hipStreamSynchronize(s.stream));
        std::memcpy(coefs[i], s.h_coefs, n * sizeof(int32_t));
        if (bucket && bucket[i]) std::memcpy(bucket[i], s.h_bucket, n);
        if (prediction && prediction[i]) std::memcpy(prediction[i], s.h_prediction, n * sizeof(int32_t));
        std::memcpy(hist[i], s.h_hist, C * 10 * 1024 * sizeof(uint32_t));
        std::memcpy(n_out_of_alphabet[i], s.h_oob, C * sizeof(uint64_t));
        if (fit) std::memcpy(params[i], s.h_params, C * sizeof(PredictParams));
        for (size_t ch = 0; ch < C; ch++)
            if (fit && s.h_oob[C + ch]) first_error = first_error ? first_error : FRI_HIP_ERR_OUT_OF_RANGE; // (the fit's range count; the scan does not check what the forward kernel wrote)
        return FRI_HIP_OK;
    };
    for (Slot &s : p->slots) s.pending = -1;
    for (uint32_t i = 0; i < n_images; i++) {
        if (!pixels[i] || !params[i] || !coefs[i] || !hist[i] || !n_out_of_alphabet[i]) return FRI_HIP_ERR_INVALID_ARGUMENT;
        Slot &s = p->slots[i % kBatchSlots];
        if (int rc = drain(s)) return rc; // the other slots' copies and kernels keep running meanwhile
        std::memcpy(s.h_pixels, pixels[i], pb);
        HIP_TRY(c, hipMemcpyAsync(s.d_pixels, s.h_pixels, pb, hipMemcpyHostToDevice, s.stream));
        if (!fit) {
            std::memcpy(s.h_params, params[i], C * sizeof(PredictParams));
            HIP_TRY(c, hipMemcpyAsync(s.d_params, s.h_params, C * sizeof(PredictParams), hipMemcpyHostToDevice, s.stream));
        }
        if (int rc = fri_hip_encode_image_batch_dev(p, 1, s.d_pixels, 0, qmatrix, fit, s.d_params, s.d_coefs, 0, bucket && bucket[i] ? s.d_bucket : nullptr,
                                                    prediction && prediction[i] ? s.d_prediction : nullptr, 0, s.d_hist, (uint64_t *)s.d_oob, (uint64_t *)(s.d_oob + C), s.stream))
            return rc;
        HIP_TRY(c, hipMemcpyAsync(s.h_coefs, s.d_coefs, n * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
        if (bucket && bucket[i]) HIP_TRY(c, hipMemcpyAsync(s.h_bucket, s.d_bucket, n, hipMemcpyDeviceToHost, s.stream));
        if (prediction && prediction[i]) HIP_TRY(c, hipMemcpyAsync(s.h_prediction, s.d_prediction, n * sizeof(int32_t), hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(c, hipMemcpyAsync(s.h_hist, s.d_hist, C * 10 * 1024 * sizeof(uint32_t), hipMemcpyDeviceToHost, s.stream));
        HIP_TRY(c, hipMemcpyAsync(s.h_oob, s.d_oob, 2 * C * sizeof(unsigned long long), hipMemcpyDeviceToHost, s.stream));
        if (fit) HIP_TRY(c, hipMemcpyAsync(s.h_params, s.d_params, C * sizeof(PredictParams), hipMemcpyDeviceToHost, s.stream));
        s.pending = (int)i;
    }
    for (Slot &s : p->slots)
        if (int rc = drain(s)) return rc;
    return first_error;
}

// ... and over the GPUs of a node: image i on devices[i mod n_devices], one host thread per device running the loop above over its shard.
int fri_hip_multi_encode_image(fri_hip_multi *m, uint32_t n_images, const uint8_t *const *pixels, const int32_t qmatrix[32], int fit, float *const *params,
                               int32_t *const *coefs, uint8_t *const *bucket, int32_t *const *prediction, uint32_t *const *hist, uint64_t *const *n_out_of_alphabet) {
    if (!m || !pixels || !params || !coefs || !hist || !n_out_of_alphabet || !qmatrix) return FRI_HIP_ERR_INVALID_ARGUMENT;
    const uint32_t n_devices = (uint32_t)m->plans.size();
    std::vector<int> rcs(n_devices, FRI_HIP_OK);
    auto work = [&](uint32_t d) {
        const uint32_t n = fri_hip_shard_size(n_images, d, n_devices);
        if (!n) return;
        std::vector<const uint8_t *> in(n);
        std::vector<float *> par(n);
        std::vector<int32_t *> co(n), pr(n);
        std::vector<uint8_t *> bu(n);
        std::vector<uint32_t *> hi(n);
        std::vector<uint64_t *> oo(n);
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t i = fri_hip_shard_image(k, d, n_devices);
            in[k] = pixels[i], par[k] = params[i], co[k] = coefs[i], hi[k] = hist[i], oo[k] = n_out_of_alphabet[i];
            bu[k] = bucket ? bucket[i] : nullptr, pr[k] = prediction ? prediction[i] : nullptr;
        }
        rcs[d] = fri_hip_encode_image_batch(m->plans[d], n, in.data(), qmatrix, fit, par.data(), co.data(), bucket ? bu.data() : nullptr, prediction ? pr.data() : nullptr, hi.data(),
                                            oo.data());
    };
    std::vector<std::thread> threads;
    for (uint32_t d = 1; d < n_devices; d++) threads.emplace_back(work, d);
    work(0); // the calling thread drives the first device
    for (auto &t : threads) t.join();
    for (int rc : rcs)
        if (rc != FRI_HIP_OK) return rc;
    return FRI_HIP_OK;
}

/* ---- the ordered symbol stream (K5) ------------------------------------------------------------------------- */
int fri_hip_plan_set_stream_order(fri_hip_plan *p, const uint32_t *order, uint64_t n) {
    if (int rc = need_device(p)) return rc;
    if (!order || n != p->geo.n_some) return FRI_HIP_ERR_INVALID_ARGUMENT;
    const size_t F = p->geo.centers.size();
    // a permutation of the Some nodes: every entry a Some node, every Some node once (the kernel indexes the planes with it)
    std::vector<uint8_t> seen(F * kCell, 0);
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t e = order[i], cell = e >> 9, heap = e & 511u;
        if (cell >= F || !((p->geo.valid_mask[(size_t)cell * 16 + (heap >> 5)] >> (heap & 31)) & 1u) || seen[e]) return FRI_HIP_ERR_INVALID_ARGUMENT;
        seen[e] = 1;
    }
    fri_hip_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!p->d_stream_order) HIP_TRY(c, hipMalloc((void **)&p->d_stream_order, (n ? n : 1) * sizeof(uint32_t)));
    HIP_TRY(c, hipMemcpy(p->d_stream_order, order, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    { // the inverse: where a node's symbol sits in the stream (the scan writes the stream directly when nobody asks for the node words)
        std::vector<uint32_t> pos(F * kCell, 0xFFFFFFFFu);
        for (uint64_t i = 0; i < n; i++) pos[order[i]] = (uint32_t)i;
        if (!p->d_stream_pos) HIP_TRY(c, hipMalloc((void **)&p->d_stream_pos, pos.size() * sizeof(uint32_t)));
        HIP_TRY(c, hipMemcpy(p->d_stream_pos, pos.data(), pos.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    return FRI_HIP_OK;
}

int fri_hip_symbol_stream_batch_dev(fri_hip_plan *p, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, const uint8_t *d_bucket, const int32_t *d_prediction,
                                    size_t out_stride, uint16_t *d_symbols, size_t symbol_stride, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!p->d_stream_order) return FRI_HIP_ERR_INVALID_ARGUMENT; // fri_hip_plan_set_stream_order first
    const size_t plane = p->geo.centers.size() * kCell;
    if (!d_coefs || !d_bucket || !d_prediction || !d_symbols || !n_planes || n_planes > 65535u) return FRI_HIP_ERR_INVALID_ARGUMENT;
    if (n_planes > 1 && (coef_stride < plane || out_stride < plane || symbol_stride < p->geo.n_some)) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    HIP_TRY(p->ctx, launch_symbol_stream(p->d_stream_order, p->geo.n_some, n_planes, d_coefs, coef_stride, d_bucket, d_prediction, out_stride, d_symbols, symbol_stride,
                                         (hipStream_t)stream));
    return FRI_HIP_OK;
}

// The device part of FRIEncoder::encode_bytes for host buffers: pixels up (1 B per pixel and channel), the whole chain and the symbol stream kernel on
// the device, and down come 2 bytes per symbol, the histograms and the parameters - the arrays of coefficients, predictions and buckets never leave.
int fri_hip_encode_image_symbols(fri_hip_plan *p, const uint8_t *pixels, const int32_t qmatrix[32], int fit, float *value_params, float *width_params, uint16_t *symbols,
                                 uint32_t *hist, uint64_t *n_out_of_alphabet) {
    if (int rc = need_device(p)) return rc;
    if (!pixels || !value_params || !width_params || !symbols || !hist || !n_out_of_alphabet || !p->d_stream_order) return FRI_HIP_ERR_INVALID_ARGUMENT;
    fri_hip_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!p->d_pixels) HIP_TRY(c, hipMalloc((void **)&p->d_pixels, fri_hip_plan_pixel_bytes(p))); // (of the single-image staging only the pixels: no int32 planes, no node arrays here)
    if (int rc = ensure_encode_staging(p, false)) return rc;
    const size_t C = p->geo.channels, n = p->geo.n_some;
    if (!p->d_symbols) HIP_TRY(c, hipMalloc((void **)&p->d_symbols, (C * n ? C * n : 1) * sizeof(uint16_t)));
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    HIP_TRY(c, hipMemcpy(p->d_pixels, pixels, fri_hip_plan_pixel_bytes(p), hipMemcpyHostToDevice));
    // nobody outside sees the coefficients of this call: compact planes (int16, None as 0) between the forward kernel, the fit and the scan
    if (int rc = ensure_coefs16(p, C)) return rc;
    if (int rc = coefs16_begin(p, nullptr)) return rc;
    HIP_TRY(c, launch_fwd_transform_quant(p->dev, 1, p->d_pixels, 0, nullptr, 0, q, nullptr, false, p->d_coefs16));
    // the scan in its stream form: no bucket / prediction arrays, no node words, no gather kernel - every symbol goes straight to its place in its channel's stream
    if (int rc = predict_image_dev(p, nullptr, fit, value_params, width_params, nullptr, nullptr, p->d_hist_all, (uint64_t *)p->d_oob_all, kPredForwardOutput, nullptr, nullptr,
                                   p->d_coefs16, p->d_symbols))
        return rc;
    if (int rc = coefs16_end(p, nullptr)) return rc;
    HIP_TRY(c, hipMemcpy(symbols, p->d_symbols, C * n * sizeof(uint16_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(hist, p->d_hist_all, C * 10 * 1024 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(n_out_of_alphabet, p->d_oob_all, C * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return FRI_HIP_OK;
}

/* ---- inverse ---------------------------------------------------------------------------------- */
int fri_hip_inverse_transform_batch_dev(fri_hip_plan *p, uint32_t n_images, const int32_t *d_coefs, size_t coef_stride, const int32_t qmatrix[32], uint8_t *d_pixels,
                                        size_t pixel_stride, void *stream) {
    if (int rc = need_device(p)) return rc;
    if (!d_coefs || !d_pixels || !n_images || n_images > 65535u) return FRI_HIP_ERR_INVALID_ARGUMENT;
    if (n_images > 1 && (pixel_stride < fri_hip_plan_pixel_bytes(p) || coef_stride < fri_hip_plan_coef_count(p))) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    HIP_TRY(p->ctx, launch_inverse_transform(p->dev_inv, n_images, d_coefs, coef_stride, q, d_pixels, pixel_stride, (hipStream_t)stream));
    return FRI_HIP_OK;
}

int fri_hip_inverse_transform_dev(fri_hip_plan *p, const int32_t *d_coefs, const int32_t qmatrix[32], uint8_t *d_pixels, void *stream) {
    return fri_hip_inverse_transform_batch_dev(p, 1, d_coefs, 0, qmatrix, d_pixels, 0, stream);
}

int fri_hip_inverse_transform(fri_hip_plan *p, const int32_t *coefs, const int32_t qmatrix[32], uint8_t *pixels) {
    if (int rc = need_device(p)) return rc;
    if (!coefs || !pixels) return FRI_HIP_ERR_INVALID_ARGUMENT;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    if (int rc = ensure_staging(p)) return rc;
    HIP_TRY(p->ctx, hipMemcpy(p->d_coefs, coefs, fri_hip_plan_coef_count(p) * sizeof(int32_t), hipMemcpyHostToDevice));
    if (int rc = fri_hip_inverse_transform_dev(p, p->d_coefs, qmatrix, p->d_pixels, nullptr)) return rc;
    HIP_TRY(p->ctx, hipMemcpy(pixels, p->d_pixels, fri_hip_plan_pixel_bytes(p), hipMemcpyDeviceToHost));
    return FRI_HIP_OK;
}

/* ---- timing helper ------------------------------------------------------------------------------ */
int fri_hip_time_transform_quant_dev(fri_hip_plan *p, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride,
                                     const int32_t qmatrix[32], int32_t *d_coefs, size_t coef_stride, uint32_t iters, void *stream,
                                     double *mean_us) {
    if (int rc = need_device(p)) return rc;
    if (!d_pixels || !d_coefs || !n_images || !iters || !mean_us) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!p->ev_begin) HIP_TRY(p->ctx, hipEventCreate(&p->ev_begin)); // normally made by fri_hip_plan_create; owned by the plan either way
    if (!p->ev_end) HIP_TRY(p->ctx, hipEventCreate(&p->ev_end));
    const hipEvent_t e0 = p->ev_begin, e1 = p->ev_end;
    HIP_TRY(p->ctx, hipEventRecord(e0, s));
    for (uint32_t i = 0; i < iters; i++) {
        const uint32_t k = i % n_images;
        HIP_TRY(p->ctx, launch_fwd_transform_quant(p->dev, 1, d_pixels + (size_t)k * pixel_stride, 0, d_coefs + (size_t)k * coef_stride, 0, q, s));
    }
    HIP_TRY(p->ctx, hipEventRecord(e1, s));
    // Poll instead of hipEventSynchronize: after a few hundred microseconds the runtime's wait goes to sleep, and the wake-up (15-20 us,
    // plus as much again in the caller's next synchronize) would be charged to a measurement of twenty 17-us launches.
    for (uint64_t spins = 0;; spins++) {
        const hipError_t q = hipEventQuery(e1);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(p->ctx, q);
        if (spins > (1ull << 22)) { // seconds: not a measurement any more
            HIP_TRY(p->ctx, hipEventSynchronize(e1));
            break;
        }
    }
    float ms = 0.f;
    HIP_TRY(p->ctx, hipEventElapsedTime(&ms, e0, e1));
    *mean_us = (double)ms * 1000.0 / iters;
    return FRI_HIP_OK;
}


/* The same loop with the launches dealt over `n_streams` streams of the library's own (launch i on stream i mod n_streams; the images are independent):
 * launch i + 1's workgroups move into the CUs launch i's early finishers leave, so the drain of one launch overlaps the ramp of the next. mean_us is the
 * launch PERIOD (time from the first launch's begin to the last one's end over iters) - not a kernel duration: with more than one stream launches overlap. */
int fri_hip_time_transform_quant_streams_dev(fri_hip_plan *p, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride, const int32_t qmatrix[32],
                                             int32_t *d_coefs, size_t coef_stride, uint32_t iters, uint32_t n_streams, double *mean_us) {
    if (int rc = need_device(p)) return rc;
    if (!d_pixels || !d_coefs || !n_images || !iters || !mean_us || !n_streams || n_streams > 8) return FRI_HIP_ERR_INVALID_ARGUMENT;
    QMatrix q;
    if (int rc = check_q(qmatrix, q)) return rc;
    fri_hip_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t st[8] = {};
    hipEvent_t done[8] = {}, e0 = nullptr, e1 = nullptr;
    int rc = FRI_HIP_OK;
    auto cleanup = [&]() {
        for (uint32_t k = 0; k < n_streams; k++) {
            if (done[k]) (void)hipEventDestroy(done[k]);
            if (st[k]) (void)hipStreamDestroy(st[k]);
        }
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define TRY_OR_CLEAN(expr)                                  \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) {                             \
            rc = fail_hip(c, e_, #expr);                    \
            (void)hipDeviceSynchronize();                   \
            cleanup();                                      \
            return rc;                                      \
        }                                                   \
    } while (0)
    for (uint32_t k = 0; k < n_streams; k++) {
        TRY_OR_CLEAN(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        TRY_OR_CLEAN(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
    }
    TRY_OR_CLEAN(hipEventCreate(&e0));
    TRY_OR_CLEAN(hipEventCreate(&e1));
    TRY_OR_CLEAN(hipDeviceSynchronize()); // whatever the caller has queued is done before the clock starts
    TRY_OR_CLEAN(hipEventRecord(e0, st[0]));
    for (uint32_t k = 1; k < n_streams; k++) TRY_OR_CLEAN(hipStreamWaitEvent(st[k], e0, 0)); // nobody starts before the begin event
    for (uint32_t i = 0; i < iters; i++) {
        const uint32_t k = i % n_images;
        TRY_OR_CLEAN(launch_fwd_transform_quant(p->dev, 1, d_pixels + (size_t)k * pixel_stride, 0, d_coefs + (size_t)k * coef_stride, 0, q, st[i % n_streams]));
    }
    for (uint32_t k = 1; k < n_streams; k++) {
        TRY_OR_CLEAN(hipEventRecord(done[k], st[k]));
        TRY_OR_CLEAN(hipStreamWaitEvent(st[0], done[k], 0)); // the end event is behind every stream's last launch
    }
    TRY_OR_CLEAN(hipEventRecord(e1, st[0]));
    TRY_OR_CLEAN(hipEventSynchronize(e1));
    float ms = 0.f;
    TRY_OR_CLEAN(hipEventElapsedTime(&ms, e0, e1));
#undef TRY_OR_CLEAN
    *mean_us = (double)ms * 1000.0 / iters;
    cleanup();
    return rc;
}

/* Measures candidate tilings of the forward kernel on this plan's device and keeps the fastest (see "forward tiling chosen by measurement" above). */
int fri_hip_plan_tune_forward(fri_hip_plan *p, uint32_t launches, char *report, size_t report_bytes) {
    if (int rc = need_device(p)) return rc;
    std::string rep;
    auto finish = [&](int rc) {
        if (report && report_bytes) {
            std::snprintf(report, report_bytes, "%s", rep.c_str());
        }
        return rc;
    };
    if (!p->fwd_tunable) {
        rep = "{\"tuned\": false, \"kept\": \"" + p->fwd_tiling_note + "\", \"why\": \"tiling pinned by the environment, shared with the inverse kernel, or a many-shares plan\"}";
        return finish(FRI_HIP_OK);
    }
    fri_hip_ctx *c = p->ctx;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!launches) launches = 96;
    const Geometry &g0 = p->geo;
    const bool rgb = g0.channels != 1;
    // candidates: the plan's tiling first, then the settings that won somewhere in round 4's sweeps (tools/r4_k1_hbm*.sh, r4_cells.sh)
    std::vector<std::unique_ptr<FwdTiling>> cand;
    auto add_tp = [&](const TilingParams &tp) {
        auto t = std::make_unique<FwdTiling>();
        t->tp = tp;
        if (t->tp.cells_per_tile < 0 || t->tp.band_rows < 0) return;
        if (!build_forward_candidate(p, *t)) return;
        t->label = tiling_label(t->tp, t->geo);
        for (const auto &o : cand)
            if (o->label == t->label) return; // the builder clamped it to something already there
        cand.push_back(std::move(t));
    };
    auto add = [&](int strided, int band, int cells_delta) {
        TilingParams tp = p->fwd_tp;
        if (strided >= 0) tp.strided_shares = strided != 0;
        if (band > 0) tp.band_rows = band;
        if (cells_delta) tp.cells_per_tile = g0.cells_per_tile + cells_delta;
        add_tp(tp);
    };
    add(-1, 0, 0);
    if (cand.empty()) {
        rep = "{\"tuned\": false, \"why\": \"the plan's own tiling could not be rebuilt\"}";
        return finish(FRI_HIP_OK);
    }
    if (!rgb) {
        add(0, 72, 0), add(1, 16, 1), add(1, 32, 0), add(1, 8, 0), add(0, 8, 0), add(1, 24, 0), add(0, 48, 0), add(0, 96, 0);
    } else {
        add(1, 12, 0), add(1, 24, 0), add(0, 16, 0), add(1, 8, 0), add(0, 32, 0);
    }
    // scratch: enough distinct images that the pixels cannot come from the 256 MiB Infinity Cache, and coefficient slots whose rewrites are > 512 MB apart
    const size_t px_bytes = fri_hip_plan_pixel_bytes(p), co_bytes = fri_hip_plan_coef_count(p) * sizeof(int32_t);
    const uint32_t px_slots = (uint32_t)std::min<size_t>(64, std::max<size_t>(2, ((size_t)420 << 20) / px_bytes + 1));
    const uint32_t co_slots = (uint32_t)std::min<size_t>(px_slots, std::max<size_t>(2, ((size_t)512 << 20) / co_bytes + 1));
    uint8_t *d_px = nullptr;
    int32_t *d_co = nullptr;
    unsigned long long *d_stat = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t stream = nullptr;
    int rc = FRI_HIP_OK;
    auto cleanup = [&]() {
        (void)hipDeviceSynchronize();
        for (auto &t : cand)
            for (void *b : t->bufs) (void)hipFree(b);
        if (d_stat) (void)hipFree(d_stat);
        if (d_px) (void)hipFree(d_px);
        if (d_co) (void)hipFree(d_co);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (stream) (void)hipStreamDestroy(stream);
    };
#define TRY_OR_CLEAN(expr)                                  \
    do {                                                    \
        hipError_t e_ = (expr);                             \
        if (e_ != hipSuccess) {                             \
            rc = fail_hip(c, e_, #expr);                    \
            cleanup();                                      \
            return finish(rc);                              \
        }                                                   \
    } while (0)
    TRY_OR_CLEAN(hipMalloc((void **)&d_px, (size_t)px_slots * px_bytes));
    TRY_OR_CLEAN(hipMalloc((void **)&d_co, (size_t)co_slots * co_bytes));
    TRY_OR_CLEAN(hipMemset(d_px, 0x5A, (size_t)px_slots * px_bytes)); // the forward kernel's time does not depend on the pixel values
    TRY_OR_CLEAN(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    TRY_OR_CLEAN(hipEventCreate(&e0));
    TRY_OR_CLEAN(hipEventCreate(&e1));
    for (auto &t : cand) TRY_OR_CLEAN(upload_forward_candidate(p, *t));
    QMatrix q;
    for (int i = 0; i < 32; i++) q.q[i] = 1;
    constexpr int kRounds = 5;
    uint32_t it = 0;
    auto run = [&](const FwdTiling &t, uint32_t n) -> hipError_t {
        for (uint32_t i = 0; i < n; i++, it++)
            if (hipError_t e = launch_fwd_transform_quant(t.dev, 1, d_px + (size_t)(it % px_slots) * px_bytes, 0, d_co + (size_t)(it % co_slots) * (co_bytes / 4), 0, q, stream)) return e;
        return hipSuccess;
    };
    // median microseconds per launch of the candidates `idx`, in interleaved rounds: a drift of the clocks hits every candidate alike
    auto measure = [&](const std::vector<size_t> &idx) -> hipError_t {
        std::vector<std::vector<double>> times(idx.size());
        for (int r = 0; r < kRounds; r++)
            for (size_t k = 0; k < idx.size(); k++) {
                if (hipError_t e = run(*cand[idx[k]], 8)) return e;
                if (hipError_t e = hipEventRecord(e0, stream)) return e;
                if (hipError_t e = run(*cand[idx[k]], launches)) return e;
                if (hipError_t e = hipEventRecord(e1, stream)) return e;
                if (hipError_t e = hipEventSynchronize(e1)) return e;
                float ms = 0.f;
                if (hipError_t e = hipEventElapsedTime(&ms, e0, e1)) return e;
                times[k].push_back((double)ms * 1000.0 / launches);
            }
        for (size_t k = 0; k < idx.size(); k++) {
            std::sort(times[k].begin(), times[k].end());
            cand[idx[k]]->us = times[k][times[k].size() / 2];
        }
        return hipSuccess;
    };
    TRY_OR_CLEAN(run(*cand[0], 4 * launches)); // clocks and translations up before anything is compared
    std::vector<size_t> all(cand.size());
    for (size_t k = 0; k < cand.size(); k++) all[k] = k;
    TRY_OR_CLEAN(measure(all));
    size_t best = 0;
    for (size_t k = 0; k < cand.size(); k++)
        if (cand[k]->us < cand[best]->us) best = k;
    // the default stays unless a candidate beats it by more than the noise between rounds (1.5 %)
    if (cand[best]->us > cand[0]->us * 0.985) best = 0;
    // Equal times: contiguous shares. They re-read less halo through HBM at every band height (4096^2, PMC passes over 21 tilings: 1.016-1.018 x the algorithmic bytes
    // against 1.044-1.056 x for interleaved shares, profiles/r05_k1_traffic_by_tiling.json) - a contiguous candidate within 1 % of an interleaved winner takes its place.
    if (cand[best]->tp.strided_shares) {
        size_t c = best;
        for (size_t k = 0; k < cand.size(); k++)
            if (!cand[k]->tp.strided_shares && cand[k]->us <= cand[best]->us * 1.01 && (c == best || cand[k]->us < cand[c]->us)) c = k;
        best = c;
    }
    // Second phase: around the winner - its neighbours in band height and tile size, for contiguous shares the XCDs taking turns over groups of tiles, flatter and
    // steeper share sizes by dispatch rank - measured together with it, interleaved like the first phase.
    {
        const TilingParams w = cand[best]->tp;
        const Geometry &gw = cand[best]->geo;
        std::vector<size_t> round2{best};
        const size_t n0 = cand.size();
        auto also = [&](TilingParams tp) { add_tp(tp); };
        for (int d : {-8, 8}) {
            TilingParams tp = w;
            tp.band_rows = gw.band_rows + d;
            if (tp.band_rows >= 8) also(tp);
        }
        for (int d : {-1, 1}) {
            TilingParams tp = w;
            tp.cells_per_tile = gw.cells_per_tile + d;
            if (tp.cells_per_tile >= 2) also(tp);
        }
        if (w.ranks == 4 && w.rank_weight[0] > 0) {
            static const float flat[4] = {1.2f, 1.05f, 0.95f, 0.8f}, steep[4] = {1.4f, 1.15f, 0.85f, 0.6f};
            for (const float *v : {flat, steep}) {
                TilingParams tp = w;
                for (int i = 0; i < 4; i++) tp.rank_weight[i] = v[i];
                also(tp);
            }
        }
        for (size_t k = n0; k < cand.size(); k++) {
            TRY_OR_CLEAN(upload_forward_candidate(p, *cand[k]));
            round2.push_back(k);
        }
        if (round2.size() > 1) {
            TRY_OR_CLEAN(measure(round2));
            size_t b2 = best;
            for (size_t k : round2)
                if (cand[k]->us < cand[b2]->us) b2 = k;
            if (cand[b2]->us < cand[best]->us * 0.99) best = b2; // (1 %: the winner of phase one was measured again in this phase, side by side)
        }
    }
    // How long each XCD's workgroups live under the winner (FwdArgs::xcd_stat; a diagnostic for the report: re-cutting the XCDs' shares in proportion was tried and
    // does not transfer from the tuner's scratch buffers to the caller's - which XCD is slow changes with the buffers' placement).
    std::string xcd_note;
    if (hipMalloc((void **)&d_stat, 16 * sizeof(unsigned long long)) == hipSuccess) {
        unsigned long long h_stat[16];
        TRY_OR_CLEAN(hipMemsetAsync(d_stat, 0, sizeof h_stat, stream));
        cand[best]->dev.k1_xcd_stat = d_stat;
        const hipError_t e_run = run(*cand[best], launches);
        cand[best]->dev.k1_xcd_stat = nullptr;
        TRY_OR_CLEAN(e_run);
        TRY_OR_CLEAN(hipStreamSynchronize(stream));
        TRY_OR_CLEAN(hipMemcpy(h_stat, d_stat, sizeof h_stat, hipMemcpyDeviceToHost));
        for (int x = 0; x < 8; x++) {
            char b[32];
            std::snprintf(b, sizeof b, "%s%.2f", x ? " " : "", h_stat[2 * x + 1] ? (double)h_stat[2 * x] / (double)h_stat[2 * x + 1] / 100.0 : 0.0);
            xcd_note += b;
        }
    }
#undef TRY_OR_CLEAN
    char b[160];
    rep = "{\"tuned\": true, \"launches\": " + std::to_string(launches) + ", \"rounds\": " + std::to_string(kRounds) + ", \"pixel_slots\": " + std::to_string(px_slots) +
          ", \"coef_slots\": " + std::to_string(co_slots) + ", \"winner\": \"" + cand[best]->label + "\", \"candidates_us\": {";
    for (size_t k = 0; k < cand.size(); k++) {
        std::snprintf(b, sizeof b, "%s\"%s\": %.3f", k ? ", " : "", cand[k]->label.c_str(), cand[k]->us);
        rep += b;
    }
    rep += "}, \"xcd_workgroup_lifetimes_us\": \"" + xcd_note + "\"}";
    (void)hipDeviceSynchronize();
    if (best != 0) adopt_forward_tiling(p, *cand[best]);
    else p->fwd_tiling_note = cand[0]->label + " (measured: the default won)";
    {
        std::lock_guard<std::mutex> lk(g_tuned_mu);
        const TunedKey k{c->device, g0.width, g0.height, g0.channels};
        auto itc = std::find_if(g_tuned.begin(), g_tuned.end(), [&](const auto &e) { return e.first == k; });
        if (itc == g_tuned.end()) g_tuned.emplace_back(k, p->fwd_tp);
        else itc->second = p->fwd_tp;
    }
    cleanup();
    return finish(FRI_HIP_OK);
}

} // extern "C"
