/*
 * fri_hip.h -- C ABI of libfri_hip.so: the MI355X (gfx950) implementation of libfri's
 * transform + quantisation + prediction/histogram hot path (and its inverse).
 *
 * This is the drop-in boundary. The reference (pagmerek/frave, crate libfri) has no FFI of its
 * own; every entry point below names the private Rust stage function whose body it replaces.
 * Paths are relative to crates/libfri/src/ of the reference. INTEGRATION.md shows the Rust
 * `extern "C"` block and the replacement stage bodies.
 *
 * Conventions
 *   - Every call returns 0 on success or a negative FRI_HIP_ERR_* code; fri_hip_strerror() gives
 *     text. Nothing panics or throws across the boundary.
 *   - All buffers are caller-owned. "host" entry points take host pointers and are synchronous.
 *     "_dev" entry points take device pointers, enqueue on `stream` (a hipStream_t passed as
 *     void*, NULL = the null stream) and return without synchronising.
 *   - One ctx per (host thread, GPU). Calls on one ctx/plan are not thread-safe; distinct ctxs are
 *     independent. There is no global mutable state.
 *   - There is NO CPU fallback: without a usable gfx950 device ctx_create fails and every compute
 *     entry point returns FRI_HIP_ERR_NO_DEVICE.
 *
 * Data layout
 *   pixels  : interleaved u8, index ((y*width + x)*channels + c)             (images.rs:94)
 *   cells   : the F retained 512-pixel tiles ("Fractal", stages/wavelet_transform.rs:29-37) in
 *             canonical order = ascending centre.im, then centre.re          (utils.rs:17-32)
 *   coefs   : int32 [channels][F][512], heap order inside a cell: index 0 = DC, 1 = root,
 *             2^l .. 2^(l+1)-1 = level l                (Fractal.coefficients, wavelet_transform.rs:32)
 *             Option::None is encoded as FRI_HIP_NONE.
 *   bucket  : u8  [F][512], prediction: int32 [F][512]   (Fractal.parameter_predictors, :33)
 *   hist    : u32 [10][1024]                             (AnsContext.freqs, stages/entropy_coding.rs:34)
 */
#ifndef FRI_HIP_H
#define FRI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what this header declares is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define FRI_HIP_NONE INT32_MIN
#define FRI_HIP_CELL_SIZE 512     /* 1 << BASE_FRAC_DEPTH, stages/wavelet_transform.rs:39 */
#define FRI_HIP_CONTEXT_AMOUNT 10 /* stages/prediction.rs:15 */
#define FRI_HIP_ALPHABET_SIZE 1024 /* stages/entropy_coding.rs:25 */

#define FRI_HIP_OK 0
#define FRI_HIP_ERR_INVALID_ARGUMENT (-1)
#define FRI_HIP_ERR_HIP (-2)          /* a HIP runtime call failed; see fri_hip_last_hip_error() */
#define FRI_HIP_ERR_NO_DEVICE (-3)    /* no gfx950 device / host-only plan used for compute */
#define FRI_HIP_ERR_OUT_OF_MEMORY (-4)
#define FRI_HIP_ERR_DIVIDE_BY_ZERO (-5) /* a used qmatrix entry is 0 (Rust: division panic, quantization.rs:17) */
#define FRI_HIP_ERR_EMPTY_LATTICE (-6)  /* no retained cell (Rust: index panic, wavelet_transform.rs:664) */
#define FRI_HIP_ERR_OUT_OF_RANGE (-7)   /* fit sums: a Some coefficient outside [-256, 255] (see fri_hip_fit_value_sums) */

typedef struct fri_hip_ctx fri_hip_ctx;
typedef struct fri_hip_plan fri_hip_plan;

const char *fri_hip_strerror(int code);
const char *fri_hip_version(void);

/* ---- context ------------------------------------------------------------------------------ */
/* Binds to HIP device `device`; fails with FRI_HIP_ERR_NO_DEVICE unless it is a gfx950 GPU. */
int fri_hip_ctx_create(int device, fri_hip_ctx **out);
int fri_hip_ctx_destroy(fri_hip_ctx *ctx);
/* "hip:gfx950" for a live ctx. */
const char *fri_hip_backend(const fri_hip_ctx *ctx);
/* Text of the last failing HIP call on this ctx ("" if none). */
const char *fri_hip_last_hip_error(const fri_hip_ctx *ctx);

/* ---- plan: geometry of one (width, height, channels), cached and reusable ------------------ */
/* Replaces WaveletImage::fractal_divide + Fractal::new + the retain() filter +
 * get_global_position_map (stages/wavelet_transform.rs:42-69, 405-484): the cell lattice, the
 * address map and the Some/None pattern depend on (width, height) only.
 * channels is 1 or 3. With channels == 1 a cell is retained iff it has >= 1 in-image leaf (the
 * reference's own Luma path drops every cell and panics).
 * ctx may be NULL: the plan is then host-only (getters work, compute returns NO_DEVICE). */
int fri_hip_plan_create(fri_hip_ctx *ctx, uint32_t width, uint32_t height, uint32_t channels, fri_hip_plan **out);
int fri_hip_plan_destroy(fri_hip_plan *plan);

uint32_t fri_hip_plan_num_cells(const fri_hip_plan *plan);     /* F, after retain()        */
uint32_t fri_hip_plan_num_bfs_cells(const fri_hip_plan *plan); /* size of fractal_divide() */
uint32_t fri_hip_plan_num_interior_cells(const fri_hip_plan *plan);
size_t fri_hip_plan_coef_count(const fri_hip_plan *plan);      /* channels * F * 512 */
size_t fri_hip_plan_pixel_bytes(const fri_hip_plan *plan);     /* width * height * channels */
/* centres[F][2] = (re, im) in canonical order. */
int fri_hip_plan_centers(const fri_hip_plan *plan, int32_t *centers);
/* mask[F][16]: bit (i & 31) of word (i >> 5) set <=> coefficient i of the cell is Some. Channel independent. */
int fri_hip_plan_valid_mask(const fri_hip_plan *plan, uint32_t *mask);
/* Number of Some coefficients per channel (= histogram total per channel). */
uint64_t fri_hip_plan_num_some(const fri_hip_plan *plan);
/* ids[F][8]: cell ids of {self, +V9[0..5]} neighbours (stages/wavelet_transform.rs:71-95), -1 = absent. */
int fri_hip_plan_neighbour_cells(const fri_hip_plan *plan, int32_t *ids);
/* table[512][6] u16: static neighbour map used by the gather kernel (context_modeling.rs:25-77):
 * bits 0-8 heap index to read, bits 9-11 index into the neighbour-cell list, bit 15 = "always 0". */
int fri_hip_plan_neighbour_table(const fri_hip_plan *plan, uint16_t *table);

/* out[8] = {workgroup shares, tiles, LDS row pitch (bytes), LDS rows, max cells per tile, band rows, cells per tile,
 * cells per workgroup}: how the forward kernel decomposes the image (diagnostics / tuning; FRI_HIP_BAND_ROWS,
 * FRI_HIP_CELLS_PER_TILE, FRI_HIP_CELLS_PER_WG override the defaults at plan creation). */
int fri_hip_plan_tiling(const fri_hip_plan *plan, int32_t out[8]);
/* The decomposition itself (any pointer may be NULL): tiles[n_tiles][6] = {x_lo, y_lo, width_px, n_rows, cell_begin,
 * cell_count}; tile_cells[F] = cell ids in tile order; wg_tiles[n_wg + 1] = tile range of each workgroup share. */
int fri_hip_plan_tile_table(const fri_hip_plan *plan, int32_t *tiles, int32_t *tile_cells, int32_t *wg_tiles);

/* ---- forward: transform + quantisation ------------------------------------------------------ */
/* Replaces wavelet_transform::encode (stages/wavelet_transform.rs:708-713: from_raster ->
 * Fractal::extract_coefficients :179-225) followed by quantization::encode
 * (stages/quantization.rs:7-25) with qmatrix = get_quantization_matrix() (:3-5, all ones today).
 * qmatrix[layer], layer = floor(log2(i + 1)) for heap index i; truncating division. */
int fri_hip_transform_quant(fri_hip_plan *plan, const uint8_t *pixels, const int32_t qmatrix[32], int32_t *coefs);
int fri_hip_transform_quant_dev(fri_hip_plan *plan, const uint8_t *d_pixels, const int32_t qmatrix[32], int32_t *d_coefs,
                                void *stream);
/* n independent images of the plan's shape: image k at d_pixels + k*pixel_stride (bytes),
 * coefficients at d_coefs + k*coef_stride (int32 elements). One launch. */
int fri_hip_transform_quant_batch_dev(fri_hip_plan *plan, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride,
                                      const int32_t qmatrix[32], int32_t *d_coefs, size_t coef_stride, void *stream);
/* Host batch: n images, pinned staging, H2D / kernel / D2H overlapped on internal streams. */
int fri_hip_transform_quant_batch(fri_hip_plan *plan, uint32_t n_images, const uint8_t *const *pixels, const int32_t qmatrix[32],
                                  int32_t *const *coefs);

/* ---- a batch of independent images over several GPUs (BASELINE config 4) ---------------------- */
/* The reference encodes a batch as a sequential loop over independent images (crates/fri-cli/src/commands/bench.rs:15-120 around
 * FRIEncoder::encode, encoder.rs:87-109). Images never exchange data (K2 needs all cells of ONE image, so an image is never
 * split), hence the batch shards by image with no collective: image i belongs to shard i mod n_shards. These two functions are
 * the partition every multi-GPU path uses (one process per GPU under torch.distributed: shard = rank; one process driving
 * several GPUs: shard = position in `devices`). */
uint32_t fri_hip_shard_size(uint32_t n_images, uint32_t shard, uint32_t n_shards);  /* images of this shard (0 on bad arguments) */
uint32_t fri_hip_shard_image(uint32_t k, uint32_t shard, uint32_t n_shards);        /* global index of the shard's k-th image */
/* One process driving several GPUs of a node: a fri_hip_multi owns one ctx + plan (+ its stream set and pinned staging) per
 * device. fri_hip_multi_transform_quant starts one host thread per device; thread d runs fri_hip_transform_quant_batch over
 * shard d of the images (image i -> devices[i mod n_devices]). No data moves between devices. Returns the first failing
 * shard's error code. fri_hip_multi_plan gives device d's plan for use with any other entry point (from one thread at a time). */
typedef struct fri_hip_multi fri_hip_multi;
int fri_hip_multi_create(const int *devices, uint32_t n_devices, uint32_t width, uint32_t height, uint32_t channels, fri_hip_multi **out);
int fri_hip_multi_destroy(fri_hip_multi *m);
uint32_t fri_hip_multi_num_devices(const fri_hip_multi *m);
fri_hip_plan *fri_hip_multi_plan(fri_hip_multi *m, uint32_t d);
int fri_hip_multi_transform_quant(fri_hip_multi *m, uint32_t n_images, const uint8_t *const *pixels, const int32_t qmatrix[32],
                                  int32_t *const *coefs);

/* ---- prediction + context bucket + ANS symbol histogram -------------------------------------- */
/* Replaces the loop body of prediction::encode (stages/prediction.rs:237-298) for one channel:
 * get_lf_context_bucket (:86-149) for heap index 0 and 1, get_hf_context_bucket (:151-207) with
 * ContextModeler::get_neighbour_values (context_modeling.rs:25-77) for levels 1..8, pack_signed
 * (utils.rs:34-40) and AnsContext::bump_freq (stages/entropy_coding.rs:98-100).
 * value_params / width_params are the [3][6] f32 sets the host fit produced (prediction.rs:232-235);
 * group 0 = level 8, 1 = level 7, 2 = levels 1..6 (prediction.rs:165-179).
 * coefs is the whole [channels][F][512] array (quantised); `channel` selects the plane.
 * hist is overwritten. Symbols >= 1024 (Rust: index panic, entropy_coding.rs:99) are not
 * histogrammed; their count is returned in *n_out_of_alphabet. bucket/prediction may be NULL.
 * Any int32 coefficient array is accepted and gives what libfri computes for it (i32 gathers, f32 predictor): a fast kernel
 * whose LDS image holds magnitudes up to 256 - all the forward transform produces - is followed by an exact int32 kernel that
 * returns at once unless the fast one met a larger value. *n_out_of_alphabet == 0 means: libfri would have produced a stream. */
int fri_hip_predict_histogram(fri_hip_plan *plan, const int32_t *coefs, uint32_t channel, const float value_params[3][6],
                              const float width_params[3][6], uint8_t *bucket, int32_t *prediction, uint32_t *hist,
                              uint64_t *n_out_of_alphabet);
/* The caller's promise that the coefficient arrays it hands to fri_hip_predict_histogram_dev / _batch_dev on this plan are outputs of
 * fri_hip_transform_quant* (every magnitude <= 255; the image holds up to 256) - the situation of the replacement stage bodies, where prediction::encode always receives
 * what wavelet_transform::encode + quantization::encode produced: with on != 0 the exact int32 kernel behind the fast one is not enqueued
 * (one launch less, ~5 us). A broken promise is detected, not obeyed: the plane reports *n_out_of_alphabet == UINT64_MAX and an all-zero
 * histogram. Default: off (any int32 array accepted). */
int fri_hip_plan_assume_forward_coefficients(fri_hip_plan *plan, int on);
/* Device form: d_hist u32[10*1024] and d_n_out_of_alphabet u64[1] are overwritten. Both must be DEVICE memory (hipMalloc): since round 4 the kernel clears
 * them itself (its first workgroups, in their prologue) and adds its counts straight into them with device-scope atomics - there is no plan-side copy of the
 * table any more. The same holds for the d_hist / d_n_out_of_alphabet arguments of every _dev / _batch_dev entry point below. */
int fri_hip_predict_histogram_dev(fri_hip_plan *plan, const int32_t *d_coefs, uint32_t channel, const float value_params[3][6],
                                  const float width_params[3][6], uint8_t *d_bucket, int32_t *d_prediction, uint32_t *d_hist,
                                  uint64_t *d_n_out_of_alphabet, void *stream);

/* The same for n_planes planes (images x channels) in ONE launch - BASELINE config 3's batch of frames, or the channels of one image:
 * plane k reads d_coefs + k * coef_stride (int32 elements), takes its parameters from d_params[k] - a DEVICE array float[n_planes][2][3][6],
 * value set then width set - writes d_bucket / d_prediction + k * out_stride (elements; either may be NULL), d_hist[k][10][1024] and
 * d_n_out_of_alphabet[k]. Replaces the channel loop of prediction::encode (stages/prediction.rs:231) and, across images, the caller's loop. */
int fri_hip_predict_histogram_batch_dev(fri_hip_plan *plan, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, const float *d_params, uint8_t *d_bucket,
                                        int32_t *d_prediction, size_t out_stride, uint32_t *d_hist, uint64_t *d_n_out_of_alphabet, void *stream);

/* ---- context-model fit: normal-equation sums ------------------ */
/* The reference fits the 3 x 6 value and 3 x 6 width parameters of a channel by building n x 6 f32 design matrices
 * (ContextModeler::get_image_neighbour_matrices, context_modeling.rs:79-142) and running an SVD least squares on them
 * (lstsq, :168, :185). These entry points return the sums from which the same least-squares problems are solved as 6 x 6
 * systems on the host; the fitted parameters are transmitted in the file, so any solution gives a decodable stream
 * (the SVD's exact f32 output is third-party arithmetic: parity unpinned).
 * Layer groups g: 0 = level 8, 1 = level 7, 2 = levels 1..6 (matrices[0..2], :87-96). Rows exist for Some coefficients of
 * heap index >= 2 only; None rows are all zero in the reference (:109-134).
 * gram[g][28] = upper triangle (row major) of sum u u^T with u = [v0..v5, value], v = get_neighbour_values:
 *              A^T A = rows/columns 0..5, A^T b = column 6, b^T b = entry (6,6). Exact integers.
 * Precondition (all fit entry points): Some coefficients lie in [-256, 255], as every output of
 * fri_hip_transform_quant does (differences of 8-bit pixels, divided by a quantiser >= 1). The kernels stage them as
 * int16 and accumulate products of pairs of rows in 32-bit partial sums (v_dot2). The kernels check the range while staging:
 * the host-pointer forms, fri_hip_encode_image and fri_hip_predict_image (with fit) return FRI_HIP_ERR_OUT_OF_RANGE instead of
 * sums that overflowed; the _dev forms cannot report it (use fri_hip_predict_histogram's n_out_of_alphabet-style checks on the
 * host side, or the host forms, for coefficients of unknown origin). */
int fri_hip_fit_value_sums(fri_hip_plan *plan, const int32_t *coefs, uint32_t channel, int64_t gram[3][28]);
int fri_hip_fit_value_sums_dev(fri_hip_plan *plan, const int32_t *d_coefs, uint32_t channel, int64_t *d_gram, void *stream);
/* Width fit (optimize_width_prediction, :144-173) for given value parameters x: residual r = |f32(value) - A x| in f32
 * (left to right like nalgebra's gemv), features w = [1, |v0-v3|, |v1-v2|, |v4-v5|, |v1-v5|, |v2-v4|].
 * wtw[g][21] = upper triangle of sum w w^T over the Some rows (exact), wtr[g][6] = sum w r: the products of the (at most 16) nodes a lane
 * holds of one tile are summed in f32 in a fixed order, each such partial sum becomes a 64-bit fixed-point number (20 fraction bits) and everything
 * beyond that is integer addition - far inside the reference's own fit, whose matrices and SVD are f32 throughout (context_modeling.rs:144-173), and
 * REPRODUCIBLE: integer adds commute, so the sums (hence the fitted parameters, the buckets and the encoder's bytes) are the same bits in every
 * run, from every entry point and for any number of planes per launch (through round 3 they were f64 atomics in arrival order). Valid while a
 * plane's sum stays below 8.8e12 (a 16384 x 16384 noise plane: ~1.6e12) and every partial sum below 2^24: a partial sum that is not (value parameters
 * that are huge, infinite or NaN) is clamped and COUNTED with the out-of-range coefficients - the host forms return FRI_HIP_ERR_OUT_OF_RANGE, the
 * device forms report the count - so a meaningless W^T r never leaves silently. rows[g] = height of the reference's matrix (F*256, F*128, F*128): its
 * all-zero rows still carry the constant feature 1 with residual 0, so add rows[g] - wtw[g][0] to entry (0,0). */
int fri_hip_fit_width_sums(fri_hip_plan *plan, const int32_t *coefs, uint32_t channel, const float value_params[3][6], int64_t wtw[3][21],
                           double wtr[3][6], uint64_t rows[3]);
int fri_hip_fit_width_sums_dev(fri_hip_plan *plan, const int32_t *d_coefs, uint32_t channel, const float value_params[3][6], int64_t *d_wtw,
                               double *d_wtr, void *stream);

/* Batch forms, one launch for n_planes planes laid out as in fri_hip_predict_histogram_batch_dev: d_gram[n_planes][3][28];
 * d_params = DEVICE float[n_planes][2][3][6] of which the value sets are used; d_wtw[n_planes][3][21], d_wtr[n_planes][3][6]. */
int fri_hip_fit_value_sums_batch_dev(fri_hip_plan *plan, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, int64_t *d_gram, void *stream);
int fri_hip_fit_width_sums_batch_dev(fri_hip_plan *plan, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, const float *d_params, int64_t *d_wtw, double *d_wtr,
                                     void *stream);
/* The 6 x 6 solves behind the fit (host, pure functions): x = pinv(M) y for a symmetric positive semi-definite M - an LDL^T factorisation
 * when every pivot stays above 1e-8 of the largest diagonal entry (any image with texture in the layer group), otherwise a cyclic Jacobi
 * eigen-decomposition in which eigenvalues <= 1e-12 of the largest are dropped: the minimum-norm solution lstsq's SVD returns, up to
 * rounding; from the sums to the parameters of optimize_value_prediction (context_modeling.rs:175-202) and optimize_width_prediction
 * (:144-173; rows[g] = F * {256, 128, 128}, the reference's matrix heights). The device-side solves of fri_hip_fit_params_batch_dev and of
 * the encode chain run the same source (csrc/solve6.hpp) and return the same bits for the same sums. */
void fri_hip_solve6(const double m[6][6], const double y[6], double x[6]);
void fri_hip_fit_value_params(const int64_t gram[3][28], float value_params[3][6]);
void fri_hip_fit_width_params(const int64_t wtw[3][21], const double wtr[3][6], const uint64_t rows[3], float width_params[3][6]);

/* The same solves on the device, for sums that are in device memory (the *_sums_batch_dev layouts): one thread per (plane, layer group) writes
 * the value set (d_params[k][0][3][6]) resp. the width set (d_params[k][1][3][6], rows = F * {256, 128, 128} of this plan) of the DEVICE array
 * float[n_planes][2][3][6]. Enqueued on `stream`, no synchronisation. */
int fri_hip_fit_value_params_batch_dev(fri_hip_plan *plan, uint32_t n_planes, const int64_t *d_gram, float *d_params, void *stream);
int fri_hip_fit_width_params_batch_dev(fri_hip_plan *plan, uint32_t n_planes, const int64_t *d_wtw, const double *d_wtr, float *d_params, void *stream);

/* ContextModeler::optimize_parameters (context_modeling.rs:204-213, called at prediction.rs:232-235) for n_planes planes, entirely on the device
 * and asynchronously: value sums -> 6 x 6 solves -> width sums (with the value parameters just found) -> 6 x 6 solves, four kernels and two tiny
 * solve kernels on `stream`, no host round trip, no synchronisation. d_params = DEVICE float[n_planes][2][3][6] (value set, then width set, per
 * plane - the array fri_hip_predict_histogram_batch_dev reads) is overwritten. d_fit_out_of_range (DEVICE u64[n_planes], may be NULL): per plane
 * the number of waves that met a Some coefficient outside [-256, 255] (non-zero: that plane's parameters are not to be trusted; the forward
 * transform never produces one). */
int fri_hip_fit_params_batch_dev(fri_hip_plan *plan, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, float *d_params, uint64_t *d_fit_out_of_range,
                                 void *stream);

/* ---- the device part of FRIEncoder::encode in one call ---------------------------------------------- */
/* Replaces the stage chain of FRIEncoder::encode (encoder.rs:19-48) up to EncoderStage::EntropyEncoding for one image, all channels:
 * wavelet_transform::encode + quantization::encode (one kernel), then per channel ContextModeler::optimize_parameters
 * (prediction.rs:232-235; the sums and the 6 x 6 solves on the device) and the scan loop of prediction::encode
 * (:237-298) - with the coefficients staying in device memory between the stages, as the reference threads ONE WaveletImage through them.
 * fit != 0: the parameters are fitted and returned in value_params / width_params (float[channels][3][6] each); fit == 0: they are inputs.
 * Outputs: coefs [C][F][512], bucket / prediction [C][F][512] (may be NULL), hist [C][10][1024], n_out_of_alphabet [C].
 * The host form uploads the pixels once and downloads each output once. The device form enqueues the whole chain on `stream` (the fit's
 * 6 x 6 solves run on the device too) and returns without synchronising when fit == 0; with fit != 0 it returns once the fitted parameters have
 * arrived in value_params / width_params - the scan kernel is queued behind them and still running. One thread / one stream per plan at a time
 * for the fit forms (the parameters travel through plan-owned buffers). FRI_HIP_ERR_OUT_OF_RANGE from the host forms: see the fit entry points.
 * Inside these chains the scan kernel does not check what it stages: the forward kernel of the same call wrote the coefficients, differences of 8-bit pixels
 * divided by a quantiser of magnitude >= 1, i.e. magnitudes <= 255, which its 16-bit staging holds exactly (the UINT64_MAX report of
 * fri_hip_plan_assume_forward_coefficients exists for coefficients the CALLER vouches for, not here). */
int fri_hip_encode_image(fri_hip_plan *plan, const uint8_t *pixels, const int32_t qmatrix[32], int fit, float *value_params, float *width_params, int32_t *coefs,
                         uint8_t *bucket, int32_t *prediction, uint32_t *hist, uint64_t *n_out_of_alphabet);
int fri_hip_encode_image_dev(fri_hip_plan *plan, const uint8_t *d_pixels, const int32_t qmatrix[32], int fit, float *value_params, float *width_params, int32_t *d_coefs,
                             uint8_t *d_bucket, int32_t *d_prediction, uint32_t *d_hist, uint64_t *d_n_out_of_alphabet, void *stream);

/* The same for n_images images with everything - parameters included - in DEVICE memory: K1 over all images, then (fit != 0) the device-side fit of
 * fri_hip_fit_params_batch_dev over all n_images * channels planes, then K2 over all planes; no host round trip and no synchronisation anywhere
 * (the call only enqueues). NOT HIP-graph capturable: the scan's histogram hand-over numbers its launches on the host, a replayed launch would reuse a
 * number - every entry point that launches the scan returns FRI_HIP_ERR_INVALID_ARGUMENT on a stream that is being captured, instead of recording a graph
 * whose replays could lose or double counts. (The forward and inverse kernels alone - fri_hip_transform_quant*_dev, fri_hip_inverse_transform*_dev - can be captured.)
 * Image k: pixels at d_pixels + k * pixel_stride (bytes), coefficients at d_coefs + k * coef_stride (int32 elements, [C][F][512] inside), bucket /
 * prediction at + k * out_stride (elements; either may be NULL), d_hist[k][C][10][1024], d_n_out_of_alphabet[k][C], d_params[k][C][2][3][6]
 * (in when fit == 0, out when fit != 0), d_fit_out_of_range[k][C] (may be NULL). With channels == 3 and n_images > 1 the images must lie back to
 * back (coef_stride == 3 * F * 512 == out_stride): the planes of the batch are then evenly spaced and every stage is one launch.
 * Replaces the per-image loop around FRIEncoder::encode (crates/fri-cli/src/commands/bench.rs:15-120; BASELINE config 3). */
int fri_hip_encode_image_batch_dev(fri_hip_plan *plan, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride, const int32_t qmatrix[32], int fit, float *d_params,
                                   int32_t *d_coefs, size_t coef_stride, uint8_t *d_bucket, int32_t *d_prediction, size_t out_stride, uint32_t *d_hist,
                                   uint64_t *d_n_out_of_alphabet, uint64_t *d_fit_out_of_range, void *stream);

/* The reference's per-image loop itself (crates/fri-cli/src/commands/bench.rs:15-120: FRIEncoder::encode per image, encoder.rs:87-109) for images in
 * HOST memory: every image runs the asynchronous chain above on one of three internal streams with pinned staging, so uploads, kernels and
 * downloads of consecutive images overlap. Per image i: params[i] = float[C][2][3][6] (value set then width set per channel: in when fit == 0,
 * out when fit != 0), coefs[i] [C][F][512], bucket[i] / prediction[i] [C][F][512] (the arrays or single entries may be NULL), hist[i] [C][10][1024],
 * n_out_of_alphabet[i] [C]. Returns the first error; FRI_HIP_ERR_OUT_OF_RANGE as in fri_hip_encode_image.
 * fri_hip_multi_encode_image: the same over the GPUs of a fri_hip_multi, image i on devices[i mod n_devices] (fri_hip_shard_*), one host thread per
 * device, no data between devices - BASELINE config 4 for the whole encoder rather than its first stage. */
int fri_hip_encode_image_batch(fri_hip_plan *plan, uint32_t n_images, const uint8_t *const *pixels, const int32_t qmatrix[32], int fit, float *const *params,
                               int32_t *const *coefs, uint8_t *const *bucket, int32_t *const *prediction, uint32_t *const *hist, uint64_t *const *n_out_of_alphabet);
int fri_hip_multi_encode_image(fri_hip_multi *m, uint32_t n_images, const uint8_t *const *pixels, const int32_t qmatrix[32], int fit, float *const *params,
                               int32_t *const *coefs, uint8_t *const *bucket, int32_t *const *prediction, uint32_t *const *hist, uint64_t *const *n_out_of_alphabet);

/* prediction::encode alone (stages/prediction.rs:224-323 minus the host's ANS models) for all channels of an image whose coefficients
 * already exist: one upload of the coefficients (host form), optional fit, the scan of every channel in one launch. Same argument
 * meaning as fri_hip_encode_image; any int32 coefficients are accepted (see fri_hip_predict_histogram). */
int fri_hip_predict_image(fri_hip_plan *plan, const int32_t *coefs, int fit, float *value_params, float *width_params, uint8_t *bucket, int32_t *prediction,
                          uint32_t *hist, uint64_t *n_out_of_alphabet);
int fri_hip_predict_image_dev(fri_hip_plan *plan, const int32_t *d_coefs, int fit, float *value_params, float *width_params, uint8_t *d_bucket, int32_t *d_prediction,
                              uint32_t *d_hist, uint64_t *d_n_out_of_alphabet, void *stream);

/* ---- the ordered symbol stream: the emitter's gather on the device ------------------------------ */
/* The reference's emitter walks a channel's Some nodes in sort_lattice order (ten scans: DC, root, levels 1..8; stages/wavelet_transform.rs:657-705,
 * stages/entropy_coding.rs:285-336) and feeds pack_signed(value - prediction) with its context bucket to the rANS coder. That walk is a permutation
 * fixed by the geometry. fri_hip_plan_set_stream_order uploads it once per plan: order[i] = cell << 9 | heap index of the i-th symbol, n =
 * fri_hip_plan_num_some (fri_emit_stream_order of include/fri_emit.h builds it; the call checks that it is a permutation of the plan's Some nodes).
 * fri_hip_symbol_stream_batch_dev then writes, per plane k, d_symbols[k * symbol_stride + i] = bucket << 10 | symbol for i < num_some, from the
 * coefficient / bucket / prediction planes laid out as in fri_hip_predict_histogram_batch_dev: 2 bytes per symbol leave the device instead of the
 * 9 bytes per node of the three arrays, and the host emitter (fri_emit_encode_image_from_streams) is the pure rANS loop. A symbol >= 1024 cannot be
 * represented (the reference panics, entropy_coding.rs:99): emit only planes whose n_out_of_alphabet is 0. */
int fri_hip_plan_set_stream_order(fri_hip_plan *plan, const uint32_t *order, uint64_t n);
int fri_hip_symbol_stream_batch_dev(fri_hip_plan *plan, uint32_t n_planes, const int32_t *d_coefs, size_t coef_stride, const uint8_t *d_bucket, const int32_t *d_prediction,
                                    size_t out_stride, uint16_t *d_symbols, size_t symbol_stride, void *stream);

/* The asynchronous chain of fri_hip_encode_image_batch_dev all the way to the emitter's input, everything in device memory: forward transform ->
 * [fit] -> the scan kernel in its halfword form -> gather into stream order. The scan then writes ONE halfword per node, d_node_words[k][C][F][512] =
 * bucket << 10 | symbol (the index of the counter the node bumped; None nodes: unspecified; a symbol >= 1024: 10 << 10, "bucket 10" - such a plane has
 * n_out_of_alphabet != 0 and must not be emitted), and neither bucket nor prediction arrays: 2 bytes per node of stores instead of 5, and the gather
 * reads 2 bytes per symbol instead of 9. d_symbols[k][C][num_some] as in fri_hip_symbol_stream_batch_dev (image k at k * symbol_stride halfwords, a
 * channel's stream directly behind the previous channel's). With channels == 3 and n_images > 1 the images must lie back to back (coef_stride ==
 * word_stride == 3 * F * 512, symbol_stride == 3 * num_some). Other arguments as in fri_hip_encode_image_batch_dev. Needs fri_hip_plan_set_stream_order.
 * d_coefs == NULL (round 5): the caller does not want the coefficients - the emitter needs the streams, the histograms and the parameters only. They then travel
 * between the kernels as int16 planes the plan owns (every coefficient of the transform fits nine bits; None as 0, which is what the fit and the scan read a None
 * neighbour as): the forward kernel writes half the bytes, the fit and the scan read half - 4096 x 4096: 99 -> 90 us with given parameters, 166 -> 154 us with the
 * fit - and everything that comes back is the same bits (tests/test_gpu_compact.py). coef_stride is ignored then. The planes are sized by the largest call so far (a
 * growing call allocates, i.e. waits for the device) and shared by the plan's calls: a chain on another stream than the previous one waits (on the device, through an
 * event) until that one is through with them.
 * d_node_words == NULL as well (needs d_coefs == NULL): the caller wants the streams and nothing else. The scan then writes every symbol straight to its place in its
 * channel's stream - a table of stream positions, the inverse of the order, built by fri_hip_plan_set_stream_order - under the Some / None masks; no node-word
 * planes, no gather kernel: 4096 x 4096: 100 -> 70 us with given parameters, 166 -> 132 us with the fit, the same streams, histograms and parameters
 * (tests/test_gpu_compact.py). word_stride is ignored then. fri_hip_encode_image_symbols always works this way. */
int fri_hip_encode_symbols_batch_dev(fri_hip_plan *plan, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride, const int32_t qmatrix[32], int fit, float *d_params,
                                     int32_t *d_coefs, size_t coef_stride, uint16_t *d_node_words, size_t word_stride, uint16_t *d_symbols, size_t symbol_stride,
                                     uint32_t *d_hist, uint64_t *d_n_out_of_alphabet, uint64_t *d_fit_out_of_range, void *stream);

/* The device part of FRIEncoder::encode all the way to the emitter's input, for host buffers: the chain above for one image; what comes back is symbols[C][num_some] (2 bytes per symbol) + hist + parameters + n_out_of_alphabet - 17 MB up and 34 MB down per
 * 4096 x 4096 plane where fri_hip_encode_image moves 17 MB up and 153 MB down. Needs fri_hip_plan_set_stream_order. Argument meaning as fri_hip_encode_image. */
int fri_hip_encode_image_symbols(fri_hip_plan *plan, const uint8_t *pixels, const int32_t qmatrix[32], int fit, float *value_params, float *width_params, uint16_t *symbols,
                                 uint32_t *hist, uint64_t *n_out_of_alphabet);

/* ---- inverse: dequantisation + inverse transform (decode side) ------------------------------ */
/* Replaces quantization::decode (stages/quantization.rs:27-45) + wavelet_transform::decode
 * (stages/wavelet_transform.rs:715-717: RasterImage::from_wavelet :308-356, extract_values
 * :358-381, set_pixel clamp images.rs:103-111). NOTE the reference's decode *divides* by
 * qmatrix[layer] (quantization.rs:37) exactly like encode; this entry point reproduces that
 * (bit-exact, and the identity for today's all-ones matrix). Pixels not covered by any Some
 * coefficient are written 0 like the reference's zero-initialised raster. */
int fri_hip_inverse_transform(fri_hip_plan *plan, const int32_t *coefs, const int32_t qmatrix[32], uint8_t *pixels);
/* Which dequantiser the inverse entry points of this plan apply. FRI_HIP_DEQUANT_REFERENCE (default): quantization::decode as the reference has it
 * (stages/quantization.rs:27-45), which DIVIDES by the matrix entry like the encoder does - bit for bit the reference's decoder, and a defect of the
 * reference as soon as the matrix is not all ones. FRI_HIP_DEQUANT_MULTIPLY: the inverse of the quantiser, coefficient x
 * qmatrix[layer] in wrapping 32-bit arithmetic - what a lossy round trip needs. With today's all-ones matrix the two are the same kernel instance. */
#define FRI_HIP_DEQUANT_REFERENCE 0
#define FRI_HIP_DEQUANT_MULTIPLY 1
int fri_hip_plan_set_dequantiser(fri_hip_plan *plan, int mode);
int fri_hip_inverse_transform_dev(fri_hip_plan *plan, const int32_t *d_coefs, const int32_t qmatrix[32], uint8_t *d_pixels,
                                  void *stream);
/* n independent images in one launch: image k at d_coefs + k * coef_stride (int32 elements), d_pixels + k * pixel_stride (bytes). */
int fri_hip_inverse_transform_batch_dev(fri_hip_plan *plan, uint32_t n_images, const int32_t *d_coefs, size_t coef_stride, const int32_t qmatrix[32], uint8_t *d_pixels,
                                        size_t pixel_stride, void *stream);

/* ---- timing helper ---------------------------------------------------------------------------- */
/* Runs the forward kernel `iters` times on `stream` bracketed by HIP events recorded on that same
 * stream and returns the mean kernel-to-kernel time per launch in microseconds (bench.py uses it
 * for roofline.achieved). Buffers rotate over n_images image/coef slots of the batch layout. */
int fri_hip_time_transform_quant_dev(fri_hip_plan *plan, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride,
                                     const int32_t qmatrix[32], int32_t *d_coefs, size_t coef_stride, uint32_t iters, void *stream,
                                     double *mean_us);

/* The same loop with launch i on stream i mod n_streams of the library's own streams (1..8; the images are independent, crates/fri-cli/src/commands/bench.rs:15-120):
 * launch i + 1's workgroups move into the CUs launch i's early finishers leave. mean_us is the launch PERIOD (first begin to last end over iters), not a
 * kernel duration - with more than one stream the launches overlap. Synchronises the device before and after. */
int fri_hip_time_transform_quant_streams_dev(fri_hip_plan *plan, uint32_t n_images, const uint8_t *d_pixels, size_t pixel_stride,
                                             const int32_t qmatrix[32], int32_t *d_coefs, size_t coef_stride, uint32_t iters, uint32_t n_streams,
                                             double *mean_us);

/* ---- forward tiling by measurement ------------------------------------------------------------- */
/* How the cells of an image are cut into tiles and dealt to workgroups (Fractal::extract_coefficients is per-cell independent,
 * stages/wavelet_transform.rs:179-225: any partition gives the same coefficients) decides the forward kernel's speed by a few percent, and which
 * partition wins depends on the image size. fri_hip_plan_create picks a default that is good everywhere; this call MEASURES a handful of candidate
 * tilings on the plan's device (`launches` launches each, 0 = 96, in five interleaved rounds, on scratch buffers of its own - about 1 GB at 4096^2,
 * freed before it returns - large enough that every byte comes from HBM), keeps the fastest and remembers it for later plans of the same shape on the
 * same device in this process. Results never change; a plan whose tiling was pinned through the tuning environment, a host-only plan's, or one the
 * inverse kernel shares (>= 400 000 cells) is left alone. `report` (may be NULL): a JSON object with the candidates' microseconds per launch.
 * Blocks for tens of milliseconds; call it once after fri_hip_plan_create, outside anything that is timed. Not thread-safe per plan. */
int fri_hip_plan_tune_forward(fri_hip_plan *plan, uint32_t launches, char *report, size_t report_bytes);

/* The inverse kernel's static write-out lists (diagnostics / tests): out[5] = {built (0/1), whole 16-byte quads, whole dwords inside
 * partly owned quads, bytes owned inside partly owned dwords, LDS bytes of the largest tile rectangle}.
 * 16 * out[1] + 4 * out[2] + out[3] equals the number of bytes of the image that belong to a retained cell. */
int fri_hip_plan_inverse_lists(const fri_hip_plan *plan, uint64_t out[5]);

/* Diagnostic timeline (plans created with FRI_HIP_TRACE=1 in the environment; INVALID_ARGUMENT otherwise): copies the
 * record of the most recent forward or inverse launch, out[n_wg][16] = {entry, prologue done, tile 0 done, ... (12 slots),
 * hardware id, exit}, time stamps in ticks of the GPU's constant 100 MHz clock. Synchronises the device. */
int fri_hip_plan_read_trace(fri_hip_plan *plan, uint64_t *out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
