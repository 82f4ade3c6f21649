/* fri_emit.h -- C ABI of the host stages behind the kernels (frave_amd/libfri_emit.so).
 *
 * Pure host code (no HIP, no GPU): what sits between the arrays include/fri_hip.h produces and a `.frv` file, and back.
 * Citations are relative to /root/reference/crates/libfri/src/. In the reference these replace
 *   WaveletImage::sort_lattice / scan_level      stages/wavelet_transform.rs:505-705   (fri_emit_symbol_order)
 *   AnsContext::finalize_context                 stages/entropy_coding.rs:82-175       (fri_emit_finalize_context)
 *   entropy_coding::encode + serialize::encode   stages/entropy_coding.rs:266-352, stages/serialize.rs:49-117
 *                                                                                      (fri_emit_encode_image)
 *   serialize::decode + entropy_coding::decode   stages/serialize.rs:119-268, stages/entropy_coding.rs:205-264, :352-443
 *                                                                                      (fri_emit_decode_image)
 * PARITY UNPINNED for the byte stream: the reference's rANS coder is the third-party crate `rans` 0.2.x (ryg_rans' rans64),
 * whose source is not part of the reference tree; see frave_amd/host/emit.hpp.
 *
 * Conventions: plain pointers and sizes, caller-owned buffers; every function returns 0 or a negative code
 * (-1 invalid argument, -2 the condition under which libfri would panic or report an error: message in `err`,
 *  -3 output buffer too small: the needed size is reported, -4 self-check mismatch). Cells are in the canonical order of
 * fri_hip_plan_centers; planes are [channels][n_cells][512] in heap order with None = INT32_MIN. */
#ifndef FRI_EMIT_H
#define FRI_EMIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* Stream order of the nodes of `level` (0..8) over all cells: out[n_cells << level] = cell << 9 | heap index
 * (level 0: heap index 1; the reference walks that list twice, for the DC and for the root, entropy_coding.rs:285-308). */
int fri_emit_symbol_order(const int32_t *centers_re_im, uint32_t n_cells, uint32_t level, uint32_t *out);

/* One ANS context. freqs: in = the counts fri_hip_predict_histogram measured for `bucket`, out = the Laplace model the coder
 * uses; cdf[1024], off[<= 1024] (off_distribution_values), *n_off, *max_freq_bits: outputs. An empty context is an error
 * (libfri divides by zero, entropy_coding.rs:123). */
int fri_emit_finalize_context(uint32_t freqs[1024], uint32_t bucket, uint32_t cdf[1024], uint16_t off[1024], uint32_t *n_off, uint32_t *max_freq_bits,
                              char *err, size_t err_cap);

/* The (symbol, bucket) sequence of one channel in stream order, None nodes skipped: what encode feeds to the coder.
 * symbols / buckets: capacity n_cells * 512; *n = entries written. */
int fri_emit_channel_symbols(const int32_t *centers_re_im, uint32_t n_cells, const int32_t *coefs, const uint8_t *bucket, const int32_t *prediction,
                             uint16_t *symbols, uint8_t *buckets, uint64_t *n);

/* The whole `.frv`. bucket / prediction: what fri_hip_predict_histogram wrote, hist: [channels][10][1024], params: the
 * [channels][3][6] f32 predictor parameters that were used (they are transmitted). Returns 0 and *len, or -3 with *len = needed size. */
int fri_emit_encode_image(uint32_t width, uint32_t height, uint32_t channels, const int32_t *centers_re_im, uint32_t n_cells, const int32_t *coefs,
                          const uint8_t *bucket, const int32_t *prediction, const uint32_t *hist, const float *value_params, const float *width_params,
                          uint8_t *out, size_t cap, size_t *len, char *err, size_t err_cap);

/* The symbol stream route (fri_hip_symbol_stream_batch_dev, include/fri_hip.h): the gather of (symbol, bucket) in sort_lattice order
 * (entropy_coding.rs:285-336 over wavelet_transform.rs:657-705) runs on the device, the host receives 2 bytes per symbol.
 * fri_emit_stream_order: the stream order with the None nodes taken out - out[i] = cell << 9 | heap index of the i-th symbol of a channel (DC scan,
 * root scan, levels 1..8); valid_mask = fri_hip_plan_valid_mask; capacity n_cells * 512, *n = fri_hip_plan_num_some. Geometry only: once per plan.
 * fri_emit_encode_image_from_streams: the whole `.frv` from streams [channels][n_symbols] u16 = bucket << 10 | symbol, byte for byte what
 * fri_emit_encode_image makes of the arrays the streams were gathered from. */
int fri_emit_stream_order(const int32_t *centers_re_im, uint32_t n_cells, const uint32_t *valid_mask, uint32_t *out, uint64_t *n);
int fri_emit_encode_image_from_streams(uint32_t width, uint32_t height, uint32_t channels, const uint16_t *streams, uint64_t n_symbols, const uint32_t *hist,
                                       const float *value_params, const float *width_params, uint8_t *out, size_t cap, size_t *len, char *err, size_t err_cap);

/* Entropy-layer self-check of a `.frv` against the arrays it was made from (parse, rebuild the models, decode every symbol
 * with the known bucket sequence, compare). */
int fri_emit_check_image(const uint8_t *frv, size_t len, uint32_t channels, const int32_t *centers_re_im, uint32_t n_cells, const int32_t *coefs,
                         const uint8_t *bucket, const int32_t *prediction, char *err, size_t err_cap);

/* A `.frv` back to the coefficient planes fri_hip_inverse_transform takes: the decoder knows only the file, it rebuilds the
 * geometry from width x height and recomputes every symbol's context from the coefficients decoded before it.
 * info = {width, height, channels, n_cells}; centers: [n_cells][2] or NULL. Returns -3 with `info` filled if coef_cap
 * (in elements) is too small: call once with coefs = NULL to size the buffer. */
int fri_emit_decode_image(const uint8_t *frv, size_t len, uint32_t info[4], int32_t *coefs, size_t coef_cap, int32_t *centers, char *err, size_t err_cap);

/* Self-check of the rANS stage: n_symbols pseudo-random symbols (seed) over ten contexts with finalised random models, coded once by the
 * plain one-loop coder (the reference's order of operations, entropy_coding.rs:332-347) and once by the library's context-parallel
 * coder; 0 = the two streams are byte-identical, -4 = they differ, -1 = invalid argument. Host only. */
int fri_emit_rans_selfcheck(uint64_t n_symbols, uint64_t seed, char *err, size_t err_cap);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
