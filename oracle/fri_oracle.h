/*
 * fri_oracle.h -- CPU restatement of libfri's transform/quant/predict/histogram path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under frave_amd/ may include, link or call this.
 * Allowed callers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * PARITY UNPINNED: the reference (pagmerek/frave, Rust) ships no golden vectors or
 * asserting tests for this path and cannot be compiled in this image (no cargo/rustc),
 * so this restatement is pinned only by (a) the reference source text it cites line by
 * line, (b) the invariants the reference itself asserts, (c) the independently derived
 * known-answer hashes recorded in SURVEY.md section 8c (checked in tests/test_oracle_kat.py).
 *
 * All "file:line" citations are relative to /root/reference/crates/libfri/src/.
 */
#ifndef FRI_ORACLE_H
#define FRI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRI_ORACLE_NONE INT32_MIN /* wire encoding of Option::None in exported arrays */

typedef struct fri_oracle_wavelet fri_oracle_wavelet; /* WaveletImage, stages/wavelet_transform.rs:384-389 */

/* WaveletImage::from_raster (stages/wavelet_transform.rs:405-432) minus sort_lattice.
 * data is interleaved u8, index ((y*w+x)*C+c) (images.rs:94). channels is 1 or 3. */
fri_oracle_wavelet *fri_oracle_from_raster(const uint8_t *data, uint32_t height, uint32_t width,
                                           uint32_t channels);
void fri_oracle_free(fri_oracle_wavelet *w);
/* from_raster over a given set of cell centres ([n][2] = (re, im), duplicates taken once) instead of fractal_divide's BFS: per cell exactly
 * the reference's Fractal::new + extract_coefficients + retain rule + position map. For sampled checks of images whose whole lattice this
 * restatement cannot hold (config 5, 16384^2): pass a cell together with its lattice neighbourhood and fri_oracle_context_at gives the
 * full image's answer for the cell's nodes. */
fri_oracle_wavelet *fri_oracle_from_raster_cells(const uint8_t *data, uint32_t height, uint32_t width, uint32_t channels, const int32_t *centers,
                                                 uint32_t n_centers);
/* One cell on its own (the transform is per-cell independent, wavelet_transform.rs:179-225): out[channels][512], heap order,
 * None = FRI_ORACLE_NONE. Returns 1 if the retain rule keeps the cell, 0 if not, -1 on bad arguments. */
int fri_oracle_cell(const uint8_t *data, uint32_t height, uint32_t width, uint32_t channels, int32_t center_re, int32_t center_im, int32_t *out);

uint32_t fri_oracle_num_cells(const fri_oracle_wavelet *w);     /* after the retain() at :415-416 */
uint32_t fri_oracle_num_bfs_cells(const fri_oracle_wavelet *w); /* fractal_divide() output size */
uint32_t fri_oracle_channels(const fri_oracle_wavelet *w);

/* Retained cell centres in canonical order: ascending im, then re (utils.rs:17-32). out[F][2]=(re,im) */
void fri_oracle_centers(const fri_oracle_wavelet *w, int32_t *out);
/* Fractal.coefficients as [C][F][512] int32 in heap order, None = FRI_ORACLE_NONE. */
void fri_oracle_coefficients(const fri_oracle_wavelet *w, int32_t *out);
/* Overwrite coefficients from the same layout (to drive the inverse on arbitrary input). */
void fri_oracle_set_coefficients(fri_oracle_wavelet *w, const int32_t *in);

/* quantization::encode (stages/quantization.rs:7-25). Returns -1 if a used divisor is 0
 * (Rust panics on division by zero). */
int fri_oracle_quantize(fri_oracle_wavelet *w, const int32_t qmatrix[32]);

/* The bucket/prediction/histogram loop of prediction::encode (stages/prediction.rs:237-298) for one
 * channel, with the parameters passed in instead of fitted (prediction.rs:232-235 is out of scope).
 * hist is [10][1024] u32 and is ADDED to (caller zeroes). Symbols >= 1024 make the reference panic
 * (entropy_coding.rs:99); here they are counted in *n_out_of_alphabet and not histogrammed. */
int fri_oracle_predict(fri_oracle_wavelet *w, uint32_t channel, const float value_params[3][6],
                       const float width_params[3][6], uint32_t *hist, uint64_t *n_out_of_alphabet);
/* Fractal.parameter_predictors for a channel: bucket[F][512] u8, prediction[F][512] i32
 * (0,0 where never written, as initialised at wavelet_transform.rs:60-64). */
void fri_oracle_predictors(const fri_oracle_wavelet *w, uint32_t channel, uint8_t *bucket,
                           int32_t *prediction);
/* The decoder's view (entropy_coding::decode_symbol, entropy_coding.rs:205-236): (bucket, prediction) of one node computed from
 * the coefficients as they are at the time of the call; cell = index in canonical order. Returns -1 if the node is None. */
int fri_oracle_context_at(const fri_oracle_wavelet *w, uint32_t channel, uint32_t cell, uint32_t heap, const float value_params[3][6],
                          const float width_params[3][6], uint32_t *bucket, int32_t *prediction);
/* coefficients[channel][heap] = Some(value) of one cell (entropy_coding.rs:387, :409, :440) */
int fri_oracle_set_coefficient(fri_oracle_wavelet *w, uint32_t channel, uint32_t cell, uint32_t heap, int32_t value);
/* ContextModeler::get_neighbour_values (context_modeling.rs:25-77) for every level>=1 node:
 * out[F][512][6]; rows for heap index 0,1 are zero. */
void fri_oracle_neighbour_values(const fri_oracle_wavelet *w, uint32_t channel, int32_t *out);

/* RasterImage::from_wavelet / extract_values (stages/wavelet_transform.rs:308-381).
 * out is h*w*C bytes, zero-filled first like the reference. */
void fri_oracle_to_raster(const fri_oracle_wavelet *w, uint8_t *out);

/* sort_lattice / scan_level (stages/wavelet_transform.rs:505-705): symbol order per level.
 * Returns the number of positions written for `level` (reference asserts == F << level), or -1.
 * out[n][2] = (re,im); pass NULL to query the count only. */
int64_t fri_oracle_sorted_level(const fri_oracle_wavelet *w, uint32_t level, int32_t *out);

/* Small pieces exported for known-answer tests. */
int fri_oracle_pair(int l_some, int32_t l, int r_some, int32_t r, int32_t *d, int32_t *s); /* :211-218 */
void fri_oracle_nearby_vectors(uint32_t depth, int32_t out[6][2]);                          /* :71-90 */
void fri_oracle_literal(uint32_t i, int32_t out[2]);                                        /* fractal.rs:51-86 */
uint32_t fri_oracle_assign_bucket(float width);                                             /* prediction.rs:55-68 */
uint32_t fri_oracle_pack_signed(int32_t k);                                                 /* utils.rs:34-40 */
int32_t fri_oracle_unpack_signed(uint32_t k);                                               /* utils.rs:42-48 */
uint32_t fri_oracle_quant_layer(uint32_t i);                                                /* quantization.rs:13 */

#ifdef __cplusplus
}
#endif
#endif
