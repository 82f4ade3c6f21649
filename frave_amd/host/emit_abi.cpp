// emit_abi.cpp -- the C entry points include/fri_emit.h declares, over emit.hpp (libfri_emit.so): what the CPU test-suite and a
// foreign-language host bind. Plain pointers and sizes; every function returns 0 or a negative code and writes a message into `err`.
#include <cstdio>
#include <cstring>

#include "emit.hpp"
#include "fri_emit.h"

using namespace libfri::emit;

namespace {
int fail(char *err, size_t cap, const std::string &msg, int code = -1) {
    if (err && cap) std::snprintf(err, cap, "%s", msg.c_str());
    return code;
}
} // namespace

extern "C" {
#pragma GCC visibility push(default)

// out[n_cells << level] = cell << 9 | heap index, in stream order (level 0: heap index 1)
int fri_emit_symbol_order(const int32_t *centers_re_im, uint32_t n_cells, uint32_t level, uint32_t *out) {
    if (!centers_re_im || !out || level >= (uint32_t)kDepth) return -1;
    const std::vector<uint32_t> v = symbol_order(centers_re_im, n_cells, (int)level);
    std::memcpy(out, v.data(), v.size() * sizeof(uint32_t));
    return 0;
}

// freqs: in = the measured counts of one context, out = the model; cdf/off/n_off/max_freq_bits: outputs
int fri_emit_finalize_context(uint32_t freqs[1024], uint32_t bucket, uint32_t cdf[1024], uint16_t off[1024], uint32_t *n_off, uint32_t *max_freq_bits, char *err,
                              size_t err_cap) {
    if (!freqs || !cdf || !off || !n_off || !max_freq_bits || bucket >= (uint32_t)kContexts) return -1;
    AnsContext c;
    uint64_t sum = 0;
    for (int j = 0; j < kAlphabet; j++) {
        c.freqs[j] = freqs[j];
        sum = (uint32_t)(sum + freqs[j]);
    }
    uint64_t n = sum;
    n |= n >> 1, n |= n >> 2, n |= n >> 4, n |= n >> 8, n |= n >> 16;
    n ^= n >> 1;
    c.max_freq_bits = n ? (uint32_t)__builtin_ctzll(n) : 64u; // prediction.rs:302-303
    const std::string e = c.finalize((int)bucket);
    if (!e.empty()) return fail(err, err_cap, e, -2);
    std::memcpy(freqs, c.freqs.data(), sizeof(uint32_t) * kAlphabet);
    std::memcpy(cdf, c.cdf.data(), sizeof(uint32_t) * kAlphabet);
    *n_off = (uint32_t)c.off_distribution_values.size();
    std::memcpy(off, c.off_distribution_values.data(), c.off_distribution_values.size() * sizeof(uint16_t));
    *max_freq_bits = c.max_freq_bits;
    return 0;
}

// symbols/buckets: capacity n_cells * 512 each; *n = symbols written (stream order)
int fri_emit_channel_symbols(const int32_t *centers_re_im, uint32_t n_cells, const int32_t *coefs, const uint8_t *bucket, const int32_t *prediction,
                             uint16_t *symbols, uint8_t *buckets, uint64_t *n) {
    if (!centers_re_im || !coefs || !bucket || !prediction || !symbols || !buckets || !n) return -1;
    std::vector<uint16_t> s;
    std::vector<uint8_t> b;
    channel_symbols(*shared_symbol_order(centers_re_im, n_cells), coefs, bucket, prediction, s, b);
    std::memcpy(symbols, s.data(), s.size() * sizeof(uint16_t));
    std::memcpy(buckets, b.data(), b.size());
    *n = s.size();
    return 0;
}

// The whole .frv: coefs/bucket/prediction are [channels][n_cells][512], hist [channels][10][1024], params [channels][3][6].
// Returns 0 and *len; -3 if `cap` is too small (*len = needed size).
int fri_emit_encode_image(uint32_t width, uint32_t height, uint32_t channels, const int32_t *centers_re_im, uint32_t n_cells, const int32_t *coefs,
                          const uint8_t *bucket, const int32_t *prediction, const uint32_t *hist, const float *value_params, const float *width_params, uint8_t *out,
                          size_t cap, size_t *len, char *err, size_t err_cap) {
    if (!centers_re_im || !coefs || !bucket || !prediction || !hist || !value_params || !width_params || !len || (channels != 1 && channels != 3))
        return fail(err, err_cap, "invalid argument");
    std::vector<ChannelStream> streams;
    std::vector<ChannelParams> params(channels);
    const auto order_ptr = shared_symbol_order(centers_re_im, n_cells); // geometry only: once for all channels, cached per image size
    const SymbolOrder &order = *order_ptr;
    const std::string e = encode_channels(order, channels, coefs, bucket, prediction, hist, streams);
    if (!e.empty()) return fail(err, err_cap, e, -2);
    for (uint32_t ch = 0; ch < channels; ch++) {
        std::memcpy(params[ch].value, value_params + (size_t)ch * 18, sizeof(params[ch].value));
        std::memcpy(params[ch].width, width_params + (size_t)ch * 18, sizeof(params[ch].width));
    }
    const std::vector<uint8_t> bytes = serialize(height, width, channels == 1 ? kLuma : kRGB, streams, params);
    *len = bytes.size();
    if (!out || cap < bytes.size()) return -3;
    std::memcpy(out, bytes.data(), bytes.size());
    return 0;
}

// The stream order without the None nodes: out[*n] = cell << 9 | heap index of the i-th symbol of a channel. Capacity n_cells * 512.
int fri_emit_stream_order(const int32_t *centers_re_im, uint32_t n_cells, const uint32_t *valid_mask, uint32_t *out, uint64_t *n) {
    if (!centers_re_im || !valid_mask || !out || !n) return -1;
    const std::vector<uint32_t> v = stream_order(*shared_symbol_order(centers_re_im, n_cells), valid_mask);
    std::memcpy(out, v.data(), v.size() * sizeof(uint32_t));
    *n = v.size();
    return 0;
}

// The whole .frv from the device's symbol streams: streams [channels][n_symbols] u16 = bucket << 10 | symbol in stream order.
int fri_emit_encode_image_from_streams(uint32_t width, uint32_t height, uint32_t channels, const uint16_t *streams, uint64_t n_symbols, const uint32_t *hist,
                                       const float *value_params, const float *width_params, uint8_t *out, size_t cap, size_t *len, char *err, size_t err_cap) {
    if (!streams || !hist || !value_params || !width_params || !len || (channels != 1 && channels != 3)) return fail(err, err_cap, "invalid argument");
    std::vector<ChannelStream> chans;
    std::vector<ChannelParams> params(channels);
    const std::string e = encode_channels_from_streams(channels, streams, (size_t)n_symbols, hist, chans);
    if (!e.empty()) return fail(err, err_cap, e, -2);
    for (uint32_t ch = 0; ch < channels; ch++) {
        std::memcpy(params[ch].value, value_params + (size_t)ch * 18, sizeof(params[ch].value));
        std::memcpy(params[ch].width, width_params + (size_t)ch * 18, sizeof(params[ch].width));
    }
    const std::vector<uint8_t> bytes = serialize(height, width, channels == 1 ? kLuma : kRGB, chans, params);
    *len = bytes.size();
    if (!out || cap < bytes.size()) return -3;
    std::memcpy(out, bytes.data(), bytes.size());
    return 0;
}

// Entropy-layer self-check of a .frv against the arrays it was made from: parse the container, rebuild every context from its
// two serialised fields, decode all symbols with the known bucket sequence and compare. 0 = identical.
int fri_emit_check_image(const uint8_t *frv, size_t len, uint32_t channels, const int32_t *centers_re_im, uint32_t n_cells, const int32_t *coefs,
                         const uint8_t *bucket, const int32_t *prediction, char *err, size_t err_cap) {
    if (!frv || !centers_re_im || !coefs || !bucket || !prediction) return fail(err, err_cap, "invalid argument");
    ParsedImage img;
    std::string e = deserialize(std::vector<uint8_t>(frv, frv + len), img);
    if (!e.empty()) return fail(err, err_cap, e, -2);
    if (img.channels.size() != channels) return fail(err, err_cap, "channel count", -2);
    const size_t plane = (size_t)n_cells * kNodes;
    const auto order_ptr = shared_symbol_order(centers_re_im, n_cells);
    const SymbolOrder &order = *order_ptr;
    for (uint32_t ch = 0; ch < channels; ch++) {
        std::vector<uint16_t> want, got;
        std::vector<uint8_t> buckets;
        channel_symbols(order, coefs + ch * plane, bucket + ch * plane, prediction + ch * plane, want, buckets);
        e = decode_symbols(img.channels[ch], buckets, got);
        if (!e.empty()) return fail(err, err_cap, "channel " + std::to_string(ch) + ": " + e, -2);
        if (got != want) return fail(err, err_cap, "channel " + std::to_string(ch) + ": decoded symbols differ", -4);
    }
    return 0;
}

// The context-parallel rANS coder against the plain one-loop coder on pseudo-random symbols (see fri_emit.h).
int fri_emit_rans_selfcheck(uint64_t n_symbols, uint64_t seed, char *err, size_t err_cap) {
    std::string e;
    const int rc = rans_selfcheck(n_symbols, seed, e);
    return rc == 0 ? 0 : fail(err, err_cap, e, rc);
}

// A .frv back to coefficient planes. info = {width, height, channels, n_cells}; coefs: [channels][n_cells][512] (None = INT32_MIN),
// centers: [n_cells][2] or null. Returns -3 with `info` filled if coef_cap (in elements) is too small: call once with coef_cap = 0.
int fri_emit_decode_image(const uint8_t *frv, size_t len, uint32_t info[4], int32_t *coefs, size_t coef_cap, int32_t *centers, char *err, size_t err_cap) {
    if (!frv || !info) return fail(err, err_cap, "invalid argument");
    if (!coefs || coef_cap == 0) { // size query: header + geometry only
        ParsedImage img;
        const std::string e = deserialize(std::vector<uint8_t>(frv, frv + len), img);
        if (!e.empty()) return fail(err, err_cap, e, -2);
        const uint32_t channels = img.colorspace == kLuma ? 1u : 3u;
        uint32_t n_cells = 0;
        const std::string ge = count_cells(img.width, img.height, channels, n_cells);
        if (!ge.empty()) return fail(err, err_cap, ge, -2);
        info[0] = img.width, info[1] = img.height, info[2] = channels, info[3] = n_cells;
        return -3;
    }
    DecodedImage d;
    const std::string e = decode_image(std::vector<uint8_t>(frv, frv + len), d);
    if (!e.empty()) return fail(err, err_cap, e, -2);
    info[0] = d.width, info[1] = d.height, info[2] = d.channels, info[3] = d.n_cells;
    if (coef_cap < d.coefs.size()) return -3;
    std::memcpy(coefs, d.coefs.data(), d.coefs.size() * sizeof(int32_t));
    if (centers) std::memcpy(centers, d.centers.data(), d.centers.size() * sizeof(int32_t));
    return 0;
}

#pragma GCC visibility pop
} // extern "C"
