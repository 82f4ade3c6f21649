// emit.hpp -- host side of libfri's encode path behind the device kernels:
// symbol order (sort_lattice, stages/wavelet_transform.rs:505-705), ANS model construction
// (AnsContext::finalize_context, stages/entropy_coding.rs:82-175), the ten interleaved rANS streams of a channel
// (entropy_coding::encode, :266-352) and the `frif` container (stages/serialize.rs:40-117).
//
// Pure host code: inputs are the arrays the C ABI of include/fri_hip.h produces (centres, coefficients, bucket,
// prediction, histogram). Nothing here touches the GPU, so the CPU test-suite exercises it against the oracle.
//
// PARITY UNPINNED for the byte stream: the rANS coder of the reference is the third-party crate `rans` (0.2.x, a
// binding of ryg_rans' 64-bit coder) whose source is not part of the reference tree, and f32 `exp` is the platform's
// libm. What is restated here is ryg_rans' published rans64 algorithm plus the call pattern of entropy_coding.rs; the
// word order of flush/init is inferred from the reference's decoder index `CONTEXT_AMOUNT - bucket - 1` (:239).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace fri {
struct Geometry;
}

namespace libfri {
namespace emit {

constexpr int kDepth = 9, kNodes = 512;
constexpr int kContexts = 10;      // CONTEXT_AMOUNT, stages/prediction.rs
constexpr int kAlphabet = 1024;    // ALPHABET_SIZE, stages/entropy_coding.rs:25
constexpr int32_t kNone = INT32_MIN;

// ---- symbol order ---------------------------------------------------------------------------------------------
// Stream order of the nodes of `level` (0..8) over all retained cells: entry = cell << 9 | heap index, heap index in
// [2^level, 2^(level+1)). Level 0 lists the cells themselves (heap index 1); the reference walks that list twice, for the
// DC (heap 0) and for the root (heap 1) (entropy_coding.rs:285-308).
// The reference finds the order by walking the level's lattice row by row (scan_level). The walk visits the lines parallel
// to nearby(9 - level)[1] one after the other in the direction of nearby(9 - level)[3], each line front to back, so the order
// is the sort by (n . p, col . p) with n normal to col; tests/test_emit.py checks that against the oracle's literal walk.
std::vector<uint32_t> symbol_order(const int32_t *centers_re_im, uint32_t n_cells, int level);
// All nine levels at once: geometry only, so a caller encoding many frames of one size builds it once.
struct SymbolOrder {
    std::vector<uint32_t> level[kDepth];
    SymbolOrder() = default;
    SymbolOrder(const int32_t *centers_re_im, uint32_t n_cells);
};
// The order of this set of centres from a small process-wide cache (built on first use): what encoders and decoders of many
// images of one size share.
std::shared_ptr<const SymbolOrder> shared_symbol_order(const int32_t *centers_re_im, uint32_t n_cells);

// ---- ANS model --------------------------------------------------------------------------------------------------
struct AnsContext {
    std::array<uint32_t, kAlphabet> freqs{};
    std::array<uint32_t, kAlphabet> cdf{};
    std::vector<uint16_t> off_distribution_values;
    uint32_t max_freq_bits = 0;
    // prediction.rs:302-305 + entropy_coding.rs:102-117. Returns "" or the reason libfri would panic.
    std::string finalize(int bucket);
};
float width_from_bucket(int bucket);                       // prediction.rs:70-84
uint32_t pack_signed(int32_t k);                           // utils.rs:34-40
int32_t unpack_signed(uint32_t k);                         // utils.rs:42-48

// ---- rANS: kContexts interleaved 64-bit states, one backwards-growing word stream (ryg_rans rans64) ---------------
class RansEncoderMulti {
  public:
    void put_at(int state, uint32_t start, uint32_t freq, uint32_t scale_bits);
    // the same step with the division precomputed per (model, symbol): see make_symbol in emit.cpp
    struct EncSymbol {
        uint64_t x_max, rcp_freq;
        uint32_t freq, bias, cmpl_freq, rcp_shift;
    };
    static EncSymbol make_symbol(uint32_t start, uint32_t freq, uint32_t scale_bits);
    void put_symbol(int state, const EncSymbol &s) {
        uint64_t x = x_[state];
        if (x >= s.x_max) {
            rev_.push_back((uint32_t)x);
            x >>= 32;
        }
        const uint64_t q = (uint64_t)(((unsigned __int128)x * s.rcp_freq) >> 64) >> s.rcp_shift;
        x_[state] = x + s.bias + q * s.cmpl_freq;
    }
    void reserve(size_t n_symbols) { rev_.reserve(n_symbols / 2 + 64); }
    void flush_all();
    std::vector<uint8_t> data() const; // little-endian words, first word of the stream first
  private:
    std::vector<uint32_t> rev_; // words in reverse stream order
    uint64_t x_[kContexts] = {1ull << 31, 1ull << 31, 1ull << 31, 1ull << 31, 1ull << 31, 1ull << 31, 1ull << 31, 1ull << 31, 1ull << 31, 1ull << 31};
};
class RansDecoderMulti {
  public:
    explicit RansDecoderMulti(const std::vector<uint8_t> &data);
    uint32_t get_at(int state, uint32_t scale_bits) const;
    void advance_at(int state, uint32_t start, uint32_t freq, uint32_t scale_bits);
    bool ok() const { return ok_; }
  private:
    std::vector<uint32_t> w_;
    size_t pos_ = 0;
    uint64_t x_[kContexts];
    bool ok_ = true;
};

// ---- one channel -------------------------------------------------------------------------------------------------
struct ChannelStream {
    std::array<AnsContext, kContexts> contexts;
    std::vector<uint8_t> data;
    uint64_t n_symbols = 0;
};
// coefs / bucket / prediction: this channel's [n_cells][512] planes; hist: [10][1024] counts of K2.
// Returns "" or an error (conditions under which the reference panics).
std::string encode_channel(const SymbolOrder &order, const int32_t *coefs, const uint8_t *bucket, const int32_t *prediction, const uint32_t *hist,
                           ChannelStream &out);
// All channels of an image, one thread per channel (they only share the read-only order). planes: [channels][n_cells][512],
// hist: [channels][10][1024]. Returns "" or "channel c: reason".
// the rANS stream of a channel's symbols (fed in reverse); sequential = the plain one-loop coder (the check of the per-context one)
std::string encode_symbols(const std::vector<uint16_t> &symbols, const std::vector<uint8_t> &buckets, const std::vector<RansEncoderMulti::EncSymbol> &tab,
                           std::vector<uint8_t> &data, bool sequential = false);
int rans_selfcheck(uint64_t n_symbols, uint64_t seed, std::string &err); // fri_emit_rans_selfcheck, fri_emit.h
std::string encode_channels(const SymbolOrder &order, uint32_t channels, const int32_t *coefs, const uint8_t *bucket, const int32_t *prediction,
                            const uint32_t *hist, std::vector<ChannelStream> &out);
// The device-side symbol stream (fri_hip_symbol_stream_batch_dev): stream_order = the stream order with the None nodes taken out (cell << 9 | heap
// per symbol, geometry only: uploaded once per plan); a channel's stream is then stream[i] = bucket << 10 | symbol, 2 bytes per symbol instead of
// the 9 bytes per node of (coefficient, prediction, bucket), and the emitter is the pure rANS loop.
std::vector<uint32_t> stream_order(const SymbolOrder &order, const uint32_t *valid_mask /* [n_cells][16] */);
std::string encode_channel_from_stream(const uint16_t *stream, size_t n_symbols, const uint32_t *hist, ChannelStream &out);
std::string encode_channels_from_streams(uint32_t channels, const uint16_t *streams /* [channels][n_symbols] */, size_t n_symbols, const uint32_t *hist,
                                         std::vector<ChannelStream> &out);
// The (symbol, bucket) sequence in stream order (what encode_channel feeds to the coder); for self-checks.
void channel_symbols(const SymbolOrder &order, const int32_t *coefs, const uint8_t *bucket, const int32_t *prediction, std::vector<uint16_t> &symbols,
                     std::vector<uint8_t> &buckets);
// Entropy-layer inverse for self-checks: given the bucket of every symbol in stream order, recover the symbols.
std::string decode_symbols(const ChannelStream &s, const std::vector<uint8_t> &buckets, std::vector<uint16_t> &symbols);

// ---- container ---------------------------------------------------------------------------------------------------
struct ChannelParams {
    float value[3][6];
    float width[3][6];
};
enum ColorSpaceCode : uint32_t { kLuma = 1, kRGB = 2, kYCbCr = 3 }; // images.rs:23-29
std::vector<uint8_t> serialize(uint32_t height, uint32_t width, ColorSpaceCode cs, const std::vector<ChannelStream> &channels,
                               const std::vector<ChannelParams> &params);
struct ParsedImage {
    uint32_t height = 0, width = 0, colorspace = 0, variant = 0;
    std::vector<ChannelStream> channels; // contexts rebuilt from (max_freq_bits, off_distribution_values) like serialize.rs:214-237
    std::vector<ChannelParams> params;
};
std::string deserialize(const std::vector<uint8_t> &bytes, ParsedImage &out);

// ---- decoder -------------------------------------------------------------------------------------------------------
// entropy_coding::decode (:352-443) for one channel: the symbols come back in stream order, each one's context (bucket and
// prediction) computed from the coefficients decoded before it, exactly as prediction.rs:86-207 computes them on the encoder
// side from the complete image - which is the same thing iff the stream order is causal (left / up_left / up_right of a node
// precede it, the level above is complete). Inherently sequential per channel; the channels run on threads of their own.
// coefs: [n_cells][512] output in heap order, None = INT32_MIN.
std::string decode_channel(const fri::Geometry &g, const SymbolOrder &order, const ChannelStream &s, const ChannelParams &p, int32_t *coefs);
struct DecodedImage {
    uint32_t height = 0, width = 0, colorspace = 0, channels = 0, n_cells = 0;
    std::vector<int32_t> centers;     // [n_cells][2], canonical order (the one fri_hip_plan_centers reports)
    std::vector<int32_t> coefs;       // [channels][n_cells][512]: what fri_hip_inverse_transform takes
    std::vector<ChannelParams> params;
};
// serialize::decode + entropy_coding::decode: a .frv back to the coefficient planes (the geometry is rebuilt from width x height)
std::string decode_image(const std::vector<uint8_t> &frv, DecodedImage &out);
std::string decode_parsed(const ParsedImage &img, DecodedImage &out); // the entropy_coding::decode half of it
std::string count_cells(uint32_t width, uint32_t height, uint32_t channels, uint32_t &n_cells); // retained cells of a width x height image

} // namespace emit
} // namespace libfri
