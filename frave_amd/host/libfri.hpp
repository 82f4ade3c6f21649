// libfri.hpp -- C++ mirror of the part of libfri's interface that sits on the hot path, implemented on top of the C ABI
// (include/fri_hip.h). Same names, argument order and error behaviour as the Rust (paths relative to
// /root/reference/crates/libfri/src/):
//   FRIEncoder::new(opts).encode(data, height, width, colorspace)   encoder.rs:82-109   (note: height before width)
//   FRIDecoder{}.decode(..)                                         decoder.rs:47-59
//   stages::wavelet_transform::{encode,decode}                      stages/wavelet_transform.rs:708-717
//   stages::quantization::{encode,decode}                           stages/quantization.rs:7-45
//   stages::prediction::encode                                      stages/prediction.rs:224-323 (scan loops only)
//   stages::entropy_coding::encode + stages::serialize::encode      stages/entropy_coding.rs:266-352, stages/serialize.rs:49-117
// FRIEncoder::encode runs the device stages as ONE call (fri_hip_encode_image: the coefficients stay in device memory from the transform
// to the histogram, one upload of the pixels, one download per output) and
// stops at the state the reference calls EncoderStage::EntropyEncoding(WaveletImage, contexts) (encoder.rs:38);
// `encode_bytes` runs the two host stages behind it (emit.hpp: symbol order, ANS model, rANS, `frif` container) and returns
// what the reference's FRIEncoder::encode returns, the file bytes.
// Errors come back as Result<T>{ok,error} with the reference's "Failed to decode: " prefix (sic, encoder.rs:106).
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "emit.hpp"
#include "fri_hip.h"

namespace libfri {

enum class ColorSpace { Luma, YCbCr, RGB }; // images.rs:8-21
inline uint32_t num_channels(ColorSpace c) { return c == ColorSpace::Luma ? 1u : 3u; }

struct ImageMetadata { // images.rs:68-79
    uint32_t height = 0, width = 0;
    ColorSpace colorspace = ColorSpace::RGB;
};
struct RasterImage { // images.rs:82-85
    ImageMetadata metadata;
    std::vector<uint8_t> data;
};

constexpr int CONTEXT_AMOUNT = FRI_HIP_CONTEXT_AMOUNT; // prediction.rs:15
constexpr int ALPHABET_SIZE = FRI_HIP_ALPHABET_SIZE;   // entropy_coding.rs:25
struct AnsContext {                                     // entropy_coding.rs:32-40, the part the device fills
    std::array<uint32_t, ALPHABET_SIZE> freqs{};
};

using PredictionParams = std::array<std::array<float, 6>, 3>; // Vec<[f32; 6]> with 3 layer groups (prediction.rs:165-179)

struct EncoderOpts { // encoder.rs:58-64
    bool emit_coefficients = false;
    bool verbose = false;
    std::array<PredictionParams, 3> value_prediction_params{}; // per channel; an INPUT here (the SVD fit stays on the host)
    std::array<PredictionParams, 3> width_prediction_params{};
    std::array<int32_t, 32> quantization_matrix;               // get_quantization_matrix(), quantization.rs:3-5
    bool fit_parameters = true; // like the reference (prediction.rs:232-235); false = use the parameters given above
    int device = 0;
    EncoderOpts() { quantization_matrix.fill(1); }
};

// WaveletImage (wavelet_transform.rs:384-389) as dense arrays in the plan's canonical cell order.
struct WaveletImage {
    ImageMetadata metadata;
    uint32_t num_cells = 0;
    std::vector<int32_t> centers;      // [F][2] (re, im)
    std::vector<int32_t> coefficients; // [C][F][512], FRI_HIP_NONE = None          (Fractal.coefficients)
    std::vector<uint8_t> bucket;       // [C][F][512]                           (Fractal.parameter_predictors.0)
    std::vector<int32_t> prediction;   // [C][F][512]                           (Fractal.parameter_predictors.1)
    bool quantized = false;
    size_t plane() const { return (size_t)num_cells * FRI_HIP_CELL_SIZE; }
    const uint8_t *bucket_of(uint32_t channel) const { return bucket.data() + channel * plane(); }
    const int32_t *prediction_of(uint32_t channel) const { return prediction.data() + channel * plane(); }
};

template <typename T>
struct Result {
    bool ok = false;
    T value{};
    std::string error;
};

// Owns the fri_hip_ctx and a cache of plans keyed by (width, height, channels).
class Device {
  public:
    explicit Device(int device);
    ~Device();
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    bool ok() const { return ctx_ != nullptr; }
    const std::string &error() const { return error_; }
    fri_hip_plan *plan(uint32_t width, uint32_t height, uint32_t channels, std::string &err);
    // Plans of this device measure their forward tiling when they are made (fri_hip_plan_tune_forward: tens of milliseconds and ~1 GB of scratch memory per
    // new shape, remembered per shape for the process) - what a batch driver wants; off by default, so that a one-image call costs what it did.
    void measure_forward_tiling(bool on) { tune_ = on; }
    // the plan of that shape with the emitter's symbol order installed (fri_hip_plan_set_stream_order; geometry only: computed and uploaded once per plan)
    fri_hip_plan *stream_plan(uint32_t width, uint32_t height, uint32_t channels, std::string &err);
    std::string describe(int code) const;

  private:
    fri_hip_ctx *ctx_ = nullptr;
    std::string error_;
    bool tune_ = false;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, fri_hip_plan *> plans_;
    std::vector<fri_hip_plan *> ordered_; // plans whose stream order is installed
};

// ContextModeler (context_modeling.rs:13-213): the least-squares fit of the value / width predictors. The device
// accumulates the normal-equation sums (fri_hip_fit_value_sums / fri_hip_fit_width_sums); the 6 x 6 systems are solved by the
// library's host functions (fri_hip_fit_value_params / fri_hip_fit_width_params: minimum-norm solution via a Jacobi
// eigen-decomposition, the counterpart of lstsq's SVD with its 1e-14 cut-off). optimize_parameters is the single-channel, stage-by-stage
// form; FRIEncoder::encode and stages::prediction::encode fit all channels inside one device-resident call instead.
struct ContextModeler {
    std::array<PredictionParams, 3> value_predictors{};
    std::array<PredictionParams, 3> width_predictors{};
    // optimize_parameters(&wavelet_image, channel), context_modeling.rs:204-213
    Result<bool> optimize_parameters(const WaveletImage &image, uint32_t channel, Device &dev);
    // x = pinv(M) y for a symmetric positive semi-definite 6 x 6 matrix (fri_hip_solve6)
    static std::array<double, 6> solve_normal_equations(const double (&m)[6][6], const double (&y)[6]);
};

namespace stages {
namespace wavelet_transform {
Result<WaveletImage> encode(const RasterImage &raster, const EncoderOpts &opts, Device &dev); // + fused quantiser, see quantization::encode
Result<RasterImage> decode(const WaveletImage &image, const EncoderOpts &opts, Device &dev);
} // namespace wavelet_transform
namespace quantization {
// The device applies the matrix inside the transform kernel; encode() only checks that this happened.
Result<WaveletImage> encode(WaveletImage image);
} // namespace quantization
namespace prediction {
Result<std::array<std::vector<AnsContext>, 3>> encode(WaveletImage &image, EncoderOpts &opts, Device &dev); // fits opts.*_prediction_params first
} // namespace prediction
} // namespace stages

// CompressedImage (images.rs:115-126): per channel the ten ANS models, the interleaved rANS stream and the predictor parameters
struct CompressedImage {
    ImageMetadata metadata;
    std::vector<emit::ChannelStream> channel_data;
    std::vector<emit::ChannelParams> params;
    uint32_t variant = 1; // FractalVariant::TameTwindragon, images.rs:49-55
};
namespace stages {
namespace entropy_coding {
// entropy_coding::encode (:266-352); the contexts are rebuilt from the device histograms inside (prediction.rs:302-305)
Result<CompressedImage> encode(const WaveletImage &image, const std::array<std::vector<AnsContext>, 3> &contexts, const EncoderOpts &opts);
} // namespace entropy_coding
namespace serialize {
std::vector<uint8_t> encode(const CompressedImage &image); // serialize.rs:49-117
Result<CompressedImage> decode(const std::vector<uint8_t> &bytes); // serialize.rs:119-268
} // namespace serialize
namespace entropy_coding {
// entropy_coding::decode (:352-443): sequential per channel on the host (every symbol's context depends on the ones before it)
Result<WaveletImage> decode(const CompressedImage &image);
} // namespace entropy_coding
} // namespace stages

struct EncodedStages { // EncoderStage::EntropyEncoding(WaveletImage, [Vec<AnsContext>; 3]), encoder.rs:12
    WaveletImage image;
    std::array<std::vector<AnsContext>, 3> contexts;
};

class FRIEncoder { // encoder.rs:66-109
  public:
    explicit FRIEncoder(EncoderOpts opts) : opts_(std::move(opts)) {}
    Result<EncodedStages> encode(std::vector<uint8_t> data, uint32_t height, uint32_t width, ColorSpace colorspace);
    // the whole pipeline of encoder.rs:19-48: ... -> EntropyEncoding -> Serialization -> EncodedImage(Vec<u8>)
    Result<std::vector<uint8_t>> encode_bytes(std::vector<uint8_t> data, uint32_t height, uint32_t width, ColorSpace colorspace);
    // the same bytes through the symbol stream route: the emitter's gather runs on the device (fri_hip_encode_image_symbols), 2 bytes per symbol come down
    Result<std::vector<uint8_t>> encode_bytes_streamed(std::vector<uint8_t> data, uint32_t height, uint32_t width, ColorSpace colorspace);
    const EncoderOpts &opts() const { return opts_; } // after encode: the fitted predictor parameters

  private:
    EncoderOpts opts_;
};

// A batch of images of one shape to .frv bytes - the loop of crates/fri-cli/src/commands/bench.rs:15-120 around FRIEncoder::encode - as a pipeline: one thread
// per device runs the stage chain up to the emitter's input (fri_hip_encode_image_symbols: K1 -> fit -> K2 -> K5, 17 MB up and 34 MB down per 4096^2 plane),
// image i on device i mod n_devices (fri_hip_shard_image), while `emit_threads` host threads turn the streams of the images before it into rANS bytes
// (entropy_coding.rs:266-352) and containers (serialize.rs:48-117). The bytes are those of FRIEncoder::encode_bytes, image for image.
struct BatchStats {
    double seconds = 0, device_seconds = 0, emit_seconds = 0; // wall clock of the batch; summed over images: the device calls / the host emits
};
Result<std::vector<std::vector<uint8_t>>> encode_batch_bytes(const std::vector<const uint8_t *> &images, uint32_t height, uint32_t width, ColorSpace colorspace,
                                                             const EncoderOpts &opts, const std::vector<int> &devices, unsigned emit_threads, BatchStats *stats = nullptr);
// The same with devices the caller keeps (one Device per producer thread; a Device caches its plans and their stream order): a service that encodes batch after
// batch pays for contexts, plans and the symbol order (0.2-0.3 s per 4096^2 shape) once, not per call.
Result<std::vector<std::vector<uint8_t>>> encode_batch_bytes(const std::vector<const uint8_t *> &images, uint32_t height, uint32_t width, ColorSpace colorspace,
                                                             const EncoderOpts &opts, const std::vector<Device *> &devices, unsigned emit_threads, BatchStats *stats = nullptr);

class FRIDecoder { // decoder.rs:44-59
  public:
    // the whole pipeline of decoder.rs:16-40: EncodedImage -> EntropyDecoding -> Dequantization -> WaveletTransform -> RawImage.
    // Container parsing and entropy decoding run on the host, dequantisation + inverse transform on the device.
    Result<RasterImage> decode(const std::vector<uint8_t> &data, const EncoderOpts &opts = EncoderOpts());
    // from the WaveletTransform stage on
    Result<RasterImage> decode(const WaveletImage &image, const EncoderOpts &opts = EncoderOpts());
};

} // namespace libfri
