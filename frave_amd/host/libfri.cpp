// libfri.cpp -- see libfri.hpp. Host glue over the C ABI; no compute here.
#include "libfri.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>

namespace libfri {

Device::Device(int device) {
    int rc = fri_hip_ctx_create(device, &ctx_);
    if (rc != FRI_HIP_OK) {
        ctx_ = nullptr;
        error_ = fri_hip_strerror(rc);
    }
}
Device::~Device() {
    for (auto &kv : plans_) fri_hip_plan_destroy(kv.second);
    if (ctx_) fri_hip_ctx_destroy(ctx_);
}
std::string Device::describe(int code) const {
    std::string s = fri_hip_strerror(code);
    if (ctx_ && code == FRI_HIP_ERR_HIP) s += std::string(": ") + fri_hip_last_hip_error(ctx_);
    return s;
}
fri_hip_plan *Device::plan(uint32_t width, uint32_t height, uint32_t channels, std::string &err) {
    if (!ctx_) {
        err = error_;
        return nullptr;
    }
    auto key = std::make_tuple(width, height, channels);
    auto it = plans_.find(key);
    if (it != plans_.end()) return it->second;
    fri_hip_plan *p = nullptr;
    int rc = fri_hip_plan_create(ctx_, width, height, channels, &p);
    if (rc != FRI_HIP_OK) {
        err = describe(rc);
        return nullptr;
    }
    if (tune_) (void)fri_hip_plan_tune_forward(p, 0, nullptr, 0); // (a failed measurement leaves the default tiling: not an error of the encode)
    plans_[key] = p;
    return p;
}

std::array<double, 6> ContextModeler::solve_normal_equations(const double (&m)[6][6], const double (&y)[6]) {
    std::array<double, 6> x{};
    fri_hip_solve6(m, y, x.data());
    return x;
}

Result<bool> ContextModeler::optimize_parameters(const WaveletImage &image, uint32_t channel, Device &dev) {
    Result<bool> r;
    const uint32_t c = num_channels(image.metadata.colorspace);
    fri_hip_plan *plan = dev.plan(image.metadata.width, image.metadata.height, c, r.error);
    if (!plan) return r;
    int64_t gram[3][28];
    int rc = fri_hip_fit_value_sums(plan, image.coefficients.data(), channel, gram);
    if (rc != FRI_HIP_OK) {
        r.error = dev.describe(rc);
        return r;
    }
    float vp[3][6], wp[3][6];
    fri_hip_fit_value_params(gram, vp); // optimize_value_prediction, context_modeling.rs:175-202
    int64_t wtw[3][21];
    double wtr[3][6];
    uint64_t rows[3];
    rc = fri_hip_fit_width_sums(plan, image.coefficients.data(), channel, vp, wtw, wtr, rows);
    if (rc != FRI_HIP_OK) {
        r.error = dev.describe(rc);
        return r;
    }
    fri_hip_fit_width_params(wtw, wtr, rows, wp); // optimize_width_prediction, context_modeling.rs:144-173
    for (int g = 0; g < 3; g++)
        for (int k = 0; k < 6; k++) value_predictors[channel][g][k] = vp[g][k], width_predictors[channel][g][k] = wp[g][k];
    r.ok = r.value = true;
    return r;
}

namespace stages {
namespace wavelet_transform {

Result<WaveletImage> encode(const RasterImage &raster, const EncoderOpts &opts, Device &dev) {
    Result<WaveletImage> r;
    const uint32_t c = num_channels(raster.metadata.colorspace);
    if (raster.data.size() != (size_t)raster.metadata.width * raster.metadata.height * c) {
        r.error = "raster size does not match its metadata";
        return r;
    }
    fri_hip_plan *plan = dev.plan(raster.metadata.width, raster.metadata.height, c, r.error);
    if (!plan) return r;
    WaveletImage &w = r.value;
    w.metadata = raster.metadata;
    w.num_cells = fri_hip_plan_num_cells(plan);
    w.centers.resize((size_t)w.num_cells * 2);
    w.coefficients.resize(fri_hip_plan_coef_count(plan));
    int rc = fri_hip_plan_centers(plan, w.centers.data());
    if (rc == FRI_HIP_OK) rc = fri_hip_transform_quant(plan, raster.data.data(), opts.quantization_matrix.data(), w.coefficients.data());
    if (rc != FRI_HIP_OK) {
        r.error = dev.describe(rc);
        return r;
    }
    w.quantized = true;
    r.ok = true;
    return r;
}

Result<RasterImage> decode(const WaveletImage &image, const EncoderOpts &opts, Device &dev) {
    Result<RasterImage> r;
    const uint32_t c = num_channels(image.metadata.colorspace);
    fri_hip_plan *plan = dev.plan(image.metadata.width, image.metadata.height, c, r.error);
    if (!plan) return r;
    if (image.coefficients.size() != fri_hip_plan_coef_count(plan)) {
        r.error = "coefficient array does not match the image geometry";
        return r;
    }
    r.value.metadata = image.metadata;
    r.value.data.resize(fri_hip_plan_pixel_bytes(plan));
    int rc = fri_hip_inverse_transform(plan, image.coefficients.data(), opts.quantization_matrix.data(), r.value.data.data());
    if (rc != FRI_HIP_OK) {
        r.error = dev.describe(rc);
        return r;
    }
    r.ok = true;
    return r;
}

} // namespace wavelet_transform

namespace quantization {
Result<WaveletImage> encode(WaveletImage image) {
    Result<WaveletImage> r;
    if (!image.quantized) {
        r.error = "coefficients were not produced by wavelet_transform::encode";
        return r;
    }
    r.value = std::move(image);
    r.ok = true;
    return r;
}
} // namespace quantization

namespace prediction {
static void contexts_from_hist(const std::vector<uint32_t> &hist, uint32_t channels, std::array<std::vector<AnsContext>, 3> &out) {
    for (uint32_t ch = 0; ch < channels; ch++) {
        out[ch].resize(CONTEXT_AMOUNT);
        for (int b = 0; b < CONTEXT_AMOUNT; b++)
            for (int s = 0; s < ALPHABET_SIZE; s++) out[ch][b].freqs[s] = hist[((size_t)ch * CONTEXT_AMOUNT + b) * ALPHABET_SIZE + s];
    }
}
static void params_to_flat(const EncoderOpts &opts, uint32_t channels, float (&vp)[3][3][6], float (&wp)[3][3][6]) {
    for (uint32_t ch = 0; ch < channels; ch++)
        for (int g = 0; g < 3; g++)
            for (int k = 0; k < 6; k++) vp[ch][g][k] = opts.value_prediction_params[ch][g][k], wp[ch][g][k] = opts.width_prediction_params[ch][g][k];
}
static void params_from_flat(EncoderOpts &opts, uint32_t channels, const float (&vp)[3][3][6], const float (&wp)[3][3][6]) {
    for (uint32_t ch = 0; ch < channels; ch++)
        for (int g = 0; g < 3; g++)
            for (int k = 0; k < 6; k++) opts.value_prediction_params[ch][g][k] = vp[ch][g][k], opts.width_prediction_params[ch][g][k] = wp[ch][g][k];
}

// One upload of the coefficients, then the fit and the scan of every channel on the device (fri_hip_predict_image).
Result<std::array<std::vector<AnsContext>, 3>> encode(WaveletImage &image, EncoderOpts &opts, Device &dev) {
    Result<std::array<std::vector<AnsContext>, 3>> r;
    const uint32_t c = num_channels(image.metadata.colorspace);
    fri_hip_plan *plan = dev.plan(image.metadata.width, image.metadata.height, c, r.error);
    if (!plan) return r;
    const size_t n = image.plane();
    std::vector<uint32_t> hist((size_t)c * CONTEXT_AMOUNT * ALPHABET_SIZE);
    image.bucket.resize(c * n);
    image.prediction.resize(c * n);
    float vp[3][3][6], wp[3][3][6];
    params_to_flat(opts, c, vp, wp);
    uint64_t oob[3] = {0, 0, 0};
    const int rc = fri_hip_predict_image(plan, image.coefficients.data(), opts.fit_parameters ? 1 : 0, &vp[0][0][0], &wp[0][0][0], image.bucket.data(), image.prediction.data(),
                                         hist.data(), oob);
    if (rc != FRI_HIP_OK) {
        r.error = dev.describe(rc);
        return r;
    }
    params_from_flat(opts, c, vp, wp);
    if (oob[0] | oob[1] | oob[2]) { // the reference panics here: index out of bounds in bump_freq (entropy_coding.rs:99)
        r.error = "symbol outside the 1024-entry alphabet";
        return r;
    }
    contexts_from_hist(hist, c, r.value);
    r.ok = true;
    return r;
}
} // namespace prediction
} // namespace stages

Result<EncodedStages> FRIEncoder::encode(std::vector<uint8_t> data, uint32_t height, uint32_t width, ColorSpace colorspace) {
    Result<EncodedStages> r;
    auto fail = [&](const std::string &msg) {
        r.error = "Failed to decode: " + msg; // sic, encoder.rs:106
        return r;
    };
    Device dev(opts_.device);
    if (!dev.ok()) return fail(dev.error());
    const uint32_t c = num_channels(colorspace);
    if (data.size() != (size_t)width * height * c) return fail("raster size does not match its metadata");
    std::string err;
    fri_hip_plan *plan = dev.plan(width, height, c, err);
    if (!plan) return fail(err);
    // RawImage -> ChannelTransform (identity, channel_transform.rs:4-10) -> WaveletTransform -> Quantization -> Prediction (encoder.rs:19-38) as
    // ONE device-resident call: the pixels go up once, the coefficients stay in device memory between the stages, every output comes down once.
    WaveletImage &w = r.value.image;
    w.metadata = ImageMetadata{height, width, colorspace};
    w.num_cells = fri_hip_plan_num_cells(plan);
    w.centers.resize((size_t)w.num_cells * 2);
    w.coefficients.resize(fri_hip_plan_coef_count(plan));
    w.bucket.resize(c * w.plane());
    w.prediction.resize(c * w.plane());
    std::vector<uint32_t> hist((size_t)c * CONTEXT_AMOUNT * ALPHABET_SIZE);
    float vp[3][3][6], wp[3][3][6];
    stages::prediction::params_to_flat(opts_, c, vp, wp);
    uint64_t oob[3] = {0, 0, 0};
    int rc = fri_hip_plan_centers(plan, w.centers.data());
    if (rc == FRI_HIP_OK)
        rc = fri_hip_encode_image(plan, data.data(), opts_.quantization_matrix.data(), opts_.fit_parameters ? 1 : 0, &vp[0][0][0], &wp[0][0][0], w.coefficients.data(),
                                  w.bucket.data(), w.prediction.data(), hist.data(), oob);
    if (rc != FRI_HIP_OK) return fail(dev.describe(rc));
    w.quantized = true;
    stages::prediction::params_from_flat(opts_, c, vp, wp);
    if (oob[0] | oob[1] | oob[2]) return fail("symbol outside the 1024-entry alphabet"); // the reference panics: bump_freq, entropy_coding.rs:99
    stages::prediction::contexts_from_hist(hist, c, r.value.contexts);
    r.ok = true;
    return r;
}

Result<CompressedImage> stages::entropy_coding::encode(const WaveletImage &image, const std::array<std::vector<AnsContext>, 3> &contexts, const EncoderOpts &opts) {
    Result<CompressedImage> r;
    const uint32_t channels = num_channels(image.metadata.colorspace);
    const size_t plane = (size_t)image.num_cells * 512;
    r.value.metadata = image.metadata;
    r.value.params.resize(channels);
    std::vector<uint32_t> hist((size_t)channels * CONTEXT_AMOUNT * ALPHABET_SIZE);
    if (image.bucket.size() != channels * plane || image.prediction.size() != channels * plane) {
        r.error = "missing predictors";
        return r;
    }
    for (uint32_t ch = 0; ch < channels; ch++) {
        if (contexts[ch].size() != (size_t)CONTEXT_AMOUNT) {
            r.error = "missing contexts";
            return r;
        }
        for (int b = 0; b < CONTEXT_AMOUNT; b++)
            std::copy(contexts[ch][b].freqs.begin(), contexts[ch][b].freqs.end(), hist.begin() + ((size_t)ch * CONTEXT_AMOUNT + b) * ALPHABET_SIZE);
        for (int g = 0; g < 3; g++)
            for (int k = 0; k < 6; k++) {
                r.value.params[ch].value[g][k] = opts.value_prediction_params[ch][g][k];
                r.value.params[ch].width[g][k] = opts.width_prediction_params[ch][g][k];
            }
    }
    const auto order_ptr = emit::shared_symbol_order(image.centers.data(), image.num_cells); // geometry only: cached per image size
    const emit::SymbolOrder &order = *order_ptr;
    const std::string err = emit::encode_channels(order, channels, image.coefficients.data(), image.bucket.data(), image.prediction.data(), hist.data(), r.value.channel_data);
    if (!err.empty()) {
        r.error = err;
        return r;
    }
    r.ok = true;
    return r;
}

std::vector<uint8_t> stages::serialize::encode(const CompressedImage &image) {
    const emit::ColorSpaceCode cs = image.metadata.colorspace == ColorSpace::Luma ? emit::kLuma : image.metadata.colorspace == ColorSpace::RGB ? emit::kRGB : emit::kYCbCr;
    return emit::serialize(image.metadata.height, image.metadata.width, cs, image.channel_data, image.params);
}

Result<CompressedImage> stages::serialize::decode(const std::vector<uint8_t> &bytes) {
    Result<CompressedImage> r;
    emit::ParsedImage p;
    r.error = emit::deserialize(bytes, p);
    if (!r.error.empty()) return r;
    r.value.metadata.height = p.height;
    r.value.metadata.width = p.width;
    r.value.metadata.colorspace = p.colorspace == emit::kLuma ? ColorSpace::Luma : p.colorspace == emit::kRGB ? ColorSpace::RGB : ColorSpace::YCbCr;
    r.value.variant = p.variant;
    r.value.channel_data = std::move(p.channels);
    r.value.params = std::move(p.params);
    r.ok = true;
    return r;
}

Result<WaveletImage> stages::entropy_coding::decode(const CompressedImage &image) {
    Result<WaveletImage> r;
    emit::ParsedImage p;
    p.height = image.metadata.height, p.width = image.metadata.width, p.variant = image.variant;
    p.colorspace = image.metadata.colorspace == ColorSpace::Luma ? emit::kLuma : image.metadata.colorspace == ColorSpace::RGB ? emit::kRGB : emit::kYCbCr;
    p.channels = image.channel_data;
    p.params = image.params;
    emit::DecodedImage d;
    r.error = emit::decode_parsed(p, d);
    if (!r.error.empty()) return r;
    r.value.metadata = image.metadata;
    r.value.num_cells = d.n_cells;
    r.value.centers = std::move(d.centers);
    r.value.coefficients = std::move(d.coefs);
    r.value.quantized = true;
    r.ok = true;
    return r;
}

Result<RasterImage> FRIDecoder::decode(const std::vector<uint8_t> &data, const EncoderOpts &opts) {
    Result<RasterImage> r;
    auto c = stages::serialize::decode(data);
    if (!c.ok) {
        r.error = "Failed to decode: " + c.error; // decoder.rs:56
        return r;
    }
    auto w = stages::entropy_coding::decode(c.value);
    if (!w.ok) {
        r.error = "Failed to decode: " + w.error;
        return r;
    }
    return decode(w.value, opts); // quantization::decode + wavelet_transform::decode: one kernel (fri_hip_inverse_transform)
}

namespace {
// what the device hands the emitter for one image
struct StreamedImage {
    size_t index = 0;
    std::vector<uint16_t> symbols; // [C][num_some]
    std::vector<uint32_t> hist;    // [C][10][1024]
    float vp[3][3][6], wp[3][3][6];
};
std::string set_plan_stream_order(fri_hip_plan *plan, Device &dev) {
    const uint32_t F = fri_hip_plan_num_cells(plan);
    std::vector<int32_t> centers((size_t)F * 2);
    std::vector<uint32_t> mask((size_t)F * 16);
    int rc = fri_hip_plan_centers(plan, centers.data());
    if (rc == FRI_HIP_OK) rc = fri_hip_plan_valid_mask(plan, mask.data());
    if (rc != FRI_HIP_OK) return dev.describe(rc);
    const std::vector<uint32_t> order = emit::stream_order(*emit::shared_symbol_order(centers.data(), F), mask.data());
    rc = fri_hip_plan_set_stream_order(plan, order.data(), order.size());
    return rc == FRI_HIP_OK ? std::string() : dev.describe(rc);
}
std::string emit_streamed(const StreamedImage &im, uint32_t c, uint64_t n, uint32_t height, uint32_t width, ColorSpace colorspace, std::vector<uint8_t> &out) {
    std::vector<emit::ChannelStream> streams;
    const std::string e = emit::encode_channels_from_streams(c, im.symbols.data(), (size_t)n, im.hist.data(), streams);
    if (!e.empty()) return e;
    std::vector<emit::ChannelParams> params(c);
    for (uint32_t ch = 0; ch < c; ch++)
        for (int g = 0; g < 3; g++)
            for (int k = 0; k < 6; k++) params[ch].value[g][k] = im.vp[ch][g][k], params[ch].width[g][k] = im.wp[ch][g][k];
    const emit::ColorSpaceCode cs = colorspace == ColorSpace::Luma ? emit::kLuma : colorspace == ColorSpace::RGB ? emit::kRGB : emit::kYCbCr;
    out = emit::serialize(height, width, cs, streams, params);
    return std::string();
}
} // namespace

fri_hip_plan *Device::stream_plan(uint32_t width, uint32_t height, uint32_t channels, std::string &err) {
    fri_hip_plan *p = plan(width, height, channels, err);
    if (!p) return nullptr;
    if (std::find(ordered_.begin(), ordered_.end(), p) != ordered_.end()) return p;
    err = set_plan_stream_order(p, *this);
    if (!err.empty()) return nullptr;
    ordered_.push_back(p);
    return p;
}

// FRIEncoder::encode (encoder.rs:87-109) end to end through the symbol stream route: the device runs the stage chain AND the emitter's gather
// (sort_lattice order, entropy_coding.rs:285-336), the host receives 2 bytes per symbol and runs the rANS loop and the serializer. Byte for byte
// the .frv of encode_bytes below (tests/test_emit.py); 34 MB instead of 153 MB come down per 4096 x 4096 plane.
Result<std::vector<uint8_t>> FRIEncoder::encode_bytes_streamed(std::vector<uint8_t> data, uint32_t height, uint32_t width, ColorSpace colorspace) {
    Result<std::vector<uint8_t>> r;
    auto fail = [&](const std::string &msg) {
        r.error = "Failed to decode: " + msg; // sic, encoder.rs:106
        return r;
    };
    Device dev(opts_.device);
    if (!dev.ok()) return fail(dev.error());
    const uint32_t c = num_channels(colorspace);
    if (data.size() != (size_t)width * height * c) return fail("raster size does not match its metadata");
    std::string err;
    fri_hip_plan *plan = dev.plan(width, height, c, err);
    if (!plan) return fail(err);
    const uint64_t n = fri_hip_plan_num_some(plan);
    if (const std::string e = set_plan_stream_order(plan, dev); !e.empty()) return fail(e); // geometry only, once per plan (the plan is new here: Device lives for this call, as in encode())
    std::vector<uint16_t> symbols((size_t)c * n);
    std::vector<uint32_t> hist((size_t)c * CONTEXT_AMOUNT * ALPHABET_SIZE);
    float vp[3][3][6], wp[3][3][6];
    stages::prediction::params_to_flat(opts_, c, vp, wp);
    uint64_t oob[3] = {0, 0, 0};
    const int rc = fri_hip_encode_image_symbols(plan, data.data(), opts_.quantization_matrix.data(), opts_.fit_parameters ? 1 : 0, &vp[0][0][0], &wp[0][0][0], symbols.data(),
                                                hist.data(), oob);
    if (rc != FRI_HIP_OK) return fail(dev.describe(rc));
    stages::prediction::params_from_flat(opts_, c, vp, wp);
    if (oob[0] | oob[1] | oob[2]) return fail("symbol outside the 1024-entry alphabet"); // the reference panics: bump_freq, entropy_coding.rs:99
    std::vector<emit::ChannelStream> streams;
    const std::string e = emit::encode_channels_from_streams(c, symbols.data(), (size_t)n, hist.data(), streams);
    if (!e.empty()) return fail(e);
    std::vector<emit::ChannelParams> params(c);
    for (uint32_t ch = 0; ch < c; ch++)
        for (int g = 0; g < 3; g++)
            for (int k = 0; k < 6; k++) params[ch].value[g][k] = vp[ch][g][k], params[ch].width[g][k] = wp[ch][g][k];
    const emit::ColorSpaceCode cs = colorspace == ColorSpace::Luma ? emit::kLuma : colorspace == ColorSpace::RGB ? emit::kRGB : emit::kYCbCr;
    r.value = emit::serialize(height, width, cs, streams, params);
    r.ok = true;
    return r;
}

Result<std::vector<std::vector<uint8_t>>> encode_batch_bytes(const std::vector<const uint8_t *> &images, uint32_t height, uint32_t width, ColorSpace colorspace,
                                                             const EncoderOpts &opts, const std::vector<int> &devices, unsigned emit_threads, BatchStats *stats) {
    std::vector<std::unique_ptr<Device>> owned;
    std::vector<Device *> devs;
    for (int d : devices) {
        owned.emplace_back(new Device(d));
        devs.push_back(owned.back().get());
    }
    return encode_batch_bytes(images, height, width, colorspace, opts, devs, emit_threads, stats);
}

Result<std::vector<std::vector<uint8_t>>> encode_batch_bytes(const std::vector<const uint8_t *> &images, uint32_t height, uint32_t width, ColorSpace colorspace,
                                                             const EncoderOpts &opts, const std::vector<Device *> &devices, unsigned emit_threads, BatchStats *stats) {
    Result<std::vector<std::vector<uint8_t>>> r;
    const size_t n_images = images.size();
    const uint32_t c = num_channels(colorspace);
    const uint32_t n_dev = (uint32_t)devices.size();
    if (!n_dev || !emit_threads) {
        r.error = "encode_batch_bytes: no device / no emitter thread";
        return r;
    }
    // One producer thread per entry drives that Device's plan and staging buffers: a Device listed twice would put two threads on one plan (a plan is
    // single-threaded, include/fri_hip.h), a null entry or a null image pointer would go straight into the C ABI. Refused up front.
    for (uint32_t d = 0; d < n_dev; d++) {
        if (!devices[d]) {
            r.error = "encode_batch_bytes: null Device";
            return r;
        }
        for (uint32_t e = 0; e < d; e++)
            if (devices[e] == devices[d]) {
                r.error = "encode_batch_bytes: the same Device is listed twice (one producer thread per entry: a plan is single-threaded)";
                return r;
            }
    }
    for (size_t i = 0; i < n_images; i++)
        if (!images[i]) {
            r.error = "encode_batch_bytes: null image pointer";
            return r;
        }
    r.value.assign(n_images, {});
    std::mutex mu;
    std::condition_variable have_work, have_room;
    std::deque<StreamedImage> queue; // bounded: the device side runs at most a few images ahead of the emitters (a 4096^2 RGB stream set is 100 MB)
    const size_t max_queued = 2 * (size_t)emit_threads + n_dev;
    uint32_t producers_left = n_dev;
    std::string first_error;
    std::atomic<long long> dev_ns{0}, emit_ns{0};
    auto fail = [&](const std::string &msg) {
        std::lock_guard<std::mutex> lk(mu);
        if (first_error.empty()) first_error = msg;
        have_work.notify_all(), have_room.notify_all();
    };
    const auto t_begin = std::chrono::steady_clock::now();
    std::atomic<uint64_t> n_some{0}; // symbols per channel: a geometry fact, published by whichever producer has its plan first (every plan of the shape gives the same number)
    std::vector<std::thread> producers, emitters;
    for (uint32_t d = 0; d < n_dev; d++)
        producers.emplace_back([&, d]() {
            Device &dev = *devices[d];
            std::string err;
            fri_hip_plan *plan = dev.ok() ? dev.stream_plan(width, height, c, err) : nullptr;
            if (!plan && err.empty()) err = dev.error();
            if (!err.empty()) fail(err);
            const uint64_t n = plan ? fri_hip_plan_num_some(plan) : 0;
            if (n) n_some.store(n, std::memory_order_relaxed);
            for (size_t k = 0; err.empty() && k < fri_hip_shard_size(n_images, d, n_dev); k++) {
                const size_t i = fri_hip_shard_image(k, d, n_dev); // image i -> device i mod n_dev: the library's partition
                StreamedImage im;
                im.index = i;
                im.symbols.resize((size_t)c * n);
                im.hist.resize((size_t)c * CONTEXT_AMOUNT * ALPHABET_SIZE);
                EncoderOpts o = opts;
                stages::prediction::params_to_flat(o, c, im.vp, im.wp);
                uint64_t oob[3] = {0, 0, 0};
                const auto t0 = std::chrono::steady_clock::now();
                const int rc = fri_hip_encode_image_symbols(plan, images[i], o.quantization_matrix.data(), o.fit_parameters ? 1 : 0, &im.vp[0][0][0], &im.wp[0][0][0], im.symbols.data(),
                                                            im.hist.data(), oob);
                dev_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                if (rc != FRI_HIP_OK) {
                    fail(dev.describe(rc));
                    break;
                }
                if (oob[0] | oob[1] | oob[2]) {
                    fail("symbol outside the 1024-entry alphabet"); // the reference panics: bump_freq, entropy_coding.rs:99
                    break;
                }
                std::unique_lock<std::mutex> lk(mu);
                have_room.wait(lk, [&] { return queue.size() < max_queued || !first_error.empty(); });
                if (!first_error.empty()) break;
                queue.push_back(std::move(im));
                have_work.notify_one();
            }
            std::lock_guard<std::mutex> lk(mu);
            if (--producers_left == 0) have_work.notify_all();
        });
    for (unsigned t = 0; t < emit_threads; t++)
        emitters.emplace_back([&]() {
            for (;;) {
                StreamedImage im;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    have_work.wait(lk, [&] { return !queue.empty() || producers_left == 0 || !first_error.empty(); });
                    if (!first_error.empty() || queue.empty()) return;
                    im = std::move(queue.front());
                    queue.pop_front();
                    have_room.notify_one();
                }
                const auto t0 = std::chrono::steady_clock::now();
                uint64_t n = n_some.load(std::memory_order_relaxed); // (set before the first image was queued: the queue's mutex orders it)
                if (!n) n = im.symbols.size() / c;
                const std::string e = emit_streamed(im, c, n, height, width, colorspace, r.value[im.index]);
                emit_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                if (!e.empty()) {
                    fail(e);
                    return;
                }
            }
        });
    for (auto &t : producers) t.join();
    for (auto &t : emitters) t.join();
    if (stats) {
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        stats->device_seconds = dev_ns.load() * 1e-9, stats->emit_seconds = emit_ns.load() * 1e-9;
    }
    if (!first_error.empty()) {
        r.error = "Failed to decode: " + first_error; // sic, encoder.rs:106
        r.value.clear();
        return r;
    }
    r.ok = true;
    return r;
}

Result<std::vector<uint8_t>> FRIEncoder::encode_bytes(std::vector<uint8_t> data, uint32_t height, uint32_t width, ColorSpace colorspace) {
    Result<std::vector<uint8_t>> r;
    auto st = encode(std::move(data), height, width, colorspace);
    if (!st.ok) {
        r.error = st.error;
        return r;
    }
    auto c = stages::entropy_coding::encode(st.value.image, st.value.contexts, opts_);
    if (!c.ok) {
        r.error = "Failed to decode: " + c.error; // sic, encoder.rs:106
        return r;
    }
    r.value = stages::serialize::encode(c.value);
    r.ok = true;
    return r;
}

Result<RasterImage> FRIDecoder::decode(const WaveletImage &image, const EncoderOpts &opts) {
    Device dev(opts.device);
    Result<RasterImage> r;
    if (!dev.ok()) {
        r.error = "Failed to decode: " + dev.error();
        return r;
    }
    r = stages::wavelet_transform::decode(image, opts, dev);
    if (!r.ok) r.error = "Failed to decode: " + r.error;
    return r;
}

} // namespace libfri
