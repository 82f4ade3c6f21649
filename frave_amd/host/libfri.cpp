// libfri.cpp -- see libfri.hpp. Host glue over the C ABI; no compute here.
#include "libfri.hpp"

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
    plans_[key] = p;
    return p;
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
Result<std::array<std::vector<AnsContext>, 3>> encode(WaveletImage &image, const EncoderOpts &opts, Device &dev) {
    Result<std::array<std::vector<AnsContext>, 3>> r;
    const uint32_t c = num_channels(image.metadata.colorspace);
    fri_hip_plan *plan = dev.plan(image.metadata.width, image.metadata.height, c, r.error);
    if (!plan) return r;
    const size_t n = (size_t)image.num_cells * FRI_HIP_CELL_SIZE;
    std::vector<uint32_t> hist((size_t)CONTEXT_AMOUNT * ALPHABET_SIZE);
    for (uint32_t ch = 0; ch < c; ch++) {
        image.bucket[ch].resize(n);
        image.prediction[ch].resize(n);
        uint64_t oob = 0;
        int rc = fri_hip_predict_histogram(plan, image.coefficients.data(), ch,
                                           reinterpret_cast<const float(*)[6]>(opts.value_prediction_params[ch].data()),
                                           reinterpret_cast<const float(*)[6]>(opts.width_prediction_params[ch].data()), image.bucket[ch].data(),
                                           image.prediction[ch].data(), hist.data(), &oob);
        if (rc != FRI_HIP_OK) {
            r.error = dev.describe(rc);
            return r;
        }
        if (oob) { // the reference panics here: index out of bounds in bump_freq (entropy_coding.rs:99)
            r.error = "symbol outside the 1024-entry alphabet";
            return r;
        }
        r.value[ch].resize(CONTEXT_AMOUNT);
        for (int b = 0; b < CONTEXT_AMOUNT; b++)
            for (int s = 0; s < ALPHABET_SIZE; s++) r.value[ch][b].freqs[s] = hist[(size_t)b * ALPHABET_SIZE + s];
    }
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
    RasterImage image{ImageMetadata{height, width, colorspace}, std::move(data)};
    // RawImage -> ChannelTransform (identity, channel_transform.rs:4-10) -> WaveletTransform -> Quantization -> Prediction
    auto w = stages::wavelet_transform::encode(image, opts_, dev);
    if (!w.ok) return fail(w.error);
    auto q = stages::quantization::encode(std::move(w.value));
    if (!q.ok) return fail(q.error);
    auto p = stages::prediction::encode(q.value, opts_, dev);
    if (!p.ok) return fail(p.error);
    r.value.image = std::move(q.value);
    r.value.contexts = std::move(p.value);
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
