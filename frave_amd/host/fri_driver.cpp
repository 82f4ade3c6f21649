// fri_driver.cpp -- command-line driver over the C++ mirror / C ABI (the counterpart of fri-cli's encode/decode/bench
// for this path; crates/fri-cli/src/commands/*.rs). Synthetic inputs only (SURVEY.md section 8d generators).
//   fri_driver roundtrip <width> <height> <channels>         encode -> predict -> decode, checks the lossless identity
//   fri_driver batch <width> <height> <channels> <n> [--gpus N] [--chain]   n images sharded by image over N GPUs: the forward stage, or (--chain) the whole encoder
//   fri_driver encode <width> <height> <channels> <out.frv>  the whole encode pipeline on a synthetic image: device stages, then
//                                                            symbol order / ANS models / rANS / frif container on the host; self-checks the stream
//   fri_driver encode-file <in.pgm|in.ppm|in.bmp> <out.frv>  the same pipeline on a binary PGM (P5, one plane), PPM (P6, RGB) or uncompressed
//                                                            24-bit BMP file, 8 bits per sample (fri-cli encode, crates/fri-cli/src/commands/encode.rs:8-54)
//   fri_driver decode-file <in.frv> <out.pgm|.ppm|.bmp>     container -> rANS / context decoding on the host -> dequantisation + inverse
//                                                            transform on the device (fri-cli decode, crates/fri-cli/src/commands/decode.rs)
//   fri_driver batch <width> <height> <channels> <n_images> [--gpus N]
//                                                            BASELINE config 3: host batch with H2D / kernel / D2H overlap; with --gpus N
//                                                            BASELINE config 4: the batch sharded over N GPUs of this node (image i -> GPU i mod N,
//                                                            one host thread + ctx per GPU, no collective; fri_hip_multi_transform_quant)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <memory>
#include <thread>
#include <vector>

#include "libfri.hpp"

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static std::vector<uint8_t> noise_image(uint32_t w, uint32_t h, uint32_t c, uint64_t index) {
    std::vector<uint8_t> v((size_t)w * h * c);
    const uint64_t seed = 0xF7A5E000ull + index;
    for (size_t i = 0; i < v.size(); i++) v[i] = (uint8_t)(splitmix64(seed + i * 0x9E3779B97F4A7C15ull) & 0xFF);
    return v;
}

// Binary PGM (P5) / PPM (P6) with maxval 255: the image I/O of fri-cli reduced to the two formats that need no library.
static bool read_pnm(const char *path, std::vector<uint8_t> &data, uint32_t &w, uint32_t &h, uint32_t &c, std::string &err) {
    FILE *f = std::fopen(path, "rb");
    if (!f) {
        err = std::string("cannot open ") + path;
        return false;
    }
    auto token = [&](std::string &t) {
        t.clear();
        int ch = std::fgetc(f);
        while (ch != EOF) {
            if (ch == '#') {
                while (ch != EOF && ch != '\n') ch = std::fgetc(f);
            } else if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r') {
                ch = std::fgetc(f);
            } else {
                break;
            }
        }
        while (ch != EOF && ch != ' ' && ch != '\t' && ch != '\n' && ch != '\r') {
            t.push_back((char)ch);
            ch = std::fgetc(f);
        }
        return !t.empty();
    };
    std::string magic, sw, sh, smax;
    bool ok = token(magic) && token(sw) && token(sh) && token(smax);
    if (ok) ok = (magic == "P5" || magic == "P6") && std::atoi(smax.c_str()) == 255 && std::atoi(sw.c_str()) > 0 && std::atoi(sh.c_str()) > 0;
    if (!ok) {
        err = "not a binary PGM/PPM with maxval 255";
        std::fclose(f);
        return false;
    }
    w = (uint32_t)std::atoi(sw.c_str()), h = (uint32_t)std::atoi(sh.c_str()), c = magic == "P5" ? 1u : 3u;
    data.resize((size_t)w * h * c);
    ok = std::fread(data.data(), 1, data.size(), f) == data.size();
    std::fclose(f);
    if (!ok) err = "file shorter than its header says";
    return ok;
}

// Uncompressed 24-bit BMP (BITMAPINFOHEADER, BI_RGB): rows bottom-up (top-down if the height is negative), BGR, padded to 4 bytes.
static bool read_bmp(const char *path, std::vector<uint8_t> &data, uint32_t &w, uint32_t &h, uint32_t &c, std::string &err) {
    FILE *f = std::fopen(path, "rb");
    if (!f) {
        err = std::string("cannot open ") + path;
        return false;
    }
    uint8_t hd[54];
    auto u32 = [&](int o) { return (uint32_t)hd[o] | (uint32_t)hd[o + 1] << 8 | (uint32_t)hd[o + 2] << 16 | (uint32_t)hd[o + 3] << 24; };
    bool ok = std::fread(hd, 1, 54, f) == 54 && hd[0] == 'B' && hd[1] == 'M';
    const int32_t sw = ok ? (int32_t)u32(18) : 0, sh = ok ? (int32_t)u32(22) : 0;
    ok = ok && u32(14) >= 40 && (hd[28] | hd[29] << 8) == 24 && u32(30) == 0 && sw > 0 && sh != 0;
    if (!ok) {
        err = "not an uncompressed 24-bit BMP";
        std::fclose(f);
        return false;
    }
    w = (uint32_t)sw, h = (uint32_t)(sh < 0 ? -(int64_t)sh : sh), c = 3;
    const size_t stride = ((size_t)w * 3 + 3) & ~(size_t)3;
    std::vector<uint8_t> row(stride);
    data.resize((size_t)w * h * 3);
    ok = std::fseek(f, (long)u32(10), SEEK_SET) == 0;
    for (uint32_t r = 0; ok && r < h; r++) {
        ok = std::fread(row.data(), 1, stride, f) == stride;
        uint8_t *dst = data.data() + (size_t)(sh < 0 ? r : h - 1 - r) * w * 3;
        for (uint32_t x = 0; x < w; x++) dst[3 * x] = row[3 * x + 2], dst[3 * x + 1] = row[3 * x + 1], dst[3 * x + 2] = row[3 * x];
    }
    std::fclose(f);
    if (!ok) err = "file shorter than its header says";
    return ok;
}
static bool write_bmp(const char *path, const std::vector<uint8_t> &rgb, uint32_t w, uint32_t h) {
    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const size_t stride = ((size_t)w * 3 + 3) & ~(size_t)3;
    uint8_t hd[54] = {'B', 'M'};
    auto put = [&](int o, uint32_t v) { hd[o] = (uint8_t)v, hd[o + 1] = (uint8_t)(v >> 8), hd[o + 2] = (uint8_t)(v >> 16), hd[o + 3] = (uint8_t)(v >> 24); };
    put(2, (uint32_t)(54 + stride * h)), put(10, 54), put(14, 40), put(18, w), put(22, h), put(34, (uint32_t)(stride * h));
    hd[26] = 1, hd[28] = 24;
    std::fwrite(hd, 1, 54, f);
    std::vector<uint8_t> row(stride, 0);
    for (uint32_t r = 0; r < h; r++) {
        const uint8_t *src = rgb.data() + (size_t)(h - 1 - r) * w * 3;
        for (uint32_t x = 0; x < w; x++) row[3 * x] = src[3 * x + 2], row[3 * x + 1] = src[3 * x + 1], row[3 * x + 2] = src[3 * x];
        std::fwrite(row.data(), 1, stride, f);
    }
    return std::fclose(f) == 0;
}
static bool has_suffix(const char *path, const char *suffix) {
    const size_t n = std::strlen(path), m = std::strlen(suffix);
    return n >= m && std::strcmp(path + n - m, suffix) == 0;
}

// `encode` / `encode-file`: the .frv comes from the symbol stream route (FRIEncoder::encode_bytes_streamed: the emitter's gather on the device, 2 bytes per symbol
// over PCIe) - the default since round 4. Self-checks: the array route (stage functions one by one, 9 bytes per node over PCIe, gather on the host) must give
// the same bytes - it does bit for bit since the fit's W^T r sums are fixed-point integers (k4_fit.hip): both routes fit the same parameters - ; the container
// parses and every symbol decodes; FRIDecoder::decode returns the input.
static int encode_image_to_file(std::vector<uint8_t> img, uint32_t w, uint32_t h, uint32_t c, const libfri::EncoderOpts &opts, const char *out_path) {
    const libfri::ColorSpace cs = c == 1 ? libfri::ColorSpace::Luma : libfri::ColorSpace::RGB;
    auto t0 = std::chrono::steady_clock::now();
    libfri::FRIEncoder streamed_encoder(opts);
    auto streamed = streamed_encoder.encode_bytes_streamed(img, h, w, cs);
    const double t_streamed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!streamed.ok) {
        std::fprintf(stderr, "%s\n", streamed.error.c_str());
        return 1;
    }
    const std::vector<uint8_t> &bytes = streamed.value;
    // the array route
    libfri::FRIEncoder encoder(opts);
    t0 = std::chrono::steady_clock::now();
    auto st = encoder.encode(img, h, w, cs);
    if (!st.ok) {
        std::fprintf(stderr, "%s\n", st.error.c_str());
        return 1;
    }
    const double t_dev = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    t0 = std::chrono::steady_clock::now();
    auto comp = libfri::stages::entropy_coding::encode(st.value.image, st.value.contexts, encoder.opts());
    if (!comp.ok) {
        std::fprintf(stderr, "%s\n", comp.error.c_str());
        return 1;
    }
    const bool same = libfri::stages::serialize::encode(comp.value) == bytes;
    const double t_host = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!same) {
        std::fprintf(stderr, "self-check failed: the array route gives different bytes\n");
        return 1;
    }
    // parse the container, rebuild the models from it, decode every symbol
    libfri::emit::ParsedImage parsed;
    std::string err = libfri::emit::deserialize(bytes, parsed);
    const size_t plane = (size_t)st.value.image.num_cells * 512;
    const auto order_ptr = libfri::emit::shared_symbol_order(st.value.image.centers.data(), st.value.image.num_cells);
    const libfri::emit::SymbolOrder &order = *order_ptr;
    for (uint32_t ch = 0; err.empty() && ch < c; ch++) {
        std::vector<uint16_t> want, got;
        std::vector<uint8_t> buckets;
        libfri::emit::channel_symbols(order, st.value.image.coefficients.data() + ch * plane,
                                      st.value.image.bucket_of(ch), st.value.image.prediction_of(ch), want, buckets);
        err = libfri::emit::decode_symbols(parsed.channels[ch], buckets, got);
        if (err.empty() && got != want) err = "decoded symbols differ";
    }
    if (!err.empty()) {
        std::fprintf(stderr, "self-check failed: %s\n", err.c_str());
        return 1;
    }
    // and the whole way back like FRIDecoder::decode (decoder.rs:47-59): every context recomputed from the symbols decoded so far
    t0 = std::chrono::steady_clock::now();
    auto back = libfri::FRIDecoder().decode(bytes, opts);
    const double t_dec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!back.ok || back.value.data != img) {
        std::fprintf(stderr, "self-check failed: %s\n", back.ok ? "decoded image differs from the input" : back.error.c_str());
        return 1;
    }
    if (FILE *f = std::fopen(out_path, "wb")) {
        std::fwrite(bytes.data(), 1, bytes.size(), f);
        std::fclose(f);
    } else {
        std::fprintf(stderr, "cannot write %s\n", out_path);
        return 1;
    }
    std::printf("%ux%ux%u: %zu bytes, %.3f bits per pixel; symbol stream route end to end (context, plan, stream order, chain, emit) %.3f s; self-checks: array route, device stages (incl. plan + PCIe) %.3f s + host emit %.3f s: same bytes; decoded back in %.3f s: lossless\n", w, h, c,
                bytes.size(), 8.0 * bytes.size() / ((double)w * h), t_streamed, t_dev, t_host, t_dec);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 4 && std::string(argv[1]) == "encode-file") {
        std::vector<uint8_t> img;
        uint32_t fw = 0, fh = 0, fc = 0;
        std::string err;
        if (!(has_suffix(argv[2], ".bmp") ? read_bmp(argv[2], img, fw, fh, fc, err) : read_pnm(argv[2], img, fw, fh, fc, err))) {
            std::fprintf(stderr, "%s\n", err.c_str());
            return 1;
        }
        libfri::EncoderOpts file_opts; // parameters are fitted on the device sums (fit_parameters defaults to true)
        return encode_image_to_file(std::move(img), fw, fh, fc, file_opts, argv[3]);
    }
    if (argc >= 4 && std::string(argv[1]) == "decode-file") {
        std::vector<uint8_t> bytes;
        if (FILE *f = std::fopen(argv[2], "rb")) {
            uint8_t buf[1 << 16];
            for (size_t n; (n = std::fread(buf, 1, sizeof buf, f)) > 0;) bytes.insert(bytes.end(), buf, buf + n);
            std::fclose(f);
        } else {
            std::fprintf(stderr, "cannot open %s\n", argv[2]);
            return 1;
        }
        auto img = libfri::FRIDecoder().decode(bytes);
        if (!img.ok) {
            std::fprintf(stderr, "%s\n", img.error.c_str());
            return 1;
        }
        const uint32_t ch = libfri::num_channels(img.value.metadata.colorspace);
        if (has_suffix(argv[3], ".bmp")) {
            if (ch != 3 || !write_bmp(argv[3], img.value.data, img.value.metadata.width, img.value.metadata.height)) {
                std::fprintf(stderr, "cannot write %s%s\n", argv[3], ch != 3 ? " (a BMP takes an RGB image)" : "");
                return 1;
            }
            std::printf("%ux%ux%u decoded\n", img.value.metadata.width, img.value.metadata.height, ch);
            return 0;
        }
        FILE *f = std::fopen(argv[3], "wb");
        if (!f) {
            std::fprintf(stderr, "cannot write %s\n", argv[3]);
            return 1;
        }
        std::fprintf(f, "%s\n%u %u\n255\n", ch == 1 ? "P5" : "P6", img.value.metadata.width, img.value.metadata.height);
        std::fwrite(img.value.data.data(), 1, img.value.data.size(), f);
        std::fclose(f);
        std::printf("%ux%ux%u decoded\n", img.value.metadata.width, img.value.metadata.height, ch);
        return 0;
    }
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s roundtrip|encode|batch|batch-frv <width> <height> <channels> [n_images | out.frv]\n       %s encode-file <in.pgm|in.ppm|in.bmp> <out.frv>\n       %s decode-file <in.frv> <out.pgm|out.ppm|out.bmp>\n", argv[0], argv[0], argv[0]);
        return 2;
    }
    const std::string cmd = argv[1];
    const uint32_t w = (uint32_t)std::atoi(argv[2]), h = (uint32_t)std::atoi(argv[3]), c = (uint32_t)std::atoi(argv[4]);
    const libfri::ColorSpace cs = c == 1 ? libfri::ColorSpace::Luma : libfri::ColorSpace::RGB;
    libfri::EncoderOpts opts;
    for (int ch = 0; ch < 3; ch++)
        for (int g = 0; g < 3; g++) {
            opts.value_prediction_params[ch][g] = {0.25f, 0.25f, 0.25f, 0.125f, 0.0625f, 0.0625f};
            opts.width_prediction_params[ch][g] = {1.0f, 0.5f, 0.25f, 0.25f, 0.125f, 0.125f};
        }
    if (cmd == "roundtrip") {
        std::vector<uint8_t> img = noise_image(w, h, c, 0);
        auto enc = libfri::FRIEncoder(opts).encode(img, h, w, cs);
        if (!enc.ok) {
            std::fprintf(stderr, "%s\n", enc.error.c_str());
            return 1;
        }
        uint64_t total = 0;
        for (auto &ctx : enc.value.contexts[0])
            for (uint32_t f : ctx.freqs) total += f;
        auto dec = libfri::FRIDecoder().decode(enc.value.image, opts);
        if (!dec.ok) {
            std::fprintf(stderr, "%s\n", dec.error.c_str());
            return 1;
        }
        const bool same = dec.value.data == img;
        std::printf("cells=%u hist_total_ch0=%llu lossless=%s (parameters fitted on the device sums)\n", enc.value.image.num_cells, (unsigned long long)total,
                    same ? "yes" : "NO");
        return same ? 0 : 1;
    }
    if (cmd == "encode") {
        if (argc < 6) {
            std::fprintf(stderr, "usage: %s encode <width> <height> <channels> <out.frv>\n", argv[0]);
            return 2;
        }
        // left half smooth, right half noise (SURVEY.md section 8d generators): fills all ten ANS contexts; pure noise leaves some
        // empty at small sizes, and libfri panics on an empty context
        std::vector<uint8_t> img = noise_image(w, h, c, 0);
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w / 2; x++)
                for (uint32_t k = 0; k < c; k++) img[((size_t)y * w + x) * c + k] = (uint8_t)((((x + 2 * y) >> 3) + (img[((size_t)y * w + x) * c + k] & 7)) & 0xFF);
        return encode_image_to_file(std::move(img), w, h, c, opts, argv[5]);
    }
    if (cmd == "batch-frv") { // n images -> n .frv byte strings, device chains and host emits pipelined (libfri::encode_batch_bytes)
        const uint32_t n = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 16;
        uint32_t gpus = 1, emitters = std::max(1u, std::thread::hardware_concurrency() / 4);
        bool same_device = false;
        for (int i = 6; i < argc; i++) {
            if (std::string(argv[i]) == "--gpus" && i + 1 < argc) gpus = (uint32_t)std::atoi(argv[i + 1]);
            if (std::string(argv[i]) == "--emitters" && i + 1 < argc) emitters = (uint32_t)std::atoi(argv[i + 1]);
            if (std::string(argv[i]) == "--same-device") same_device = true; // every "GPU" is device 0: the pipeline's threading on a one-GPU box
        }
        if (!n || !gpus || gpus > 64 || !emitters || emitters > 256) {
            std::fprintf(stderr, "usage: %s batch-frv <width> <height> <channels> <n_images> [--gpus N] [--emitters T] [--same-device]\n", argv[0]);
            return 2;
        }
        // the devices live for the whole run (contexts, plans and the plans' symbol order are built once, by the warm-up): the timed call is the steady state of a
        // service that encodes batch after batch
        std::vector<std::unique_ptr<libfri::Device>> owned;
        std::vector<libfri::Device *> devices;
        for (uint32_t d = 0; d < gpus; d++) {
            owned.emplace_back(new libfri::Device(same_device ? 0 : (int)d));
            owned.back()->measure_forward_tiling(true); // a service's plans are made once: they measure their forward tiling (fri_hip_plan_tune_forward)
            devices.push_back(owned.back().get());
        }
        const uint32_t distinct = n < 4 ? n : 4;
        std::vector<std::vector<uint8_t>> in(distinct);
        for (uint32_t i = 0; i < distinct; i++) { // left half smooth, right half noise, as `encode`: every context is populated
            in[i] = noise_image(w, h, c, i);
            for (uint32_t y = 0; y < h; y++)
                for (uint32_t x = 0; x < w / 2; x++)
                    for (uint32_t k = 0; k < c; k++) in[i][((size_t)y * w + x) * c + k] = (uint8_t)((((x + 2 * y + 5 * i) >> 3) + (in[i][((size_t)y * w + x) * c + k] & 7)) & 0xFF);
        }
        std::vector<const uint8_t *> pin(n);
        for (uint32_t i = 0; i < n; i++) pin[i] = in[i % distinct].data();
        libfri::EncoderOpts fit_opts; // parameters fitted per image on the device
        {   // warm-up: contexts, plans, stream order, pinned staging, the emitter's cached symbol order
            std::vector<const uint8_t *> few(pin.begin(), pin.begin() + std::min<size_t>(n, gpus));
            auto warm = libfri::encode_batch_bytes(few, h, w, cs, fit_opts, devices, emitters);
            if (!warm.ok) {
                std::fprintf(stderr, "%s\n", warm.error.c_str());
                return 1;
            }
        }
        libfri::BatchStats stats;
        auto out = libfri::encode_batch_bytes(pin, h, w, cs, fit_opts, devices, emitters, &stats);
        if (!out.ok) {
            std::fprintf(stderr, "%s\n", out.error.c_str());
            return 1;
        }
        size_t total = 0;
        for (uint32_t i = 0; i < n; i++) {
            total += out.value[i].size();
            if (i >= distinct && out.value[i] != out.value[i % distinct]) {
                std::fprintf(stderr, "image %u: bytes differ from image %u (same input)\n", i, i % distinct);
                return 1;
            }
        }
        // one image through the single-image API, and back: the batch's bytes are FRIEncoder's, and they decode to the input
        auto single = libfri::FRIEncoder(fit_opts).encode_bytes_streamed(in[0], h, w, cs);
        if (!single.ok || single.value != out.value[0]) {
            std::fprintf(stderr, "self-check failed: image 0 of the batch differs from FRIEncoder::encode_bytes_streamed\n");
            return 1;
        }
        auto back = libfri::FRIDecoder().decode(out.value[0], fit_opts);
        if (!back.ok || back.value.data != in[0]) {
            std::fprintf(stderr, "self-check failed: image 0 does not decode to its input\n");
            return 1;
        }
        std::printf("batch-frv %u x %ux%ux%u on %u device thread(s)%s + %u emitter thread(s): %.3f s = %.1f images/s = %.1f Mpixels/s pixels-to-.frv (PCIe and host rANS inclusive); "
                    "summed over images: device calls %.3f s, host emits %.3f s; %.3f bits per pixel; image 0 = the single-image API's bytes, decodes losslessly\n",
                    n, w, h, c, gpus, same_device ? " (all on device 0)" : "", emitters, stats.seconds, n / stats.seconds, (double)n * w * h / stats.seconds / 1e6, stats.device_seconds,
                    stats.emit_seconds, 8.0 * total / ((double)n * w * h));
        return 0;
    }
    if (cmd == "batch") {
        const uint32_t n = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 16;
        uint32_t gpus = 1;
        for (int i = 6; i + 1 < argc; i++)
            if (std::string(argv[i]) == "--gpus") gpus = (uint32_t)std::atoi(argv[i + 1]);
        if (!gpus || gpus > 64) {
            std::fprintf(stderr, "--gpus must be 1..64\n");
            return 2;
        }
        std::vector<int> devices(gpus);
        for (uint32_t d = 0; d < gpus; d++) devices[d] = (int)d;
        fri_hip_multi *multi = nullptr;
        if (int rc = fri_hip_multi_create(devices.data(), gpus, w, h, c, &multi)) {
            std::fprintf(stderr, "%s\n", fri_hip_strerror(rc));
            return 1;
        }
        const size_t coef_count = fri_hip_plan_coef_count(fri_hip_multi_plan(multi, 0));
        const uint32_t distinct = n < 8 ? n : 8; // a few distinct inputs, every image gets its own output
        std::vector<std::vector<uint8_t>> in(distinct);
        for (uint32_t i = 0; i < distinct; i++) in[i] = noise_image(w, h, c, i);
        std::vector<std::vector<int32_t>> out(n, std::vector<int32_t>(coef_count));
        std::vector<const uint8_t *> pin(n);
        std::vector<int32_t *> pout(n);
        for (uint32_t i = 0; i < n; i++) {
            pin[i] = in[i % distinct].data();
            pout[i] = out[i].data();
        }
        std::vector<int32_t> q(32, 1);
        bool chain = false;
        for (int i = 6; i < argc; i++) chain = chain || std::string(argv[i]) == "--chain";
        if (chain) { // the whole encoder per image (K1 -> fit -> K2), sharded the same way: fri_hip_multi_encode_image
            const size_t C = c;
            std::vector<std::vector<float>> par(n, std::vector<float>(C * 36));
            std::vector<std::vector<uint32_t>> hist(n, std::vector<uint32_t>(C * 10 * 1024));
            std::vector<std::vector<uint64_t>> oob(n, std::vector<uint64_t>(C));
            std::vector<float *> ppar(n);
            std::vector<uint32_t *> phist(n);
            std::vector<uint64_t *> poob(n);
            for (uint32_t i = 0; i < n; i++) ppar[i] = par[i].data(), phist[i] = hist[i].data(), poob[i] = oob[i].data();
            int rc = fri_hip_multi_encode_image(multi, n < 3 * gpus ? n : 3 * gpus, pin.data(), q.data(), 1, ppar.data(), pout.data(), nullptr, nullptr, phist.data(), poob.data());
            auto t0 = std::chrono::steady_clock::now();
            if (rc == FRI_HIP_OK) rc = fri_hip_multi_encode_image(multi, n, pin.data(), q.data(), 1, ppar.data(), pout.data(), nullptr, nullptr, phist.data(), poob.data());
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const uint64_t some = fri_hip_plan_num_some(fri_hip_multi_plan(multi, 0));
            fri_hip_multi_destroy(multi);
            if (rc != FRI_HIP_OK) {
                std::fprintf(stderr, "%s\n", fri_hip_strerror(rc));
                return 1;
            }
            for (uint32_t i = 0; i < n; i++) {
                for (size_t ch = 0; ch < C; ch++) {
                    uint64_t total = oob[i][ch];
                    for (size_t k = 0; k < 10 * 1024; k++) total += hist[i][ch * 10 * 1024 + k];
                    if (total != some) {
                        std::fprintf(stderr, "image %u channel %zu: histogram total %llu, expected %llu\n", i, ch, (unsigned long long)total, (unsigned long long)some);
                        return 1;
                    }
                }
                if (i >= distinct && (out[i] != out[i % distinct] || hist[i] != hist[i % distinct])) {
                    std::fprintf(stderr, "image %u differs from image %u (same input)\n", i, i % distinct);
                    return 1;
                }
            }
            std::printf("batch --chain %u x %ux%ux%u on %u GPU(s) (image i -> GPU i mod %u): whole encoder chain with the fit, %.3f s, %.1f Mpixels/s host-to-host (PCIe inclusive)\n", n, w,
                        h, c, gpus, gpus, s, (double)n * w * h / s / 1e6);
            return 0;
        }
        // warm-up pass on a few images per GPU (allocates the pinned staging), then the timed pass
        int rc = fri_hip_multi_transform_quant(multi, n < 3 * gpus ? n : 3 * gpus, pin.data(), q.data(), pout.data());
        auto t0 = std::chrono::steady_clock::now();
        if (rc == FRI_HIP_OK) rc = fri_hip_multi_transform_quant(multi, n, pin.data(), q.data(), pout.data());
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        fri_hip_multi_destroy(multi);
        if (rc != FRI_HIP_OK) {
            std::fprintf(stderr, "%s\n", fri_hip_strerror(rc));
            return 1;
        }
        // images that share an input must have produced identical coefficients, whichever GPU they ran on
        for (uint32_t i = distinct; i < n; i++)
            if (out[i] != out[i % distinct]) {
                std::fprintf(stderr, "image %u differs from image %u (same input)\n", i, i % distinct);
                return 1;
            }
        std::printf("batch %u x %ux%ux%u on %u GPU(s) (image i -> GPU i mod %u): %.3f s, %.1f Mpixels/s host-to-host (PCIe inclusive)\n", n, w, h, c, gpus, gpus,
                    s, (double)n * w * h / s / 1e6);
        return 0;
    }
    std::fprintf(stderr, "unknown command %s\n", cmd.c_str());
    return 2;
}
