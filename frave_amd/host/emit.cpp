// emit.cpp -- see emit.hpp. Citations are relative to /root/reference/crates/libfri/src/.
#include "emit.hpp"

#include <atomic>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <memory>
#include <mutex>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

#include "../csrc/geometry.hpp"

namespace libfri {
namespace emit {

// ---- symbol order ---------------------------------------------------------------------------------------------------
std::vector<uint32_t> symbol_order(const int32_t *centers, uint32_t n_cells, int level) {
    if (level < 0 || level >= kDepth || !n_cells) return {};
    const fri::StaticTables &st = fri::static_tables();
    fri::Int2 nv[6];
    fri::nearby_vectors(kDepth - level, nv);
    const fri::Int2 row = nv[3], col = nv[1];
    // normal of the scan lines, pointing the way the reference advances from line to line (row_dir, :513-517, :619-628)
    int64_t nx = -col.y, ny = col.x;
    if ((int64_t)row.x * nx + (int64_t)row.y * ny < 0) nx = -nx, ny = -ny;
    // offset of node i of `level` from its cell centre (Fractal::new, :42-69): child 2p+1 of a level-v node adds LITERALS[8 - v]
    const uint32_t per_cell = 1u << level;
    std::vector<fri::Int2> node_off(per_cell);
    for (uint32_t i = 0; i < per_cell; i++) {
        fri::Int2 o{0, 0};
        for (int j = 0; j < level; j++)
            if (i >> j & 1) o = {o.x + st.literals[kDepth - level + j].x, o.y + st.literals[kDepth - level + j].y};
        node_off[i] = o;
    }
    // key = (line, along), both shifted to start at 0 and packed into 64 bits; LSD radix sort of (key, id) pairs - the order is
    // geometry only but 8.7 M entries at level 8 of a 4096^2 image make a comparison sort the slowest part of the whole emitter
    const size_t n = (size_t)n_cells * per_cell;
    std::vector<int64_t> line(n), along(n);
    int64_t lmin = INT64_MAX, lmax = INT64_MIN, amin = INT64_MAX, amax = INT64_MIN;
    for (uint32_t c = 0; c < n_cells; c++)
        for (uint32_t i = 0; i < per_cell; i++) {
            const int64_t x = (int64_t)centers[2 * c] + node_off[i].x, y = (int64_t)centers[2 * c + 1] + node_off[i].y;
            const int64_t l = x * nx + y * ny, a = x * col.x + y * col.y;
            line[(size_t)c * per_cell + i] = l, along[(size_t)c * per_cell + i] = a;
            lmin = std::min(lmin, l), lmax = std::max(lmax, l), amin = std::min(amin, a), amax = std::max(amax, a);
        }
    int abits = 1, lbits = 1;
    while (((uint64_t)(amax - amin) >> abits) != 0) abits++;
    while (((uint64_t)(lmax - lmin) >> lbits) != 0) lbits++;
    std::vector<uint64_t> key(n), key2(n);
    std::vector<uint32_t> id(n), id2(n);
    for (uint32_t c = 0; c < n_cells; c++)
        for (uint32_t i = 0; i < per_cell; i++) {
            const size_t k = (size_t)c * per_cell + i;
            key[k] = (uint64_t)(line[k] - lmin) << abits | (uint64_t)(along[k] - amin);
            id[k] = c << 9 | (per_cell + i);
        }
    line.clear(), line.shrink_to_fit(), along.clear(), along.shrink_to_fit();
    constexpr int kDigit = 11;
    for (int shift = 0; shift < abits + lbits; shift += kDigit) {
        size_t count[(1 << kDigit) + 1] = {};
        for (size_t k = 0; k < n; k++) count[((key[k] >> shift) & ((1u << kDigit) - 1)) + 1]++;
        for (int d = 0; d < (1 << kDigit); d++) count[d + 1] += count[d];
        for (size_t k = 0; k < n; k++) {
            const size_t at = count[(key[k] >> shift) & ((1u << kDigit) - 1)]++;
            key2[at] = key[k], id2[at] = id[k];
        }
        key.swap(key2), id.swap(id2);
    }
    return id;
}

std::shared_ptr<const SymbolOrder> shared_symbol_order(const int32_t *centers, uint32_t n_cells) {
    // The order depends on the cell centres alone, i.e. on (width, height): every image of a size shares one. A small process-wide
    // cache keyed by the centres themselves (hash + full compare); 68 MB per 4096^2 entry, four entries.
    struct Entry {
        uint64_t hash;
        std::vector<int32_t> centers;
        std::shared_ptr<const SymbolOrder> order;
    };
    static std::mutex mu;
    static std::vector<Entry> cache;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < (size_t)n_cells * 2; i++) h = (h ^ (uint32_t)centers[i]) * 1099511628211ull;
    {
        std::lock_guard<std::mutex> lock(mu);
        for (size_t i = 0; i < cache.size(); i++)
            if (cache[i].hash == h && cache[i].centers.size() == (size_t)n_cells * 2 && std::equal(cache[i].centers.begin(), cache[i].centers.end(), centers)) {
                Entry e = std::move(cache[i]); // most recently used last
                cache.erase(cache.begin() + (long)i);
                cache.push_back(std::move(e));
                return cache.back().order;
            }
    }
    auto order = std::make_shared<const SymbolOrder>(centers, n_cells); // built outside the lock: two threads may build the same order once
    std::lock_guard<std::mutex> lock(mu);
    if (cache.size() >= 4) cache.erase(cache.begin());
    cache.push_back(Entry{h, std::vector<int32_t>(centers, centers + (size_t)n_cells * 2), order});
    return order;
}

SymbolOrder::SymbolOrder(const int32_t *centers, uint32_t n_cells) {
    // the levels are independent; level l holds 2^l entries per cell, so levels 8 and 7 get threads of their own
    std::thread t8([&] { level[8] = symbol_order(centers, n_cells, 8); });
    std::thread t7([&] { level[7] = symbol_order(centers, n_cells, 7); });
    for (int l = 0; l < 7; l++) level[l] = symbol_order(centers, n_cells, l);
    t7.join();
    t8.join();
}

// ---- small helpers ----------------------------------------------------------------------------------------------------
float width_from_bucket(int bucket) {
    static const float w[kContexts] = {2.5f, 4.5f, 6.3f, 8.5f, 12.7f, 16.f, 20.f, 24.f, 28.f, 36.f};
    return bucket >= 0 && bucket < kContexts ? w[bucket] : 50.f;
}
uint32_t pack_signed(int32_t k) { return ((uint32_t)k << 1) ^ (uint32_t)(k >> 31); }
int32_t unpack_signed(uint32_t k) { return (k & 1u) ? (int32_t)(k + 1) / -2 : (int32_t)(k / 2); }

namespace {
// utils.rs:5-14 (the or-cascade stops at 16: exact below 2^32, which is all a u32 sum can reach)
uint64_t prev_power_two(uint64_t x) {
    uint64_t n = x;
    n |= n >> 1, n |= n >> 2, n |= n >> 4, n |= n >> 8, n |= n >> 16;
    return n ^ (n >> 1);
}
uint32_t trailing_zeros64(uint64_t x) { return x ? (uint32_t)__builtin_ctzll(x) : 64u; }
// `1 << bits` on a 32-bit integer in a release build: the shift amount wraps modulo 32 (debug builds panic from 32 up)
uint32_t shl1_release(uint32_t bits) { return 1u << (bits & 31u); }
// Rust `f32 as u32`: truncating, saturating, NaN -> 0
uint32_t f32_as_u32(float v) {
    if (!(v > 0.0f)) return 0;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
} // namespace

// ---- AnsContext -----------------------------------------------------------------------------------------------------
std::string AnsContext::finalize(int bucket) {
    if (max_freq_bits < 8) max_freq_bits = 8; // :103-105
    // fill_with_laplace, :82-96. f32 throughout; `exp` is the platform libm's expf, as it is for the reference.
    const float width = width_from_bucket(bucket);
    const float scale = (float)(int32_t)shl1_release(max_freq_bits);
    for (int j = 0; j < kAlphabet; j++) {
        const float x = (float)unpack_signed((uint32_t)j);
        const float lap = std::exp(-std::fabs(x - 0.0f) / width) / (2.0f * width); // prediction.rs:219-221
        const uint32_t lv = f32_as_u32(lap * scale);
        uint32_t &f = freqs[j];
        if (lv == 0 && f == 0 && std::find(off_distribution_values.begin(), off_distribution_values.end(), (uint16_t)j) != off_distribution_values.end()) {
            f = 1;
        } else if (f != 0 && lv == 0) {
            f = 1;
            off_distribution_values.push_back((uint16_t)j);
        } else {
            f = lv;
        }
    }
    // normalize_freqs(1 << max_freq_bits), :119-159
    const uint32_t target = shl1_release(max_freq_bits);
    std::array<uint32_t, kAlphabet> cum{};
    {
        uint32_t acc = 0;
        for (int i = 0; i < kAlphabet; i++) {
            cum[i] = acc;
            acc += freqs[i];
        }
    }
    const uint32_t cur_total = cum[kAlphabet - 1] + freqs[kAlphabet - 1];
    if (cur_total == 0) return "empty context: libfri divides by zero here (entropy_coding.rs:123)";
    for (int i = 1; i < kAlphabet; i++) cum[i] = (uint32_t)(((uint64_t)target * cum[i]) / cur_total);
    for (int i = 0; i < kAlphabet - 1; i++) { // a used symbol whose slot collapsed steals one count from the smallest slot > 1
        if (freqs[i] != 0 && cum[i + 1] == cum[i]) {
            uint32_t best_freq = 0xFFFFFFFFu;
            int best = -1;
            for (int j = 0; j < kAlphabet - 1; j++) {
                const uint32_t f = cum[j + 1] - cum[j];
                if (f > 1 && f < best_freq) best_freq = f, best = j;
            }
            if (best < 0) continue;
            if (best < i)
                for (int j = best + 1; j <= i; j++) cum[j] -= 1;
            else
                for (int j = i + 1; j <= best; j++) cum[j] += 1;
        }
    }
    for (int i = 0; i < kAlphabet - 1; i++) freqs[i] = cum[i + 1] - cum[i];
    freqs[kAlphabet - 1] = cum[kAlphabet - 1] - target; // :156 as written (wraps in a release build unless the last slot is empty)
    cdf = cum;
    uint64_t sum = 0; // `iter().sum::<u32>()` wraps in a release build
    for (uint32_t f : freqs) sum = (uint32_t)(sum + f);
    max_freq_bits = trailing_zeros64(prev_power_two(sum)); // :113-114
    return "";
}

// ---- rANS (ryg_rans rans64: state in [2^31, 2^63), 32-bit renormalisation) ----------------------------------------------
void RansEncoderMulti::put_at(int s, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    uint64_t x = x_[s];
    const uint64_t x_max = ((((uint64_t)1 << 31) >> scale_bits) << 32) * freq;
    if (x >= x_max) {
        rev_.push_back((uint32_t)x);
        x >>= 32;
    }
    x_[s] = ((x / freq) << scale_bits) + (x % freq) + start;
}
// ryg_rans' Rans64EncSymbol: the division and the remainder of put_at replaced by a multiplication with a precomputed 64-bit
// reciprocal (rans64.h, Rans64EncSymbolInit / Rans64EncPutSymbol). Exact: q below equals x / freq for every state x < 2^63, so
// x + bias + q * cmpl_freq is the same new state as ((x / freq) << scale_bits) + x % freq + start.
RansEncoderMulti::EncSymbol RansEncoderMulti::make_symbol(uint32_t start, uint32_t freq, uint32_t scale_bits) {
    EncSymbol s;
    s.freq = freq;
    s.cmpl_freq = (uint32_t)((1ull << scale_bits) - freq);
    s.x_max = ((((uint64_t)1 << 31) >> scale_bits) << 32) * freq;
    if (freq < 2) { // q = mul_hi(x, ~0) = x - 1 for x > 0; new state = x + start + (2^scale - 1) + (x - 1) * (2^scale - 1) = (x << scale) + start
        s.rcp_freq = ~0ull;
        s.rcp_shift = 0;
        s.bias = start + (uint32_t)((1ull << scale_bits) - 1);
    } else {
        uint32_t shift = 0;
        while (freq > (1u << shift)) shift++;
        // ceil(2^(shift + 63) / freq) by long division in two 32-bit digits
        uint64_t x0 = freq - 1;
        const uint64_t x1 = 1ull << (shift + 31);
        const uint64_t t1 = x1 / freq;
        x0 += (x1 % freq) << 32;
        const uint64_t t0 = x0 / freq;
        s.rcp_freq = t0 + (t1 << 32);
        s.rcp_shift = shift - 1;
        s.bias = start;
    }
    return s;
}
void RansEncoderMulti::flush_all() {
    for (int s = 0; s < kContexts; s++) { // each flush prepends two words: state kContexts-1 ends up first in the stream
        rev_.push_back((uint32_t)(x_[s] >> 32));
        rev_.push_back((uint32_t)x_[s]);
    }
}
std::vector<uint8_t> RansEncoderMulti::data() const {
    std::vector<uint8_t> out(rev_.size() * 4);
    for (size_t i = 0; i < rev_.size(); i++) {
        const uint32_t w = rev_[rev_.size() - 1 - i];
        out[4 * i] = (uint8_t)w, out[4 * i + 1] = (uint8_t)(w >> 8), out[4 * i + 2] = (uint8_t)(w >> 16), out[4 * i + 3] = (uint8_t)(w >> 24);
    }
    return out;
}
RansDecoderMulti::RansDecoderMulti(const std::vector<uint8_t> &d) {
    w_.resize(d.size() / 4);
    for (size_t i = 0; i < w_.size(); i++) w_[i] = (uint32_t)d[4 * i] | (uint32_t)d[4 * i + 1] << 8 | (uint32_t)d[4 * i + 2] << 16 | (uint32_t)d[4 * i + 3] << 24;
    for (int s = 0; s < kContexts; s++) {
        if (pos_ + 2 > w_.size()) {
            ok_ = false;
            x_[s] = 1ull << 31;
            continue;
        }
        x_[s] = (uint64_t)w_[pos_] | (uint64_t)w_[pos_ + 1] << 32;
        pos_ += 2;
    }
}
uint32_t RansDecoderMulti::get_at(int s, uint32_t scale_bits) const { return (uint32_t)(x_[s] & (((uint64_t)1 << scale_bits) - 1)); }
void RansDecoderMulti::advance_at(int s, uint32_t start, uint32_t freq, uint32_t scale_bits) {
    const uint64_t mask = ((uint64_t)1 << scale_bits) - 1;
    uint64_t x = x_[s];
    x = (uint64_t)freq * (x >> scale_bits) + (x & mask) - start;
    if (x < (1ull << 31)) {
        if (pos_ < w_.size())
            x = (x << 32) | w_[pos_++];
        else
            ok_ = false;
    }
    x_[s] = x;
}

// ---- one channel -------------------------------------------------------------------------------------------------------
void channel_symbols(const SymbolOrder &order, const int32_t *coefs, const uint8_t *bucket, const int32_t *prediction, std::vector<uint16_t> &symbols,
                     std::vector<uint8_t> &buckets) {
    // The stream is the concatenation of ten scans (DC, root, levels 1..8), each a gather through the order. The gather is cut into
    // chunks that threads fill on their own (a None coefficient yields no symbol, so a chunk's output length is known only afterwards),
    // then the chunks are laid end to end: same sequence as the reference's single loop (entropy_coding.rs:285-330).
    struct Scan {
        const uint32_t *list;
        size_t n;
        int fixed_heap; // >= 0: every entry is a cell whose node `fixed_heap` is taken (the two level-0 scans)
    };
    std::vector<Scan> scans;
    scans.push_back({order.level[0].data(), order.level[0].size(), 0});
    scans.push_back({order.level[0].data(), order.level[0].size(), 1});
    for (int level = 1; level < kDepth; level++) scans.push_back({order.level[level].data(), order.level[level].size(), -1});
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t max_threads = hw ? std::min(hw, 8u) : 4u;
    auto parallel_for = [&](size_t n_items, auto &&body) { // body(i) for i < n_items, items handed out one at a time
        const size_t n_threads = std::min(n_items, max_threads);
        if (n_threads <= 1) {
            for (size_t i = 0; i < n_items; i++) body(i);
            return;
        }
        std::atomic<size_t> next{0};
        std::vector<std::thread> workers;
        for (size_t t = 0; t < n_threads; t++)
            workers.emplace_back([&] {
                for (size_t i; (i = next.fetch_add(1)) < n_items;) body(i);
            });
        for (std::thread &t : workers) t.join();
    };
    // First a streaming pass that folds a node's three arrays into one word - bucket << 16 | symbol, all ones for a None - so that the
    // gather below takes one cache miss per symbol instead of three (the order jumps from cell to cell).
    size_t n_nodes = 0; // every retained cell is in the level-0 list
    for (uint32_t e : order.level[0]) n_nodes = std::max(n_nodes, ((size_t)(e >> 9) + 1) * kNodes);
    constexpr uint32_t kNoSymbol = 0xFFFFFFFFu;
    std::vector<uint32_t> packed(n_nodes);
    constexpr size_t kPackChunk = 1u << 18;
    parallel_for((n_nodes + kPackChunk - 1) / kPackChunk, [&](size_t i) {
        const size_t end = std::min(n_nodes, (i + 1) * kPackChunk);
        for (size_t at = i * kPackChunk; at < end; at++) {
            const uint32_t sym = pack_signed((int32_t)((uint32_t)coefs[at] - (uint32_t)prediction[at]));
            packed[at] = coefs[at] == kNone ? kNoSymbol : ((uint32_t)bucket[at] << 16) | std::min(sym, 0xFFFFu); // (a symbol >= 1024 is an error later on)
        }
    });
    constexpr size_t kChunk = 1u << 18;
    struct Chunk {
        size_t scan, begin, end;
        std::vector<uint16_t> sym;
        std::vector<uint8_t> bkt;
    };
    std::vector<Chunk> chunks;
    for (size_t sc = 0; sc < scans.size(); sc++)
        for (size_t b = 0; b < scans[sc].n; b += kChunk) chunks.push_back(Chunk{sc, b, std::min(scans[sc].n, b + kChunk), {}, {}});
    parallel_for(chunks.size(), [&](size_t i) {
        Chunk &c = chunks[i];
        const Scan &sc = scans[c.scan];
        c.sym.resize(c.end - c.begin);
        c.bkt.resize(c.end - c.begin);
        size_t n = 0;
        for (size_t k = c.begin; k < c.end; k++) {
            const uint32_t e = sc.list[k];
            const uint32_t v = packed[(size_t)(e >> 9) * kNodes + (sc.fixed_heap >= 0 ? (uint32_t)sc.fixed_heap : (e & 511u))];
            c.sym[n] = (uint16_t)v, c.bkt[n] = (uint8_t)(v >> 16);
            n += v != kNoSymbol; // `if let Some(value)`, entropy_coding.rs:288, :300, :317: a None yields no symbol
        }
        c.sym.resize(n);
        c.bkt.resize(n);
    });
    size_t total = 0;
    for (const Chunk &c : chunks) total += c.sym.size();
    symbols.resize(total);
    buckets.resize(total);
    size_t at = 0;
    for (const Chunk &c : chunks) {
        std::copy(c.sym.begin(), c.sym.end(), symbols.begin() + (long)at);
        std::copy(c.bkt.begin(), c.bkt.end(), buckets.begin() + (long)at);
        at += c.sym.size();
    }
}

// Highest index i in [lo, hi) with bytes[i] == value, SIZE_MAX if there is none: 16 bytes per step where SSE2 is there (x86-64 always).
static inline size_t prev_equal(const uint8_t *bytes, size_t lo, size_t hi, uint8_t value) {
#if defined(__SSE2__)
    const __m128i needle = _mm_set1_epi8((char)value);
    while (hi - lo >= 16) {
        const unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(bytes + hi - 16)), needle));
        if (m) return hi - 16 + (31u - (unsigned)__builtin_clz(m));
        hi -= 16;
    }
#endif
    while (hi > lo)
        if (bytes[--hi] == value) return hi;
    return SIZE_MAX;
}

// The symbols fed in reverse (entropy_coding.rs:332-334) into ten rANS states, one per context, that share one word stream. A state's
// history depends on its own symbols only, and a step emits at most one 32-bit word, so the coder runs context by context on
// threads of its own - each context notes AT WHICH symbols it emitted a word - and the shared stream is put together afterwards: the
// words in order of decreasing symbol index, then the flush. Byte for byte what the one-loop coder (`sequential`) produces.
std::string encode_symbols(const std::vector<uint16_t> &symbols, const std::vector<uint8_t> &buckets, const std::vector<RansEncoderMulti::EncSymbol> &tab,
                           std::vector<uint8_t> &data, bool sequential) {
    const size_t n = symbols.size();
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t n_threads = std::min<size_t>(kContexts, hw ? std::min(hw, 8u) : 4u);
    if (sequential || n < (1u << 16) || n >= (1ull << 32) || n_threads <= 1) {
        RansEncoderMulti enc;
        enc.reserve(n);
        for (size_t k = n; k-- > 0;) {
            const uint32_t sym = symbols[k];
            if (sym >= (uint32_t)kAlphabet) return "symbol outside the alphabet (libfri panics, entropy_coding.rs:99)";
            if (buckets[k] >= kContexts) return "bucket outside 0..9";
            const RansEncoderMulti::EncSymbol &e = tab[(size_t)buckets[k] * kAlphabet + sym];
            if (e.freq == 0) return "symbol with zero model frequency";
            enc.put_symbol(buckets[k], e);
        }
        enc.flush_all();
        data = enc.data();
        return "";
    }
    constexpr size_t kSpan = 1u << 16; // symbols per span of the stitching pass
    const size_t n_spans = (n + kSpan - 1) / kSpan;
    struct PerContext {
        std::vector<uint32_t> words;  // emitted words, in emission order (decreasing symbol index)
        std::vector<uint32_t> before; // [span]: words emitted before the coder entered the span (spans are entered last to first)
        uint64_t x = 1ull << 31;
        size_t n_symbols = 0, error_at = SIZE_MAX;
        const char *error = nullptr;
    };
    std::vector<PerContext> ctx(kContexts);
    std::vector<uint8_t> emitted(n, 0); // 1: the step of symbol k emitted a word
    auto run_context = [&](int b) {
        PerContext &c = ctx[(size_t)b];
        c.before.assign(n_spans, 0);
        const RansEncoderMulti::EncSymbol *t = tab.data() + (size_t)b * kAlphabet;
        uint64_t x = c.x;
        for (size_t sp = n_spans; sp-- > 0;) {
            c.before[sp] = (uint32_t)c.words.size();
            const size_t lo = sp * kSpan, hi = std::min(n, lo + kSpan);
            size_t k = hi;
            while (k > lo) {
                k = prev_equal(buckets.data(), lo, k, (uint8_t)b); // the context's next symbol below k
                if (k == SIZE_MAX) break;
                c.n_symbols++;
                const uint32_t sym = symbols[k];
                if (sym >= (uint32_t)kAlphabet) {
                    c.error_at = k, c.error = "symbol outside the alphabet (libfri panics, entropy_coding.rs:99)";
                    return;
                }
                const RansEncoderMulti::EncSymbol &e = t[sym];
                if (e.freq == 0) {
                    c.error_at = k, c.error = "symbol with zero model frequency";
                    return;
                }
                if (x >= e.x_max) {
                    c.words.push_back((uint32_t)x);
                    emitted[k] = 1;
                    x >>= 32;
                }
                const uint64_t q = (uint64_t)(((unsigned __int128)x * e.rcp_freq) >> 64) >> e.rcp_shift;
                x = x + e.bias + q * e.cmpl_freq;
            }
        }
        c.x = x;
    };
    {
        std::atomic<int> next{0};
        std::vector<std::thread> workers;
        for (size_t t = 0; t < n_threads; t++)
            workers.emplace_back([&] {
                for (int b; (b = next.fetch_add(1)) < kContexts;) run_context(b);
            });
        for (std::thread &t : workers) t.join();
    }
    // the error the one-loop coder would have met first: the one at the highest symbol index
    const PerContext *bad = nullptr;
    size_t seen = 0;
    for (const PerContext &c : ctx) {
        seen += c.n_symbols;
        if (c.error && (!bad || c.error_at > bad->error_at)) bad = &c;
    }
    if (bad) return bad->error;
    if (seen != n) return "bucket outside 0..9";
    // stitching: span sp's words start at position (words of all later spans) in emission order; inside a span the contexts' words
    // interleave as their symbols do
    std::vector<size_t> first(n_spans + 1, 0); // first[sp]: emission-order position of span sp's first word; spans are emitted last to first
    size_t total = 0;
    for (size_t sp = n_spans; sp-- > 0;) {
        first[sp] = total;
        for (const PerContext &c : ctx) total += (sp ? c.before[sp - 1] : c.words.size()) - c.before[sp];
    }
    const size_t n_words = total + 2 * kContexts;
    data.assign(n_words * 4, 0);
    auto put = [&](size_t emission_pos, uint32_t w) { // the stream is the emission order backwards (RansEncoderMulti::data)
        uint8_t *d = data.data() + 4 * (n_words - 1 - emission_pos);
        d[0] = (uint8_t)w, d[1] = (uint8_t)(w >> 8), d[2] = (uint8_t)(w >> 16), d[3] = (uint8_t)(w >> 24);
    };
    auto stitch = [&](size_t sp) {
        size_t cur[kContexts];
        for (int b = 0; b < kContexts; b++) cur[b] = ctx[(size_t)b].before[sp];
        size_t pos = first[sp];
        const size_t lo = sp * kSpan, hi = std::min(n, lo + kSpan);
        for (size_t k = hi; k > lo;) {
            k = prev_equal(emitted.data(), lo, k, 1);
            if (k == SIZE_MAX) break;
            const int b = buckets[k];
            put(pos++, ctx[(size_t)b].words[cur[b]++]);
        }
    };
    {
        std::atomic<size_t> next{0};
        std::vector<std::thread> workers;
        for (size_t t = 0; t < n_threads; t++)
            workers.emplace_back([&] {
                for (size_t sp; (sp = next.fetch_add(1)) < n_spans;) stitch(sp);
            });
        for (std::thread &t : workers) t.join();
    }
    size_t pos = total;
    for (int b = 0; b < kContexts; b++) { // flush_all: each flush prepends two words, state kContexts - 1 ends up first in the stream
        put(pos++, (uint32_t)(ctx[(size_t)b].x >> 32));
        put(pos++, (uint32_t)ctx[(size_t)b].x);
    }
    return "";
}

// The context-parallel rANS coder against the plain one-loop coder on pseudo-random symbols (see fri_emit.h).
int rans_selfcheck(uint64_t n_symbols, uint64_t seed, std::string &err) {
    if (n_symbols == 0 || n_symbols >= (1ull << 31)) return err = "invalid argument", -1;
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 0x2545F4914F6CDD1Dull;
    auto rnd = [&]() {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        return (uint32_t)(s >> 24);
    };
    std::vector<uint16_t> symbols(n_symbols);
    std::vector<uint8_t> buckets(n_symbols);
    std::vector<uint32_t> hist((size_t)kContexts * kAlphabet, 0);
    for (uint64_t k = 0; k < n_symbols; k++) {
        const uint32_t r = rnd();
        const int b = (int)((r >> 8) % 13u < 10u ? (r >> 8) % 13u : (r >> 12) % 3u); // uneven context sizes
        const uint32_t spread = 1u << (2 + b % 8);                                   // narrow and wide distributions
        const uint32_t sym = std::min<uint32_t>((rnd() % spread) + (rnd() % spread) * (rnd() & 1u), kAlphabet - 1);
        symbols[k] = (uint16_t)sym, buckets[k] = (uint8_t)b;
        hist[(size_t)b * kAlphabet + sym]++;
    }
    std::vector<RansEncoderMulti::EncSymbol> tab((size_t)kContexts * kAlphabet, RansEncoderMulti::EncSymbol{0, 0, 0, 0, 0, 0});
    for (int b = 0; b < kContexts; b++) {
        AnsContext c;
        uint64_t sum = 0;
        for (int j = 0; j < kAlphabet; j++) c.freqs[j] = hist[(size_t)b * kAlphabet + j], sum += c.freqs[j];
        if (sum == 0) continue; // (a context without symbols is never looked up)
        c.max_freq_bits = trailing_zeros64(prev_power_two(sum));
        const std::string e = c.finalize(b);
        if (!e.empty()) return err = "context " + std::to_string(b) + ": " + e, -2;
        for (int j = 0; j < kAlphabet; j++)
            if (c.freqs[j]) tab[(size_t)b * kAlphabet + j] = RansEncoderMulti::make_symbol(c.cdf[j], c.freqs[j], c.max_freq_bits);
    }
    std::vector<uint8_t> plain, parallel;
    std::string e = encode_symbols(symbols, buckets, tab, plain, true);
    if (!e.empty()) return err = "one-loop coder: " + e, -2;
    e = encode_symbols(symbols, buckets, tab, parallel, false);
    if (!e.empty()) return err = "context-parallel coder: " + e, -2;
    if (plain != parallel) return err = "the two coders' streams differ (" + std::to_string(plain.size()) + " / " + std::to_string(parallel.size()) + " bytes)", -4;
    return 0;
}

static std::string contexts_from_hist(const uint32_t *hist, ChannelStream &out);
std::string encode_channel(const SymbolOrder &order, const int32_t *coefs, const uint8_t *bucket, const int32_t *prediction, const uint32_t *hist,
                           ChannelStream &out) {
    const std::string cerr = contexts_from_hist(hist, out);
    if (!cerr.empty()) return cerr;
    std::vector<uint16_t> symbols;
    std::vector<uint8_t> buckets;
    channel_symbols(order, coefs, bucket, prediction, symbols, buckets);
    // per (context, symbol): the coder step with its division precomputed; a zero-frequency symbol keeps freq = 0 and is an error when met
    std::vector<RansEncoderMulti::EncSymbol> tab((size_t)kContexts * kAlphabet);
    for (int b = 0; b < kContexts; b++)
        for (int j = 0; j < kAlphabet; j++) {
            const AnsContext &c = out.contexts[b];
            tab[(size_t)b * kAlphabet + j] = c.freqs[j] ? RansEncoderMulti::make_symbol(c.cdf[j], c.freqs[j], c.max_freq_bits) : RansEncoderMulti::EncSymbol{0, 0, 0, 0, 0, 0};
        }
    const std::string err = encode_symbols(symbols, buckets, tab, out.data);
    if (!err.empty()) return err;
    out.n_symbols = symbols.size();
    return "";
}

// The channel's ANS models from the histogram K2 measured (prediction.rs:302-305)
static std::string contexts_from_hist(const uint32_t *hist, ChannelStream &out) {
    for (int b = 0; b < kContexts; b++) {
        AnsContext &c = out.contexts[b];
        uint64_t sum = 0;
        for (int j = 0; j < kAlphabet; j++) {
            c.freqs[j] = hist[b * kAlphabet + j];
            sum = (uint32_t)(sum + c.freqs[j]);
        }
        c.off_distribution_values.clear();
        c.max_freq_bits = trailing_zeros64(prev_power_two(sum));
        const std::string err = c.finalize(b);
        if (!err.empty()) return "context " + std::to_string(b) + ": " + err;
    }
    return "";
}

// The stream order with the None nodes taken out (which nodes are None is geometry, like the order itself): entry = cell << 9 | heap index of
// the i-th symbol of a channel - the DC scan, the root scan, then levels 1..8 (entropy_coding.rs:285-330). This is what the device's symbol
// stream kernel (fri_hip_symbol_stream_batch_dev) walks.
std::vector<uint32_t> stream_order(const SymbolOrder &order, const uint32_t *valid_mask /* [n_cells][16] */) {
    std::vector<uint32_t> out;
    size_t total = 2 * order.level[0].size();
    for (int level = 1; level < kDepth; level++) total += order.level[level].size();
    out.reserve(total);
    auto some = [&](uint32_t cell, uint32_t heap) { return (valid_mask[(size_t)cell * 16 + (heap >> 5)] >> (heap & 31)) & 1u; };
    for (uint32_t heap0 = 0; heap0 < 2; heap0++)
        for (uint32_t e : order.level[0])
            if (some(e >> 9, heap0)) out.push_back((e & ~511u) | heap0);
    for (int level = 1; level < kDepth; level++)
        for (uint32_t e : order.level[level])
            if (some(e >> 9, e & 511u)) out.push_back(e);
    return out;
}

// One channel from the symbol stream the device wrote: stream[i] = bucket << 10 | symbol of the i-th Some node in stream order.
std::string encode_channel_from_stream(const uint16_t *stream, size_t n, const uint32_t *hist, ChannelStream &out) {
    const std::string cerr = contexts_from_hist(hist, out);
    if (!cerr.empty()) return cerr;
    std::vector<uint16_t> symbols(n);
    std::vector<uint8_t> buckets(n);
    for (size_t i = 0; i < n; i++) {
        symbols[i] = stream[i] & 1023u, buckets[i] = (uint8_t)(stream[i] >> 10);
        if (buckets[i] >= kContexts) return "symbol stream: bucket out of range";
    }
    std::vector<RansEncoderMulti::EncSymbol> tab((size_t)kContexts * kAlphabet);
    for (int b = 0; b < kContexts; b++)
        for (int j = 0; j < kAlphabet; j++) {
            const AnsContext &c = out.contexts[b];
            tab[(size_t)b * kAlphabet + j] = c.freqs[j] ? RansEncoderMulti::make_symbol(c.cdf[j], c.freqs[j], c.max_freq_bits) : RansEncoderMulti::EncSymbol{0, 0, 0, 0, 0, 0};
        }
    const std::string err = encode_symbols(symbols, buckets, tab, out.data);
    if (!err.empty()) return err;
    out.n_symbols = n;
    return "";
}

std::string encode_channels_from_streams(uint32_t channels, const uint16_t *streams, size_t n_symbols, const uint32_t *hist, std::vector<ChannelStream> &out) {
    out.assign(channels, ChannelStream{});
    std::vector<std::string> errs(channels);
    std::vector<std::thread> workers;
    for (uint32_t ch = 1; ch < channels; ch++)
        workers.emplace_back([&, ch] { errs[ch] = encode_channel_from_stream(streams + (size_t)ch * n_symbols, n_symbols, hist + (size_t)ch * kContexts * kAlphabet, out[ch]); });
    if (channels) errs[0] = encode_channel_from_stream(streams, n_symbols, hist, out[0]);
    for (std::thread &t : workers) t.join();
    for (uint32_t ch = 0; ch < channels; ch++)
        if (!errs[ch].empty()) return "channel " + std::to_string(ch) + ": " + errs[ch];
    return "";
}

std::string encode_channels(const SymbolOrder &order, uint32_t channels, const int32_t *coefs, const uint8_t *bucket, const int32_t *prediction,
                            const uint32_t *hist, std::vector<ChannelStream> &out) {
    out.assign(channels, ChannelStream{});
    const size_t plane = order.level[0].size() * kNodes;
    std::vector<std::string> errs(channels);
    std::vector<std::thread> workers;
    for (uint32_t ch = 1; ch < channels; ch++)
        workers.emplace_back([&, ch] { errs[ch] = encode_channel(order, coefs + ch * plane, bucket + ch * plane, prediction + ch * plane, hist + (size_t)ch * kContexts * kAlphabet, out[ch]); });
    if (channels) errs[0] = encode_channel(order, coefs, bucket, prediction, hist, out[0]);
    for (std::thread &t : workers) t.join();
    for (uint32_t ch = 0; ch < channels; ch++)
        if (!errs[ch].empty()) return "channel " + std::to_string(ch) + ": " + errs[ch];
    return "";
}

std::string decode_symbols(const ChannelStream &s, const std::vector<uint8_t> &buckets, std::vector<uint16_t> &symbols) {
    RansDecoderMulti dec(s.data);
    symbols.resize(buckets.size());
    for (size_t k = 0; k < buckets.size(); k++) {
        const int b = buckets[k];
        if (b < 0 || b >= kContexts) return "bucket out of range";
        const AnsContext &c = s.contexts[b];
        const int state = kContexts - b - 1; // entropy_coding.rs:239
        const uint32_t v = dec.get_at(state, c.max_freq_bits);
        // the slot containing v: the last symbol with cdf <= v and a non-empty slot (find_nearest_or_equal + the skip loop, :244-256)
        int sym = (int)(std::upper_bound(c.cdf.begin(), c.cdf.end(), v) - c.cdf.begin()) - 1;
        if (sym < 0 || c.freqs[sym] == 0) return "stream does not match the model";
        dec.advance_at(state, c.cdf[sym], c.freqs[sym], c.max_freq_bits);
        if (!dec.ok()) return "stream truncated";
        symbols[k] = (uint16_t)sym;
    }
    return "";
}

// ---- decoder (entropy_coding::decode, :352-443) ------------------------------------------------------------------------------
namespace {
// Rust `f32 as i32`: truncating, saturating, NaN -> 0
int32_t f32_as_i32(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
int bucket_of(uint32_t w) { // assign_bucket, prediction.rs:55-68
    return w < 3 ? 0 : w < 5 ? 1 : w < 6 ? 2 : w < 8 ? 3 : w < 12 ? 4 : w < 16 ? 5 : w < 20 ? 6 : w < 25 ? 7 : w < 30 ? 8 : 9;
}
int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); } // release-build wrapping
int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
int32_t wabs(int32_t a) { return a < 0 ? (int32_t)(0u - (uint32_t)a) : a; }
} // namespace

std::string decode_channel(const fri::Geometry &g, const SymbolOrder &order, const ChannelStream &s, const ChannelParams &prm, int32_t *coefs) {
    const fri::StaticTables &st = fri::static_tables();
    const size_t F = g.centers.size();
    if (order.level[0].size() != F) return "symbol order does not match the geometry";
    // WaveletImage::from_metadata (wavelet_transform.rs:392-404): the transform of an all-zero image, i.e. Some(0) wherever the
    // geometry has a node and None elsewhere. Values not decoded yet are read as that 0 by the context of later symbols.
    for (size_t c = 0; c < F; c++)
        for (int p = 0; p < kNodes; p++) coefs[c * kNodes + p] = (g.valid_mask[c * 16 + (p >> 5)] >> (p & 31) & 1u) ? 0 : kNone;
    RansDecoderMulti dec(s.data);
    if (!dec.ok()) return "stream truncated";
    const char *failure = nullptr;
    auto gather = [&](uint32_t cell, uint32_t heap, int n, int32_t *v) { // context_modeling.rs:25-77 through the plan's static neighbour table
        for (int k = 0; k < n; k++) {
            const uint16_t e = st.nbr_table[heap][k];
            v[k] = 0;
            if (e & 0x8000u) continue; // not a node of that level anywhere
            const int slot = (e >> 9) & 7;
            const int32_t nc = g.nbr_cells[(size_t)cell * fri::kNbr + slot];
            if (nc < 0) continue; // no retained cell there
            const int32_t x = coefs[(size_t)nc * kNodes + (e & 511u)];
            v[k] = x == kNone ? 0 : x; // .unwrap_or(0)
        }
    };
    // Symbol of a slot value (:243-256: the last symbol whose cumulative start is <= v): a 4096-entry table per context over the slot's
    // top bits gives the answer for the first value of the table cell, a short forward walk the rest - instead of a ten-step
    // binary search with unpredictable branches per symbol.
    constexpr int kLutBits = 12;
    std::vector<uint16_t> lut((size_t)kContexts << kLutBits);
    int lut_shift[kContexts];
    for (int b = 0; b < kContexts; b++) {
        const AnsContext &c = s.contexts[b];
        lut_shift[b] = c.max_freq_bits > (uint32_t)kLutBits ? (int)c.max_freq_bits - kLutBits : 0;
        int sym = 0;
        for (uint32_t i = 0; i < (1u << kLutBits); i++) {
            const uint64_t v0 = (uint64_t)i << lut_shift[b];
            while (sym + 1 < kAlphabet && c.cdf[sym + 1] <= v0) sym++;
            lut[((size_t)b << kLutBits) + i] = (uint16_t)sym;
        }
    }
    auto decode_one = [&](uint32_t cell, uint32_t heap, int bucket, int32_t prediction) { // decode_symbol, :205-264
        const AnsContext &c = s.contexts[bucket];
        const int state = kContexts - bucket - 1; // :239
        const uint32_t v = dec.get_at(state, c.max_freq_bits);
        int sym = c.cdf[0] <= v ? (int)lut[((size_t)bucket << kLutBits) + std::min<uint32_t>(v >> lut_shift[bucket], (1u << kLutBits) - 1)] : -1;
        while (sym >= 0 && sym + 1 < kAlphabet && c.cdf[sym + 1] <= v) sym++;
        if (sym < 0 || c.freqs[sym] == 0) {
            failure = "stream does not match the model";
            return;
        }
        dec.advance_at(state, c.cdf[sym], c.freqs[sym], c.max_freq_bits);
        if (!dec.ok()) {
            failure = "stream truncated";
            return;
        }
        coefs[(size_t)cell * kNodes + heap] = wadd(unpack_signed((uint32_t)sym), prediction); // :263
    };
    auto lf = [&](uint32_t cell, uint32_t heap) { // get_lf_context_bucket, prediction.rs:86-149
        int32_t v[3];
        gather(cell, heap, 3, v);
        const uint32_t width = (uint32_t)wabs(wsub(v[0], v[2]));
        const int32_t mx = std::max(v[0], v[2]), mn = std::min(v[0], v[2]);
        const int32_t pred = v[1] >= mx ? mx : v[1] <= mn ? mn : wsub(wadd(v[0], v[2]), v[1]);
        decode_one(cell, heap, bucket_of(f32_as_u32((float)width)), pred);
    };
    for (uint32_t e : order.level[0]) { // first scan: DC (:369-388). Every retained cell has a root, hence a DC.
        lf(e >> 9, 0);
        if (failure) return failure;
    }
    for (uint32_t e : order.level[0]) { // second scan: root (:391-410)
        lf(e >> 9, 1);
        if (failure) return failure;
    }
    for (int level = 1; level < kDepth; level++) { // :413-443
        const int grp = level < kDepth - 2 ? 2 : level == kDepth - 2 ? 1 : 0; // prediction.rs:165-179
        const float *wp = prm.width[grp], *vp = prm.value[grp];
        for (uint32_t e : order.level[level]) {
            const uint32_t cell = e >> 9, heap = e & 511u;
            if (coefs[(size_t)cell * kNodes + heap] == kNone) continue; // :421-425
            int32_t v[6];
            gather(cell, heap, 6, v);
            // get_hf_context_bucket, prediction.rs:190-206: f32, left to right, one rounding per operation (this file is built with
            // -ffp-contract=off)
            float width = wp[0];
            width = width + wp[1] * (float)wabs(wsub(v[0], v[3]));
            width = width + wp[2] * (float)wabs(wsub(v[1], v[2]));
            width = width + wp[3] * (float)wabs(wsub(v[4], v[5]));
            width = width + wp[4] * (float)wabs(wsub(v[1], v[5]));
            width = width + wp[5] * (float)wabs(wsub(v[2], v[4]));
            float pf = (float)v[0] * vp[0];
            pf = pf + (float)v[1] * vp[1];
            pf = pf + (float)v[2] * vp[2];
            pf = pf + (float)v[3] * vp[3];
            pf = pf + (float)v[4] * vp[4];
            pf = pf + (float)v[5] * vp[5];
            decode_one(cell, heap, bucket_of(f32_as_u32(width)), f32_as_i32(pf));
            if (failure) return failure;
        }
    }
    return "";
}

std::string count_cells(uint32_t width, uint32_t height, uint32_t channels, uint32_t &n_cells) {
    fri::Geometry g;
    const std::string err = fri::build_geometry(width, height, channels, fri::TilingParams{}, g);
    n_cells = (uint32_t)g.centers.size();
    return err;
}

std::string decode_image(const std::vector<uint8_t> &frv, DecodedImage &out) {
    ParsedImage img;
    const std::string err = deserialize(frv, img);
    return err.empty() ? decode_parsed(img, out) : err;
}

std::string decode_parsed(const ParsedImage &img, DecodedImage &out) {
    std::string err;
    const uint32_t channels = img.colorspace == kLuma ? 1u : 3u; // ColorSpace::num_channels, images.rs:31-38
    if (img.channels.size() != channels) return "Malformed image bytes";
    for (const ChannelStream &c : img.channels)
        for (const AnsContext &a : c.contexts)
            if (a.max_freq_bits == 0) return "Malformed image bytes"; // fewer than ten EHD segments
    fri::Geometry g;
    err = fri::build_geometry(img.width, img.height, channels, fri::TilingParams{}, g);
    if (!err.empty()) return err;
    const size_t F = g.centers.size(), plane = F * kNodes;
    out.height = img.height, out.width = img.width, out.colorspace = img.colorspace, out.channels = channels, out.n_cells = (uint32_t)F;
    out.centers.resize(F * 2);
    for (size_t c = 0; c < F; c++) out.centers[2 * c] = g.centers[c].x, out.centers[2 * c + 1] = g.centers[c].y;
    out.coefs.assign(channels * plane, 0);
    out.params = img.params;
    const auto order_ptr = shared_symbol_order(out.centers.data(), (uint32_t)F);
    const SymbolOrder &order = *order_ptr;
    std::vector<std::string> errs(channels);
    std::vector<std::thread> workers;
    for (uint32_t ch = 1; ch < channels; ch++)
        workers.emplace_back([&, ch] { errs[ch] = decode_channel(g, order, img.channels[ch], img.params[ch], out.coefs.data() + ch * plane); });
    errs[0] = decode_channel(g, order, img.channels[0], img.params[0], out.coefs.data());
    for (std::thread &t : workers) t.join();
    for (uint32_t ch = 0; ch < channels; ch++)
        if (!errs[ch].empty()) return "channel " + std::to_string(ch) + ": " + errs[ch];
    return "";
}

// ---- container ---------------------------------------------------------------------------------------------------------
namespace {
void put_u16(std::vector<uint8_t> &v, uint16_t x) { v.push_back((uint8_t)x), v.push_back((uint8_t)(x >> 8)); }
void put_u32(std::vector<uint8_t> &v, uint32_t x) {
    for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i)));
}
void put_u64(std::vector<uint8_t> &v, uint64_t x) {
    for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i)));
}
void put_f32(std::vector<uint8_t> &v, float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    put_u32(v, u);
}
constexpr uint8_t kEHD[2] = {0xFF, 0xB2}, kDAT[2] = {0xFF, 0xB4}, kEOC[2] = {0xFF, 0xB8}, kPRD[2] = {0xFF, 0xBB}, kEOI[2] = {0xFF, 0xDF}; // serialize.rs:41-47
} // namespace

std::vector<uint8_t> serialize(uint32_t height, uint32_t width, ColorSpaceCode cs, const std::vector<ChannelStream> &channels, const std::vector<ChannelParams> &params) {
    std::vector<uint8_t> s;
    s.insert(s.end(), {'f', 'r', 'i', 'f'});
    put_u32(s, height);
    put_u32(s, width);
    put_u32(s, (uint32_t)cs << 30 | 1u << 28); // variant: TameTwindragon = 0b01 (images.rs:49-55)
    for (size_t ch = 0; ch < channels.size(); ch++) {
        s.insert(s.end(), kPRD, kPRD + 2);
        for (int g = 0; g < 3; g++)
            for (int k = 0; k < 6; k++) put_f32(s, params[ch].value[g][k]);
        for (int g = 0; g < 3; g++)
            for (int k = 0; k < 6; k++) put_f32(s, params[ch].width[g][k]);
        for (const AnsContext &c : channels[ch].contexts) {
            s.insert(s.end(), kEHD, kEHD + 2);
            put_u32(s, c.max_freq_bits);
            put_u64(s, c.off_distribution_values.size()); // usize
            for (uint16_t v : c.off_distribution_values) put_u16(s, v);
        }
        s.insert(s.end(), kDAT, kDAT + 2);
        put_u64(s, channels[ch].data.size());
        s.insert(s.end(), channels[ch].data.begin(), channels[ch].data.end());
        s.insert(s.end(), kEOC, kEOC + 2);
    }
    s.insert(s.end(), kEOI, kEOI + 2);
    return s;
}

std::string deserialize(const std::vector<uint8_t> &b, ParsedImage &out) {
    size_t o = 0;
    auto need = [&](size_t n) { return o + n <= b.size(); };
    auto u32 = [&]() {
        uint32_t v = (uint32_t)b[o] | (uint32_t)b[o + 1] << 8 | (uint32_t)b[o + 2] << 16 | (uint32_t)b[o + 3] << 24;
        o += 4;
        return v;
    };
    auto u64 = [&]() {
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)b[o + i] << (8 * i);
        o += 8;
        return v;
    };
    if (!need(16) || std::memcmp(b.data(), "frif", 4) != 0) return "Invalid signature for FRIF image.";
    o = 4;
    out.height = u32();
    out.width = u32();
    const uint32_t mdat = u32();
    out.colorspace = mdat >> 30 & 3u;
    out.variant = mdat >> 28 & 3u;
    if (out.colorspace == 0 || out.variant == 0) return "Invalid metadata";
    ChannelStream cur;
    ChannelParams prm{};
    int n_ctx = 0;
    for (;;) {
        if (!need(2)) return "Malformed image bytes";
        const uint8_t m0 = b[o], m1 = b[o + 1];
        o += 2;
        if (m0 != 0xFF) return "Malformed image bytes";
        if (m1 == kPRD[1]) {
            if (!need(36 * 4)) return "Malformed image bytes";
            for (int g = 0; g < 3; g++)
                for (int k = 0; k < 6; k++) {
                    const uint32_t u = u32();
                    std::memcpy(&prm.value[g][k], &u, 4);
                }
            for (int g = 0; g < 3; g++)
                for (int k = 0; k < 6; k++) {
                    const uint32_t u = u32();
                    std::memcpy(&prm.width[g][k], &u, 4);
                }
        } else if (m1 == kEHD[1]) {
            if (!need(12) || n_ctx >= kContexts) return "Malformed image bytes";
            AnsContext c;
            c.max_freq_bits = u32();
            const uint64_t n = u64();
            if (n > kAlphabet || !need(n * 2)) return "Malformed image bytes";
            for (uint64_t i = 0; i < n; i++) {
                c.off_distribution_values.push_back((uint16_t)(b[o] | b[o + 1] << 8));
                o += 2;
            }
            const std::string err = c.finalize(n_ctx); // serialize.rs:232: the decoder rebuilds the table from the two fields
            if (!err.empty()) return err;
            cur.contexts[n_ctx++] = c;
        } else if (m1 == kDAT[1]) {
            if (!need(8)) return "Malformed image bytes";
            const uint64_t n = u64();
            if (!need(n)) return "Malformed image bytes";
            cur.data.assign(b.begin() + o, b.begin() + o + n);
            o += n;
        } else if (m1 == kEOC[1]) {
            out.channels.push_back(cur);
            out.params.push_back(prm);
            cur = ChannelStream{};
            prm = ChannelParams{};
            n_ctx = 0;
        } else if (m1 == kEOI[1]) {
            return "";
        } else {
            return "Malformed image bytes";
        }
    }
}

} // namespace emit
} // namespace libfri
