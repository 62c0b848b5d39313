// Host-side range coder of libpcc_geo_hip.so.
//
// The reference codes symbols with tensorflow-compression 1.3's C++ ops
// `unbounded_index_range_encode/decode` (call sites: src/utils/patch_gaussian_conditional.py:27-31,
// src/model_types.py:291-292,382-387,404-407) which run on the CPU there too
// (patch_gaussian_conditional.py:105-106).  tfc's source is not under /root/reference; this file
// implements the published algorithm of that coder: a 32-bit range coder with 16-bit
// renormalisation and a delayed-carry counter, per-index quantised CDFs of `precision` bits, and
// out-of-range symbols escaped through the last CDF bin followed by an Elias-gamma-like sequence of
// `overflow_width`-bit digits.  Byte-compatibility with tfc 1.3 is unpinned (no golden bitstreams).
//
// Streams (one per block and per string) are independent and are coded concurrently by a
// persistent pool of host threads.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/pcc_geo.h"

void pcc_set_error(const char* fmt, ...);
#define PCC_API extern "C" __attribute__((visibility("default")))

namespace {

// ------------------------------------------------------------------------------------------
class Encoder {
  public:
    Encoder(uint8_t* out, size_t cap) : out_(out), cap_(cap) {}
    inline void encode(int32_t lower, int32_t upper, int precision) {
        const uint64_t size = (uint64_t)size_minus1_ + 1;
        const uint32_t a = (uint32_t)((size * (uint64_t)(uint32_t)lower) >> precision);
        const uint32_t b = (uint32_t)(((size * (uint64_t)(uint32_t)upper) >> precision) - 1);
        base_ += a;
        size_minus1_ = b - a;
        const bool base_overflow = base_ < a;
        if ((uint32_t)(base_ + size_minus1_) < base_) {  // interval straddles 2^32: carry undecided
            if ((size_minus1_ >> 16) == 0) {
                base_ <<= 16;
                size_minus1_ = (size_minus1_ << 16) | 0xFFFFu;
                delay_ += 0x20000;
            }
            return;
        }
        if (delay_ != 0) {
            if (base_overflow) {
                put((uint8_t)(delay_ >> 8)); put((uint8_t)delay_); fill(delay_ >> 16, 0x00);
            } else {
                --delay_;
                put((uint8_t)(delay_ >> 8)); put((uint8_t)delay_); fill(delay_ >> 16, 0xFF);
            }
            delay_ = 0;
        }
        if ((size_minus1_ >> 16) == 0) {
            const uint32_t top = base_ >> 16;
            base_ <<= 16;
            size_minus1_ = (size_minus1_ << 16) | 0xFFFFu;
            if (base_ <= (uint32_t)(base_ + size_minus1_)) {
                put((uint8_t)(top >> 8)); put((uint8_t)top);
            } else {
                delay_ = top + 1;
            }
        }
    }
    void finalize() {
        if (delay_ != 0) {
            put((uint8_t)(delay_ >> 8));
            if ((delay_ & 0xFF) != 0) put((uint8_t)delay_);
        } else if (base_ != 0) {
            const uint32_t mid = ((base_ - 1) >> 16) + 1;
            put((uint8_t)(mid >> 8));
            if ((mid & 0xFF) != 0) put((uint8_t)mid);
        }
    }
    size_t size() const { return n_; }
    bool overflowed() const { return n_ > cap_; }

  private:
    inline void put(uint8_t b) { if (n_ < cap_) out_[n_] = b; ++n_; }
    inline void fill(uint64_t cnt, uint8_t b) { while (cnt--) put(b); }
    uint32_t base_ = 0, size_minus1_ = 0xFFFFFFFFu;
    uint64_t delay_ = 0;
    uint8_t* out_;
    size_t cap_, n_ = 0;
};

class Decoder {
  public:
    Decoder(const uint8_t* s, size_t n) : cur_(s), end_(s + n) { read16(); read16(); }
    // cdf[0..len-1], cdf[0] = 0, cdf[len-1] = 2^precision.  `guess` = the most probable symbol of the row (the centre of a Gaussian
    // table, i.e. the value 0): ~97 % of the y symbols of a trained codec are it, and testing its interval first (two multiplies)
    // gives the same answer as the bisection over up to 1481 entries (eleven) -- the interval test IS the bisection's exit condition.
    inline int32_t decode(const int32_t* cdf, int len, int precision, int guess = -1) {
        const uint64_t size = (uint64_t)size_minus1_ + 1;
        const uint64_t offset = (((uint64_t)(uint32_t)(value_ - base_) + 1) << precision) - 1;
        if (guess >= 0 && guess + 1 < len) {
            const uint64_t plo = size * (uint64_t)(uint32_t)cdf[guess], phi = size * (uint64_t)(uint32_t)cdf[guess + 1];
            if (plo <= offset && phi > offset) {
                narrow((uint32_t)(plo >> precision), (uint32_t)((phi >> precision) - 1));      // = update(cdf[guess], cdf[guess + 1]) with the products at hand
                return guess;
            }
        }
        const int32_t* pv = cdf + 1;
        int n = len - 1;
        do {
            const int half = n >> 1;
            const int32_t* mid = pv + half;
            if (size * (uint64_t)(uint32_t)(*mid) <= offset) { pv = mid + 1; n -= half + 1; }
            else n = half;
        } while (n > 0);
        if (pv >= cdf + len) { corrupt_ = true; pv = cdf + len - 1; }
        update(pv[-1], pv[0], precision, size);
        return (int32_t)(pv - cdf - 1);
    }
    // 2^w equiprobable values: no table needed
    inline int32_t decode_uniform(int w) {
        const uint64_t size = (uint64_t)size_minus1_ + 1;
        const uint64_t offset = (((uint64_t)(uint32_t)(value_ - base_) + 1) << w) - 1;
        // smallest v in [1, 2^w] with size * v > offset
        uint64_t v = offset / size + 1;
        const uint64_t vmax = (uint64_t)1 << w;
        if (v > vmax) { corrupt_ = true; v = vmax; }
        update((int32_t)(v - 1), (int32_t)v, w, size);
        return (int32_t)(v - 1);
    }
    bool corrupt() const { return corrupt_; }

  private:
    inline void update(int32_t lo, int32_t hi, int precision, uint64_t size) {
        narrow((uint32_t)((size * (uint64_t)(uint32_t)lo) >> precision), (uint32_t)(((size * (uint64_t)(uint32_t)hi) >> precision) - 1));
    }
    inline void narrow(uint32_t a, uint32_t b) {
        base_ += a;
        size_minus1_ = b - a;
        if ((size_minus1_ >> 16) == 0) {
            base_ <<= 16;
            size_minus1_ = (size_minus1_ << 16) | 0xFFFFu;
            read16();
        }
    }
    inline void read16() {
        value_ <<= 8;
        if (cur_ != end_) value_ |= *cur_++;
        value_ <<= 8;
        if (cur_ != end_) value_ |= *cur_++;
    }
    uint32_t base_ = 0, size_minus1_ = 0xFFFFFFFFu, value_ = 0;
    const uint8_t* cur_;
    const uint8_t* end_;
    bool corrupt_ = false;
};

template <class DataT, class IndexT>
long encode_stream(const pcc_cdf_table& t, const DataT* data, const IndexT* index, int index_mod, size_t n,
                   uint8_t* out, size_t cap) {
    Encoder e(out, cap);
    const int ow = t.overflow_width;
    const uint32_t omax = (1u << ow) - 1;
    // per-row constants side by side (as in decode_stream)
    struct Row { const int32_t* c; int32_t max_value, offset; };
    std::vector<Row> rows((size_t)t.rows);
    for (int r = 0; r < t.rows; ++r) rows[(size_t)r] = Row{t.cdf + (size_t)r * t.cdf_stride, t.cdf_size[r] - 2, t.offset[r]};
    int row_mod = 0;
    for (size_t i = 0; i < n; ++i) {
        int row;
        if (index) row = (int)index[i];
        else { row = row_mod; if (++row_mod == index_mod) row_mod = 0; }
        if ((unsigned)row >= (unsigned)t.rows) return -2;
        const Row& R = rows[(size_t)row];
        const int32_t max_value = R.max_value;
        int32_t value = (int32_t)data[i] - R.offset;
        uint32_t overflow = 0;
        if (value < 0) { overflow = (uint32_t)(-2 * (int64_t)value - 1); value = max_value; }
        else if (value >= max_value) { overflow = (uint32_t)(2 * ((int64_t)value - max_value)); value = max_value; }
        const int32_t* c = R.c;
        e.encode(c[value], c[value + 1], t.precision);
        if (value != max_value) continue;
        int widths = 0;
        while (widths * ow < 32 && (overflow >> (widths * ow)) != 0) ++widths;
        uint32_t val = (uint32_t)widths;
        while (val >= omax) { e.encode((int32_t)omax, (int32_t)omax + 1, ow); val -= omax; }
        e.encode((int32_t)val, (int32_t)val + 1, ow);
        for (int j = 0; j < widths; ++j) {
            const uint32_t dgt = (overflow >> (j * ow)) & omax;
            e.encode((int32_t)dgt, (int32_t)dgt + 1, ow);
        }
    }
    e.finalize();
    return e.overflowed() ? -1 : (long)e.size();
}

// returns 0, -1 (corrupt stream), -2 (row out of range), -3 (a symbol does not fit OutT)
template <class IndexT, class OutT>
int decode_stream(const pcc_cdf_table& t, const uint8_t* str, size_t len, const IndexT* index, int index_mod,
                  size_t n, OutT* out) {
    Decoder d(str, len);
    const int ow = t.overflow_width;
    const uint32_t omax = (1u << ow) - 1;
    // per-row constants side by side (one cache line per four rows instead of three arrays + a multiply per symbol)
    struct Row { const int32_t* c; int32_t size, max_value, zero_at, offset; };
    std::vector<Row> rows((size_t)t.rows);
    for (int r = 0; r < t.rows; ++r) {
        const int32_t mv = t.cdf_size[r] - 2, z = -t.offset[r];      // z: table position of the value 0 (the mode of every prior in use)
        rows[(size_t)r] = Row{t.cdf + (size_t)r * t.cdf_stride, t.cdf_size[r], mv, z >= 0 && z < mv ? z : -1, t.offset[r]};
    }
    int row_mod = 0;
    for (size_t i = 0; i < n; ++i) {
        int row;
        if (index) row = (int)index[i];
        else { row = row_mod; if (++row_mod == index_mod) row_mod = 0; }
        if ((unsigned)row >= (unsigned)t.rows) return -2;
        const Row& R = rows[(size_t)row];
        const int32_t max_value = R.max_value;
        int32_t value = d.decode(R.c, R.size, t.precision, R.zero_at);
        if (value == max_value) {
            int widths = 0;
            uint32_t val;
            do { val = (uint32_t)d.decode_uniform(ow); widths += (int)val; } while (val == omax && widths < 64);
            uint32_t overflow = 0;
            for (int j = 0; j < widths; ++j) {
                const uint32_t dgt = (uint32_t)d.decode_uniform(ow);
                if (j * ow < 32) overflow |= dgt << (j * ow);
            }
            value = (int32_t)(overflow >> 1);
            if (overflow & 1) value = -value - 1;
            else value += max_value;
        }
        const int32_t sym = value + R.offset;
        out[i] = (OutT)sym;
        if (sizeof(OutT) < 4 && (int32_t)out[i] != sym) return -3;
    }
    return d.corrupt() ? -1 : 0;
}

// ------------------------------------------------------------------------------------------
// persistent pool: parallel_for(n, fn) runs fn(i) for i in [0,n) on up to `threads` workers.  One pool per CALLING thread
// (thread_local): a host that range-encodes one chunk on a helper thread while it range-decodes another on the main thread
// gets two sets of workers instead of one lock.
class Pool {
  public:
    static Pool& get() { static thread_local Pool p; return p; }
    void parallel_for(int n, int threads, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
        if (threads <= 0) threads = hw;
        threads = std::min(threads, n);
        if (threads <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        std::unique_lock<std::mutex> call_lock(call_mu_);  // one parallel_for at a time
        ensure_workers(threads - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; n_ = n; next_.store(0); active_ = threads - 1; want_ = threads - 1; ++epoch_;
        }
        cv_.notify_all();
        run();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return active_ == 0; });
        fn_ = nullptr;
    }

  private:
    Pool() = default;
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++epoch_; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void ensure_workers(int k) {
        while ((int)workers_.size() < k) {
            const int id = (int)workers_.size();
            workers_.emplace_back([this, id] { worker(id); });
        }
    }
    void run() {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) break;
            (*fn_)(i);
        }
    }
    void worker(int id) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return epoch_ != seen; });
            seen = epoch_;
            if (stop_) return;
            if (id >= want_) continue;  // not needed for this call
            lk.unlock();
            run();
            lk.lock();
            if (--active_ == 0) done_cv_.notify_all();
        }
    }
    std::mutex call_mu_, mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, active_ = 0, want_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

bool table_ok(const pcc_cdf_table* t) {
    return t && t->cdf && t->cdf_size && t->offset && t->rows > 0 && t->cdf_stride >= 2 && t->precision >= 1 &&
           t->precision <= 16 && t->overflow_width >= 1 && t->overflow_width <= 8;
}

}  // namespace

PCC_API int pcc_range_encode_batch(const pcc_cdf_table* t, int32_t n_streams, const int32_t* const* data,
                                   const int32_t* const* index, int32_t index_mod, const size_t* n, uint8_t* const* out,
                                   const size_t* cap, size_t* out_len, int32_t n_threads) {
    if (!table_ok(t) || n_streams < 0 || !data || !n || !out || !cap || !out_len) {
        pcc_set_error("pcc_range_encode_batch: bad argument");
        return PCC_ERR_ARG;
    }
    if (!index && index_mod <= 0) { pcc_set_error("pcc_range_encode_batch: index NULL needs index_mod > 0"); return PCC_ERR_ARG; }
    std::atomic<int> status{0};
    Pool::get().parallel_for(n_streams, n_threads, [&](int s) {
        const long r = encode_stream(*t, data[s], index ? index[s] : nullptr, index_mod, n[s], out[s], cap[s]);
        if (r < 0) { status.store(r == -1 ? PCC_ERR_SPACE : PCC_ERR_ARG); out_len[s] = 0; }
        else out_len[s] = (size_t)r;
    });
    if (status.load() != 0) {
        pcc_set_error("pcc_range_encode_batch: %s", status.load() == PCC_ERR_SPACE ? "output buffer too small" : "CDF row index out of range");
        return status.load();
    }
    return PCC_OK;
}

PCC_API int pcc_range_decode_batch(const pcc_cdf_table* t, int32_t n_streams, const uint8_t* const* str,
                                   const size_t* str_len, const int32_t* const* index, int32_t index_mod, const size_t* n,
                                   int32_t* const* out, int32_t n_threads) {
    if (!table_ok(t) || n_streams < 0 || !str || !str_len || !n || !out) {
        pcc_set_error("pcc_range_decode_batch: bad argument");
        return PCC_ERR_ARG;
    }
    if (!index && index_mod <= 0) { pcc_set_error("pcc_range_decode_batch: index NULL needs index_mod > 0"); return PCC_ERR_ARG; }
    std::atomic<int> status{0};
    Pool::get().parallel_for(n_streams, n_threads, [&](int s) {
        const int r = decode_stream(*t, str[s], str_len[s], index ? index[s] : nullptr, index_mod, n[s], out[s]);
        if (r == -1) status.store(PCC_ERR_CORRUPT);
        else if (r == -2) status.store(PCC_ERR_ARG);
    });
    if (status.load() != 0) {
        pcc_set_error("pcc_range_decode_batch: %s", status.load() == PCC_ERR_CORRUPT ? "corrupt stream" : "CDF row index out of range");
        return status.load();
    }
    return PCC_OK;
}

// Narrow host arrays (round 3): symbols as int16 and CDF rows as uint8 -- what crosses PCIe in the codec's hot loop (the Gaussian
// conditional's 64 scale rows fit a byte; symbols beyond int16 make the caller fall back to the 32-bit entry points).
PCC_API int pcc_range_encode_batch_n(const pcc_cdf_table* t, int32_t n_streams, const void* const* data, int32_t data_bytes,
                                     const void* const* index, int32_t index_bytes, int32_t index_mod, const size_t* n,
                                     uint8_t* const* out, const size_t* cap, size_t* out_len, int32_t n_threads) {
    if (!table_ok(t) || n_streams < 0 || !data || !n || !out || !cap || !out_len || (data_bytes != 2 && data_bytes != 4) ||
        (index && index_bytes != 1 && index_bytes != 4)) {
        pcc_set_error("pcc_range_encode_batch_n: bad argument (data_bytes 2|4, index_bytes 1|4)");
        return PCC_ERR_ARG;
    }
    if (!index && index_mod <= 0) { pcc_set_error("pcc_range_encode_batch_n: index NULL needs index_mod > 0"); return PCC_ERR_ARG; }
    std::atomic<int> status{0};
    Pool::get().parallel_for(n_streams, n_threads, [&](int s) {
        long r;
        const void* ix = index ? index[s] : nullptr;
        if (data_bytes == 2) r = (ix && index_bytes == 1) ? encode_stream(*t, (const int16_t*)data[s], (const uint8_t*)ix, index_mod, n[s], out[s], cap[s])
                                                          : encode_stream(*t, (const int16_t*)data[s], (const int32_t*)ix, index_mod, n[s], out[s], cap[s]);
        else r = (ix && index_bytes == 1) ? encode_stream(*t, (const int32_t*)data[s], (const uint8_t*)ix, index_mod, n[s], out[s], cap[s])
                                          : encode_stream(*t, (const int32_t*)data[s], (const int32_t*)ix, index_mod, n[s], out[s], cap[s]);
        if (r < 0) { status.store(r == -1 ? PCC_ERR_SPACE : PCC_ERR_ARG); out_len[s] = 0; }
        else out_len[s] = (size_t)r;
    });
    if (status.load() != 0) {
        pcc_set_error("pcc_range_encode_batch_n: %s", status.load() == PCC_ERR_SPACE ? "output buffer too small" : "CDF row index out of range");
        return status.load();
    }
    return PCC_OK;
}

// out_bytes 2: returns PCC_ERR_SPACE when a decoded symbol does not fit int16 (decode again with out_bytes 4)
PCC_API int pcc_range_decode_batch_n(const pcc_cdf_table* t, int32_t n_streams, const uint8_t* const* str, const size_t* str_len,
                                     const void* const* index, int32_t index_bytes, int32_t index_mod, const size_t* n,
                                     void* const* out, int32_t out_bytes, int32_t n_threads) {
    if (!table_ok(t) || n_streams < 0 || !str || !str_len || !n || !out || (out_bytes != 2 && out_bytes != 4) ||
        (index && index_bytes != 1 && index_bytes != 4)) {
        pcc_set_error("pcc_range_decode_batch_n: bad argument (out_bytes 2|4, index_bytes 1|4)");
        return PCC_ERR_ARG;
    }
    if (!index && index_mod <= 0) { pcc_set_error("pcc_range_decode_batch_n: index NULL needs index_mod > 0"); return PCC_ERR_ARG; }
    std::atomic<int> status{0};
    Pool::get().parallel_for(n_streams, n_threads, [&](int s) {
        int r;
        const void* ix = index ? index[s] : nullptr;
        if (out_bytes == 2) r = (ix && index_bytes == 1) ? decode_stream(*t, str[s], str_len[s], (const uint8_t*)ix, index_mod, n[s], (int16_t*)out[s])
                                                         : decode_stream(*t, str[s], str_len[s], (const int32_t*)ix, index_mod, n[s], (int16_t*)out[s]);
        else r = (ix && index_bytes == 1) ? decode_stream(*t, str[s], str_len[s], (const uint8_t*)ix, index_mod, n[s], (int32_t*)out[s])
                                          : decode_stream(*t, str[s], str_len[s], (const int32_t*)ix, index_mod, n[s], (int32_t*)out[s]);
        if (r == -1) status.store(PCC_ERR_CORRUPT);
        else if (r == -2) status.store(PCC_ERR_ARG);
        else if (r == -3) { int z = 0; status.compare_exchange_strong(z, PCC_ERR_SPACE); }
    });
    if (status.load() != 0) {
        pcc_set_error("pcc_range_decode_batch_n: %s", status.load() == PCC_ERR_CORRUPT ? "corrupt stream" :
                      status.load() == PCC_ERR_SPACE ? "a symbol does not fit the 16-bit output" : "CDF row index out of range");
        return status.load();
    }
    return PCC_OK;
}

// tfc `pmf_to_quantized_cdf`: round(pmf * 2^precision) floored at 1, then the sum is repaired by
// repeatedly adjusting the entry whose code-length penalty is smallest (sorted-queue formulation).
PCC_API int pcc_pmf_to_quantized_cdf(const float* pmf, int32_t n, int32_t precision, int32_t* cdf) {
    if (!pmf || !cdf || n <= 0 || precision < 1 || precision > 16 || n > (1 << precision)) {
        pcc_set_error("pcc_pmf_to_quantized_cdf: bad argument");
        return PCC_ERR_ARG;
    }
    const int64_t target = (int64_t)1 << precision;
    int32_t* q = cdf + 1;
    int64_t sum = 0;
    for (int i = 0; i < n; ++i) {
        int32_t v = (int32_t)std::rint((double)pmf[i] * (double)target);
        q[i] = v < 1 ? 1 : v;
        sum += q[i];
    }
    struct Item { int i; double key; };
    std::vector<Item> items((size_t)n);
    if (sum > target) {
        auto penalty = [&](int i) { return q[i] <= 1 ? INFINITY : (double)pmf[i] * (std::log2((double)q[i]) - std::log2((double)q[i] - 1)); };
        for (int i = 0; i < n; ++i) items[i] = {i, penalty(i)};
        std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.key < b.key; });
        while (sum > target) {
            Item it = items[0];
            if (!(it.key < INFINITY)) { pcc_set_error("pcc_pmf_to_quantized_cdf: cannot normalise"); return PCC_ERR_ARG; }
            --q[it.i]; --sum;
            it.key = penalty(it.i);
            size_t j = 1;  // re-insert: linear search (the new position is almost always near the front)
            while (j < items.size() && !(it.key < items[j].key)) { items[j - 1] = items[j]; ++j; }
            items[j - 1] = it;
        }
    } else if (sum < target) {
        auto gain = [&](int i) { return (double)pmf[i] * (std::log2((double)q[i] + 1) - std::log2((double)q[i])); };
        for (int i = 0; i < n; ++i) items[i] = {i, gain(i)};
        std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.key > b.key; });
        while (sum < target) {
            Item it = items[0];
            ++q[it.i]; ++sum;
            it.key = gain(it.i);
            size_t j = 1;
            while (j < items.size() && !(it.key > items[j].key)) { items[j - 1] = items[j]; ++j; }
            items[j - 1] = it;
        }
    }
    cdf[0] = 0;
    for (int i = 0; i < n; ++i) cdf[i + 1] += cdf[i];
    return PCC_OK;
}
