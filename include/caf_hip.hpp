// caf_hip.hpp -- header-only C++ mirror of the reference's operator surface over the C ABI.
//
// Reference (caf_rust/src/caf/mod.rs): `trait CafSurface` with associated functions
// caf_surface (:26-27), find_peak (:31-42), apply_freq_shift (:46-65), the row record
// CafSurfaceRow (:17-22), and xcor_rustfft::Xcor::{new, run, clone} (xcor_rustfft.rs:14-93).
// `caf::CafHip` is the backend an eighth `impl CafSurface for CafHip` would be (the Rust
// stub is in INTEGRATION.md); the reference's other helpers used by its callers
// (utils.rs read_file_c64, test.rs gen_float_shifts/load_files) are mirrored too so that
// tests and the demo read like the reference's.  Errors: the reference panics
// (assert!/unwrap); here a std::runtime_error carries caf_last_error_string().
#pragma once
#include <complex>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "caf_hip.h"

namespace caf {

using Complex64 = std::complex<double>;  // num_complex::Complex64 == {re: f64, im: f64}

struct CafSurfaceRow {  // mod.rs:17-22
    double freq;
    std::vector<double> xcor_mag;
    std::size_t xcor_peak_idx;
    double xcor_peak_val;
};

inline void check(int rc, const char *what)
{
    if (rc != CAF_OK) throw std::runtime_error(std::string(what) + ": " + caf_last_error_string());
}

// One context per thread (the C ABI's contexts are not thread-safe).
inline caf_ctx *default_ctx()
{
    struct Holder {
        caf_ctx *c = nullptr;
        Holder() { check(caf_ctx_create(0, &c), "caf_ctx_create"); }
        ~Holder() { caf_ctx_destroy(c); }
    };
    static thread_local Holder h;
    return h.c;
}

// Per-thread pinned arena (caf_host_alloc) the row kernel writes the surface into in place: a fresh
// std::vector of 26 MB per call would cost more in page faults than the whole CAF.  Grows, never shrinks; freed
// with the context.
inline double *surface_arena(std::size_t values)
{
    static thread_local double *arena = nullptr;
    static thread_local std::size_t cap = 0;
    if (values > cap) {
        if (arena) check(caf_host_free(default_ctx(), arena), "caf_host_free");
        arena = nullptr;
        cap = 0;
        void *p = nullptr;
        check(caf_host_alloc(default_ctx(), values * sizeof(double), &p), "caf_host_alloc");
        arena = static_cast<double *>(p);
        cap = values;
    }
    return arena;
}

struct CafHip {  // `pub struct CafHip {}  impl CafSurface for CafHip`
    // mod.rs:26-27.  Rows come back in freq-list order (like CafRustFFT / the rayon collect).
    static std::vector<CafSurfaceRow> caf_surface(const std::vector<Complex64> &needle,
                                                  const std::vector<Complex64> &haystack,
                                                  const std::vector<double> &freqs_hz, uint32_t fs)
    {
        if (needle.size() != haystack.size())  // Xcor::run's assert (xcor_rustfft.rs:54-55)
            throw std::runtime_error("assertion failed: a.len() == self.n");
        const std::size_t n = needle.size(), F = freqs_hz.size(), L = 2 * n;
        std::vector<double> val(F);
        std::vector<uint64_t> idx(F);
        double *surf = (F > 0 && L > 0) ? surface_arena(F * L) : nullptr;
        caf_peak pk;
        check(caf_surface_c128(default_ctx(), reinterpret_cast<const double *>(needle.data()),
                               reinterpret_cast<const double *>(haystack.data()), n, freqs_hz.data(), F, fs,
                               surf, idx.data(), val.data(), &pk),
              "caf_surface_c128");
        std::vector<CafSurfaceRow> rows(F);
        for (std::size_t r = 0; r < F; ++r)  // the row Vecs of mod.rs:156-161
            rows[r] = CafSurfaceRow{freqs_hz[r], std::vector<double>(surf + r * L, surf + (r + 1) * L),
                                    static_cast<std::size_t>(idx[r]), val[r]};
        return rows;
    }

    // mod.rs:31-42: consumes the surface; first strictly-greater row wins; empty -> (0.0, 0).
    static std::pair<double, std::size_t> find_peak(std::vector<CafSurfaceRow> arr)
    {
        std::vector<double> fr(arr.size()), val(arr.size());
        std::vector<uint64_t> idx(arr.size());
        for (std::size_t r = 0; r < arr.size(); ++r) {
            fr[r] = arr[r].freq;
            idx[r] = arr[r].xcor_peak_idx;
            val[r] = arr[r].xcor_peak_val;
        }
        caf_peak pk;
        check(caf_find_peak(default_ctx(), fr.data(), idx.data(), val.data(), arr.size(), &pk), "caf_find_peak");
        return {pk.freq, static_cast<std::size_t>(pk.idx)};
    }

    // mod.rs:46-65
    static std::vector<Complex64> apply_freq_shift(const std::vector<Complex64> &samples, double freq_shift,
                                                   uint32_t fs)
    {
        std::vector<Complex64> out(samples.size());
        check(caf_apply_freq_shift_c128(default_ctx(), reinterpret_cast<const double *>(samples.data()),
                                        samples.size(), freq_shift, fs, reinterpret_cast<double *>(out.data())),
              "caf_apply_freq_shift_c128");
        return out;
    }
};

class Xcor {  // xcor_rustfft.rs:14-93
  public:
    explicit Xcor(std::size_t n) : n_(n)
    {
        if (n == 0 || (n & (n - 1))) throw std::runtime_error("Xcor::new: n must be a power of two");
    }
    std::vector<Complex64> run(const std::vector<Complex64> &a, const std::vector<Complex64> &b) const
    {
        if (a.size() != n_) throw std::runtime_error("assertion failed: a.len() == self.n");
        if (b.size() != n_) throw std::runtime_error("assertion failed: b.len() == self.n");
        std::vector<Complex64> out(n_);
        check(caf_xcor_c128(default_ctx(), reinterpret_cast<const double *>(a.data()),
                            reinterpret_cast<const double *>(b.data()), n_, reinterpret_cast<double *>(out.data())),
              "caf_xcor_c128");
        return out;
    }
    Xcor clone() const { return Xcor(n_); }  // plans are cached per n inside the context

  private:
    std::size_t n_;
};

// Back-to-back surfaces from host memory (BASELINE configs[4]; the reference has no counterpart: its
// benches call caf_surface once per iteration, caf_bench.rs:150-168).  One plan + one caf_stream, RAII;
// run() is one native loop over all pairs (caf_stream_run) and returns find_peak's answer per pair.
class CafHipStream {
  public:
    CafHipStream(std::size_t n, const std::vector<double> &freqs_hz, uint32_t fs, int nslots = 2) : n_(n)
    {
        check(caf_plan_create(default_ctx(), n, freqs_hz.data(), freqs_hz.size(), fs, CAF_C128, 0, freqs_hz.size(), &plan_),
              "caf_plan_create");
        if (int rc = caf_stream_create(plan_, 1, nslots, 0, &stream_)) {
            caf_plan_destroy(plan_);
            check(rc, "caf_stream_create");
        }
    }
    CafHipStream(const CafHipStream &) = delete;
    CafHipStream &operator=(const CafHipStream &) = delete;
    ~CafHipStream()
    {
        caf_stream_destroy(stream_);
        caf_plan_destroy(plan_);
    }
    std::vector<std::pair<double, std::size_t>> run(const std::vector<std::vector<Complex64>> &needles,
                                                    const std::vector<std::vector<Complex64>> &haystacks)
    {
        if (needles.size() != haystacks.size()) throw std::runtime_error("CafHipStream::run: needles vs haystacks");
        std::vector<Complex64> a(needles.size() * n_), b(needles.size() * n_);
        for (std::size_t k = 0; k < needles.size(); ++k) {
            if (needles[k].size() != n_ || haystacks[k].size() != n_) throw std::runtime_error("CafHipStream::run: length");
            std::copy(needles[k].begin(), needles[k].end(), a.begin() + k * n_);
            std::copy(haystacks[k].begin(), haystacks[k].end(), b.begin() + k * n_);
        }
        std::vector<caf_peak> pk(needles.size());
        check(caf_stream_run(stream_, a.data(), b.data(), needles.size(), pk.data(), nullptr, nullptr), "caf_stream_run");
        std::vector<std::pair<double, std::size_t>> out;
        for (const caf_peak &p : pk) out.emplace_back(p.freq, static_cast<std::size_t>(p.idx));
        return out;
    }

  private:
    std::size_t n_;
    caf_plan *plan_ = nullptr;
    caf_stream *stream_ = nullptr;
};

// The same over several GPUs (surface-parallel decomposition): one context + plan + stream per entry
// of `devices` (an id may repeat), whole surfaces round-robin, answers in input order, no collective.
class CafHipMultiStream {
  public:
    CafHipMultiStream(const std::vector<int> &devices, std::size_t n, const std::vector<double> &freqs_hz, uint32_t fs,
                      int nslots = 3)
        : n_(n)
    {
        check(caf_multi_stream_create(devices.data(), static_cast<int>(devices.size()), n, freqs_hz.data(), freqs_hz.size(), fs,
                                      CAF_C128, nslots, 0, &ms_),
              "caf_multi_stream_create");
    }
    CafHipMultiStream(const CafHipMultiStream &) = delete;
    CafHipMultiStream &operator=(const CafHipMultiStream &) = delete;
    ~CafHipMultiStream() { caf_multi_stream_destroy(ms_); }
    int devices() const { return caf_multi_stream_devices(ms_); }
    // every later run throws (CAF_ERR_TIMEOUT) instead of waiting longer than `seconds` for a device; 0 = no deadline
    void set_timeout(double seconds) { check(caf_multi_stream_set_timeout(ms_, seconds), "caf_multi_stream_set_timeout"); }
    std::vector<std::pair<double, std::size_t>> run(const std::vector<std::vector<Complex64>> &needles,
                                                    const std::vector<std::vector<Complex64>> &haystacks)
    {
        if (needles.size() != haystacks.size()) throw std::runtime_error("CafHipMultiStream::run: needles vs haystacks");
        std::vector<Complex64> a(needles.size() * n_), b(needles.size() * n_);
        for (std::size_t k = 0; k < needles.size(); ++k) {
            if (needles[k].size() != n_ || haystacks[k].size() != n_) throw std::runtime_error("CafHipMultiStream::run: length");
            std::copy(needles[k].begin(), needles[k].end(), a.begin() + k * n_);
            std::copy(haystacks[k].begin(), haystacks[k].end(), b.begin() + k * n_);
        }
        std::vector<caf_peak> pk(needles.size());
        check(caf_multi_stream_run(ms_, a.data(), b.data(), needles.size(), pk.data(), nullptr, nullptr), "caf_multi_stream_run");
        std::vector<std::pair<double, std::size_t>> out;
        for (const caf_peak &p : pk) out.emplace_back(p.freq, static_cast<std::size_t>(p.idx));
        return out;
    }

  private:
    std::size_t n_;
    caf_multi_stream *ms_ = nullptr;
};

// `impl CafSurface for CafHipMulti`: ONE caf_surface call whose Doppler rows are sharded over several GPUs inside the
// operator, like CafRustFFTThreadpool spreads them over pool workers (mod.rs:391-461): worker r computes the contiguous rows
// [r*F/G, (r+1)*F/G) and writes them into the shared pinned arena in place; find_peak over all rows is joined on the host
// or, with `rccl`, by ncclAllReduce(max) + ncclAllReduce(min key) over xGMI (distinct devices only).  An instance fixes
// (devices, n, freq list, fs) the way Xcor::new fixes n; RAII.
class CafHipMulti {
  public:
    CafHipMulti(const std::vector<int> &devices, std::size_t n, const std::vector<double> &freqs_hz, uint32_t fs, bool rccl = false)
        : n_(n), freqs_(freqs_hz)
    {
        check(caf_multi_surface_create(devices.data(), static_cast<int>(devices.size()), n, freqs_hz.data(), freqs_hz.size(), fs,
                                       CAF_C128, rccl ? CAF_MULTI_REDUCE_RCCL : 0u, &h_),
              "caf_multi_surface_create");
        const std::size_t values = freqs_hz.size() * 2 * n;
        if (values) {
            void *p = nullptr;
            if (int rc = caf_multi_surface_host_alloc(h_, values * sizeof(double), &p)) {
                caf_multi_surface_destroy(h_);
                check(rc, "caf_multi_surface_host_alloc");
            }
            arena_ = static_cast<double *>(p);
        }
    }
    CafHipMulti(const CafHipMulti &) = delete;
    CafHipMulti &operator=(const CafHipMulti &) = delete;
    ~CafHipMulti() { caf_multi_surface_destroy(h_); }  // frees the arena too
    int devices() const { return caf_multi_surface_devices(h_); }
    // the join that cannot be skipped must not wait for ever either (mod.rs:452-457 panics on a dead worker): every later
    // call throws (CAF_ERR_TIMEOUT) instead of waiting longer than `seconds` for a device; 0 = no deadline
    void set_timeout(double seconds) { check(caf_multi_surface_set_timeout(h_, seconds), "caf_multi_surface_set_timeout"); }

    // mod.rs:26-27: rows in freq-list order; the global (freq, idx) of find_peak comes back in `peak`
    std::vector<CafSurfaceRow> caf_surface(const std::vector<Complex64> &needle, const std::vector<Complex64> &haystack,
                                           std::pair<double, std::size_t> *peak = nullptr)
    {
        if (needle.size() != n_ || haystack.size() != n_)  // Xcor::run's assert (xcor_rustfft.rs:54-55)
            throw std::runtime_error("assertion failed: a.len() == self.n");
        const std::size_t F = freqs_.size(), L = 2 * n_;
        std::vector<double> val(F);
        std::vector<uint64_t> idx(F);
        caf_peak pk;
        check(caf_multi_surface_run(h_, needle.data(), haystack.data(), arena_, idx.data(), val.data(), &pk), "caf_multi_surface_run");
        if (peak) *peak = {pk.freq, static_cast<std::size_t>(pk.idx)};
        std::vector<CafSurfaceRow> rows(F);
        for (std::size_t r = 0; r < F; ++r)
            rows[r] = CafSurfaceRow{freqs_[r], std::vector<double>(arena_ + r * L, arena_ + (r + 1) * L),
                                    static_cast<std::size_t>(idx[r]), val[r]};
        return rows;
    }

    // The loop of benches/caf_bench.rs:150-168 as ONE call (caf_multi_surface_run_batch): B pairs in, the B (freq, idx)
    // answers of find_peak out; every device runs ONE launch over its row shard of all B surfaces.  `row_val` (optional)
    // receives the [B][F] row peak values.
    std::vector<std::pair<double, std::size_t>> find_peaks_batch(const std::vector<std::vector<Complex64>> &needles,
                                                                 const std::vector<std::vector<Complex64>> &haystacks,
                                                                 std::vector<double> *row_val = nullptr)
    {
        if (needles.size() != haystacks.size()) throw std::runtime_error("CafHipMulti::find_peaks_batch: needles vs haystacks");
        const std::size_t B = needles.size(), F = freqs_.size();
        std::vector<Complex64> a(B * n_), b(B * n_);
        for (std::size_t k = 0; k < B; ++k) {
            if (needles[k].size() != n_ || haystacks[k].size() != n_)  // Xcor::run's assert (xcor_rustfft.rs:54-55)
                throw std::runtime_error("assertion failed: a.len() == self.n");
            std::copy(needles[k].begin(), needles[k].end(), a.begin() + k * n_);
            std::copy(haystacks[k].begin(), haystacks[k].end(), b.begin() + k * n_);
        }
        std::vector<caf_peak> pk(B);
        std::vector<uint64_t> idx(row_val ? B * F : 0);
        if (row_val) row_val->assign(B * F, 0.0);
        check(caf_multi_surface_run_batch(h_, a.data(), b.data(), B, row_val ? idx.data() : nullptr, row_val ? row_val->data() : nullptr,
                                          pk.data()),
              "caf_multi_surface_run_batch");
        std::vector<std::pair<double, std::size_t>> out;
        for (const caf_peak &p : pk) out.emplace_back(p.freq, static_cast<std::size_t>(p.idx));
        return out;
    }

  private:
    std::size_t n_;
    std::vector<double> freqs_;
    caf_multi_surface *h_ = nullptr;
    double *arena_ = nullptr;
};

// The bench loop of the reference (benches/caf_bench.rs:150-168: one caf_surface + find_peak per iteration) handed over as
// ONE call per B pairs: caf_multi_surface_run_batch over every device given, each device one launch over its Doppler rows of
// all B surfaces, surfaces kept in the devices' HBM (CAF_MULTI_SURFACE_ON_DEVICE), the B peaks joined on the host or, with
// `rccl`, by one grouped ncclAllReduce(max) + ncclAllReduce(min key) per call.  `upload` copies the pairs to every device;
// `find_peaks` then re-runs the resident pairs (inputs already in HBM) and returns the B (freq, idx) answers.  RAII.
class CafHipBatch {
  public:
    CafHipBatch(const std::vector<int> &devices, std::size_t n, const std::vector<double> &freqs_hz, uint32_t fs, bool rccl = false)
        : n_(n)
    {
        check(caf_multi_surface_create(devices.data(), static_cast<int>(devices.size()), n, freqs_hz.data(), freqs_hz.size(), fs,
                                       CAF_C128, CAF_MULTI_SURFACE_ON_DEVICE | (rccl ? CAF_MULTI_REDUCE_RCCL : 0u), &h_),
              "caf_multi_surface_create");
    }
    CafHipBatch(const CafHipBatch &) = delete;
    CafHipBatch &operator=(const CafHipBatch &) = delete;
    ~CafHipBatch() { caf_multi_surface_destroy(h_); }
    // every later call throws (CAF_ERR_TIMEOUT) instead of waiting longer than `seconds` for a device; 0 = no deadline
    void set_timeout(double seconds) { check(caf_multi_surface_set_timeout(h_, seconds), "caf_multi_surface_set_timeout"); }

    // needles / haystacks: B * n contiguous samples each; one call = upload + compute (PCIe-inclusive)
    std::vector<std::pair<double, std::size_t>> upload(const std::vector<Complex64> &needles, const std::vector<Complex64> &haystacks)
    {
        if (needles.size() != haystacks.size() || needles.size() % n_)  // Xcor::run's assert (xcor_rustfft.rs:54-55)
            throw std::runtime_error("assertion failed: a.len() == self.n");
        batch_ = needles.size() / n_;
        return run(needles.data(), haystacks.data());
    }
    // the pairs of the last upload, from HBM
    std::vector<std::pair<double, std::size_t>> find_peaks() { return run(nullptr, nullptr); }
    std::size_t batch() const { return batch_; }

  private:
    std::vector<std::pair<double, std::size_t>> run(const Complex64 *a, const Complex64 *b)
    {
        std::vector<caf_peak> pk(batch_);
        check(caf_multi_surface_run_batch(h_, a, b, batch_, nullptr, nullptr, pk.data()), "caf_multi_surface_run_batch");
        std::vector<std::pair<double, std::size_t>> out;
        out.reserve(batch_);
        for (const caf_peak &p : pk) out.emplace_back(p.freq, static_cast<std::size_t>(p.idx));
        return out;
    }
    std::size_t n_, batch_ = 0;
    caf_multi_surface *h_ = nullptr;
};

// utils.rs:10-35: packed LE f32 I/Q pairs -> Complex64
inline std::vector<Complex64> read_file_c64(const std::string &filename)
{
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("read_file_c64: cannot open " + filename);
    std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (buf.size() % 8) throw std::runtime_error("read_file_c64: trailing partial sample");
    const float *p = reinterpret_cast<const float *>(buf.data());
    std::vector<Complex64> out(buf.size() / 8);
    for (std::size_t i = 0; i < out.size(); ++i) out[i] = Complex64(p[2 * i], p[2 * i + 1]);
    return out;
}

// test.rs:319-331: haystack.resize(needle.len(), 0)
inline std::pair<std::vector<Complex64>, std::vector<Complex64>> load_files(const std::string &needle_filename,
                                                                           const std::string &haystack_filename)
{
    auto needle = read_file_c64(needle_filename);
    auto haystack = read_file_c64(haystack_filename);
    haystack.resize(needle.size(), Complex64(0.0, 0.0));
    return {needle, haystack};
}

// test.rs:335-352
inline std::vector<double> gen_float_shifts(double start, double end, double step)
{
    const int32_t s = static_cast<int32_t>(start * 1000.0), e = static_cast<int32_t>(end * 1000.0);
    const std::size_t st = static_cast<std::size_t>(step * 1000.0);
    if (st == 0) throw std::runtime_error("step_by(0)");
    std::vector<double> shifts;
    for (int64_t m = s; m < e; m += static_cast<int64_t>(st)) shifts.push_back(static_cast<double>(m) / 1e3);
    return shifts;
}

}  // namespace caf
