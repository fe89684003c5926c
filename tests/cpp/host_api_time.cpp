// host_api_time.cpp -- the literal drop-in calls timed from C: what a compiled host pays per call of
//   caf_surface_c128 (peaks only / with the 26 MB surface), caf_find_peak, caf_apply_freq_shift_c128, caf_xcor_c128
// on the reference's bench shape (benches/caf_bench.rs:23-40,170-179: n = 4096, 400 shifts -100..99.5 Hz,
// fs = 48000, apply_freq_shift(needle, 77.77 Hz)), next to the PCIe floor of this box (one pinned D2H copy of the
// same 26 214 400 bytes).  Prints one JSON object; bench.py puts it into `extra.host_api`.
// build: hipcc -O2 -std=c++17 -I include -o host_api_time host_api_time.cpp -L caf_cookoff_amd -lcaf_hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "caf_hip.h"

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define CAFCK(x)                                                                                   \
    do {                                                                                           \
        int rc_ = (x);                                                                             \
        if (rc_ != CAF_OK) { std::fprintf(stderr, "%s: %s\n", #x, caf_last_error_string()); return 1; } \
    } while (0)
#define HIPCK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

struct Stat { double median, best, mean; };
template <typename F>
static Stat timeit(int warm, int reps, F fn)
{
    for (int i = 0; i < warm; ++i) fn();
    std::vector<double> t(reps);
    for (int i = 0; i < reps; ++i) {
        const double t0 = now_us();
        fn();
        t[i] = now_us() - t0;
    }
    std::sort(t.begin(), t.end());
    double sum = 0;
    for (double v : t) sum += v;
    return {t[reps / 2], t[0], sum / reps};
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? std::atoi(argv[1]) : 200;
    const size_t n = 4096, F = 400, L = 2 * n;
    const uint32_t fs = 48000;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> g(0.0, 0.25);
    // a needle and the same signal delayed by 202 samples and shifted by 69.25 Hz (the chirp_0 truth of test.rs:17-30)
    std::vector<double> needle(2 * n), hay(2 * n, 0.0);
    for (auto &v : needle) v = g(rng);
    for (size_t i = 202; i < n; ++i) {
        const double ph = 2.0 * M_PI * 69.25 * (double)i / fs, c = std::cos(ph), s = std::sin(ph);
        const double re = needle[2 * (i - 202)], im = needle[2 * (i - 202) + 1];
        hay[2 * i] = re * c - im * s;
        hay[2 * i + 1] = re * s + im * c;
    }
    std::vector<double> freqs(F), freqs2(100);
    for (size_t k = 0; k < F; ++k) freqs[k] = (-100000 + 500 * (long)k) / 1e3;  // caf_bench.rs:31-35
    for (size_t k = 0; k < 100; ++k) freqs2[k] = (60000 + 200 * (long)k) / 1e3;
    caf_ctx *ctx = nullptr;
    CAFCK(caf_ctx_create(0, &ctx));
    std::vector<uint64_t> ridx(F);
    std::vector<double> rval(F);
    caf_peak pk{}, pk2{};
    const size_t surf_bytes = F * L * sizeof(double);
    double *surf_page = (double *)std::aligned_alloc(4096, surf_bytes);
    std::memset(surf_page, 0, surf_bytes);
    double *surf_pin = nullptr;
    CAFCK(caf_host_alloc(ctx, surf_bytes, (void **)&surf_pin));
    double *surf_reg = (double *)std::aligned_alloc(4096, surf_bytes);
    std::memset(surf_reg, 0, surf_bytes);
    const double t_reg0 = now_us();
    CAFCK(caf_host_register(ctx, surf_reg, surf_bytes));
    const double register_us = now_us() - t_reg0;

    // ---- peaks only: caf_surface (no surface copy) [+ find_peak over the returned rows, as the callers do]
    const Stat peaks = timeit(10, reps, [&] {
        caf_surface_c128(ctx, needle.data(), hay.data(), n, freqs.data(), F, fs, nullptr, ridx.data(), rval.data(), &pk);
    });
    if (!(pk.freq == 69.0 || pk.freq == 69.5) || pk.idx != 202) { std::fprintf(stderr, "wrong peak (%g, %llu)\n", pk.freq, (unsigned long long)pk.idx); return 1; }
    const Stat peaks_fp = timeit(10, reps, [&] {
        caf_surface_c128(ctx, needle.data(), hay.data(), n, freqs.data(), F, fs, nullptr, ridx.data(), rval.data(), &pk);
        caf_find_peak(ctx, freqs.data(), ridx.data(), rval.data(), F, &pk2);
    });
    if (pk2.freq != pk.freq || pk2.idx != pk.idx) { std::fprintf(stderr, "find_peak disagrees\n"); return 1; }
    const Stat find_peak = timeit(10, reps, [&] { caf_find_peak(ctx, freqs.data(), ridx.data(), rval.data(), F, &pk2); });
    // ---- with the surface
    const int sreps = std::max(20, reps / 4);
    const Stat s_page = timeit(3, sreps, [&] {
        caf_surface_c128(ctx, needle.data(), hay.data(), n, freqs.data(), F, fs, surf_page, ridx.data(), rval.data(), &pk);
    });
    const Stat s_pin = timeit(3, sreps, [&] {
        caf_surface_c128(ctx, needle.data(), hay.data(), n, freqs.data(), F, fs, surf_pin, ridx.data(), rval.data(), &pk);
    });
    const Stat s_reg = timeit(3, sreps, [&] {
        caf_surface_c128(ctx, needle.data(), hay.data(), n, freqs.data(), F, fs, surf_reg, ridx.data(), rval.data(), &pk);
    });
    // the three surfaces are the same bits, and row 338 (69.0 Hz) peaks where the row record says
    if (std::memcmp(surf_page, surf_pin, surf_bytes) || std::memcmp(surf_page, surf_reg, surf_bytes)) { std::fprintf(stderr, "surfaces differ\n"); return 1; }
    if (surf_page[(size_t)pk.row * L + pk.idx] != pk.val) { std::fprintf(stderr, "surface[peak] != peak value\n"); return 1; }
    // ---- PCIe floor: one pinned D2H copy of the same number of bytes
    void *d = nullptr, *hp = nullptr;
    hipStream_t st;
    HIPCK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIPCK(hipMalloc(&d, surf_bytes));
    HIPCK(hipMemset(d, 0, surf_bytes));
    HIPCK(hipHostMalloc(&hp, surf_bytes, hipHostMallocDefault));
    std::memset(hp, 0, surf_bytes);
    const Stat floor = timeit(3, sreps, [&] {
        (void)hipMemcpyAsync(hp, d, surf_bytes, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
    });
    // ---- alternating two cached shapes (n, freq list): no plan rebuild between calls
    std::vector<double> needle2(2 * 1024), hay2(2 * 1024);
    for (auto &v : needle2) v = g(rng);
    for (auto &v : hay2) v = g(rng);
    std::vector<uint64_t> ridx2(100);
    std::vector<double> rval2(100);
    const Stat alt = timeit(4, reps, [&] {
        caf_surface_c128(ctx, needle.data(), hay.data(), n, freqs.data(), F, fs, nullptr, ridx.data(), rval.data(), &pk);
        caf_surface_c128(ctx, needle2.data(), hay2.data(), 1024, freqs2.data(), 100, fs, nullptr, ridx2.data(), rval2.data(), &pk2);
    });
    // ---- apply_freq_shift (benches/caf_bench.rs:170-179; README.md:117-120: rust 120 us) and xcor
    std::vector<double> shifted(2 * n), xc(2 * L), xa(2 * L), xb(2 * L);
    const Stat shift = timeit(10, reps, [&] { caf_apply_freq_shift_c128(ctx, needle.data(), n, 77.77, fs, shifted.data()); });
    if (shifted[0] != needle[0] || shifted[1] != needle[1]) { std::fprintf(stderr, "apply_shift: sample 0 changed\n"); return 1; }
    for (auto &v : xa) v = g(rng);
    for (auto &v : xb) v = g(rng);
    const Stat xcor = timeit(5, std::max(20, reps / 4), [&] { caf_xcor_c128(ctx, xa.data(), xb.data(), L, xc.data()); });

    std::printf("{\"shape\": \"400x8192 complex128, chirp_0-like pair\", \"reps\": %d, "
                "\"peaks_only_us\": %.2f, \"peaks_only_best_us\": %.2f, \"peaks_only_plus_find_peak_us\": %.2f, \"find_peak_us\": %.2f, "
                "\"with_surface_ms\": %.4f, \"with_surface_best_ms\": %.4f, "
                "\"with_surface_in_place_ms\": %.4f, \"with_surface_registered_ms\": %.4f, "
                "\"pcie_floor_ms\": %.4f, \"with_surface_over_floor\": %.3f, \"in_place_over_floor\": %.3f, "
                "\"host_register_26MB_us\": %.1f, \"alternating_two_shapes_us_per_pair\": %.2f, "
                "\"apply_shift_4096_us\": %.2f, \"apply_shift_published_rust_us\": 120, \"xcor_8192_us\": %.2f, "
                "\"surface_bytes\": %zu}\n",
                reps, peaks.median, peaks.best, peaks_fp.median, find_peak.median, s_page.median / 1e3, s_page.best / 1e3,
                s_pin.median / 1e3, s_reg.median / 1e3, floor.median / 1e3, s_page.median / floor.median,
                s_pin.median / floor.median, register_us, alt.median, shift.median, xcor.median, surf_bytes);
    CAFCK(caf_host_unregister(ctx, surf_reg));
    CAFCK(caf_host_free(ctx, surf_pin));
    CAFCK(caf_ctx_destroy(ctx));
    return 0;
}
