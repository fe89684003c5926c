// Integration tests for chirp 0-9 through the C++ mirror of the CafSurface trait --
// the C++ counterpart of the reference's caf_rust/tests/test.rs (same files, same shift
// lists, same exact-equality assertions on (freq, samp_idx)).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "caf_hip.hpp"

using namespace caf;

static int failures = 0;
static std::string data_dir = "tests/golden/data/";

#define ASSERT_EQ(a, b)                                                                              \
    do {                                                                                             \
        if (!((a) == (b))) {                                                                         \
            std::printf("  assertion failed: `(left == right)` left: `%.17g`, right: `%.17g` (%s:%d)\n", \
                        (double)(a), (double)(b), __FILE__, __LINE__);                               \
            ++failures;                                                                              \
        }                                                                                            \
    } while (0)

static void kat(const char *name, const char *needle_file, const char *haystack_file, double start, double end,
                double step, double want_freq, std::size_t want_idx)
{
    // Read the chirp reference and modified files (test.rs:150-153 style)
    auto files = load_files(data_dir + needle_file, data_dir + haystack_file);
    auto shifts = gen_float_shifts(start, end, step);
    // Get the CAF estimates
    auto surface = CafHip::caf_surface(files.first, files.second, shifts, 48000);
    auto peak = CafHip::find_peak(std::move(surface));
    // Confirm correct results
    const int before = failures;
    ASSERT_EQ(peak.first, want_freq);
    ASSERT_EQ(peak.second, want_idx);
    std::printf("test %s ... %s\n", name, failures == before ? "ok" : "FAILED");
}

int main(int argc, char **argv)
{
    if (argc > 1) data_dir = std::string(argv[1]) + "/";
    kat("test_hip_chirp0", "chirp_0_raw.c64", "chirp_0_T+202samp_F+69.25Hz.c64", -100.0, 100.0, 0.25, 69.25, 202);
    kat("test_hip_chirp1", "chirp_1_raw.c64", "chirp_1_T+78samp_F+35.99Hz.c64", -50.0, 50.0, 1.0, 36.0, 78);
    kat("test_hip_chirp2", "chirp_2_raw.c64", "chirp_2_T+169samp_F+32.16Hz.c64", 30.0, 35.0, 0.05, 32.15, 169);
    kat("test_hip_chirp3", "chirp_3_raw.c64", "chirp_3_T+151samp_F-76.22Hz.c64", -100.0, 100.0, 0.25, -76.25, 151);
    kat("test_hip_chirp4", "chirp_4_raw.c64", "chirp_4_T+70samp_F+82.89Hz.c64", 80.0, 100.0, 0.1, 82.9, 70);
    kat("test_hip_chirp5", "chirp_5_raw.c64", "chirp_5_T+177samp_F-92.72Hz.c64", -100.0, 100.0, 0.25, -92.75, 177);
    kat("test_hip_chirp6", "chirp_6_raw.c64", "chirp_6_T+15samp_F-49.69Hz.c64", -100.0, 100.0, 0.25, -49.75, 15);
    kat("test_hip_chirp7", "chirp_7_raw.c64", "chirp_7_T+84samp_F+68.26Hz.c64", -100.0, 100.0, 0.25, 68.25, 84);
    kat("test_hip_chirp8", "chirp_8_raw.c64", "chirp_8_T+80samp_F-46.28Hz.c64", -100.0, 100.0, 0.25, -46.25, 80);
    kat("test_hip_chirp9", "chirp_9_raw.c64", "chirp_9_T+176samp_F+61.49Hz.c64", -100.0, 100.0, 0.5, 61.5, 176);
    // apply_freq_shift: sample 0 untouched, |out| == |in| (mod.rs:57-61)
    {
        auto needle = read_file_c64(data_dir + "chirp_0_raw.c64");
        auto out = CafHip::apply_freq_shift(needle, 77.77, 48000);  // caf_bench.rs:172-177
        const int before = failures;
        ASSERT_EQ(out[0].real(), needle[0].real());
        ASSERT_EQ(out[0].imag(), needle[0].imag());
        ASSERT_EQ(out.size(), needle.size());
        std::printf("test apply_freq_shift ... %s\n", failures == before ? "ok" : "FAILED");
    }
    // Xcor length assert (xcor_rustfft.rs:54-55)
    {
        bool threw = false;
        try {
            Xcor x(8);
            x.run(std::vector<Complex64>(8), std::vector<Complex64>(4));
        } catch (const std::runtime_error &) { threw = true; }
        ASSERT_EQ(threw, true);
        std::printf("test xcor_length_assert ... %s\n", threw ? "ok" : "FAILED");
    }
    // streaming (BASELINE configs[4]): the ten pairs back to back through one caf_stream on chirp 9's shift
    // list give, pair by pair, what caf_surface + find_peak give on that list; pair 9 is the reference's KAT
    {
        const char *hay[10] = {"chirp_0_T+202samp_F+69.25Hz.c64", "chirp_1_T+78samp_F+35.99Hz.c64",
                               "chirp_2_T+169samp_F+32.16Hz.c64", "chirp_3_T+151samp_F-76.22Hz.c64",
                               "chirp_4_T+70samp_F+82.89Hz.c64", "chirp_5_T+177samp_F-92.72Hz.c64",
                               "chirp_6_T+15samp_F-49.69Hz.c64", "chirp_7_T+84samp_F+68.26Hz.c64",
                               "chirp_8_T+80samp_F-46.28Hz.c64", "chirp_9_T+176samp_F+61.49Hz.c64"};
        auto shifts = gen_float_shifts(-100.0, 100.0, 0.5);
        std::vector<std::vector<Complex64>> nd, hs;
        for (int k = 0; k < 10; ++k) {
            auto files = load_files(data_dir + "chirp_" + std::to_string(k) + "_raw.c64", data_dir + hay[k]);
            nd.push_back(files.first);
            hs.push_back(files.second);
        }
        CafHipStream stream(nd[0].size(), shifts, 48000);
        auto got = stream.run(nd, hs);
        const int before = failures;
        for (int k = 0; k < 10; ++k) {
            auto want = CafHip::find_peak(CafHip::caf_surface(nd[k], hs[k], shifts, 48000));
            ASSERT_EQ(got[k].first, want.first);
            ASSERT_EQ(got[k].second, want.second);
        }
        ASSERT_EQ(got[9].first, 61.5);
        ASSERT_EQ(got[9].second, 176);
        std::printf("test hip_stream_ten_pairs ... %s\n", failures == before ? "ok" : "FAILED");
        // the same ten pairs, whole surfaces round-robin over TWO contexts on GPU 0 (caf_multi_stream_*: the
        // surface-parallel multi-GPU driver, one host thread per context): same answers, in input order
        CafHipMultiStream multi({0, 0}, nd[0].size(), shifts, 48000, 2);
        multi.set_timeout(30.0);  // ABI 5: a deadline changes nothing about a healthy run (polled waits, same answers)
        auto got2 = multi.run(nd, hs);
        const int before2 = failures;
        ASSERT_EQ(multi.devices(), 2);
        for (int k = 0; k < 10; ++k) {
            ASSERT_EQ(got2[k].first, got[k].first);
            ASSERT_EQ(got2[k].second, got[k].second);
        }
        std::printf("test hip_multi_stream_ten_pairs ... %s\n", failures == before2 ? "ok" : "FAILED");
    }
    // the literal drop-in call with the surface written in place into pinned memory (caf_host_alloc) and into a
    // pageable buffer: same bits, and the row record points at the row's own maximum (mod.rs:143-151)
    {
        auto files = load_files(data_dir + "chirp_4_raw.c64", data_dir + "chirp_4_T+70samp_F+82.89Hz.c64");
        auto shifts = gen_float_shifts(80.0, 100.0, 0.1);
        const std::size_t n = files.first.size(), F = shifts.size(), L = 2 * n;
        std::vector<double> pageable(F * L), val(F), val2(F);
        std::vector<uint64_t> idx(F), idx2(F);
        caf_peak pk, pk2;
        void *pinned = nullptr;
        check(caf_host_alloc(default_ctx(), F * L * sizeof(double), &pinned), "caf_host_alloc");
        const double *nd = reinterpret_cast<const double *>(files.first.data());
        const double *hs = reinterpret_cast<const double *>(files.second.data());
        check(caf_surface_c128(default_ctx(), nd, hs, n, shifts.data(), F, 48000, pageable.data(), idx.data(), val.data(), &pk), "caf_surface_c128");
        check(caf_surface_c128(default_ctx(), nd, hs, n, shifts.data(), F, 48000, static_cast<double *>(pinned), idx2.data(), val2.data(), &pk2), "caf_surface_c128 (in place)");
        const int before = failures;
        ASSERT_EQ(std::memcmp(pageable.data(), pinned, F * L * sizeof(double)), 0);
        ASSERT_EQ(pk.freq, 82.9);   // test.rs:207-220
        ASSERT_EQ(pk.idx, 70u);
        ASSERT_EQ(pk2.freq, pk.freq);
        ASSERT_EQ(pk2.idx, pk.idx);
        ASSERT_EQ(pageable[static_cast<std::size_t>(pk.row) * L + pk.idx], pk.val);
        check(caf_host_free(default_ctx(), pinned), "caf_host_free");
        std::printf("test hip_surface_in_place ... %s\n", failures == before ? "ok" : "FAILED");
    }
    // `impl CafSurface for CafHipMulti`: the rows of ONE surface sharded over workers inside the operator (the
    // reference's threadpool shape, mod.rs:391-461), here two and three contexts on GPU 0: every row record and every
    // surface value equal to the unsharded call's, and the reference's own answers on KAT 0, 2 (tightest margin), 4
    {
        struct K { const char *nd, *hs; double a, b, st, f; std::size_t idx; };
        const K ks[3] = {{"chirp_0_raw.c64", "chirp_0_T+202samp_F+69.25Hz.c64", -100.0, 100.0, 0.25, 69.25, 202},
                         {"chirp_2_raw.c64", "chirp_2_T+169samp_F+32.16Hz.c64", 30.0, 35.0, 0.05, 32.15, 169},
                         {"chirp_4_raw.c64", "chirp_4_T+70samp_F+82.89Hz.c64", 80.0, 100.0, 0.1, 82.9, 70}};
        const int before = failures;
        for (const K &k : ks) {
            auto files = load_files(data_dir + k.nd, data_dir + k.hs);
            auto shifts = gen_float_shifts(k.a, k.b, k.st);
            auto want = CafHip::caf_surface(files.first, files.second, shifts, 48000);
            for (int workers = 2; workers <= 3; ++workers) {
                CafHipMulti multi(std::vector<int>(workers, 0), files.first.size(), shifts, 48000);
                if (workers == 3) multi.set_timeout(30.0);  // ABI 5: with and without a deadline, the same bits as the unsharded call
                ASSERT_EQ(multi.devices(), workers);
                std::pair<double, std::size_t> pk;
                auto got = multi.caf_surface(files.first, files.second, &pk);
                ASSERT_EQ(pk.first, k.f);
                ASSERT_EQ(pk.second, k.idx);
                ASSERT_EQ(got.size(), want.size());
                for (std::size_t r = 0; r < want.size() && r < got.size(); ++r) {
                    ASSERT_EQ(got[r].xcor_peak_idx, want[r].xcor_peak_idx);
                    ASSERT_EQ(got[r].xcor_peak_val, want[r].xcor_peak_val);
                    ASSERT_EQ(std::memcmp(got[r].xcor_mag.data(), want[r].xcor_mag.data(), want[r].xcor_mag.size() * sizeof(double)), 0);
                }
                auto pk2 = CafHip::find_peak(std::move(got));  // the joined rows scanned like mod.rs:31-42
                ASSERT_EQ(pk2.first, k.f);
                ASSERT_EQ(pk2.second, k.idx);
            }
        }
        std::printf("test hip_multi_row_shards ... %s\n", failures == before ? "ok" : "FAILED");
    }
    // The bench loop as ONE call (caf_multi_surface_run_batch): the five chirp pairs that share the (-100, 100, 0.25) grid of
    // test.rs (k = 0, 3, 5, 6, 7) as one batch over three workers; the reference's answers, and every row peak value equal to
    // the single-surface call's to the last bit or two (the batched launch and the one-launch surface are two kernels of
    // the same arithmetic: they need not contract multiply-adds alike; bit-equality with the batched device path is what
    // tests/test_gpu_multi.py checks)
    {
        struct K { const char *nd, *hs; double f; std::size_t idx; };
        const K ks[5] = {{"chirp_0_raw.c64", "chirp_0_T+202samp_F+69.25Hz.c64", 69.25, 202},    // test.rs:17-30
                         {"chirp_3_raw.c64", "chirp_3_T+151samp_F-76.22Hz.c64", -76.25, 151},   // :188-201
                         {"chirp_5_raw.c64", "chirp_5_T+177samp_F-92.72Hz.c64", -92.75, 177},   // :226-239
                         {"chirp_6_raw.c64", "chirp_6_T+15samp_F-49.69Hz.c64", -49.75, 15},     // :245-258
                         {"chirp_7_raw.c64", "chirp_7_T+84samp_F+68.26Hz.c64", 68.25, 84}};     // :264-277
        const int before = failures;
        auto shifts = gen_float_shifts(-100.0, 100.0, 0.25);
        std::vector<std::vector<Complex64>> nds, hss;
        for (const K &k : ks) {
            auto files = load_files(data_dir + k.nd, data_dir + k.hs);
            nds.push_back(files.first);
            hss.push_back(files.second);
        }
        CafHipMulti multi(std::vector<int>(3, 0), nds[0].size(), shifts, 48000);
        multi.set_timeout(30.0);
        std::vector<double> row_val;
        auto peaks = multi.find_peaks_batch(nds, hss, &row_val);
        ASSERT_EQ(peaks.size(), 5u);
        for (std::size_t b = 0; b < 5 && b < peaks.size(); ++b) {
            ASSERT_EQ(peaks[b].first, ks[b].f);
            ASSERT_EQ(peaks[b].second, ks[b].idx);
            auto want = CafHip::caf_surface(nds[b], hss[b], shifts, 48000);
            for (std::size_t r = 0; r < want.size(); ++r) {
                const double got = row_val[b * shifts.size() + r], ref = want[r].xcor_peak_val;
                ASSERT_EQ(std::fabs(got - ref) <= 1e-12 * std::fabs(ref), true);
            }
        }
        std::printf("test hip_multi_batch ... %s\n", failures == before ? "ok" : "FAILED");
    }
    std::printf("test result: %s. %d failed\n", failures ? "FAILED" : "ok", failures);
    return failures ? 1 : 0;
}
