// C++ counterpart of the reference's multi-thread bench (caf_rust/benches/caf_bench.rs:150-168, bench_rustfft_threadpool): the
// chirp_0 pair, 400 shifts -100 ... 99.5 Hz (caf_bench.rs:26-35), fs = 48000 -- through the C ABI only, as a compiled host
// reaches the engine:
//   (1) the literal loop: one caf_surface + find_peak per iteration (host pointers, the 26 MB surface back as row Vecs)
//   (2) the same call peaks-only (caf_surface_c128 with surface = NULL)
//   (3) the loop handed over as ONE call per B pairs: caf_multi_surface_run_batch on every visible GPU (Doppler rows sharded
//       over the GPUs, surfaces kept in HBM, peaks joined by in-library RCCL when there is more than one GPU), uploading the
//       pairs with every call and from HBM
// usage: caf_bench [data_dir] [B = 256] [calls = 20] [join = auto | rccl | host]
//   join: auto = the in-library RCCL join when more than one GPU is visible, the host join on one GPU (loading librccl and
//   creating a communicator costs a cold process about five seconds and a one-rank exchange moves nothing); rccl / host force one
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "caf_hip.hpp"

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    using namespace caf;
    const std::string dir = argc > 1 ? argv[1] : "tests/golden/data";
    const std::size_t B = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 256;
    const int calls = argc > 3 ? std::atoi(argv[3]) : 20;
    const std::string join_arg = argc > 4 ? argv[4] : "auto";
    const double t_start = now_s();
    auto files = load_files(dir + "/chirp_0_raw.c64", dir + "/chirp_0_T+202samp_F+69.25Hz.c64");
    const auto &needle = files.first, &hay = files.second;
    const std::size_t n = needle.size();
    std::vector<double> shifts;
    for (int m = -100000; m < 100000; m += 500) shifts.push_back(m / 1e3);

    // (1) the literal loop
    std::pair<double, std::size_t> pk;
    for (int i = 0; i < 3; ++i) pk = CafHip::find_peak(CafHip::caf_surface(needle, hay, shifts, 48000));
    double t0 = now_s();
    const int lit = 10;
    for (int i = 0; i < lit; ++i) pk = CafHip::find_peak(CafHip::caf_surface(needle, hay, shifts, 48000));
    const double ms_literal = (now_s() - t0) / lit * 1e3;
    if (pk.first != 69.0 || pk.second != 202) { std::fprintf(stderr, "wrong answer (%g, %zu)\n", pk.first, pk.second); return 1; }

    // (2) peaks only
    std::vector<uint64_t> idx(shifts.size());
    std::vector<double> val(shifts.size());
    caf_peak p{};
    auto peaks_only = [&] {
        check(caf_surface_c128(default_ctx(), reinterpret_cast<const double *>(needle.data()), reinterpret_cast<const double *>(hay.data()), n,
                               shifts.data(), shifts.size(), 48000, nullptr, idx.data(), val.data(), &p), "caf_surface_c128");
    };
    for (int i = 0; i < 10; ++i) peaks_only();
    t0 = now_s();
    const int po = 500;
    for (int i = 0; i < po; ++i) peaks_only();
    const double us_peaks = (now_s() - t0) / po * 1e6;
    if (p.freq != 69.0 || p.idx != 202) { std::fprintf(stderr, "wrong answer (peaks only)\n"); return 1; }

    // (3) B pairs per call: pair b = the chirp_0 pair with the haystack delayed by (b % 64) more samples -> tau = 202 + b % 64
    std::vector<Complex64> nds(B * n), hss(B * n, Complex64(0.0, 0.0));
    for (std::size_t b = 0; b < B; ++b) {
        const std::size_t d = b % 64;
        std::copy(needle.begin(), needle.end(), nds.begin() + b * n);
        std::copy(hay.begin(), hay.end() - d, hss.begin() + b * n + d);
    }
    const int ndev = caf_device_count();
    std::vector<int> devices;
    for (int d = 0; d < ndev; ++d) devices.push_back(d);
    // the in-library RCCL join (if librccl can be loaded: the library dlopen()s it), or the host join
    std::unique_ptr<CafHipBatch> bp;
    const bool want_rccl = join_arg == "rccl" || (join_arg == "auto" && ndev > 1);
    const char *join = want_rccl ? "rccl" : "host";
    const double t_create = now_s();
    if (want_rccl) {
        try {
            bp.reset(new CafHipBatch(devices, n, shifts, 48000, /*rccl=*/true));
        } catch (const std::runtime_error &e) {
            std::fprintf(stderr, "RCCL join not available (%s): joining the peaks on the host\n", e.what());
            join = "host";
        }
    }
    if (!bp) bp.reset(new CafHipBatch(devices, n, shifts, 48000, /*rccl=*/false));
    const double create_s = now_s() - t_create;
    CafHipBatch &batch = *bp;
    // a compiled host has no watchdog of its own: the library's deadline ends a call that waits for a device that does not
    // answer (CAF_ERR_TIMEOUT -> the wrapper throws), instead of hanging this loop for ever
    batch.set_timeout(60.0);
    auto got = batch.upload(nds, hss);  // allocations + first upload
    t0 = now_s();
    for (int i = 0; i < calls; ++i) got = batch.upload(nds, hss);
    const double ms_upload = (now_s() - t0) / calls * 1e3;
    for (int i = 0; i < 3; ++i) got = batch.find_peaks();
    t0 = now_s();
    for (int i = 0; i < calls; ++i) got = batch.find_peaks();
    const double ms_resident = (now_s() - t0) / calls * 1e3;
    for (std::size_t b = 0; b < B; ++b)
        if (got[b].second != 202 + b % 64 || (got[b].first != 69.0 && got[b].first != 69.5)) {
            std::fprintf(stderr, "pair %zu: wrong answer (%g, %zu)\n", b, got[b].first, got[b].second);
            return 1;
        }
    std::printf("{\"shape\": \"400x8192 complex128, chirp_0 pair\", \"gpus\": %d, \"peak_join\": \"%s\", \"literal_loop_ms_per_surface\": %.3f, "
                "\"peaks_only_us_per_surface\": %.1f, \"batch\": %zu, \"batch_with_upload_ms_per_call\": %.3f, "
                "\"batch_with_upload_surfaces_per_s\": %.0f, \"batch_resident_ms_per_call\": %.3f, \"batch_resident_surfaces_per_s\": %.0f, "
                "\"call_timeout_s\": 60, \"create_s\": %.2f, \"process_s\": %.2f, \"published_rust_threadpool_ms_per_surface_R9_3900X\": 28}\n",
                ndev, join, ms_literal, us_peaks, B, ms_upload, B / ms_upload * 1e3, ms_resident, B / ms_resident * 1e3,
                create_s, now_s() - t_start);
    return 0;
}
