// pcie_rates.hip -- what the host-pointer API (caf_surface_c128 with a host surface) can reach on this box:
//   * D2H of one 400 x 8192 f64 surface (26 214 400 B) into pinned / pageable / registered memory
//   * a kernel storing the same bytes straight into pinned host memory (zero-copy)
//   * hipHostRegister cost of a caller buffer
//   * memcpy pinned -> pageable with 1..8 threads
//   * launch + completion latencies (stream sync vs a polled pinned word)
// build: hipcc -O3 --offload-arch=gfx950 -o pcie_rates pcie_rates.hip -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                               \
    do {                                                                                    \
        hipError_t e_ = (x);                                                                \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }   \
    } while (0)

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
__global__ void k_flag(unsigned long long *w, unsigned long long v)
{
    __hip_atomic_store(w, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_empty() {}

static void par_memcpy(char *dst, const char *src, size_t bytes, int nthr)
{
    if (nthr <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t per = (bytes / nthr + 4095) & ~(size_t)4095;
    for (int t = 0; t < nthr; ++t) {
        const size_t o = (size_t)t * per;
        if (o >= bytes) break;
        const size_t k = bytes - o < per ? bytes - o : per;
        th.emplace_back([=] { memcpy(dst + o, src + o, k); });
    }
    for (auto &t : th) t.join();
}

int main()
{
    const size_t B = 400ull * 8192 * 8;
    CK(hipSetDevice(0));
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    char *d = nullptr, *hp = nullptr, *hpage = nullptr, *hreg = nullptr;
    CK(hipMalloc(&d, B));
    CK(hipMemset(d, 1, B));
    CK(hipHostMalloc(&hp, B, hipHostMallocDefault));
    hpage = (char *)aligned_alloc(4096, B);
    hreg = (char *)aligned_alloc(4096, B);
    memset(hp, 0, B); memset(hpage, 0, B); memset(hreg, 0, B);
    printf("host threads available: %u\n", std::thread::hardware_concurrency());

    auto rep = [&](const char *name, int reps, auto fn) {
        fn();
        double best = 1e30, tot = 0;
        for (int i = 0; i < reps; ++i) {
            const double t0 = now_us();
            fn();
            const double dt = now_us() - t0;
            best = dt < best ? dt : best;
            tot += dt;
        }
        printf("%-58s best %9.1f us  mean %9.1f us  (%6.1f GB/s best)\n", name, best, tot / reps, B / best / 1e3);
    };
    rep("D2H 26.2 MB -> pinned (hipMemcpyAsync + sync)", 20, [&] {
        CK(hipMemcpyAsync(hp, d, B, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
    rep("D2H 26.2 MB -> pageable (hipMemcpyAsync + sync)", 10, [&] {
        CK(hipMemcpyAsync(hpage, d, B, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
    {
        const double t0 = now_us();
        CK(hipHostRegister(hreg, B, hipHostRegisterDefault));
        const double t1 = now_us();
        printf("hipHostRegister(26.2 MB): %.1f us\n", t1 - t0);
        rep("D2H 26.2 MB -> registered (hipMemcpyAsync + sync)", 20, [&] {
            CK(hipMemcpyAsync(hreg, d, B, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
        char *dreg = nullptr;
        CK(hipHostGetDevicePointer((void **)&dreg, hreg, 0));
        rep("kernel stores -> registered host memory (1024 x 256)", 20, [&] {
            k_copy16<<<1024, 256, 0, s>>>((const uint4 *)d, (uint4 *)dreg, B / 16); CK(hipStreamSynchronize(s)); });
        const double t2 = now_us();
        CK(hipHostUnregister(hreg));
        printf("hipHostUnregister: %.1f us\n", now_us() - t2);
        for (int i = 0; i < 3; ++i) {
            const double a = now_us();
            CK(hipHostRegister(hreg, B, hipHostRegisterDefault));
            const double b = now_us();
            CK(hipHostUnregister(hreg));
            printf("  register again %.1f us, unregister %.1f us\n", b - a, now_us() - b);
        }
    }
    char *dhp = nullptr;
    CK(hipHostGetDevicePointer((void **)&dhp, hp, 0));
    for (int grid : {64, 256, 1024, 4096})
        for (int sc = 0; sc < 1; ++sc) {
            char nm[96];
            snprintf(nm, sizeof nm, "kernel stores -> pinned host memory (%d x 256)", grid);
            rep(nm, 10, [&] { k_copy16<<<grid, 256, 0, s>>>((const uint4 *)d, (uint4 *)dhp, B / 16); CK(hipStreamSynchronize(s)); });
        }
    for (int nthr : {1, 2, 3, 4, 6, 8, 12}) {
        char nm[96];
        snprintf(nm, sizeof nm, "memcpy pinned -> pageable, %d thread(s) (spawn per call)", nthr);
        rep(nm, 10, [&] { par_memcpy(hpage, hp, B, nthr); });
    }
    // chunked D2H into pinned + memcpy by a helper pool, overlapped
    for (int chunks : {8, 16, 32})
        for (int nthr : {2, 4, 8}) {
            std::vector<hipEvent_t> ev(chunks);
            for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            char nm[96];
            snprintf(nm, sizeof nm, "D2H in %d chunks -> pinned, %d copier threads -> pageable", chunks, nthr);
            rep(nm, 8, [&] {
                const size_t per = B / chunks;
                std::atomic<int> ready{0};
                std::vector<std::thread> th;
                for (int t = 0; t < nthr; ++t)
                    th.emplace_back([&, t] {
                        for (int c = 0; c < chunks; ++c) {
                            while (ready.load(std::memory_order_acquire) <= c) __builtin_ia32_pause();
                            const size_t sub = (per / nthr + 63) & ~(size_t)63, o = (size_t)t * sub;
                            if (o < per) memcpy(hpage + c * per + o, hp + c * per + o, per - o < sub ? per - o : sub);
                        }
                    });
                for (int c = 0; c < chunks; ++c) {
                    CK(hipMemcpyAsync(hp + c * per, d + c * per, per, hipMemcpyDeviceToHost, s));
                    CK(hipEventRecord(ev[c], s));
                }
                for (int c = 0; c < chunks; ++c) {
                    CK(hipEventSynchronize(ev[c]));
                    ready.store(c + 1, std::memory_order_release);
                }
                for (auto &t : th) t.join();
            });
            for (auto &e : ev) CK(hipEventDestroy(e));
        }
    // latencies
    unsigned long long *hw = nullptr, *dw = nullptr;
    CK(hipHostMalloc((void **)&hw, 64, hipHostMallocDefault));
    CK(hipHostGetDevicePointer((void **)&dw, hw, 0));
    *hw = 0;
    unsigned long long seq = 0;
    auto lat = [&](const char *name, int reps, auto fn) {
        fn();
        double best = 1e30, tot = 0;
        for (int i = 0; i < reps; ++i) {
            const double t0 = now_us();
            fn();
            const double dt = now_us() - t0;
            best = dt < best ? dt : best;
            tot += dt;
        }
        printf("%-58s best %7.2f us  mean %7.2f us\n", name, best, tot / reps);
    };
    lat("empty kernel + hipStreamSynchronize", 200, [&] { k_empty<<<1, 64, 0, s>>>(); CK(hipStreamSynchronize(s)); });
    lat("flag kernel + poll pinned word", 200, [&] {
        ++seq;
        k_flag<<<1, 64, 0, s>>>(dw, seq);
        while (__atomic_load_n(hw, __ATOMIC_ACQUIRE) < seq) __builtin_ia32_pause();
    });
    {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        k_empty<<<1, 64, 0, s>>>();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        lat("graph(empty kernel) launch + hipStreamSynchronize", 200, [&] { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); });
    }
    lat("memcpy 2 x 64 KiB pageable -> pinned (1 thread)", 200, [&] { memcpy(hp, hpage, 65536); memcpy(hp + 65536, hpage + 65536, 65536); });
    lat("hipMemcpyAsync 128 KiB pageable H2D + sync", 100, [&] { CK(hipMemcpyAsync(d, hpage, 131072, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); });
    lat("hipMemcpyAsync 6.4 KB D2H pinned + sync", 100, [&] { CK(hipMemcpyAsync(hp, d, 6400, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); });
    hipPointerAttribute_t at;
    lat("hipPointerGetAttributes(pageable)", 200, [&] { (void)hipPointerGetAttributes(&at, hpage); (void)hipGetLastError(); });
    lat("hipPointerGetAttributes(pinned)", 200, [&] { (void)hipPointerGetAttributes(&at, hp); });
    return 0;
}
