// caf_api.hip -- C-ABI implementation (include/caf_hip.h) over the gfx950 kernels.
// No torch types, no CPU fallback: every entry point fails loudly without a GPU.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/caf_hip.h"
#include "kernels_fused4096.hpp"
#include "kernels_seq4096.hpp"
#include "kernels_surf4096.hpp"
#include "kernels_duo4096.hpp"
#include "kernels_chain.hpp"
#include "kernels_small.hpp"
#include "kernels_xcor.hpp"
#include "kernels_generic.hpp"

using namespace caf;

// ------------------------------------------------------------------ errors --
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return fail(CAF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                              \
    } while (0)

#define KCHK() HIPCHK(hipGetLastError())

// No C++ exception leaves the library (include/caf_hip.h: "never unwind across the boundary"): every int-returning
// entry point runs inside this guard (std::bad_alloc of a container -> CAF_ERR_NOMEM).
#define CAF_GUARD_BEGIN try {
#define CAF_GUARD_END                                                                                 \
    }                                                                                                 \
    catch (const std::bad_alloc &) { return fail(CAF_ERR_NOMEM, "%s: out of host memory", __func__); } \
    catch (const std::exception &e) { return fail(CAF_ERR_STATE, "%s: C++ exception: %s", __func__, e.what()); } \
    catch (...) { return fail(CAF_ERR_STATE, "%s: unknown C++ exception", __func__); }


static bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
static size_t elem_size(int dtype) { return dtype == CAF_C128 ? 16 : 8; }
static size_t real_size(int dtype) { return dtype == CAF_C128 ? 8 : 4; }

// -------------------------------------------------------------- allocations --
// Every device / pinned allocation of the library goes through these four functions.  With red zones switched on
// (caf_debug_guard_bands, a process-wide debug setting; GPU AddressSanitizer does not exist on this hardware pool) an
// allocation of `bytes` becomes [guard | bytes | guard] with both guards filled with 0xA5, and
// caf_debug_check_guards() verifies every live allocation's guards: a kernel that stores outside a table, a slab, a
// staging buffer or a surface is caught after the fact.  Off (the default) they are hipMalloc / hipHostMalloc.
struct GuardRec {
    char *base = nullptr;  // what the runtime returned
    size_t bytes = 0, guard = 0;
    int device = 0;
    bool pinned = false;
    const char *file = "";  // allocation site
    int line = 0;
};
static std::mutex g_guard_mu;
static std::map<void *, GuardRec> g_guarded;  // user pointer -> record
static std::atomic<size_t> g_guard_bytes{0};
static std::atomic<size_t> g_guard_live{0};
static constexpr unsigned char GUARD_FILL = 0xA5;

static hipError_t raw_pinned_alloc(void **p, size_t bytes)
{
    // explicitly coherent (fine-grained) and mapped, not "whatever the runtime's default is": the kernels read and
    // write this memory in place and the host polls it; portable: every device of a caf_multi_* object may address it
    return hipHostMalloc(p, bytes, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable);
}

static hipError_t guarded_alloc(void **p, size_t bytes, bool pinned, const char *file, int line)
{
    const size_t g = g_guard_bytes.load(std::memory_order_relaxed);
    if (g == 0) return pinned ? raw_pinned_alloc(p, bytes) : hipMalloc(p, bytes);
    char *base = nullptr;
    hipError_t e = pinned ? raw_pinned_alloc((void **)&base, bytes + 2 * g) : hipMalloc((void **)&base, bytes + 2 * g);
    if (e != hipSuccess) return e;
    if (pinned) {
        memset(base, GUARD_FILL, g);
        memset(base + g + bytes, GUARD_FILL, g);
    } else {
        e = hipMemset(base, GUARD_FILL, g);
        if (e == hipSuccess) e = hipMemset(base + g + bytes, GUARD_FILL, g);
        if (e == hipSuccess) e = hipDeviceSynchronize();  // the library's streams do not wait for the null stream
        if (e != hipSuccess) { (void)hipFree(base); return e; }
    }
    GuardRec r;
    r.base = base; r.bytes = bytes; r.guard = g; r.pinned = pinned; r.line = line;
    r.file = strrchr(file, '/') ? strrchr(file, '/') + 1 : file;
    (void)hipGetDevice(&r.device);
    try {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        g_guarded[base + g] = r;
    } catch (...) {  // (the registry could not grow: give the memory back rather than hand out an unchecked block)
        (void)(pinned ? hipHostFree(base) : hipFree(base));
        return hipErrorOutOfMemory;
    }
    g_guard_live.fetch_add(1);
    *p = base + g;
    return hipSuccess;
}

static hipError_t guarded_free(void *p, bool pinned)
{
    if (!p) return hipSuccess;
    if (g_guard_live.load() != 0) {
        void *base = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_guard_mu);
            auto it = g_guarded.find(p);
            if (it != g_guarded.end()) { base = it->second.base; g_guarded.erase(it); }
        }
        if (base) { g_guard_live.fetch_sub(1); p = base; }
    }
    return pinned ? hipHostFree(p) : hipFree(p);
}

#define dev_alloc(pp, bytes) guarded_alloc((void **)(pp), (bytes), false, __FILE__, __LINE__)
#define pinned_alloc(pp, bytes) guarded_alloc((void **)(pp), (bytes), true, __FILE__, __LINE__)
static hipError_t dev_free(void *p) { return guarded_free(p, false); }
static hipError_t pin_free(void *p) { return guarded_free(p, true); }

// ----------------------------------------------------------------- structs --
struct MeasureCtx;   // measurement build only: measure/dispatch.inc
struct MeasurePlan;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return CAF_OK;
        if (p) { (void)dev_free(p); p = nullptr; cap = 0; }
        if (bytes == 0) return CAF_OK;
        hipError_t e = dev_alloc(&p, bytes);
        if (e != hipSuccess) { p = nullptr; return fail(CAF_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
        cap = bytes;
        return CAF_OK;
    }
    void release() { if (p) (void)dev_free(p); p = nullptr; cap = 0; }
};

// a pinned staging buffer of the host-pointer entry points and its device mapping (grows, never shrinks)
struct PinBuf {
    void *h = nullptr, *m = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return CAF_OK;
        release();
        hipError_t e = pinned_alloc(&h, bytes);
        if (e != hipSuccess) { h = nullptr; return fail(CAF_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
        e = hipHostGetDevicePointer(&m, h, 0);
        if (e != hipSuccess) { (void)pin_free(h); h = nullptr; return fail(CAF_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
        cap = bytes;
        return CAF_OK;
    }
    void release() { if (h) (void)pin_free(h); h = m = nullptr; cap = 0; }
};

// One cached (n, freq list, fs, dtype) of the host-pointer caf_surface_* entry points: its plan (what
// Xcor::new buys, xcor_rustfft.rs:29-46) plus one slot of pinned staging, device buffers and counters --
// the machinery of a caf_stream slot, launched directly instead of through a graph.  A context keeps the
// four most recently used.
struct HostSlot {
    caf_plan *plan = nullptr;
    std::vector<double> freqs;
    unsigned long long stamp = 0;  // LRU clock
    bool one_launch = false;       // n = 4096: the whole surface is ONE launch (k_seq_surface), completion is a polled word
    char *h_base = nullptr, *m_base = nullptr;  // one pinned allocation: needle | haystack | peak | row_idx | row_val | status | seq
    size_t o_hay = 0, o_peak = 0, o_ridx = 0, o_rval = 0, o_status = 0, o_seq = 0;
    void *d_needle = nullptr, *d_ridx = nullptr, *d_rval = nullptr, *d_peak = nullptr, *d_spec = nullptr, *d_slab = nullptr;
    unsigned *d_sync = nullptr;
    unsigned long long launches = 0;
};

struct HostRange {  // caller memory this context may write in place (caf_host_alloc / caf_host_register)
    size_t bytes = 0;
    char *dev = nullptr;
    bool owned = false;     // caf_host_alloc: freed with the context
    bool borrowed = false;  // memory of a caf_multi_surface (registered in every worker's context, owned by that object)
};

struct caf_ctx {
    int device = 0;
    int cu_count = 0;
    std::string name;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // row-independent fused tables per dtype
    void *tw4096[2] = {nullptr, nullptr};
    void *th[2] = {nullptr, nullptr};
    // generic FFT twiddles per (L, dtype)
    std::map<std::pair<size_t, int>, void *> tw_cache;
    // small-path tables per (L, dtype): e^{2 pi i m / L}, m < L
    std::map<std::pair<size_t, int>, void *> small_tabs;
    // chain-path tables per (LOGM, R, dtype): {twM, th}
    std::map<std::tuple<int, int, int>, std::pair<void *, void *>> chain_tabs;
    // host-pointer entry points: cached plans + their staging slots (LRU), shared work buffers
    std::vector<HostSlot *> host_slots;
    unsigned long long host_clock = 0;
    DevBuf io_surface, io_a, io_b;
    PinBuf pin_a, pin_b;
    std::map<char *, HostRange> host_ranges;
    std::vector<caf_plan *> plans;  // every live plan of this context (destroyed with it)
    // Streams of caf_stream slots are pooled per context and reused by later caf_stream objects: how
    // the runtime spreads streams over its few hardware queues depends on creation order, and a slot
    // stream that lands on a queue another slot uses serialises the two slots.
    std::vector<hipStream_t> slot_pool;
    std::vector<bool> slot_busy;
    std::map<std::pair<hipStream_t, hipStream_t>, bool> overlap;  // streams_overlap() results
    MeasureCtx *mz = nullptr;  // measurement build only (measure/dispatch.inc); always NULL in the product
};

struct caf_plan {
    caf_ctx *ctx = nullptr;
    size_t n = 0, L = 0;
    int dtype = CAF_C128;
    uint32_t fs = 0;
    size_t nfreq_total = 0, row_begin = 0, rows = 0;
    bool fused = false;
    bool small = false;         // n <= 512: lane-group rows (kernels_small.hpp)
    void *s_twL = nullptr;      //   ... its W_L table (borrowed from the ctx cache)
    bool chain = false;         // LDS-resident chain path (kernels_chain.hpp): R chains of 2^logm points
    int clogm = 0, cR = 0;
    void *c_twM = nullptr, *c_th = nullptr;  // borrowed from the ctx cache
    DevBuf slab;                             // R = 4: per-workgroup scratch of the last radix-4 stage
    void *slab_override = nullptr;           // streaming slots bring their own
    PeakStageOut stage_out = {nullptr, nullptr, nullptr};  // streaming capture: k_peak also writes the pinned result buffers
    const void *stage_in_src = nullptr;  // streaming capture, fused path: the spectrum kernel also stages the needles in
    void *stage_in_dst = nullptr;
    size_t stage_in_bytes = 0;
    double *d_freqs = nullptr;  // this shard's slice
    double *d_ph = nullptr;
    // fused
    void *d_phasor = nullptr;
    DevBuf spec;
    void *spec_override = nullptr;  // streaming slots bring their own spectrum buffer (they run concurrently)
    // generic
    void *d_tw = nullptr;  // borrowed from ctx cache
    DevBuf wx, wy, hx, hy;
    int live_streams = 0;               // caf_stream objects whose graphs hold this plan's buffers
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    MeasurePlan *mz = nullptr;  // measurement build only (measure/dispatch.inc); always NULL in the product
};

// forward declarations the measurement hooks use
template <typename T>
static int build_fused_tables(caf_ctx *c, int dt);
static int timing_mark(caf_plan *p);

// ------------------------------------------------- measurement-build hooks --
// -DCAF_MEASURE (libcaf_hip_measure.so, used by tools/ and the variant tests only) compiles the rejected kernel
// variants, the ablation instantiations (WRONG results, timing only) and the CAF_* environment switches that select
// them: all of that lives in measure/dispatch.inc, behind the hooks below.  The product library has the no-op hooks,
// contains none of those kernels and reads no environment variable.  This is the only conditional in this file.
#ifdef CAF_MEASURE
#include "measure/dispatch.inc"
#else
#define CHAIN_CASES_MEASURE_F32(STMT)
static int measure_ctx_init(caf_ctx *) { return CAF_OK; }
static void measure_ctx_free(caf_ctx *) {}
static size_t measure_chain_mmax(int, size_t m_max) { return m_max; }
static int measure_plan_select(caf_plan *) { return CAF_OK; }
static void measure_plan_free(caf_plan *) {}
static const char *measure_plan_path(const caf_plan *) { return nullptr; }
static const char *measure_kernel_name(const caf_plan *) { return nullptr; }
static bool measure_keeps_own_kernels(const caf_plan *) { return false; }
static bool measure_own_stage_in(const caf_plan *) { return false; }
template <typename T> static bool measure_plan_tables(caf_plan *, int *) { return false; }
template <typename T> static void measure_fused_args(const caf_plan *, FusedArgs<T> &) {}
template <typename T> static bool measure_fused_prepare(caf_plan *, FusedArgs<T> &, size_t) { return false; }
static void measure_fused_tuning(bool *, size_t *) {}
template <typename T> static bool measure_fused_rows(caf_plan *, FusedArgs<T> &, unsigned, size_t, int *) { return false; }
template <typename T> static bool measure_chain_rows(caf_plan *, ChainArgs<T> &, const cpx<T> *, unsigned, int, int *) { return false; }
static bool measure_surface_dev(caf_plan *, const void *, const void *, size_t, void *, uint64_t *, void *, int *) { return false; }
#endif

// ------------------------------------------------------------ small helpers --
extern "C" int caf_abi_version(void) { return CAF_ABI_VERSION; }
extern "C" const char *caf_last_error_string(void) { return g_err; }

extern "C" int caf_device_count(void)
{
    CAF_GUARD_BEGIN
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
    CAF_GUARD_END
}

template <typename T>
static int build_fused_tables(caf_ctx *c, int dt)
{
    if (c->tw4096[dt]) return CAF_OK;
    HIPCHK(dev_alloc(&c->tw4096[dt], 4096 * sizeof(cpx<T>)));
    HIPCHK(dev_alloc(&c->th[dt], 512 * sizeof(cpx<T>)));
    k_fused_tables<T><<<16, 256, 0, c->stream>>>((cpx<T> *)c->tw4096[dt], (cpx<T> *)c->th[dt]);
    KCHK();
    return CAF_OK;
}

template <typename T>
static int get_generic_tw(caf_ctx *c, size_t L, int dt, void **out)
{
    auto key = std::make_pair(L, dt);
    auto it = c->tw_cache.find(key);
    if (it != c->tw_cache.end()) { *out = it->second; return CAF_OK; }
    const size_t half = L / 2 ? L / 2 : 1;
    void *p = nullptr;
    HIPCHK(dev_alloc(&p, half * sizeof(cpx<T>)));
    k_twiddle<T><<<(unsigned)((half + 255) / 256), 256, 0, c->stream>>>((cpx<T> *)p, half, L);
    KCHK();
    c->tw_cache[key] = p;
    *out = p;
    return CAF_OK;
}

// Stockham passes over HBM, ping-pong x<->y: radix 16 while at least 16 points remain, then one radix-8 / 4 / 2 pass
// (log16(L) passes instead of the log2(L) radix-2 stages of rounds 1-2); returns the buffer holding the result.
template <typename T, int R>
static void fft_pass(caf_ctx *c, const cpx<T> *x, cpx<T> *y, const cpx<T> *tw, size_t L, size_t nrows, size_t n_cur, int inverse)
{
    const size_t nb = L / R;
    for (size_t r0 = 0; r0 < nrows; r0 += 65535) {
        const size_t nr = nrows - r0 < 65535 ? nrows - r0 : 65535;
        dim3 grid((unsigned)((nb + 255) / 256), (unsigned)nr);
        k_fft_pass<T, R><<<grid, 256, 0, c->stream>>>(x + r0 * L, y + r0 * L, tw, L, n_cur, inverse);
    }
}
template <typename T>
static int run_fft(caf_ctx *c, cpx<T> *x, cpx<T> *y, const cpx<T> *tw, size_t L, size_t nrows,
                   int inverse, cpx<T> **res)
{
    for (size_t n_cur = L; n_cur >= 2;) {
        const size_t R = n_cur >= 16 ? 16 : n_cur;
        switch (R) {
        case 16: fft_pass<T, 16>(c, x, y, tw, L, nrows, n_cur, inverse); break;
        case 8: fft_pass<T, 8>(c, x, y, tw, L, nrows, n_cur, inverse); break;
        case 4: fft_pass<T, 4>(c, x, y, tw, L, nrows, n_cur, inverse); break;
        default: fft_pass<T, 2>(c, x, y, tw, L, nrows, n_cur, inverse); break;
        }
        KCHK();
        std::swap(x, y);
        n_cur /= R;
    }
    *res = x;
    return CAF_OK;
}

// ------------------------------------------------------------------ context --
extern "C" int caf_ctx_create(int device_id, caf_ctx **out)
{
    CAF_GUARD_BEGIN
    if (!out) return fail(CAF_ERR_BAD_ARG, "caf_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(CAF_ERR_NO_DEVICE, "no HIP device visible (%s); this engine has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return fail(CAF_ERR_NO_DEVICE, "device id %d out of range [0,%d)", device_id, ndev);
    HIPCHK(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    // the library holds gfx950 code objects only ("gfx950:sramecc+:xnack-" is what the runtime reports for an MI355X)
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CAF_ERR_NO_DEVICE, "device %d is %s, not gfx950 (MI355X): this library carries no code for it and has no CPU fallback",
                    device_id, prop.gcnArchName);
    caf_ctx *c = new (std::nothrow) caf_ctx;
    if (!c) return fail(CAF_ERR_NOMEM, "out of host memory");
    c->device = device_id;
    c->cu_count = prop.multiProcessorCount;
    c->name = prop.gcnArchName;
    if (int mrc = measure_ctx_init(c)) { delete c; return mrc; }
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { measure_ctx_free(c); delete c; return fail(CAF_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->stream = c->own_stream;
    *out = c;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_plan_destroy(caf_plan *p);
static void host_slot_free(HostSlot *s);

extern "C" int caf_ctx_destroy(caf_ctx *c)
{
    CAF_GUARD_BEGIN
    if (!c) return CAF_OK;
    for (caf_plan *p : c->plans)
        if (p->live_streams)
            return fail(CAF_ERR_STATE, "caf_ctx_destroy: a plan still has %d live caf_stream(s); destroy them first",
                        p->live_streams);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    while (!c->host_slots.empty()) { host_slot_free(c->host_slots.back()); c->host_slots.pop_back(); }
    while (!c->plans.empty()) caf_plan_destroy(c->plans.back());  // user plans too: their tables live in this context
    for (int d = 0; d < 2; ++d) {
        if (c->tw4096[d]) (void)dev_free(c->tw4096[d]);
        if (c->th[d]) (void)dev_free(c->th[d]);
    }
    measure_ctx_free(c);
    for (auto st_ : c->slot_pool)
        if (st_ != c->own_stream) (void)hipStreamDestroy(st_);
    for (auto &kv : c->tw_cache) (void)dev_free(kv.second);
    for (auto &kv : c->small_tabs) (void)dev_free(kv.second);
    for (auto &kv : c->chain_tabs) { (void)dev_free(kv.second.first); (void)dev_free(kv.second.second); }
    c->io_surface.release(); c->io_a.release(); c->io_b.release();
    c->pin_a.release(); c->pin_b.release();
    for (auto &kv : c->host_ranges) {
        if (kv.second.borrowed) continue;
        if (kv.second.owned) (void)pin_free(kv.first);
        else (void)hipHostUnregister(kv.first);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_ctx_set_stream(caf_ctx *c, void *hip_stream)
{
    CAF_GUARD_BEGIN
    if (!c) return fail(CAF_ERR_BAD_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)hip_stream;  // NULL == the null stream
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_ctx_reset_stream(caf_ctx *c)
{
    CAF_GUARD_BEGIN
    if (!c) return fail(CAF_ERR_BAD_ARG, "ctx is NULL");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->stream = c->own_stream;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_ctx_synchronize(caf_ctx *c)
{
    CAF_GUARD_BEGIN
    if (!c) return fail(CAF_ERR_BAD_ARG, "ctx is NULL");
    HIPCHK(hipStreamSynchronize(c->stream));
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_ctx_device_info(caf_ctx *c, int *cu_count, char *name_buf, size_t name_cap)
{
    CAF_GUARD_BEGIN
    if (!c) return fail(CAF_ERR_BAD_ARG, "ctx is NULL");
    if (cu_count) *cu_count = c->cu_count;
    if (name_buf && name_cap) {
        strncpy(name_buf, c->name.c_str(), name_cap - 1);
        name_buf[name_cap - 1] = 0;
    }
    return CAF_OK;
    CAF_GUARD_END
}

// --------------------------------------------------------- apply_freq_shift --
template <typename T>
static int apply_shift_impl(caf_ctx *c, const T *in, size_t n, double f, uint32_t fs, T *out)
{
    if (!c || !out || (!in && n)) return fail(CAF_ERR_BAD_ARG, "apply_freq_shift: NULL argument");
    if (fs == 0) return fail(CAF_ERR_BAD_ARG, "apply_freq_shift: fs == 0");
    if (n == 0) return CAF_OK;  // empty slice in, empty Vec out (mod.rs:50)
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = n * sizeof(cpx<T>);
    int rc;
    if ((rc = c->pin_a.ensure(bytes))) return rc;
    if ((rc = c->pin_b.ensure(bytes))) return rc;
    // same left-to-right f64 evaluation as mod.rs:54-56 (host IEEE arithmetic)
    const double dt = 1.0 / (double)fs;
    volatile double two_pi_f = (2.0 * 3.14159265358979323846264338327950288) * f;
    const double ph = two_pi_f * dt;
    // the kernel reads the pinned copy of `in` and writes the pinned result in place (one pass over PCIe each
    // way, no copy-engine hop and no pageable hipMemcpyAsync: 64 KiB each way at n = 4096)
    memcpy(c->pin_a.h, in, bytes);
    k_apply_shift<T><<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>((const cpx<T> *)c->pin_a.m, n, ph,
                                                                         (cpx<T> *)c->pin_b.m);
    KCHK();
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(out, c->pin_b.h, bytes);
    return CAF_OK;
}

extern "C" int caf_apply_freq_shift_c128(caf_ctx *c, const double *in, size_t n, double f, uint32_t fs,
                                         double *out)
{
    CAF_GUARD_BEGIN
    return apply_shift_impl<double>(c, in, n, f, fs, out);
    CAF_GUARD_END
}
extern "C" int caf_apply_freq_shift_c64(caf_ctx *c, const float *in, size_t n, double f, uint32_t fs,
                                        float *out)
{
    CAF_GUARD_BEGIN
    return apply_shift_impl<float>(c, in, n, f, fs, out);
    CAF_GUARD_END
}

// -------------------------------------------------------------------- xcor --
// full-length table e^{2 pi i m / n}, m < n, cached per (n, dtype) in the context (shared with the small-row plans)
template <typename T>
static int get_full_tw(caf_ctx *c, size_t n, int dt, void **out)
{
    auto key = std::make_pair(n, dt);
    auto it = c->small_tabs.find(key);
    if (it == c->small_tabs.end()) {
        void *tw = nullptr;
        HIPCHK(dev_alloc(&tw, n * sizeof(cpx<T>)));
        k_twiddle<T><<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>((cpx<T> *)tw, n, n);
        KCHK();
        it = c->small_tabs.emplace(key, tw).first;
    }
    *out = it->second;
    return CAF_OK;
}

// one-launch forms (kernels_xcor.hpp); returns false if n has none
template <typename T>
static bool xcor_one_launch(caf_ctx *c, const cpx<T> *a, const cpx<T> *b, const cpx<T> *tw, size_t n, cpx<T> *out)
{
    int lg = 0;
    while (((size_t)1 << lg) < n) ++lg;
#define XS(LG) case LG: k_xcor_small<T, LG><<<1, 64, 0, c->stream>>>(a, b, tw, out); return true;
#define XC(LG) case LG: k_xcor_chain<T, LG><<<1, ChainGeo<LG>::W, 0, c->stream>>>(a, b, tw, out); return true;
    switch (lg) {
        XS(1) XS(2) XS(3) XS(4) XS(5) XS(6) XS(7) XS(8) XS(9) XS(10)
        XC(11) XC(12) XC(13)
    case 14:
        if constexpr (sizeof(T) == 4) { k_xcor_chain<T, 14><<<1, ChainGeo<14>::W, 0, c->stream>>>(a, b, tw, out); return true; }
        return false;  // complex128: a 16384-point chain does not fit in LDS
    default: return false;
    }
#undef XS
#undef XC
}

template <typename T>
static int xcor_impl(caf_ctx *c, const T *a, const T *b, size_t n, T *out, int dt)
{
    if (!c || !a || !b || !out) return fail(CAF_ERR_BAD_ARG, "xcor: NULL argument");
    if (!is_pow2(n)) return fail(CAF_ERR_LENGTH, "xcor: n=%zu is not a power of two", n);
    HIPCHK(hipSetDevice(c->device));
    const size_t bytes = n * sizeof(cpx<T>);
    int rc;
    if ((rc = c->pin_a.ensure(2 * bytes))) return rc;
    if ((rc = c->pin_b.ensure(bytes))) return rc;
    memcpy(c->pin_a.h, a, bytes);                  // row 0 = a
    memcpy((char *)c->pin_a.h + bytes, b, bytes);  // row 1 = b
    if (n == 1) {  // out[0] = a[0] conj(b[0])   (a 1-point transform is the identity)
        const T ar = a[0], ai = a[1], br = b[0], bi = b[1];
        out[0] = ar * br + ai * bi;
        out[1] = ai * br - ar * bi;
        return CAF_OK;
    }
    if (n <= 16384 && !(n == 16384 && dt == CAF_C128)) {
        // ONE launch: the kernel reads the pinned copies of a and b and writes the pinned result in place
        void *twf = nullptr;
        if ((rc = get_full_tw<T>(c, n, dt, &twf))) return rc;
        if (xcor_one_launch<T>(c, (const cpx<T> *)c->pin_a.m, (const cpx<T> *)c->pin_a.m + n, (const cpx<T> *)twf, n,
                               (cpx<T> *)c->pin_b.m)) {
            KCHK();
            HIPCHK(hipStreamSynchronize(c->stream));
            memcpy(out, c->pin_b.h, bytes);
            return CAF_OK;
        }
    }
    if ((rc = c->io_a.ensure(2 * bytes))) return rc;
    if ((rc = c->io_b.ensure(2 * bytes))) return rc;
    void *tw = nullptr;
    if ((rc = get_generic_tw<T>(c, n, dt, &tw))) return rc;
    cpx<T> *x = (cpx<T> *)c->io_a.p, *y = (cpx<T> *)c->io_b.p;
    const size_t in16 = (2 * bytes / 16 + 255) / 256;
    const unsigned cgrid = (unsigned)(in16 < 1 ? 1 : in16 > 1024 ? 1024 : in16);
    k_stage_copy<<<cgrid, 256, 0, c->stream>>>(CopyJobs{{c->pin_a.m, nullptr, nullptr}, {x, nullptr, nullptr}, {2 * bytes, 0, 0}});
    KCHK();
    cpx<T> *spec = nullptr;
    if ((rc = run_fft<T>(c, x, y, (const cpx<T> *)tw, n, 2, 0, &spec))) return rc;  // xcor_rustfft.rs:58-61
    cpx<T> *other = spec == x ? y : x;
    k_mul_conj<T><<<dim3((unsigned)((n + 255) / 256), 1), 256, 0, c->stream>>>(spec, spec + n, n, 1);  // :64-73
    KCHK();
    cpx<T> *res = nullptr;
    if ((rc = run_fft<T>(c, spec + n, other + n, (const cpx<T> *)tw, n, 1, 1, &res))) return rc;  // :76
    k_stage_copy<<<cgrid, 256, 0, c->stream>>>(CopyJobs{{res, nullptr, nullptr}, {c->pin_b.m, nullptr, nullptr}, {bytes, 0, 0}});
    KCHK();
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(out, c->pin_b.h, bytes);
    return CAF_OK;
}

extern "C" int caf_xcor_c128(caf_ctx *c, const double *a, const double *b, size_t n, double *out)
{
    CAF_GUARD_BEGIN
    return xcor_impl<double>(c, a, b, n, out, CAF_C128);
    CAF_GUARD_END
}
extern "C" int caf_xcor_c64(caf_ctx *c, const float *a, const float *b, size_t n, float *out)
{
    CAF_GUARD_BEGIN
    return xcor_impl<float>(c, a, b, n, out, CAF_C64);
    CAF_GUARD_END
}

template <typename T>
static int get_full_tw(caf_ctx *c, size_t n, int dt, void **out);

// ----------------------------------------------------------- chain path set-up --
// Which padded lengths the LDS-resident chain kernels (kernels_chain.hpp) cover: L = 2n = R * M
// with one chain of M points (plus its padding and twiddle tables) inside 160 KiB of LDS:
//   complex64:  M <= 16384  -> n = 1024 ... 16384 (R = 2), 32768 (R = 4, BASELINE configs[3]), 65536 (R = 8), 131072 (R = 16)
//   complex128: M <=  8192  -> n = 1024 ... 8192 (R = 2), 16384 (R = 4), 32768 (R = 8), 65536 (R = 16)
// n = 4096 keeps its tuned kernels (kernels_seq4096.hpp / kernels_duo4096.hpp).
static bool chain_config(size_t n, int dtype, int *logm, int *R)
{
    if (n < 1024 || n == (size_t)F_N || !is_pow2(n)) return false;
    size_t m_max = dtype == CAF_C64 ? 16384 : 8192;
    m_max = measure_chain_mmax(dtype, m_max);
    size_t M;
    if (n <= m_max) { M = n; *R = 2; }
    else if (n / 2 <= m_max) { M = n / 2; *R = 4; }
    else if (n / 4 <= m_max) { M = n / 4; *R = 8; }
    else if (n / 8 <= m_max) { M = n / 8; *R = 16; }
    else return false;
    int l = 0;
    while (((size_t)1 << l) < M) ++l;
    *logm = l;
    return true;
}

// CHAIN_DISPATCH(T, logm, R, STMT): run STMT with constexpr LOGM_ / R_ for the instantiated combinations
#define CHAIN_CASE(LG, RR, STMT)                                                                  \
    if (logm_ == LG && R_rt == RR) {                                                             \
        constexpr int LOGM_ = LG;                                                                \
        constexpr int R_ = RR;                                                                   \
        constexpr int NB_ = chain_nb_v(LG, sizeof(cpx<T>));                                      \
        (void)NB_;                                                                               \
        STMT;                                                                                    \
    } else
#define CHAIN_DISPATCH(T, logm, R, STMT)                                                               \
    do {                                                                                               \
        const int logm_ = (logm), R_rt = (R);                                                          \
        if constexpr (sizeof(T) == 4) {                                                                \
            CHAIN_CASE(10, 2, STMT) CHAIN_CASE(11, 2, STMT) CHAIN_CASE(13, 2, STMT) CHAIN_CASE(14, 2, STMT) \
            CHAIN_CASE(14, 4, STMT) CHAIN_CASE(14, 8, STMT) CHAIN_CASE(14, 16, STMT) CHAIN_CASES_MEASURE_F32(STMT) \
            return fail(CAF_ERR_STATE, "chain path: no kernel for M=2^%d R=%d", logm_, R_rt);           \
        } else {                                                                                       \
            CHAIN_CASE(10, 2, STMT) CHAIN_CASE(11, 2, STMT) CHAIN_CASE(13, 2, STMT) CHAIN_CASE(13, 4, STMT) \
            CHAIN_CASE(13, 8, STMT) CHAIN_CASE(13, 16, STMT)                                           \
            return fail(CAF_ERR_STATE, "chain path: no kernel for M=2^%d R=%d", logm_, R_rt);           \
        }                                                                                              \
    } while (0)

template <typename T>
static int build_chain_tables(caf_plan *p)
{
    caf_ctx *c = p->ctx;
    const int M = 1 << p->clogm, R = p->cR, W = M / 16;
    auto key = std::make_tuple(p->clogm, R, p->dtype);
    auto it = c->chain_tabs.find(key);
    if (it == c->chain_tabs.end()) {
        void *twM = nullptr, *th = nullptr;
        HIPCHK(dev_alloc(&twM, (size_t)M * sizeof(cpx<T>)));
        HIPCHK(dev_alloc(&th, (size_t)(R - 1) * W * sizeof(cpx<T>)));
        k_chain_tables<T><<<(unsigned)((M + 255) / 256), 256, 0, c->stream>>>((cpx<T> *)twM, (cpx<T> *)th, M, R);
        KCHK();
        it = c->chain_tabs.emplace(key, std::make_pair(twM, th)).first;
    }
    p->c_twM = it->second.first;
    p->c_th = it->second.second;
    const size_t nr = p->rows + 1;  // +1: the f = 0 row for the haystack transform
    const size_t PH = (size_t)chain_ph_v(R);
    HIPCHK(dev_alloc(&p->d_phasor, nr * PH * sizeof(cpx<T>)));
    k_chain_phasors<T><<<(unsigned)((nr * PH + 255) / 256), 256, 0, c->stream>>>(p->d_ph, (int)p->rows, M, R,
                                                                                 (cpx<T> *)p->d_phasor);
    KCHK();
    return CAF_OK;
}

// -------------------------------------------------------------------- plan --
template <typename T>
static int plan_build_tables(caf_plan *p)
{
    caf_ctx *c = p->ctx;
    const int dt = p->dtype;
    int rc;
    if (measure_plan_tables<T>(p, &rc)) return rc;  // (measurement build: tables of a selected variant)
    if (p->chain) return build_chain_tables<T>(p);
    if (p->small) {
        if ((rc = get_full_tw<T>(c, p->L, dt, &p->s_twL))) return rc;
        // k_small_rows (L >= 16): w^tl and w^TPR of every row, [rows][TPR + 1] complex f64.  Beyond 256 MiB the kernel
        // runs the two sincos itself (same function, same arguments: same bits).
        const size_t tpr = p->L / 16, entries = p->rows * (tpr + 1);
        if (p->L >= 16 && p->rows && entries * sizeof(cpx<double>) <= ((size_t)256 << 20)) {
            HIPCHK(dev_alloc(&p->d_phasor, entries * sizeof(cpx<double>)));
            k_small_phasors<<<(unsigned)((entries + 255) / 256), 256, 0, c->stream>>>(p->d_ph, (int)p->rows, (int)tpr,
                                                                                    (cpx<double> *)p->d_phasor);
            KCHK();
        }
        return CAF_OK;
    }
    if (p->fused) {
        if ((rc = build_fused_tables<T>(c, dt))) return rc;
        const size_t nr = p->rows + 1;  // +1: the f = 0 row for the haystack transform
        HIPCHK(dev_alloc(&p->d_phasor, nr * 64 * sizeof(cpx<T>)));
        const size_t threads = nr * 64;
        k_fused_phasors<T><<<(unsigned)((threads + 255) / 256), 256, 0, c->stream>>>(p->d_ph, (int)p->rows,
                                                                                    (cpx<T> *)p->d_phasor);
        KCHK();
    } else {
        if ((rc = get_generic_tw<T>(c, p->L, dt, &p->d_tw))) return rc;
    }
    return CAF_OK;
}

extern "C" int caf_plan_create(caf_ctx *c, size_t n, const double *freqs_hz, size_t nfreq, uint32_t fs,
                               int dtype, size_t row_begin, size_t row_end, caf_plan **out)
{
    CAF_GUARD_BEGIN
    if (!c || !out) return fail(CAF_ERR_BAD_ARG, "plan_create: NULL argument");
    *out = nullptr;
    if (!freqs_hz && nfreq) return fail(CAF_ERR_BAD_ARG, "plan_create: freqs_hz is NULL");
    if (dtype != CAF_C128 && dtype != CAF_C64) return fail(CAF_ERR_BAD_ARG, "plan_create: bad dtype %d", dtype);
    if (!is_pow2(n)) return fail(CAF_ERR_LENGTH, "caf_surface: n=%zu is not a power of two >= 1", n);
    if (fs == 0) return fail(CAF_ERR_BAD_ARG, "plan_create: fs == 0");
    if (row_begin > row_end || row_end > nfreq)
        return fail(CAF_ERR_BAD_ARG, "plan_create: bad shard [%zu,%zu) of %zu", row_begin, row_end, nfreq);
    if (row_end - row_begin > 0x7fffffffu / 2) return fail(CAF_ERR_BAD_ARG, "plan_create: too many rows");
    HIPCHK(hipSetDevice(c->device));
    caf_plan *p = new (std::nothrow) caf_plan;
    if (!p) return fail(CAF_ERR_NOMEM, "out of host memory");
    p->ctx = c;
    p->n = n;
    p->L = 2 * n;  // mod.rs:130-131
    p->dtype = dtype;
    p->fs = fs;
    p->nfreq_total = nfreq;
    p->row_begin = row_begin;
    p->rows = row_end - row_begin;
    p->fused = (n == (size_t)F_N);
    p->small = n <= 512;                                         // lane-group rows (kernels_small.hpp)
    p->chain = chain_config(n, dtype, &p->clogm, &p->cR);        // LDS-resident chains (kernels_chain.hpp)
    int rc = CAF_OK;
    auto bail = [&](int code) { caf_plan_destroy(p); return code; };
    if ((rc = measure_plan_select(p))) return bail(rc);  // (measurement build: the environment may pick a variant)
    if (p->rows) {
        hipError_t e;
        if ((e = dev_alloc((void **)&p->d_freqs, p->rows * sizeof(double))) != hipSuccess ||
            (e = dev_alloc((void **)&p->d_ph, p->rows * sizeof(double))) != hipSuccess)
            return bail(fail(CAF_ERR_NOMEM, "hipMalloc: %s", hipGetErrorString(e)));
        // pageable H2D on a stream is synchronous w.r.t. the host buffer: safe to borrow
        if ((e = hipMemcpyAsync(p->d_freqs, freqs_hz + row_begin, p->rows * sizeof(double),
                                hipMemcpyHostToDevice, c->stream)) != hipSuccess)
            return bail(fail(CAF_ERR_HIP, "hipMemcpyAsync(freqs): %s", hipGetErrorString(e)));
        k_phase<<<(unsigned)((p->rows + 255) / 256), 256, 0, c->stream>>>(p->d_freqs, (int)p->rows, fs, p->d_ph);
        if ((e = hipGetLastError()) != hipSuccess)
            return bail(fail(CAF_ERR_HIP, "k_phase launch: %s", hipGetErrorString(e)));
    }
    rc = dtype == CAF_C128 ? plan_build_tables<double>(p) : plan_build_tables<float>(p);
    if (rc) return bail(rc);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return bail(fail(CAF_ERR_HIP, "plan tables: %s", hipGetErrorString(e)));
    c->plans.push_back(p);
    *out = p;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_plan_destroy(caf_plan *p)
{
    CAF_GUARD_BEGIN
    if (!p) return CAF_OK;
    // the captured graphs of a caf_stream hold raw pointers into this plan's tables and workspaces
    if (p->live_streams)
        return fail(CAF_ERR_STATE, "caf_plan_destroy: %d caf_stream(s) of this plan are still alive", p->live_streams);
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    for (auto it = p->ctx->plans.begin(); it != p->ctx->plans.end(); ++it)
        if (*it == p) { p->ctx->plans.erase(it); break; }
    if (p->d_freqs) (void)dev_free(p->d_freqs);
    if (p->d_ph) (void)dev_free(p->d_ph);
    if (p->d_phasor) (void)dev_free(p->d_phasor);
    p->spec.release(); p->wx.release(); p->wy.release(); p->hx.release(); p->hy.release();
    p->slab.release();
    measure_plan_free(p);
    for (auto ev : p->ev) (void)hipEventDestroy(ev);
    delete p;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" const char *caf_plan_path(const caf_plan *p)
{
    if (!p) return "";
    if (const char *m = measure_plan_path(p)) return m;
    return p->fused ? "fused4096" : p->chain ? "chain" : p->small ? "small" : "generic";
}
extern "C" size_t caf_plan_rows(const caf_plan *p) { return p ? p->rows : 0; }
extern "C" const char *caf_plan_kernel_name(const caf_plan *p)
{
    if (!p) return "";
    const bool f64 = p->dtype == CAF_C128;
    if (const char *m = measure_kernel_name(p)) return m;
    if (p->chain) {
        static thread_local char name[64];
        snprintf(name, sizeof name, "caf::k_chain_rows<%s, %d, %d, %d, 0>", f64 ? "double" : "float", p->clogm, p->cR,
                 chain_nb_v(p->clogm, f64 ? 16 : 8));
        return name;
    }
    if (p->small) {
        static thread_local char name[64];
        int lg = 0;
        while (((size_t)1 << lg) < p->L) ++lg;
        if (lg >= 4) snprintf(name, sizeof name, "caf::k_small_rows<%s, %d>", f64 ? "double" : "float", lg);
        else snprintf(name, sizeof name, "caf::k_small<%s, %d, false>", f64 ? "double" : "float", lg);
        return name;
    }
    if (!p->fused) return f64 ? "caf::k_fft_pass<double, 16>" : "caf::k_fft_pass<float, 16>";
    return f64 ? "caf::k_seq_rows<double, 15, caf::SeqIo<double> >" : "caf::k_duo_rows<float, caf::DuoIo<float> >";
}

static int timing_mark(caf_plan *p)
{
    if (!p->timing) return CAF_OK;
    if (p->ev_used == p->ev.size()) {
        hipEvent_t ev;
        HIPCHK(hipEventCreate(&ev));
        p->ev.push_back(ev);
    }
    HIPCHK(hipEventRecord(p->ev[p->ev_used++], p->ctx->stream));
    return CAF_OK;
}

extern "C" int caf_plan_timing_begin(caf_plan *p)
{
    CAF_GUARD_BEGIN
    if (!p) return fail(CAF_ERR_BAD_ARG, "plan is NULL");
    p->timing = true;
    p->ev_used = 0;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_plan_timing_end(caf_plan *p, double *ms_total, uint64_t *launches)
{
    CAF_GUARD_BEGIN
    if (!p) return fail(CAF_ERR_BAD_ARG, "plan is NULL");
    HIPCHK(hipSetDevice(p->ctx->device));
    HIPCHK(hipStreamSynchronize(p->ctx->stream));
    double tot = 0.0;
    for (size_t i = 0; i + 1 < p->ev_used; i += 2) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, p->ev[i], p->ev[i + 1]));
        tot += ms;
    }
    if (ms_total) *ms_total = tot;
    if (launches) *launches = p->ev_used / 2;
    p->timing = false;
    p->ev_used = 0;
    return CAF_OK;
    CAF_GUARD_END
}

// ------------------------------------------------------------ surface (dev) --
template <typename T>
static int surface_dev_fused(caf_plan *p, const void *d_needle, const void *d_hay, size_t batch,
                             void *d_surface, uint64_t *d_ridx, void *d_rval)
{
    caf_ctx *c = p->ctx;
    int rc;
    const size_t spec_bytes = batch * 2 * 16 * 256 * sizeof(cpx<T>);  // + 256 B: the row-ticket counter
    if (!p->spec_override && (rc = p->spec.ensure(spec_bytes + 256))) return rc;
    FusedArgs<T> a;
    a.phasor = (const cpx<T> *)p->d_phasor;
    a.tab.tw4096 = (const cpx<T> *)c->tw4096[p->dtype];
    a.tab.th = (const cpx<T> *)c->th[p->dtype];
    a.spec = (cpx<T> *)(p->spec_override ? p->spec_override : p->spec.p);
    a.work = (unsigned *)((unsigned char *)a.spec + spec_bytes);
    a.rows = (int)p->rows;
    a.surface = nullptr; a.row_idx = nullptr; a.row_val = nullptr;
    a.dbg = nullptr;
    measure_fused_args(p, a);
    a.stage_src = (const uint4 *)p->stage_in_src;
    a.stage_dst = (uint4 *)p->stage_in_dst;
    a.stage_n16 = (unsigned)(p->stage_in_bytes / 16);
    // haystack spectrum, once per surface (the reference recomputes it per row,
    // xcor_rustfft.rs:58-59)
    a.sig = (const cpx<T> *)d_hay;
    a.total = (int)batch;
    if (!measure_fused_prepare<T>(p, a, batch)) {  // one workgroup per (surface, chain): halves the latency of a single-surface call
        const size_t want = 2 * batch, cap2 = 2 * (size_t)c->cu_count;
        a.fft_blocks = (unsigned)(want < cap2 ? want : cap2);
        const size_t copy_want = ((size_t)a.stage_n16 + S_THREADS - 1) / S_THREADS;  // streaming slots only
        const unsigned copy_blocks = (unsigned)(copy_want < 256 ? copy_want : 256);
        k_seq_prepare<T><<<a.fft_blocks + copy_blocks, S_THREADS, 0, c->stream>>>(a, a.phasor);
    }
    KCHK();
    const size_t total = batch * p->rows;
    if (total == 0) return CAF_OK;
    // Dynamic row tickets pay off from ~6 rows per resident workgroup; below that the static
    // stride (no atomic, no LDS round trip per row) is 2-7 % faster (measured at batch 1-16).
    bool static_rows = total <= 4 * (size_t)c->cu_count * 2;
    // resident workgroups per CU: LDS- and VGPR-limited (2 in f64, 3 in f32)
    size_t per_cu = 160 * 1024 / seq_lds_bytes<T>();
    if (per_cu > (size_t)seq_waves_per_simd<T>()) per_cu = seq_waves_per_simd<T>();
    measure_fused_tuning(&static_rows, &per_cu);
    if (static_rows) a.work = nullptr;
    a.sig = (const cpx<T> *)d_needle;
    a.total = (int)total;
    a.surface = (T *)d_surface;
    a.row_idx = d_ridx;
    a.row_val = (T *)d_rval;
    if ((rc = timing_mark(p))) return rc;
    const size_t cap = (size_t)c->cu_count * per_cu;
    const unsigned grid = (unsigned)(total < cap ? total : cap);
    if (measure_fused_rows<T>(p, a, grid, total, &rc)) {  // (measurement build: a selected variant / ablation launched instead)
        if (rc) return rc;
    } else if constexpr (sizeof(T) == 4) {
        k_duo_rows<T><<<grid, S_THREADS, 0, c->stream>>>(a, a.phasor);  // complex64 product kernel: two chains in flight
    } else {
        k_seq_rows<T><<<grid, S_THREADS, 0, c->stream>>>(a, a.phasor);  // complex128 product kernel: sequential chains
    }
    KCHK();
    if ((rc = timing_mark(p))) return rc;
    return CAF_OK;
}

// One surface = ONE launch (kernels_surf4096.hpp): needle staging, haystack spectrum, Doppler rows and
// find_peak as roles of one grid.  Used by single-surface streaming chains (caf_stream_*).
template <typename T>
static int surface_single_launch(caf_plan *p, hipStream_t on, const void *needle_src, void *d_needle, const void *hay,
                                 void *spec, void *d_surface, uint64_t *d_ridx, void *d_rval, caf_peak *d_peak,
                                 const PeakStageOut &host, unsigned *sync, unsigned *status, unsigned long long *h_seq,
                                 bool two_nodes)
{
    caf_ctx *c = p->ctx;
    FusedArgs<T> a{};
    a.phasor = (const cpx<T> *)p->d_phasor;
    a.tab.tw4096 = (const cpx<T> *)c->tw4096[p->dtype];
    a.tab.th = (const cpx<T> *)c->th[p->dtype];
    a.spec = (cpx<T> *)spec;
    a.sig = (const cpx<T> *)d_needle;
    a.rows = (int)p->rows;
    a.total = (int)p->rows;
    a.surface = (T *)d_surface;
    a.row_idx = d_ridx;
    a.row_val = (T *)d_rval;
    a.dbg = nullptr;
    a.work = nullptr;
    a.stage_src = (const uint4 *)needle_src;
    a.stage_dst = (uint4 *)d_needle;
    a.stage_n16 = needle_src ? (unsigned)(F_N * sizeof(cpx<T>) / 16) : 0u;
    a.fft_blocks = 2;
    SurfArgs<T> s{};
    s.hay = (const cpx<T> *)hay;
    s.sync = sync;
    s.status = status;
    if (two_nodes) {
        // node 1: needle staging + haystack spectrum (k_seq_prepare, 16 + 2 workgroups); node 2 below then never waits
        FusedArgs<T> pa = a;
        pa.sig = (const cpx<T> *)hay;
        pa.total = 1;
        pa.rows = (int)p->rows;
        pa.work = nullptr;
        pa.surface = nullptr; pa.row_idx = nullptr; pa.row_val = nullptr;
        const unsigned cb = (a.stage_n16 + S_THREADS - 1) / S_THREADS;
        k_seq_prepare<T><<<2u + cb, S_THREADS, 0, on>>>(pa, pa.phasor);
        KCHK();
        a.stage_src = nullptr; a.stage_dst = nullptr; a.stage_n16 = 0;
    }
    s.copy_blocks = (a.stage_n16 + S_THREADS - 1) / S_THREADS;  // one 16-byte element per thread: all reads in flight at once
    s.prep_blocks = two_nodes ? 0u : 2u;
    s.freqs = p->d_freqs;
    s.row_base = (int64_t)p->row_begin;
    s.peak = d_peak;
    s.h_peak = host.peak;
    s.h_ridx = host.row_idx;
    s.h_rval = (T *)host.row_val;
    s.h_seq = h_seq;
    const unsigned grid = s.copy_blocks + s.prep_blocks + (unsigned)p->rows;
    k_seq_surface<T><<<grid, S_THREADS, 0, on>>>(a, a.phasor, s);
    KCHK();
    return CAF_OK;
}

// LDS-resident chain path (kernels_chain.hpp)
template <typename T>
static int surface_dev_chain(caf_plan *p, const void *d_needle, const void *d_hay, size_t batch, void *d_surface,
                             uint64_t *d_ridx, void *d_rval)
{
    caf_ctx *c = p->ctx;
    const int R = p->cR, M = 1 << p->clogm, W = M / 16;
    const size_t total = batch * p->rows;
    int rc;
    if (!p->spec_override && (rc = p->spec.ensure(batch * (size_t)R * M * sizeof(cpx<T>)))) return rc;
    ChainArgs<T> a;
    a.twM = (const cpx<T> *)p->c_twM;
    a.th = (const cpx<T> *)p->c_th;
    a.spec = (cpx<T> *)(p->spec_override ? p->spec_override : p->spec.p);
    a.rows = (int)p->rows;
    a.surface = nullptr; a.row_idx = nullptr; a.row_val = nullptr; a.slab = nullptr;
    const cpx<T> *phasor = (const cpx<T> *)p->d_phasor;
    const int nb = chain_nb_v(p->clogm, sizeof(cpx<T>));
    const size_t cap = (size_t)c->cu_count * chain_wg_per_cu_v(p->clogm, sizeof(cpx<T>), nb);
    const size_t cap_prep = (size_t)c->cu_count * chain_wg_per_cu_v(p->clogm, sizeof(cpx<T>), 1);
    // haystack spectrum, once per surface (the reference recomputes it per row, xcor_rustfft.rs:58-59)
    a.sig = (const cpx<T> *)d_hay;
    a.total = (int)batch;
    {
        const size_t want = (size_t)R * batch;
        const unsigned grid = (unsigned)(want < cap_prep ? want : cap_prep);
        CHAIN_DISPATCH(T, p->clogm, R, (k_chain_prepare<T, LOGM_, R_><<<grid, W, 0, c->stream>>>(a, phasor)));
    }
    KCHK();
    if (total == 0) return CAF_OK;
    const unsigned grid = (unsigned)(total < cap ? total : cap);
    if (R >= 4) {
        const size_t slab_bytes = (size_t)cap * chain_slab_arrays_v(R) * 16 * W * sizeof(cpx<T>);
        if (!p->slab_override && (rc = p->slab.ensure(slab_bytes))) return rc;
        a.slab = (cpx<T> *)(p->slab_override ? p->slab_override : p->slab.p);
    }
    a.sig = (const cpx<T> *)d_needle;
    a.total = (int)total;
    a.surface = (T *)d_surface;
    a.row_idx = d_ridx;
    a.row_val = (T *)d_rval;
    if ((rc = timing_mark(p))) return rc;
    if (measure_chain_rows<T>(p, a, phasor, grid, W, &rc)) {  // (measurement build: ablations of the configs[3] kernel)
        if (rc) return rc;
        KCHK();
        return timing_mark(p);
    }
    CHAIN_DISPATCH(T, p->clogm, R, (k_chain_rows<T, LOGM_, R_, NB_><<<grid, W / NB_, 0, c->stream>>>(a, phasor)));
    KCHK();
    if ((rc = timing_mark(p))) return rc;
    return CAF_OK;
}

// n <= 512: lane-group rows (kernels_small.hpp), one launch for the spectra, one for the rows
template <typename T, int LOGL>
static int small_launch(caf_plan *p, SmallArgs<T> &a, const void *d_needle, const void *d_hay, size_t batch, size_t total,
                        void *d_surface, uint64_t *d_ridx, void *d_rval)
{
    using G = SmallGeo<LOGL>;
    caf_ctx *c = p->ctx;
    const size_t cap = (size_t)c->cu_count * 8;
    a.sig = (const cpx<T> *)d_hay;
    a.total = (int)batch;
    const size_t g0 = (batch + G::RPW - 1) / G::RPW;
    k_small<T, LOGL, true><<<(unsigned)(g0 < cap ? g0 : cap), G::THREADS, 0, c->stream>>>(a);
    KCHK();
    if (total == 0) return CAF_OK;
    a.sig = (const cpx<T> *)d_needle;
    a.total = (int)total;
    a.surface = (T *)d_surface;
    a.row_idx = d_ridx;
    a.row_val = (T *)d_rval;
    int rc;
    if ((rc = timing_mark(p))) return rc;
    if constexpr (LOGL >= 4) {  // transforms in registers
        using GR = SmallRowGeo<LOGL>;
        const size_t g1 = (total + GR::RPW - 1) / GR::RPW;
        k_small_rows<T, LOGL><<<(unsigned)(g1 < cap ? g1 : cap), GR::THREADS, 0, c->stream>>>(a, (const cpx<double> *)p->d_phasor);
    } else {
        const size_t g1 = (total + G::RPW - 1) / G::RPW;
        k_small<T, LOGL, false><<<(unsigned)(g1 < cap ? g1 : cap), G::THREADS, 0, c->stream>>>(a);
    }
    KCHK();
    return timing_mark(p);
}

template <typename T>
static int surface_dev_small(caf_plan *p, const void *d_needle, const void *d_hay, size_t batch, void *d_surface,
                             uint64_t *d_ridx, void *d_rval)
{
    int rc;
    if (!p->spec_override && (rc = p->spec.ensure(batch * p->L * sizeof(cpx<T>)))) return rc;
    SmallArgs<T> a;
    a.spec = (cpx<T> *)(p->spec_override ? p->spec_override : p->spec.p);
    a.twL = (const cpx<T> *)p->s_twL;
    a.ph = p->d_ph;
    a.rows = (int)p->rows;
    a.surface = nullptr; a.row_idx = nullptr; a.row_val = nullptr;
    const size_t total = batch * p->rows;
    int lg = 0;
    while (((size_t)1 << lg) < p->L) ++lg;
    switch (lg) {
    case 1: return small_launch<T, 1>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    case 2: return small_launch<T, 2>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    case 3: return small_launch<T, 3>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    case 4: return small_launch<T, 4>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    case 5: return small_launch<T, 5>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    case 6: return small_launch<T, 6>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    case 7: return small_launch<T, 7>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    case 8: return small_launch<T, 8>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    case 9: return small_launch<T, 9>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    case 10: return small_launch<T, 10>(p, a, d_needle, d_hay, batch, total, d_surface, d_ridx, d_rval);
    default: return fail(CAF_ERR_STATE, "small path: no kernel for L = %zu", p->L);
    }
}

template <typename T>
static int surface_dev_generic(caf_plan *p, const void *d_needle, const void *d_hay, size_t batch,
                               void *d_surface, uint64_t *d_ridx, void *d_rval)
{
    caf_ctx *c = p->ctx;
    const size_t L = p->L, n = p->n, rows = p->rows, total = batch * rows;
    int rc;
    if ((rc = p->hx.ensure(batch * L * sizeof(cpx<T>)))) return rc;
    if ((rc = p->hy.ensure(batch * L * sizeof(cpx<T>)))) return rc;
    const cpx<T> *tw = (const cpx<T> *)p->d_tw;
    const unsigned gx = (unsigned)((L + 255) / 256);
    // H = FFT(haystack ++ zeros)   (mod.rs:131, xcor_rustfft.rs:58-59 hoisted)
    for (size_t b0 = 0; b0 < batch; b0 += 65535) {
        const size_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
        k_mix_pad<T><<<dim3(gx, (unsigned)nb), 256, 0, c->stream>>>((const cpx<T> *)d_hay + b0 * n, n, L, nullptr, 1,
                                                                   (cpx<T> *)p->hx.p + b0 * L);
    }
    KCHK();
    cpx<T> *H = nullptr;
    if ((rc = run_fft<T>(c, (cpx<T> *)p->hx.p, (cpx<T> *)p->hy.p, tw, L, batch, 0, &H))) return rc;
    if (total == 0) return CAF_OK;
    if ((rc = p->wx.ensure(total * L * sizeof(cpx<T>)))) return rc;
    if ((rc = p->wy.ensure(total * L * sizeof(cpx<T>)))) return rc;
    cpx<T> *x = (cpx<T> *)p->wx.p, *y = (cpx<T> *)p->wy.p;
    if ((rc = timing_mark(p))) return rc;
    // shifted = apply_freq_shift(needle ++ zeros)   (mod.rs:130,138); one batch entry per launch
    for (size_t b = 0; b < batch; ++b)
        for (size_t r0 = 0; r0 < rows; r0 += 65535) {
            const size_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
            k_mix_pad<T><<<dim3(gx, (unsigned)nr), 256, 0, c->stream>>>(
                (const cpx<T> *)d_needle + b * n, n, L, p->d_ph + r0, nr, x + (b * rows + r0) * L);
        }
    KCHK();
    cpx<T> *S = nullptr;
    if ((rc = run_fft<T>(c, x, y, tw, L, total, 0, &S))) return rc;  // xcor_rustfft.rs:60-61
    cpx<T> *other = S == x ? y : x;
    for (size_t b = 0; b < batch; ++b)
        for (size_t r0 = 0; r0 < rows; r0 += 65535) {
            const size_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
            k_mul_conj<T><<<dim3(gx, (unsigned)nr), 256, 0, c->stream>>>(H + b * L, S + (b * rows + r0) * L, L, nr);
        }
    KCHK();
    cpx<T> *res = nullptr;
    if ((rc = run_fft<T>(c, S, other, tw, L, total, 1, &res))) return rc;  // xcor_rustfft.rs:76
    k_mag_argmax<T><<<(unsigned)total, 256, 0, c->stream>>>(res, L, (T *)d_surface, d_ridx, (T *)d_rval);
    KCHK();
    if ((rc = timing_mark(p))) return rc;
    return CAF_OK;
}

extern "C" int caf_surface_dev(caf_plan *p, const void *d_needle, const void *d_hay, size_t batch,
                               void *d_surface, uint64_t *d_ridx, void *d_rval, caf_peak *d_peak)
{
    CAF_GUARD_BEGIN
    if (!p || !d_needle || !d_hay || !d_peak) return fail(CAF_ERR_BAD_ARG, "caf_surface_dev: NULL argument");
    if (p->rows && (!d_ridx || !d_rval)) return fail(CAF_ERR_BAD_ARG, "caf_surface_dev: row outputs are NULL");
    if (batch == 0) return CAF_OK;
    // (the row kernels index rows with int, some after rounding the count up to a workgroup's worth of rows)
    if (batch * (p->rows ? p->rows : 1) > 0x7fffffffu - 65536u) return fail(CAF_ERR_BAD_ARG, "caf_surface_dev: batch too large");
    caf_ctx *c = p->ctx;
    HIPCHK(hipSetDevice(c->device));
    int rc;
    if (measure_surface_dev(p, d_needle, d_hay, batch, d_surface, d_ridx, d_rval, &rc)) {
        // (measurement build: one of the older whole-surface paths ran instead)
    } else if (p->dtype == CAF_C128)
        rc = p->fused ? surface_dev_fused<double>(p, d_needle, d_hay, batch, d_surface, d_ridx, d_rval)
             : p->chain ? surface_dev_chain<double>(p, d_needle, d_hay, batch, d_surface, d_ridx, d_rval)
             : p->small ? surface_dev_small<double>(p, d_needle, d_hay, batch, d_surface, d_ridx, d_rval)
                      : surface_dev_generic<double>(p, d_needle, d_hay, batch, d_surface, d_ridx, d_rval);
    else
        rc = p->fused ? surface_dev_fused<float>(p, d_needle, d_hay, batch, d_surface, d_ridx, d_rval)
             : p->chain ? surface_dev_chain<float>(p, d_needle, d_hay, batch, d_surface, d_ridx, d_rval)
             : p->small ? surface_dev_small<float>(p, d_needle, d_hay, batch, d_surface, d_ridx, d_rval)
                      : surface_dev_generic<float>(p, d_needle, d_hay, batch, d_surface, d_ridx, d_rval);
    if (rc) return rc;
    // find_peak (mod.rs:31-42)
    if (p->dtype == CAF_C128)
        k_peak<double><<<(unsigned)batch, 256, 0, c->stream>>>(p->d_freqs, d_ridx, (const double *)d_rval, (int)p->rows,
                                                               (int64_t)p->row_begin, d_peak, p->stage_out);
    else
        k_peak<float><<<(unsigned)batch, 256, 0, c->stream>>>(p->d_freqs, d_ridx, (const float *)d_rval, (int)p->rows,
                                                              (int64_t)p->row_begin, d_peak, p->stage_out);
    KCHK();
    return CAF_OK;
    CAF_GUARD_END
}

// ----------------------------------------------------------- surface (host) --
// The literal drop-in call: every caller of the reference does `X::caf_surface(..)` then `X::find_peak(..)`
// with host slices (main.rs:25-26, tests/test.rs:25-26, benches/caf_bench.rs:39-40).  Built on the machinery of
// the streaming slots: per cached (n, freq list, fs, dtype) a plan + pinned staging + device buffers; the inputs
// are copied into the pinned buffers by the CPU (2 x 64 KiB: ~3 us) and read from there by the kernels, the row
// peaks and the caf_peak record are written to pinned memory by the kernels themselves.  No pageable
// hipMemcpyAsync, no allocation per call.  n = 4096: the whole surface is ONE direct launch of k_seq_surface
// (needle staging, haystack spectrum, rows, find_peak as roles of one grid) whose completion the host reads
// from a pinned sequence word.  A host surface is written IN PLACE by the row kernel when the caller's buffer
// is memory this context may address (caf_host_alloc / caf_host_register: the stores cross PCIe while the
// other rows compute), else into a context-owned device slab followed by one D2H copy.
static void host_slot_free(HostSlot *s)
{
    if (!s) return;
    if (s->plan) caf_plan_destroy(s->plan);
    if (s->h_base) (void)pin_free(s->h_base);
    for (void *p : {s->d_needle, s->d_ridx, s->d_rval, s->d_peak, s->d_spec, s->d_slab, (void *)s->d_sync})
        if (p) (void)dev_free(p);
    delete s;
}

static constexpr size_t HOST_SLOTS_MAX = 4;

// A plan over rows [row_begin, row_end) of the freq list + the staging slot the host-pointer calls run it through.
// (row_begin, row_end) = (0, nfreq) for caf_surface_*; a proper shard for the workers of caf_multi_surface_*.
static int make_host_slot(caf_ctx *c, size_t n, const double *freqs, size_t nfreq, uint32_t fs, int dtype, size_t row_begin,
                          size_t row_end, HostSlot **out)
{
    std::vector<double> fcopy(freqs, freqs + nfreq);  // (before anything is allocated: bad_alloc leaks nothing)
    HostSlot *s = new (std::nothrow) HostSlot;
    if (!s) return fail(CAF_ERR_NOMEM, "out of host memory");
    auto bail = [&](int code) { host_slot_free(s); return code; };
    int rc = caf_plan_create(c, n, freqs, nfreq, fs, dtype, row_begin, row_end, &s->plan);
    if (rc) return bail(rc);
    caf_plan *p = s->plan;
    s->freqs = std::move(fcopy);
    s->one_launch = p->fused && p->rows > 0;
    if (measure_keeps_own_kernels(p)) s->one_launch = false;
    const size_t esz = elem_size(dtype), rsz = real_size(dtype), in1 = n * esz, rows = p->rows ? p->rows : 1;
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    s->o_hay = up(in1);
    s->o_peak = s->o_hay + up(in1);
    s->o_ridx = s->o_peak + 256;
    s->o_rval = s->o_ridx + up(rows * sizeof(uint64_t));
    s->o_status = s->o_rval + up(rows * rsz);
    s->o_seq = s->o_status + 256;
    const size_t pin_bytes = s->o_seq + 256;
#define HCHK(expr)                                                                                         \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) return bail(fail(CAF_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__))); \
    } while (0)
    HCHK(pinned_alloc((void **)&s->h_base, pin_bytes));
    memset(s->h_base, 0, pin_bytes);
    HCHK(hipHostGetDevicePointer((void **)&s->m_base, s->h_base, 0));
    HCHK(dev_alloc(&s->d_needle, in1 < 16 ? 16 : in1));
    HCHK(dev_alloc(&s->d_ridx, rows * sizeof(uint64_t)));
    HCHK(dev_alloc(&s->d_rval, rows * rsz));
    HCHK(dev_alloc(&s->d_peak, sizeof(caf_peak)));
    const size_t spec1 = p->fused ? (size_t)2 * 16 * 256 * esz : (p->chain || p->small) ? p->L * esz : 0;
    if (spec1) HCHK(dev_alloc(&s->d_spec, spec1 + 256));
    if (p->chain && p->cR >= 4)
        HCHK(dev_alloc(&s->d_slab, (size_t)c->cu_count * chain_wg_per_cu_v(p->clogm, esz, chain_nb_v(p->clogm, esz)) *
                                       chain_slab_arrays_v(p->cR) * 16 * (((size_t)1 << p->clogm) / 16) * esz));
    if (s->one_launch) {
        HCHK(dev_alloc((void **)&s->d_sync, 512));
        HCHK(hipMemsetAsync(s->d_sync, 0, 512, c->stream));
        HCHK(hipStreamSynchronize(c->stream));
    }
#undef HCHK
    *out = s;
    return CAF_OK;
}

static int get_host_slot(caf_ctx *c, size_t n, const double *freqs, size_t nfreq, uint32_t fs, int dtype, HostSlot **out)
{
    for (HostSlot *s : c->host_slots) {
        const caf_plan *p = s->plan;
        if (p->n == n && p->fs == fs && p->dtype == dtype && p->nfreq_total == nfreq && s->freqs.size() == nfreq &&
            (nfreq == 0 || memcmp(s->freqs.data(), freqs, nfreq * sizeof(double)) == 0)) {
            s->stamp = ++c->host_clock;
            *out = s;
            return CAF_OK;
        }
    }
    HostSlot *s = nullptr;
    int rc = make_host_slot(c, n, freqs, nfreq, fs, dtype, 0, nfreq, &s);
    if (rc) return rc;  // (a failed creation leaves the cache as it was)
    c->host_slots.reserve(HOST_SLOTS_MAX + 1);
    if (c->host_slots.size() >= HOST_SLOTS_MAX) {  // evict the least recently used, now that its replacement exists
        size_t lru = 0;
        for (size_t i = 1; i < c->host_slots.size(); ++i)
            if (c->host_slots[i]->stamp < c->host_slots[lru]->stamp) lru = i;
        (void)hipStreamSynchronize(c->stream);
        host_slot_free(c->host_slots[lru]);
        c->host_slots.erase(c->host_slots.begin() + (long)lru);
    }
    s->stamp = ++c->host_clock;
    c->host_slots.push_back(s);
    *out = s;
    return CAF_OK;
}

// device address for [p, p + bytes) if it lies inside memory of caf_host_alloc / caf_host_register, else NULL
static char *host_range_dev(caf_ctx *c, const void *ptr, size_t bytes)
{
    if (c->host_ranges.empty()) return nullptr;
    char *q = (char *)ptr;
    auto it = c->host_ranges.upper_bound(q);
    if (it == c->host_ranges.begin()) return nullptr;
    --it;
    if (q < it->first || q + bytes > it->first + it->second.bytes) return nullptr;
    return it->second.dev + (q - it->first);
}

// copy between caller (host) memory and a device buffer.  hipMemcpyAsync rejects a host range that straddles the edge
// of a registered range ("invalid argument"), so the copy is cut at the edges of this context's registered ranges: every
// piece lies wholly inside one range or wholly in ordinary memory.
static int host_copy(caf_ctx *c, void *host, void *dev, size_t bytes, bool to_host)
{
    char *h = (char *)host, *d = (char *)dev;
    size_t done = 0;
    while (done < bytes) {
        size_t piece = bytes - done;
        char *q = h + done;
        auto it = c->host_ranges.upper_bound(q);  // first range starting beyond q
        if (it != c->host_ranges.end() && (size_t)(it->first - q) < piece) piece = (size_t)(it->first - q);
        if (it != c->host_ranges.begin()) {
            auto in = std::prev(it);
            char *end = in->first + in->second.bytes;
            if (q < end && (size_t)(end - q) < piece) piece = (size_t)(end - q);
        }
        if (to_host) HIPCHK(hipMemcpyAsync(q, d + done, piece, hipMemcpyDeviceToHost, c->stream));
        else HIPCHK(hipMemcpyAsync(d + done, q, piece, hipMemcpyHostToDevice, c->stream));
        done += piece;
    }
    return CAF_OK;
}
static int d2h_copy(caf_ctx *c, void *dst, const void *src, size_t bytes) { return host_copy(c, dst, (void *)src, bytes, true); }

static void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
}

// wait until a single-launch surface has published launch number `want` in its pinned sequence word; if the
// poll runs out of patience, synchronise the stream and look again
static int poll_seq(const unsigned long long *h_seq, size_t count, unsigned long long want, hipStream_t stream, const char *who)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long spins = 0;; ++spins) {
        bool done = true;
        for (size_t j = 0; j < count; ++j) done = done && __atomic_load_n(&h_seq[j], __ATOMIC_ACQUIRE) >= want;
        if (done) return CAF_OK;
        cpu_relax();
        if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
    }
    HIPCHK(hipStreamSynchronize(stream));
    for (size_t j = 0; j < count; ++j)
        if (__atomic_load_n(&h_seq[j], __ATOMIC_ACQUIRE) < want)
            return fail(CAF_ERR_HIP, "%s: the launch finished without publishing its results (sequence word %llu, expected %llu)",
                        who, (unsigned long long)h_seq[j], want);
    return CAF_OK;
}

// One surface (or row shard of one) through a staging slot: `surface`, `row_idx`, `row_val` point at the slot's FIRST row
// (rows = the plan's shard); `peak` is the shard's find_peak record with GLOBAL row positions.  `dev_surface` (device
// memory of this context's GPU, rows x 2n) replaces `surface`: the rows stay on the device.
template <typename T>
static int host_slot_run(caf_ctx *c, HostSlot &s, const T *needle, const T *hay, T *surface, uint64_t *row_idx, T *row_val,
                         caf_peak *peak, T *dev_surface = nullptr)
{
    caf_plan *p = s.plan;
    const size_t n = p->n, nfreq = p->rows;
    int rc;
    const size_t L = 2 * n, in1 = n * sizeof(cpx<T>), surf_bytes = nfreq * L * sizeof(T);
    const bool want_surface = (surface || dev_surface) && nfreq;
    HIPCHK(hipSetDevice(c->device));
    // where the row kernel stores the surface: the caller's buffer itself, or a device slab + one D2H copy
    void *surf_target = nullptr;
    bool in_place = false;
    if (want_surface && dev_surface) {
        surf_target = dev_surface;
        in_place = true;
    } else if (want_surface) {
        surf_target = host_range_dev(c, surface, surf_bytes);
        in_place = surf_target != nullptr;
        if (!in_place) {
            if ((rc = c->io_surface.ensure(surf_bytes))) return rc;
            surf_target = c->io_surface.p;
        }
    }
    memcpy(s.h_base, needle, in1);
    memcpy(s.h_base + s.o_hay, hay, in1);
    const PeakStageOut ho{(caf_peak *)(s.m_base + s.o_peak), (uint64_t *)(s.m_base + s.o_ridx), (void *)(s.m_base + s.o_rval)};
    unsigned *const h_status = (unsigned *)(s.h_base + s.o_status);
    unsigned long long *const h_seq = (unsigned long long *)(s.h_base + s.o_seq);
    if (s.one_launch) {
        rc = surface_single_launch<T>(p, c->stream, s.m_base, s.d_needle, s.m_base + s.o_hay, s.d_spec, surf_target,
                                      (uint64_t *)s.d_ridx, s.d_rval, (caf_peak *)s.d_peak, ho, s.d_sync,
                                      (unsigned *)(s.m_base + s.o_status), (unsigned long long *)(s.m_base + s.o_seq), false);
        if (rc) return rc;
        ++s.launches;
    } else {
        hipStream_t on = c->stream;
        p->spec_override = s.d_spec;
        p->slab_override = s.d_slab;
        p->stage_out = ho;
        hipError_t e1 = hipSuccess;
        if (p->fused && in1 % 16 == 0 && !measure_own_stage_in(p)) {  // the spectrum kernel stages the needle in itself
            p->stage_in_src = s.m_base;
            p->stage_in_dst = s.d_needle;
            p->stage_in_bytes = in1;
        } else {
            const size_t in16 = (in1 / 16 + 255) / 256;
            k_stage_copy<<<(unsigned)(in16 < 1 ? 1 : in16 > 1024 ? 1024 : in16), 256, 0, on>>>(
                CopyJobs{{s.m_base, nullptr, nullptr}, {s.d_needle, nullptr, nullptr}, {in1, 0, 0}});
            e1 = hipGetLastError();
        }
        rc = caf_surface_dev(p, s.d_needle, s.m_base + s.o_hay, 1, surf_target, (uint64_t *)s.d_ridx, s.d_rval,
                             (caf_peak *)s.d_peak);
        p->spec_override = nullptr;
        p->slab_override = nullptr;
        p->stage_out = PeakStageOut{nullptr, nullptr, nullptr};
        p->stage_in_src = nullptr;
        p->stage_in_dst = nullptr;
        p->stage_in_bytes = 0;
        if (rc) return rc;
        if (e1 != hipSuccess) return fail(CAF_ERR_HIP, "stage copy launch: %s", hipGetErrorString(e1));
    }
    if (want_surface && !in_place && (rc = d2h_copy(c, surface, c->io_surface.p, surf_bytes))) return rc;
    if (s.one_launch && !want_surface) {
        if ((rc = poll_seq(h_seq, 1, s.launches, c->stream, "caf_surface"))) return rc;
    } else {
        // (a surface written in place is complete at the END of the kernel: its stores are not system-scope ones)
        HIPCHK(hipStreamSynchronize(c->stream));
        if (s.one_launch && __atomic_load_n(h_seq, __ATOMIC_ACQUIRE) < s.launches)
            return fail(CAF_ERR_HIP, "caf_surface: the launch finished without publishing its results");
    }
    if (s.one_launch && __atomic_load_n(h_status, __ATOMIC_ACQUIRE)) {  // a role ran into its wait bound
        HIPCHK(hipStreamSynchronize(c->stream));
        __atomic_store_n(h_status, 0u, __ATOMIC_RELEASE);
        HIPCHK(hipMemsetAsync(s.d_sync, 0, 512, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        *h_seq = 0;
        s.launches = 0;
        return fail(CAF_ERR_HIP, "caf_surface: the surface launch ran into its wait bound; results discarded");
    }
    if (row_idx && nfreq) memcpy(row_idx, s.h_base + s.o_ridx, nfreq * sizeof(uint64_t));
    if (row_val && nfreq) memcpy(row_val, s.h_base + s.o_rval, nfreq * sizeof(T));
    memcpy(peak, s.h_base + s.o_peak, sizeof(caf_peak));
    return CAF_OK;
}

template <typename T>
static int surface_host_impl(caf_ctx *c, const T *needle, const T *hay, size_t n, const double *freqs,
                             size_t nfreq, uint32_t fs, T *surface, uint64_t *row_idx, T *row_val,
                             caf_peak *peak, int dtype)
{
    if (!c || !needle || !hay || !peak) return fail(CAF_ERR_BAD_ARG, "caf_surface: NULL argument");
    if (!freqs && nfreq) return fail(CAF_ERR_BAD_ARG, "caf_surface: freqs_hz is NULL");
    if (!is_pow2(n)) return fail(CAF_ERR_LENGTH, "caf_surface: n=%zu is not a power of two >= 1", n);
    HIPCHK(hipSetDevice(c->device));
    HostSlot *sp = nullptr;
    int rc = get_host_slot(c, n, freqs, nfreq, fs, dtype, &sp);
    if (rc) return rc;
    return host_slot_run<T>(c, *sp, needle, hay, surface, row_idx, row_val, peak);
}

extern "C" int caf_surface_c128(caf_ctx *c, const double *needle, const double *hay, size_t n,
                                const double *freqs, size_t nfreq, uint32_t fs, double *surface,
                                uint64_t *row_idx, double *row_val, caf_peak *peak)
{
    CAF_GUARD_BEGIN
    return surface_host_impl<double>(c, needle, hay, n, freqs, nfreq, fs, surface, row_idx, row_val, peak, CAF_C128);
    CAF_GUARD_END
}

extern "C" int caf_surface_c64(caf_ctx *c, const float *needle, const float *hay, size_t n, const double *freqs,
                               size_t nfreq, uint32_t fs, float *surface, uint64_t *row_idx, float *row_val,
                               caf_peak *peak)
{
    CAF_GUARD_BEGIN
    return surface_host_impl<float>(c, needle, hay, n, freqs, nfreq, fs, surface, row_idx, row_val, peak, CAF_C64);
    CAF_GUARD_END
}

// ---- caller memory the kernels may write in place ------------------------------------------------------
extern "C" int caf_host_alloc(caf_ctx *c, size_t bytes, void **out)
{
    CAF_GUARD_BEGIN
    if (!c || !out || !bytes) return fail(CAF_ERR_BAD_ARG, "caf_host_alloc: NULL argument or zero size");
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    void *h = nullptr, *m = nullptr;
    hipError_t e = pinned_alloc(&h, bytes);
    if (e != hipSuccess) return fail(CAF_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    e = hipHostGetDevicePointer(&m, h, 0);
    if (e != hipSuccess) { (void)pin_free(h); return fail(CAF_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
    try {
        c->host_ranges[(char *)h] = HostRange{bytes, (char *)m, true, false};
    } catch (...) {
        (void)pin_free(h);
        throw;
    }
    *out = h;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_host_register(caf_ctx *c, void *ptr, size_t bytes)
{
    CAF_GUARD_BEGIN
    if (!c || !ptr || !bytes) return fail(CAF_ERR_BAD_ARG, "caf_host_register: NULL argument or zero size");
    HIPCHK(hipSetDevice(c->device));
    if (c->host_ranges.count((char *)ptr)) return fail(CAF_ERR_STATE, "caf_host_register: %p is already registered", ptr);
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    void *m = nullptr;
    hipError_t e = hipHostGetDevicePointer(&m, ptr, 0);
    if (e != hipSuccess) { (void)hipHostUnregister(ptr); return fail(CAF_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
    try {
        c->host_ranges[(char *)ptr] = HostRange{bytes, (char *)m, false, false};
    } catch (...) {
        (void)hipHostUnregister(ptr);
        throw;
    }
    return CAF_OK;
    CAF_GUARD_END
}

static int host_range_drop(caf_ctx *c, void *ptr, bool owned, const char *who)
{
    if (!c) return fail(CAF_ERR_BAD_ARG, "%s: ctx is NULL", who);
    if (!ptr) return CAF_OK;
    auto it = c->host_ranges.find((char *)ptr);
    if (it == c->host_ranges.end() || it->second.owned != owned || it->second.borrowed)
        return fail(CAF_ERR_BAD_ARG, "%s: %p did not come from this context's %s", who, ptr, owned ? "caf_host_alloc" : "caf_host_register");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    // the range is forgotten only once the runtime has let go of it: after a failed call it is still pinned, still
    // addressable by the kernels and still released by caf_ctx_destroy
    if (owned) HIPCHK(pin_free(ptr));
    else HIPCHK(hipHostUnregister(ptr));
    c->host_ranges.erase(it);
    return CAF_OK;
}
extern "C" int caf_host_free(caf_ctx *c, void *ptr)
{
    CAF_GUARD_BEGIN
    return host_range_drop(c, ptr, true, "caf_host_free");
    CAF_GUARD_END
}
extern "C" int caf_host_unregister(caf_ctx *c, void *ptr)
{
    CAF_GUARD_BEGIN
    return host_range_drop(c, ptr, false, "caf_host_unregister");
    CAF_GUARD_END
}

// --------------------------------------------------------------- find_peak --
extern "C" int caf_find_peak(caf_ctx *c, const double *freqs, const uint64_t *row_idx, const double *row_val,
                             size_t nfreq, caf_peak *peak)
{
    CAF_GUARD_BEGIN
    if (!c || !peak) return fail(CAF_ERR_BAD_ARG, "caf_find_peak: NULL argument");
    if (nfreq && (!freqs || !row_idx || !row_val)) return fail(CAF_ERR_BAD_ARG, "caf_find_peak: NULL rows");
    if (nfreq > 0x7fffffffu) return fail(CAF_ERR_BAD_ARG, "caf_find_peak: too many rows");
    HIPCHK(hipSetDevice(c->device));
    int rc;
    // pinned: [freqs | row_idx | row_val | caf_peak]; k_peak reads the rows and writes the record in place
    const size_t m = nfreq ? nfreq : 1, col = (m * 8 + 255) & ~(size_t)255;
    if ((rc = c->pin_a.ensure(3 * col + 256))) return rc;
    char *h = (char *)c->pin_a.h, *d = (char *)c->pin_a.m;
    if (nfreq) {
        memcpy(h, freqs, nfreq * sizeof(double));
        memcpy(h + col, row_idx, nfreq * sizeof(uint64_t));
        memcpy(h + 2 * col, row_val, nfreq * sizeof(double));
    }
    k_peak<double><<<1, 256, 0, c->stream>>>((const double *)d, (const uint64_t *)(d + col), (const double *)(d + 2 * col),
                                             (int)nfreq, 0, (caf_peak *)(d + 3 * col), PeakStageOut{nullptr, nullptr, nullptr});
    KCHK();
    HIPCHK(hipStreamSynchronize(c->stream));
    memcpy(peak, h + 3 * col, sizeof(caf_peak));
    return CAF_OK;
    CAF_GUARD_END
}

// ------------------------------------------------------------------- views --
template <typename T>
static int view_impl(caf_ctx *c, const T *surface, size_t rows, size_t n, int view, T *out)
{
    const size_t L = 2 * n, width = view == CAF_VIEW_GO ? L : n, off = view == CAF_VIEW_GO ? n : n / 2;
    const size_t in_bytes = rows * L * sizeof(T), out_bytes = rows * width * sizeof(T);
    int rc;
    // memory of caf_host_alloc / caf_host_register is read / written IN PLACE by the kernel (each value crosses PCIe once,
    // under the kernel); anything else goes through the context's device buffers (the runtime's pageable path runs at the
    // pinned rate once the pages are resident: tools/ubench/pcie_rates.hip)
    const T *src = (const T *)host_range_dev(c, surface, in_bytes);
    T *dst = (T *)host_range_dev(c, out, out_bytes);
    if (!src) {
        if ((rc = c->io_surface.ensure(in_bytes))) return rc;
        if ((rc = host_copy(c, (void *)surface, c->io_surface.p, in_bytes, false))) return rc;
        src = (const T *)c->io_surface.p;
    }
    const bool copy_out = dst == nullptr;
    if (copy_out) {
        if ((rc = c->io_a.ensure(out_bytes))) return rc;
        dst = (T *)c->io_a.p;
    }
    for (size_t r0 = 0; r0 < rows; r0 += 65535) {
        const size_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
        k_view<T><<<dim3((unsigned)((width + 255) / 256), (unsigned)nr), 256, 0, c->stream>>>(src + r0 * L, L, width, off,
                                                                                          dst + r0 * width);
    }
    KCHK();
    if (copy_out && (rc = d2h_copy(c, out, c->io_a.p, out_bytes))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    return CAF_OK;
}

extern "C" int caf_surface_view(caf_ctx *c, int dtype, const void *surface, size_t rows, size_t n, int view, void *out)
{
    CAF_GUARD_BEGIN
    if (!c || !out || (!surface && rows)) return fail(CAF_ERR_BAD_ARG, "caf_surface_view: NULL argument");
    if (view != CAF_VIEW_GO && view != CAF_VIEW_PYTHON) return fail(CAF_ERR_BAD_ARG, "caf_surface_view: bad view %d", view);
    if (dtype != CAF_C128 && dtype != CAF_C64) return fail(CAF_ERR_BAD_ARG, "caf_surface_view: bad dtype %d", dtype);
    if (n == 0) return fail(CAF_ERR_LENGTH, "caf_surface_view: n == 0");
    if (rows == 0) return CAF_OK;
    HIPCHK(hipSetDevice(c->device));
    return dtype == CAF_C128 ? view_impl<double>(c, (const double *)surface, rows, n, view, (double *)out)
                             : view_impl<float>(c, (const float *)surface, rows, n, view, (float *)out);
    CAF_GUARD_END
}

// --------------------------------------------------------------- streaming --
struct StreamSlot {
    hipStream_t stream = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    void *h_needle = nullptr, *h_hay = nullptr;            // pinned
    void *h_peak = nullptr, *h_ridx = nullptr, *h_rval = nullptr;  // pinned results
    void *d_needle = nullptr, *d_surface = nullptr;  // (the haystack is read in place from h_hay)
    void *d_ridx = nullptr, *d_rval = nullptr, *d_peak = nullptr;
    void *d_spec = nullptr;  // fused / chain plans: this slot's haystack spectra
    void *d_slab = nullptr;  // chain plans with R = 4: this slot's radix-4 scratch
    unsigned *d_sync = nullptr;    // single-launch surfaces: 128 counter words (four 128-byte lines) per surface of the slot, zero between launches
    unsigned *h_status = nullptr;  // pinned: set by a single-launch surface whose bounded wait ran out
    unsigned long long *h_seq = nullptr;  // pinned [batch]: launches completed per single-launch surface (polled by caf_stream_wait)
    unsigned long long submits = 0;       // replays of this slot's graph so far
    bool own_stream = true;
};

struct caf_stream {
    caf_plan *plan = nullptr;
    size_t batch = 0;
    double run_stats[4] = {0, 0, 0, 0};  // last caf_stream_run: seconds in {fill (memcpy into pinned), launch, wait (poll / sync), collect}
    bool counted = false;  // registered in plan->live_streams
    std::vector<StreamSlot> slots;
};

static void stream_free(caf_stream *st)
{
    if (!st) return;
    for (auto &s : st->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.exec) (void)hipGraphExecDestroy(s.exec);
        if (s.graph) (void)hipGraphDestroy(s.graph);
        for (void *p : {s.h_needle, s.h_hay, s.h_peak, s.h_ridx, s.h_rval, (void *)s.h_status, (void *)s.h_seq})
            if (p) (void)pin_free(p);
        for (void *p : {s.d_needle, s.d_surface, s.d_ridx, s.d_rval, s.d_peak, s.d_spec, s.d_slab, (void *)s.d_sync})
            if (p) (void)dev_free(p);
        if (s.stream && s.own_stream && st->plan) {  // back to the context's pool
            caf_ctx *c = st->plan->ctx;
            for (size_t i = 0; i < c->slot_pool.size(); ++i)
                if (c->slot_pool[i] == s.stream) c->slot_busy[i] = false;
        }
    }
    if (st->counted && st->plan) --st->plan->live_streams;
    delete st;
}

// Do kernels on streams a and b run at the same time?  HIP multiplexes its streams onto a few hardware
// queues (four by default) in an order this library does not control, and two slot streams that share a
// queue serialise their slots (rocprofv3 Queue_Id: with four slots two of them sat on one queue and the run
// dropped from 58 k to 37 k surfaces/s).  Measured directly: an idle kernel of ~0.2 ms on a, an empty kernel
// on b; b's finishing while a is still busy proves separate queues.
static bool streams_overlap(caf_ctx *c, hipStream_t a, hipStream_t b)
{
    if (a == b) return false;
    const auto key = a < b ? std::make_pair(a, b) : std::make_pair(b, a);
    auto it = c->overlap.find(key);
    if (it != c->overlap.end()) return it->second;
    bool concurrent = false;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&eb, hipEventDisableTiming) == hipSuccess) {
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        k_idle<<<1, 64, 0, a>>>(60u);  // 60 x s_sleep 127 = ~0.5 M cycles
        (void)hipEventRecord(ea, a);
        k_empty<<<1, 64, 0, b>>>();
        (void)hipEventRecord(eb, b);
        if (hipEventSynchronize(eb) == hipSuccess) concurrent = hipEventQuery(ea) == hipErrorNotReady;
        (void)hipEventSynchronize(ea);
        (void)hipGetLastError();
    }
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    c->overlap[key] = concurrent;
    return concurrent;
}

extern "C" int caf_stream_create_ex(caf_plan *p, size_t batch, int nslots, int want_surface, unsigned flags,
                                    caf_stream **out)
{
    CAF_GUARD_BEGIN
    if (!p || !out) return fail(CAF_ERR_BAD_ARG, "caf_stream_create: NULL argument");
    *out = nullptr;
    if (batch == 0 || nslots < 2 || nslots > 16) return fail(CAF_ERR_BAD_ARG, "caf_stream_create: batch >= 1, 2 <= nslots <= 16");
    if (flags & ~(unsigned)(CAF_STREAM_SPLIT | CAF_STREAM_THREE_KERNELS | CAF_STREAM_TWO_KERNELS | CAF_STREAM_ONE_KERNEL))
        return fail(CAF_ERR_BAD_ARG, "caf_stream_create_ex: unknown flags 0x%x", flags);
    const bool split = (flags & CAF_STREAM_SPLIT) && batch > 1;
    if (split && batch > 16) return fail(CAF_ERR_BAD_ARG, "caf_stream_create_ex: CAF_STREAM_SPLIT supports at most 16 surfaces per slot");
    caf_ctx *c = p->ctx;
    HIPCHK(hipSetDevice(c->device));
    caf_stream *st = new (std::nothrow) caf_stream;
    if (!st) return fail(CAF_ERR_NOMEM, "out of host memory");
    // which streams share a hardware queue is the runtime's business and changes when streams come and go: probe
    // afresh for every caf_stream instead of trusting results from an earlier one (a few 0.2 ms probes per creation)
    c->overlap.clear();
    st->plan = p;
    st->batch = batch;
    st->slots.resize(nslots);
    const bool private_state = p->fused || p->chain || p->small;  // slots (and split branches) own their spectra / scratch
    // single-surface chains of the tuned n = 4096 path are ONE kernel node (kernels_surf4096.hpp)
    bool one_launch = p->fused && p->rows > 0 && !(flags & CAF_STREAM_THREE_KERNELS) && (batch == 1 || split);
    // One node {staging, spectrum, rows, find_peak} or two {staging + spectrum | rows + find_peak}?  In the one-node
    // form the row workgroups of a launch hold their CU slots while the needle crosses PCIe (~5 us of 33); with
    // two surfaces in flight that costs less than a second node's launch gap, from three on it is the other way
    // round (MI355X: 2 slots 45.5 k vs 38.5 k surfaces/s, 3 slots 50 k vs 55 k, 4 slots 52 k vs 58 k).
    const size_t in_flight = (size_t)nslots * (split ? batch : 1);
    const bool two_nodes = (flags & CAF_STREAM_TWO_KERNELS) || (!(flags & CAF_STREAM_ONE_KERNEL) && in_flight > 2);
    if (measure_keeps_own_kernels(p)) one_launch = false;
    const size_t esz = elem_size(p->dtype), rsz = real_size(p->dtype);
    const size_t in1 = p->n * esz, in_bytes = batch * in1;
    const size_t rows = p->rows ? p->rows : 1;
    const size_t ridx1 = rows * sizeof(uint64_t), rval1 = rows * rsz, surf1 = rows * p->L * rsz;
    const size_t ridx_bytes = batch * ridx1, rval_bytes = batch * rval1, surf_bytes = batch * surf1;
    // per-surface spectrum bytes (+256: the fused path's row-ticket word); split branches get one each
    const size_t spec1 = p->fused ? (size_t)2 * 16 * 256 * esz : (p->chain || p->small) ? p->L * esz : 0;
    const size_t spec_stride = split ? spec1 + 256 : 0;
    const size_t slab1 = p->chain && p->cR >= 4
                             ? (size_t)c->cu_count * chain_wg_per_cu_v(p->clogm, esz, chain_nb_v(p->clogm, esz)) *
                                   chain_slab_arrays_v(p->cR) * 16 * (((size_t)1 << p->clogm) / 16) * esz
                             : 0;
    hipStream_t saved = c->stream;
    std::vector<hipStream_t> aux;   // capture-time fork streams of the split mode
    std::vector<hipEvent_t> evs;
    int rc = CAF_OK;
    auto cleanup_aux = [&]() {
        for (auto e : evs) (void)hipEventDestroy(e);
        for (auto a : aux) (void)hipStreamDestroy(a);
        evs.clear();
        aux.clear();
    };
    auto bail = [&](int code) {
        c->stream = saved; p->spec_override = nullptr; p->slab_override = nullptr;
        p->stage_out = PeakStageOut{nullptr, nullptr, nullptr};
        p->stage_in_src = nullptr; p->stage_in_dst = nullptr; p->stage_in_bytes = 0;
        cleanup_aux();
        stream_free(st);
        return code;
    };
#define SCHK(expr)                                                                                         \
    do {                                                                                                   \
        hipError_t e__ = (expr);                                                                           \
        if (e__ != hipSuccess) return bail(fail(CAF_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__))); \
    } while (0)
    for (size_t si = 0; si < st->slots.size(); ++si) {
        StreamSlot &s = st->slots[si];
        // Fused / chain plans: every slot has private device state -> slots run concurrently on
        // their own streams.  tiled65536 / generic plans share the plan's pass workspaces -> one stream.
        if (private_state || si == 0) {
            if (c->slot_pool.empty()) {  // the context's own stream is the first pooled one: one hardware queue saved
                c->slot_pool.push_back(c->own_stream);
                c->slot_busy.push_back(false);
            }
            // first free pooled stream that runs concurrently with every slot chosen so far (the pool grows up
            // to 12 streams while looking); if the hardware queues are exhausted, the first free one
            const size_t none = (size_t)-1;
            size_t pi = none, fallback = none;
            for (size_t cand = 0; cand < 12 && pi == none; ++cand) {
                if (cand == c->slot_pool.size()) {
                    hipStream_t ns = nullptr;
                    SCHK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
                    c->slot_pool.push_back(ns);
                    c->slot_busy.push_back(false);
                }
                if (c->slot_busy[cand]) continue;
                if (fallback == none) fallback = cand;
                bool ok = true;
                for (size_t sj = 0; sj < si && ok; ++sj) ok = streams_overlap(c, c->slot_pool[cand], st->slots[sj].stream);
                if (ok) pi = cand;
            }
            if (pi == none) {
                if (fallback == none) {  // every pooled stream is busy and the pool is at its cap
                    hipStream_t ns = nullptr;
                    SCHK(hipStreamCreateWithFlags(&ns, hipStreamNonBlocking));
                    c->slot_pool.push_back(ns);
                    c->slot_busy.push_back(false);
                    fallback = c->slot_pool.size() - 1;
                }
                pi = fallback;
            }
            c->slot_busy[pi] = true;
            s.stream = c->slot_pool[pi];
        } else {
            s.stream = st->slots[0].stream;
            s.own_stream = false;
        }
        if (spec1) SCHK(dev_alloc(&s.d_spec, split ? batch * spec_stride : batch * spec1 + 256));
        if (slab1) SCHK(dev_alloc(&s.d_slab, (split ? batch : 1) * slab1));
        SCHK(pinned_alloc(&s.h_needle, in_bytes));
        SCHK(pinned_alloc(&s.h_hay, in_bytes));
        SCHK(pinned_alloc(&s.h_peak, batch * sizeof(caf_peak)));
        SCHK(pinned_alloc(&s.h_ridx, ridx_bytes));
        SCHK(pinned_alloc(&s.h_rval, rval_bytes));
        SCHK(dev_alloc(&s.d_needle, in_bytes));
        SCHK(dev_alloc(&s.d_ridx, ridx_bytes));
        SCHK(dev_alloc(&s.d_rval, rval_bytes));
        SCHK(dev_alloc(&s.d_peak, batch * sizeof(caf_peak)));
        if (want_surface) SCHK(dev_alloc(&s.d_surface, surf_bytes));
        if (one_launch) {
            SCHK(dev_alloc((void **)&s.d_sync, batch * 512));
            SCHK(hipMemset(s.d_sync, 0, batch * 512));
            SCHK(pinned_alloc((void **)&s.h_status, 64));
            memset(s.h_status, 0, 64);
            SCHK(pinned_alloc((void **)&s.h_seq, batch * sizeof(unsigned long long)));
            memset(s.h_seq, 0, batch * sizeof(unsigned long long));
        }
        memset(s.h_needle, 0, in_bytes);
        memset(s.h_hay, 0, in_bytes);
    }
    const bool was_timing = p->timing;
    const bool fork = split && private_state;  // parallel branches need private spectra; otherwise the chains run in series
    if (fork) {
        aux.resize(batch - 1);
        for (auto &a : aux) { a = nullptr; SCHK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); }
        evs.resize(batch);
        for (auto &e : evs) { e = nullptr; SCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
    }
    // One node chain = {[stage-in,] haystack spectrum, row kernel(s), find_peak}: kernels only, which read
    // and write the slot's pinned host buffers through their device mappings.  Batched slot: ONE chain
    // over `batch` surfaces.  CAF_STREAM_SPLIT: `batch` single-surface chains in the slot's one graph, as
    // parallel branches when the plan's state is private.
    auto chain = [&](StreamSlot &s, hipStream_t on, size_t first, size_t nsurf, void *spec, void *slab) -> int {
        char *m_needle = nullptr, *m_hay = nullptr, *m_peak = nullptr, *m_ridx = nullptr, *m_rval = nullptr;
        HIPCHK(hipHostGetDevicePointer((void **)&m_needle, s.h_needle, 0));
        HIPCHK(hipHostGetDevicePointer((void **)&m_hay, s.h_hay, 0));
        HIPCHK(hipHostGetDevicePointer((void **)&m_peak, s.h_peak, 0));
        HIPCHK(hipHostGetDevicePointer((void **)&m_ridx, s.h_ridx, 0));
        HIPCHK(hipHostGetDevicePointer((void **)&m_rval, s.h_rval, 0));
        char *dn = (char *)s.d_needle + first * in1;
        char *dp = (char *)s.d_peak + first * sizeof(caf_peak), *di = (char *)s.d_ridx + first * ridx1;
        char *dv = (char *)s.d_rval + first * rval1;
        char *ds = s.d_surface ? (char *)s.d_surface + first * surf1 : nullptr;
        const size_t inb = nsurf * in1;
        // needle: read twice per Doppler row -> staged into device memory, by extra workgroups of the
        // haystack-spectrum launch on the fused path (k_seq_prepare), by a k_stage_copy node elsewhere;
        // haystack: read once, by the haystack-spectrum kernel -> that kernel reads the pinned host
        // buffer in place; row peaks + caf_peak: written to the pinned result buffers by find_peak
        // itself.  A slot's chain is 3 kernel nodes on the fused path, 4+ on the others.
        CopyJobs jin = {{m_needle + first * in1, nullptr, nullptr}, {dn, nullptr, nullptr}, {inb, 0, 0}};
        const size_t in16 = (inb / 16 + 255) / 256;
        c->stream = on;
        p->spec_override = spec;
        p->slab_override = slab;
        p->stage_out = PeakStageOut{(caf_peak *)(m_peak + first * sizeof(caf_peak)), (uint64_t *)(m_ridx + first * ridx1),
                                    (void *)(m_rval + first * rval1)};
        if (one_launch && nsurf == 1) {
            unsigned *m_status = nullptr;
            unsigned long long *m_seq = nullptr;
            HIPCHK(hipHostGetDevicePointer((void **)&m_status, s.h_status, 0));
            HIPCHK(hipHostGetDevicePointer((void **)&m_seq, s.h_seq, 0));
            const PeakStageOut ho = p->stage_out;
            p->spec_override = nullptr;
            p->slab_override = nullptr;
            p->stage_out = PeakStageOut{nullptr, nullptr, nullptr};
            unsigned *sy = s.d_sync + first * 128;
            return p->dtype == CAF_C128
                       ? surface_single_launch<double>(p, on, jin.src[0], dn, m_hay + first * in1, spec, ds, (uint64_t *)di, dv,
                                                       (caf_peak *)dp, ho, sy, m_status, m_seq + first, two_nodes)
                       : surface_single_launch<float>(p, on, jin.src[0], dn, m_hay + first * in1, spec, ds, (uint64_t *)di, dv,
                                                      (caf_peak *)dp, ho, sy, m_status, m_seq + first, two_nodes);
        }
        hipError_t e1 = hipSuccess;
        if (p->fused && !measure_own_stage_in(p) && inb % 16 == 0) {  // the spectrum kernel stages the needles in itself
            p->stage_in_src = jin.src[0];
            p->stage_in_dst = jin.dst[0];
            p->stage_in_bytes = inb;
        } else {
            k_stage_copy<<<(unsigned)(in16 < 1 ? 1 : in16 > 1024 ? 1024 : in16), 256, 0, on>>>(jin);
            e1 = hipGetLastError();
        }
        int r = caf_surface_dev(p, dn, m_hay + first * in1, nsurf, ds, (uint64_t *)di, dv, (caf_peak *)dp);
        p->spec_override = nullptr;
        p->slab_override = nullptr;
        p->stage_out = PeakStageOut{nullptr, nullptr, nullptr};
        p->stage_in_src = nullptr;
        p->stage_in_dst = nullptr;
        p->stage_in_bytes = 0;
        if (r) return r;
        if (e1 != hipSuccess) return fail(CAF_ERR_HIP, "stage copy launch: %s", hipGetErrorString(e1));
        return CAF_OK;
    };
    // Warm-up outside capture: lets the plan allocate its workspaces (hipMalloc is not capturable)
    {
        StreamSlot &s = st->slots[0];
        rc = chain(s, s.stream, 0, split ? 1 : batch, s.d_spec, s.d_slab);
        if (rc) return bail(rc);
        SCHK(hipStreamSynchronize(s.stream));
        if (s.h_seq) {  // the warm-up was launch 1 of surface 0 only: start every surface of every slot from zero again
            SCHK(hipMemset(s.d_sync, 0, batch * 512));
            memset(s.h_seq, 0, batch * sizeof(unsigned long long));
        }
    }
    p->timing = false;  // event records are not wanted inside the graphs
    for (auto &s : st->slots) {
        hipError_t eb = hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal);
        if (eb != hipSuccess) { p->timing = was_timing; return bail(fail(CAF_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(eb))); }
        hipError_t ef = hipSuccess;
        if (!split) {
            rc = chain(s, s.stream, 0, batch, s.d_spec, s.d_slab);
        } else {
            if (fork) ef = hipEventRecord(evs[0], s.stream);
            for (size_t i = 0; i < batch && rc == CAF_OK && ef == hipSuccess; ++i) {
                hipStream_t on = (fork && i > 0) ? aux[i - 1] : s.stream;
                if (fork && i > 0) ef = hipStreamWaitEvent(on, evs[0], 0);
                if (ef != hipSuccess) break;
                rc = chain(s, on, i, 1, s.d_spec ? (char *)s.d_spec + i * spec_stride : nullptr,
                           s.d_slab ? (char *)s.d_slab + i * slab1 : nullptr);
                if (fork && i > 0 && rc == CAF_OK) {
                    ef = hipEventRecord(evs[i], on);
                    if (ef == hipSuccess) ef = hipStreamWaitEvent(s.stream, evs[i], 0);
                }
            }
        }
        hipError_t ec = hipStreamEndCapture(s.stream, &s.graph);
        p->timing = was_timing;
        if (rc) return bail(rc);
        for (hipError_t e : {ef, ec})
            if (e != hipSuccess) return bail(fail(CAF_ERR_HIP, "graph capture: %s", hipGetErrorString(e)));
        SCHK(hipGraphInstantiate(&s.exec, s.graph, nullptr, nullptr, 0));
        p->timing = false;
    }
    p->timing = was_timing;
#undef SCHK
    cleanup_aux();
    c->stream = saved;
    st->counted = true;
    ++p->live_streams;  // caf_plan_destroy / caf_ctx_destroy refuse while the graphs hold the plan's buffers
    *out = st;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_stream_create(caf_plan *p, size_t batch, int nslots, int want_surface, caf_stream **out)
{
    CAF_GUARD_BEGIN
    return caf_stream_create_ex(p, batch, nslots, want_surface, 0u, out);
    CAF_GUARD_END
}

extern "C" int caf_stream_destroy(caf_stream *st)
{
    CAF_GUARD_BEGIN
    if (!st) return CAF_OK;
    (void)hipSetDevice(st->plan->ctx->device);
    stream_free(st);
    return CAF_OK;
    CAF_GUARD_END
}

static int slot_ok(caf_stream *st, int slot)
{
    if (!st) return fail(CAF_ERR_BAD_ARG, "stream is NULL");
    if (slot < 0 || slot >= (int)st->slots.size()) return fail(CAF_ERR_BAD_ARG, "slot %d out of range", slot);
    return CAF_OK;
}

extern "C" int caf_stream_host_buffers(caf_stream *st, int slot, void **needle, void **haystack)
{
    CAF_GUARD_BEGIN
    int rc = slot_ok(st, slot);
    if (rc) return rc;
    if (needle) *needle = st->slots[slot].h_needle;
    if (haystack) *haystack = st->slots[slot].h_hay;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_stream_submit(caf_stream *st, int slot)
{
    CAF_GUARD_BEGIN
    int rc = slot_ok(st, slot);
    if (rc) return rc;
    HIPCHK(hipSetDevice(st->plan->ctx->device));
    HIPCHK(hipGraphLaunch(st->slots[slot].exec, st->slots[slot].stream));
    ++st->slots[slot].submits;  // counted only once the replay is really enqueued (caf_stream_wait polls for this number)
    return CAF_OK;
    CAF_GUARD_END
}

// Completion of a slot whose surfaces are single launches: each writes its launch count to pinned memory
// behind its results, so the host polls that word (~1 us after the last row) instead of waiting for the
// stream's completion signal; the stream itself is only synchronised when the poll runs out of patience.
static int slot_wait(StreamSlot &s, size_t batch)
{
    if (s.h_seq) return poll_seq(s.h_seq, batch, s.submits, s.stream, "caf_stream_wait");
    HIPCHK(hipStreamSynchronize(s.stream));
    return CAF_OK;
}

extern "C" int caf_stream_wait(caf_stream *st, int slot, caf_peak *peaks, uint64_t *row_idx, void *row_val)
{
    CAF_GUARD_BEGIN
    int rc = slot_ok(st, slot);
    if (rc) return rc;
    StreamSlot &s = st->slots[slot];
    HIPCHK(hipSetDevice(st->plan->ctx->device));
    if ((rc = slot_wait(s, st->batch))) return rc;
    if (s.h_status && __atomic_load_n(s.h_status, __ATOMIC_ACQUIRE)) {  // a single-launch surface gave up waiting for its own lower tickets
        __atomic_store_n(s.h_status, 0u, __ATOMIC_RELEASE);
        HIPCHK(hipStreamSynchronize(s.stream));
        HIPCHK(hipMemset(s.d_sync, 0, st->batch * 512));
        memset(s.h_seq, 0, st->batch * sizeof(unsigned long long));
        s.submits = 0;
        return fail(CAF_ERR_HIP, "caf_stream_wait: slot %d: a surface launch ran into its wait bound; results discarded", slot);
    }
    const size_t rows = st->plan->rows;
    if (peaks) memcpy(peaks, s.h_peak, st->batch * sizeof(caf_peak));
    if (row_idx && rows) memcpy(row_idx, s.h_ridx, st->batch * rows * sizeof(uint64_t));
    if (row_val && rows) memcpy(row_val, s.h_rval, st->batch * rows * real_size(st->plan->dtype));
    return CAF_OK;
    CAF_GUARD_END
}

// The whole streaming loop in native code (BASELINE configs[4]: `count` host-resident pairs, one after the
// other): fill slot k's pinned buffers, replay its graph, collect slot k - nslots + 1 ... so that the
// caller pays one call for the run instead of three per step.  This worker handles the pairs
// first, first + stride, ... (`items` of them) of the caller's arrays and writes their results at the same
// positions: stride 1 is caf_stream_run, stride = number of devices is one worker of caf_multi_stream_run.
static int stream_run_strided(caf_stream *st, const void *needles, const void *haystacks, size_t first0, size_t stride,
                              size_t items, caf_peak *peaks, uint64_t *row_idx, void *row_val)
{
    const caf_plan *p = st->plan;
    const size_t batch = st->batch, nslots = st->slots.size(), rows = p->rows;
    const size_t in1 = p->n * elem_size(p->dtype), rsz = real_size(p->dtype);
    const size_t nsteps = (items + batch - 1) / batch;
    HIPCHK(hipSetDevice(p->ctx->device));
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    double t_fill = 0, t_launch = 0, t_wait = 0, t_collect = 0;
    for (size_t step = 0; step < nsteps + nslots; ++step) {
        StreamSlot &s = st->slots[step % nslots];
        if (step >= nslots && step - nslots < nsteps) {  // collect what this slot carried nslots steps ago
            const size_t j0 = (step - nslots) * batch, k = items - j0 < batch ? items - j0 : batch;
            const auto w0 = clk::now();
            int rc = caf_stream_wait(st, (int)(step % nslots), nullptr, nullptr, nullptr);
            if (rc) return rc;
            const auto w1 = clk::now();
            t_wait += secs(w0, w1);
            for (size_t j = 0; j < k; ++j) {
                const size_t g = first0 + (j0 + j) * stride;
                memcpy(peaks + g, (const caf_peak *)s.h_peak + j, sizeof(caf_peak));
                if (row_idx && rows) memcpy(row_idx + g * rows, (const uint64_t *)s.h_ridx + j * rows, rows * sizeof(uint64_t));
                if (row_val && rows) memcpy((char *)row_val + g * rows * rsz, (const char *)s.h_rval + j * rows * rsz, rows * rsz);
            }
            t_collect += secs(w1, clk::now());
        }
        if (step < nsteps) {
            const size_t j0 = step * batch, k = items - j0 < batch ? items - j0 : batch;
            const auto f0 = clk::now();
            for (size_t j = 0; j < k; ++j) {
                const size_t g = first0 + (j0 + j) * stride;
                memcpy((char *)s.h_needle + j * in1, (const char *)needles + g * in1, in1);
                memcpy((char *)s.h_hay + j * in1, (const char *)haystacks + g * in1, in1);
            }
            if (k < batch) {  // ragged tail: the unused surfaces of the slot run on zeros, their results are dropped
                memset((char *)s.h_needle + k * in1, 0, (batch - k) * in1);
                memset((char *)s.h_hay + k * in1, 0, (batch - k) * in1);
            }
            const auto f1 = clk::now();
            HIPCHK(hipGraphLaunch(s.exec, s.stream));
            ++s.submits;
            t_fill += secs(f0, f1);
            t_launch += secs(f1, clk::now());
        }
    }
    st->run_stats[0] = t_fill; st->run_stats[1] = t_launch; st->run_stats[2] = t_wait; st->run_stats[3] = t_collect;
    return CAF_OK;
}

extern "C" int caf_stream_run_stats(caf_stream *st, double *seconds4)
{
    CAF_GUARD_BEGIN
    if (!st || !seconds4) return fail(CAF_ERR_BAD_ARG, "caf_stream_run_stats: NULL argument");
    for (int i = 0; i < 4; ++i) seconds4[i] = st->run_stats[i];
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_stream_run(caf_stream *st, const void *needles, const void *haystacks, size_t count, caf_peak *peaks,
                              uint64_t *row_idx, void *row_val)
{
    CAF_GUARD_BEGIN
    if (!st) return fail(CAF_ERR_BAD_ARG, "stream is NULL");
    if (count && (!needles || !haystacks || !peaks)) return fail(CAF_ERR_BAD_ARG, "caf_stream_run: NULL argument");
    return stream_run_strided(st, needles, haystacks, 0, 1, count, peaks, row_idx, row_val);
    CAF_GUARD_END
}

// ------------------------------------------------------- surface-parallel multi-GPU --
// The second multi-GPU decomposition (SURVEY.md section 8e "record both"): whole surfaces round-robin over the
// devices -- the unit the reference's own pool hands out when many surfaces are wanted (one caf_surface call
// per bench iteration, benches/caf_bench.rs:150-168; independent tasks, mod.rs:404-457).  One caf_ctx + caf_plan +
// caf_stream per entry of `device_ids`, each driven by its own host thread for the length of a run; no
// collective at all: a surface's (tau, f) is complete on the device that computed it.
extern "C" int caf_multi_stream_share(size_t count, int nworkers, int worker, size_t *first, size_t *stride, size_t *items)
{
    CAF_GUARD_BEGIN
    if (nworkers <= 0 || worker < 0 || worker >= nworkers) return fail(CAF_ERR_BAD_ARG, "caf_multi_stream_share: worker %d of %d", worker, nworkers);
    const size_t w = (size_t)worker, nw = (size_t)nworkers;
    if (first) *first = w;
    if (stride) *stride = nw;
    if (items) *items = count > w ? (count - w + nw - 1) / nw : 0;  // pairs w, w + nw, w + 2 nw, ... < count
    return CAF_OK;
    CAF_GUARD_END
}

struct MultiWorker {
    int device = 0;
    caf_ctx *ctx = nullptr;
    caf_plan *plan = nullptr;
    caf_stream *stream = nullptr;
    int rc = CAF_OK;
    std::string err;
};
struct caf_multi_stream {
    std::vector<MultiWorker> workers;
    size_t surf1 = 0;  // bytes of one surface (0: created without surfaces)
};
static constexpr size_t MULTI_STREAM_BATCH = 8;

extern "C" int caf_multi_stream_destroy(caf_multi_stream *ms)
{
    CAF_GUARD_BEGIN
    if (!ms) return CAF_OK;
    for (auto &w : ms->workers) {
        if (w.stream) caf_stream_destroy(w.stream);
        if (w.plan) caf_plan_destroy(w.plan);
        if (w.ctx) caf_ctx_destroy(w.ctx);
    }
    delete ms;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_multi_stream_create(const int *device_ids, int ndev, size_t n, const double *freqs_hz, size_t nfreq,
                                       uint32_t fs, int dtype, int nslots, int want_surface, caf_multi_stream **out)
{
    CAF_GUARD_BEGIN
    if (!out || !device_ids || ndev <= 0 || ndev > 64) return fail(CAF_ERR_BAD_ARG, "caf_multi_stream_create: bad device list");
    *out = nullptr;
    caf_multi_stream *ms = new (std::nothrow) caf_multi_stream;
    if (!ms) return fail(CAF_ERR_NOMEM, "out of host memory");
    ms->workers.resize(ndev);
    for (int i = 0; i < ndev; ++i) {
        MultiWorker &w = ms->workers[i];
        w.device = device_ids[i];
        int rc = caf_ctx_create(w.device, &w.ctx);
        if (!rc) rc = caf_plan_create(w.ctx, n, freqs_hz, nfreq, fs, dtype, 0, nfreq, &w.plan);
        // eight surfaces per replay: profiles/r03_stream/form_stability.txt
        if (!rc) rc = caf_stream_create(w.plan, MULTI_STREAM_BATCH, nslots, want_surface, &w.stream);
        if (rc) {  // g_err of this thread holds the failing call's message
            caf_multi_stream_destroy(ms);
            return rc;
        }
    }
    ms->surf1 = want_surface ? nfreq * 2 * n * real_size(dtype) : 0;
    *out = ms;
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_multi_stream_devices(const caf_multi_stream *ms) { return ms ? (int)ms->workers.size() : 0; }

extern "C" int caf_multi_stream_run(caf_multi_stream *ms, const void *needles, const void *haystacks, size_t count,
                                    caf_peak *peaks, uint64_t *row_idx, void *row_val)
{
    CAF_GUARD_BEGIN
    if (!ms) return fail(CAF_ERR_BAD_ARG, "multi stream is NULL");
    if (count && (!needles || !haystacks || !peaks)) return fail(CAF_ERR_BAD_ARG, "caf_multi_stream_run: NULL argument");
    const int nw = (int)ms->workers.size();
    std::vector<std::thread> threads;
    bool spawn_failed = false;
    for (int i = 0; i < nw; ++i) {
        MultiWorker *w = &ms->workers[i];
        w->rc = CAF_OK;
        w->err.clear();
        try {  // nothing may unwind across the C boundary: a thread that cannot be started fails the call instead
            threads.emplace_back([=] {  // one host thread per device: the C ABI's contexts are single-threaded objects
                try {  // (an exception escaping a thread function would terminate the process)
                    size_t first = 0, stride = 1, items = 0;
                    caf_multi_stream_share(count, nw, i, &first, &stride, &items);
                    w->rc = stream_run_strided(w->stream, needles, haystacks, first, stride, items, peaks, row_idx, row_val);
                    if (w->rc) w->err = g_err;  // thread-local message of the worker thread
                } catch (...) {
                    w->rc = CAF_ERR_NOMEM;
                }
            });
        } catch (...) {
            spawn_failed = true;
            break;
        }
    }
    for (auto &t : threads) t.join();
    if (spawn_failed) return fail(CAF_ERR_NOMEM, "caf_multi_stream_run: could not start a worker thread (results are incomplete)");
    for (int i = 0; i < nw; ++i)
        if (ms->workers[i].rc)  // per-device error propagation: the first failing device, by position
            return fail(ms->workers[i].rc, "caf_multi_stream_run: worker %d (device %d): %s", i, ms->workers[i].device,
                        ms->workers[i].err.c_str());
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" void *caf_stream_surface(caf_stream *st, int slot)
{
    if (slot_ok(st, slot)) return nullptr;
    return st->slots[slot].d_surface;
}

// Device address of a worker's slot slab ([8][rows][2n] of the dtype's real type), NULL if the object was created
// without surfaces.
extern "C" void *caf_multi_stream_surface(caf_multi_stream *ms, int worker, int slot)
{
    if (!ms || worker < 0 || worker >= (int)ms->workers.size()) return nullptr;
    return caf_stream_surface(ms->workers[(size_t)worker].stream, slot);
}

// Where the surface of pair `pair` of the last caf_multi_stream_run over `count` pairs lives: pair k went to worker
// k % ndev as that worker's item j = k / ndev, in replay j / 8 -> slot (j / 8) % nslots, position j % 8 of the slab;
// *resident = 0 if a later replay of the same run has overwritten that slot since.
extern "C" int caf_multi_stream_locate(const caf_multi_stream *ms, size_t count, size_t pair, int *worker, int *slot,
                                       size_t *index, int *resident)
{
    CAF_GUARD_BEGIN
    if (!ms || pair >= count) return fail(CAF_ERR_BAD_ARG, "caf_multi_stream_locate: pair %zu of %zu", pair, count);
    const size_t nw = ms->workers.size(), w = pair % nw, j = pair / nw;
    const size_t nslots = ms->workers[w].stream->slots.size();
    const size_t items = count > w ? (count - w + nw - 1) / nw : 0;
    const size_t steps = (items + MULTI_STREAM_BATCH - 1) / MULTI_STREAM_BATCH, step = j / MULTI_STREAM_BATCH;
    if (worker) *worker = (int)w;
    if (slot) *slot = (int)(step % nslots);
    if (index) *index = j % MULTI_STREAM_BATCH;
    if (resident) *resident = step + nslots >= steps;
    return CAF_OK;
    CAF_GUARD_END
}

// ------------------------------------------------------------- debug: red zones --
extern "C" int caf_debug_guard_bands(size_t bytes)
{
    CAF_GUARD_BEGIN
    if (bytes > ((size_t)64 << 20)) return fail(CAF_ERR_BAD_ARG, "caf_debug_guard_bands: %zu bytes per guard is more than 64 MiB", bytes);
    g_guard_bytes.store((bytes + 4095) & ~(size_t)4095);  // whole pages: the alignment of every allocation is kept
    return CAF_OK;
    CAF_GUARD_END
}

extern "C" int caf_debug_check_guards(size_t *allocations_checked, size_t *violations)
{
    CAF_GUARD_BEGIN
    size_t checked = 0, bad = 0;
    std::string first;
    int saved = 0;
    (void)hipGetDevice(&saved);
    std::vector<unsigned char> buf;
    std::lock_guard<std::mutex> lk(g_guard_mu);
    for (const auto &kv : g_guarded) {
        const GuardRec &r = kv.second;
        HIPCHK(hipSetDevice(r.device));
        HIPCHK(hipDeviceSynchronize());
        const unsigned char *head, *tail;
        if (r.pinned) {
            head = (const unsigned char *)r.base;
            tail = (const unsigned char *)r.base + r.guard + r.bytes;
        } else {
            buf.resize(2 * r.guard);
            HIPCHK(hipMemcpy(buf.data(), r.base, r.guard, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(buf.data() + r.guard, r.base + r.guard + r.bytes, r.guard, hipMemcpyDeviceToHost));
            head = buf.data();
            tail = buf.data() + r.guard;
        }
        ++checked;
        long off = 0;
        bool hit = false;
        for (size_t i = r.guard; i-- > 0 && !hit;)  // nearest byte first on the head side
            if (head[i] != GUARD_FILL) { hit = true; off = -(long)(r.guard - i); }
        for (size_t i = 0; i < r.guard && !hit; ++i)
            if (tail[i] != GUARD_FILL) { hit = true; off = (long)(r.bytes + i); }
        if (hit) {
            ++bad;
            if (first.empty()) {
                char msg[256];
                snprintf(msg, sizeof msg, "%s allocation of %zu bytes made at %s:%d (device %d) was written at byte offset %ld",
                         r.pinned ? "pinned" : "device", r.bytes, r.file, r.line, r.device, off);
                first = msg;
            }
        }
    }
    (void)hipSetDevice(saved);
    if (allocations_checked) *allocations_checked = checked;
    if (violations) *violations = bad;
    if (bad) return fail(CAF_ERR_STATE, "caf_debug_check_guards: %zu of %zu allocations have a damaged red zone; first: %s", bad, checked, first.c_str());
    return CAF_OK;
    CAF_GUARD_END
}

// ------------------------------------------------ row shards of ONE surface over devices --
#include "caf_multi.inc"
