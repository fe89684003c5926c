// kernels_surf4096.hpp -- ONE launch = ONE whole surface, n = 4096 (streaming slots, caf_stream_*).
//
// A streaming slot that carries one surface per replay used to be three dependent kernel nodes
// {haystack spectrum (+ needle staging), Doppler rows, find_peak}: 10.5 + 23 + 4 us back to back, of
// which only the middle one fills the chip.  Here the three are one launch of
//     copy_blocks + 2 + rows   workgroups,
// whose ROLE is the order in which they start (one atomic ticket each, so "lower ticket" always means
// "already running": the ordered-ticket rule of decoupled look-back scans; blockIdx is not used):
//   tickets [0, copy_blocks)        stage the needle from the slot's pinned host buffer into device memory
//   tickets copy_blocks + {0, 1}    haystack spectrum of the even / odd bins (reads the pinned haystack in place)
//   the rest                        one Doppler row each: waits for the staged needle, runs the even chain's
//                                   forward transform WHILE the spectrum workgroups are still busy, waits for
//                                   the spectrum just before multiplying by it, finishes the row, publishes
//                                   its peak; the LAST row to finish runs find_peak over all rows
//                                   (mod.rs:31-42), writes the caf_peak record and re-arms the counters.
// Waiting is only ever for lower tickets, so the launch cannot deadlock whatever else occupies the GPU;
// every wait is bounded (~1 s) and reports through SurfArgs::status instead of hanging.
// Cross-workgroup data (needle, spectrum, row peaks) travels through device memory between CUs of
// different XCDs, whose L2s are not coherent with one another.  Agent-scope release / acquire FENCES are
// whole-L2 operations (buffer_wbl2 / buffer_inv): with ~800 of them per surface the first version of this
// kernel took 70 us.  Instead the coherence is per access:
//   * producers store WRITE-THROUGH (sc1: the data is in memory, not only in their XCD's L2), wait for the
//     stores (s_waitcnt vmcnt(0)), then bump their counter with a relaxed agent-scope atomic;
//   * consumers poll the counter with agent-scope loads and only then issue their first load of the data.
//     No L2 or L1 can hold an older copy of it: every launch starts with clean caches (the kernel-boundary
//     acquire), and inside the launch nobody reads needle / spectrum before the counter says so;
//   * the few words another workgroup may share a cache line with (row_val / row_idx, read by the last
//     row) are read with agent-scope (sc1) loads.
// The row arithmetic is seq_chain / the k_seq_rows epilogue unchanged (same bits as the batched path:
// tests/test_gpu_round2.py::test_stream_single_launch_surface).
#pragma once
#include "kernels_generic.hpp"
#include "kernels_seq4096.hpp"

namespace caf {

template <typename T>
struct SurfArgs {
    const cpx<T> *hay;     // haystack [4096]: pinned host buffer (device mapping) or device memory, read once
    unsigned *sync;        // device words [0], [32], [64], [96] (a 128-byte line each: polls of one do not queue up in front of
                           // atomics on another) = {ticket, needle blocks done, spectrum chains done, rows done}; zero between launches
    unsigned *status;      // device word: set non-zero if a wait ran into its bound (results of that launch are invalid)
    unsigned copy_blocks;  // workgroups that stage the needle (FusedArgs::stage_*); 0 = the needle is already at FusedArgs::sig
    unsigned prep_blocks;  // 2 = this launch computes the haystack spectrum itself; 0 = a k_seq_prepare node in front of it did
    const double *freqs;   // [rows]
    int64_t row_base;      // global list position of row 0 (row shards)
    caf_peak *peak;        // device record
    caf_peak *h_peak;      // pinned result buffers (device mappings), or nullptr
    uint64_t *h_ridx;
    T *h_rval;
    unsigned long long *h_seq;  // pinned: receives the number of launches this surface slot has completed, AFTER all results
                                // (the host polls it instead of synchronising the stream), or nullptr
};

// wave-uniform bounded wait until *p >= target (agent-scope relaxed polls: they bypass the non-coherent
// caches); the compiler barrier keeps the data loads that follow behind the last poll
__device__ __forceinline__ bool surf_wait(const unsigned *p, unsigned target)
{
    bool ok = false;
#pragma unroll 1
    for (unsigned it = 0; it < (1u << 20); ++it) {
        const unsigned v = __builtin_amdgcn_readfirstlane(
            __hip_atomic_load(const_cast<unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        if (v >= target) { ok = true; break; }
        __builtin_amdgcn_s_sleep(24);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return ok;
}
// the same for a whole workgroup: wave 0 polls (1 600 waves polling one word would queue up in front of the
// very atomics they wait for), the others sleep at the barrier
__device__ __forceinline__ bool surf_wait_wg(const unsigned *p, unsigned target, volatile unsigned *flag, int wave)
{
    if (__builtin_amdgcn_readfirstlane(wave) == 0) {
        const bool ok = surf_wait(p, target);
        *flag = ok ? 1u : 0u;
    }
    __syncthreads();
    const bool ok = *flag != 0u;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return ok;
}
// producer side: every thread's write-through stores have reached memory, then one relaxed count
__device__ __forceinline__ void surf_publish(unsigned *counter, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_real_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off, double *)
{
    const caf_v2u r = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, CAF_AUX_SC1);
    return __longlong_as_double(((long long)r.y << 32) | r.x);
}
__device__ __forceinline__ float load_real_sc1(__amdgpu_buffer_rsrc_t rs, unsigned off, float *)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, CAF_AUX_SC1));
}

template <typename T>
__global__ __launch_bounds__(S_THREADS, seq_waves_per_simd<T>()) void k_seq_surface(const FusedArgs<T> A,
                                                                                const cpx<T> *__restrict__ phasor,
                                                                                const SurfArgs<T> S)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[seq_lds_bytes<T>()];
    C *const Lc = reinterpret_cast<C *>(smem);
    C *const twb = Lc + F_CHAIN;
    unsigned char *const scratch = smem + (F_CHAIN + 256) * sizeof(C);
    volatile unsigned *const sh = reinterpret_cast<volatile unsigned *>(scratch + 112);
    const SeqLane L;
    if (L.tid == 0) sh[0] = __hip_atomic_fetch_add(&S.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    twb[L.tid] = A.tab.tw4096[16 * (L.tid & 15) * (L.tid >> 4)];
    __syncthreads();
    const unsigned ticket = __builtin_amdgcn_readfirstlane(sh[0]);

    // ---- role 1: needle staging ------------------------------------------------------------------
    if (ticket < S.copy_blocks) {
        const unsigned nthr = S.copy_blocks * S_THREADS;
        const __amdgpu_buffer_rsrc_t rs_dst =
            __builtin_amdgcn_make_buffer_rsrc((void *)A.stage_dst, 0, (int)(A.stage_n16 * 16u), 0x00020000);
        for (unsigned i = ticket * S_THREADS + L.tid; i < A.stage_n16; i += nthr) {
            const uint4 x = A.stage_src[i];
            store_vec_aux<CAF_AUX_SC1>(rs_dst, i * 16u, caf_v4u{x.x, x.y, x.z, x.w});
        }
        surf_publish(&S.sync[32], L.tid);
        return;
    }
    TwSet<T> tw;
    tw.w1 = A.tab.tw4096[L.t * 1];
    tw.w2 = A.tab.tw4096[L.t * 2];
    tw.w3 = A.tab.tw4096[L.t * 3];
    tw.w4 = A.tab.tw4096[L.t * 4];
    tw.w8 = A.tab.tw4096[L.t * 8];
    tw.w12 = A.tab.tw4096[L.t * 12];
    const C *const twB = twb + L.lo4;
    const unsigned role = ticket - S.copy_blocks;

    // ---- role 2: haystack spectrum of one chain (same arithmetic as k_seq_prepare) -----------------
    if (role < S.prep_blocks) {
        const int chain = (int)role;
        const C *__restrict__ ph = phasor + (size_t)A.rows * 64;  // the f = 0 row
        const T inv = T(1.0 / 8192.0);
        C pb = cmul(ph[L.lo4], ph[16 + L.hi4]);
        if (chain) pb = cmulc(pb, A.tab.th[L.t]);
        const C *ps = ph + 32 + 16 * chain;
        C v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = conj(cmul(cmul(S.hay[L.t + 256 * q], pb), ps[q]));
        dft16(v);
        apply_twA(v, tw);
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[L.pA + k * F_BLK] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pB + 17 * k];
        dft16(v);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twB[16 * k]);
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[L.pB + 17 * k] = v[k];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pC + k];
        dft16(v);
        const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.spec + chain * (16 * 256)), 0, 16 * 256 * (int)sizeof(C), 0x00020000);
#pragma unroll
        for (int k = 0; k < 16; ++k)
            store_pair_aux<CAF_AUX_SC1>(rs_h, (unsigned)((k * 256 + L.t) * sizeof(C)), v[k].x * inv, -v[k].y * inv);
        surf_publish(&S.sync[64], L.tid);
        return;
    }

    // ---- role 3: one Doppler row -----------------------------------------------------------------------
    const int r = (int)(role - S.prep_blocks);  // < A.rows by construction of the grid
    if (r >= A.rows) return;
    const C th = A.tab.th[L.t];
    const C cfac = conj(th);
    const int mpair = L.t & ~1;
    const bool odd = L.lane & 1;
    constexpr unsigned long long EVEN_LANES = 0x5555555555555555ull;
    const C *__restrict__ ph = phasor + (size_t)r * 64;
    const C pb = cmul(ph[L.lo4], ph[16 + L.hi4]);
    bool ok = true;
    if (S.copy_blocks) ok = surf_wait_wg(&S.sync[32], S.copy_blocks, sh + 2, L.wave);
    const __amdgpu_buffer_rsrc_t rs_sig =
        __builtin_amdgcn_make_buffer_rsrc((void *)A.sig, 0, F_N * (int)sizeof(C), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_spec =
        __builtin_amdgcn_make_buffer_rsrc((void *)A.spec, 0, 2 * 16 * 256 * (int)sizeof(C), 0x00020000);
    C a[16];
    load_samples(a, rs_sig, L);
    C e[16], o[16];
    // PF 8: samples preloaded; no early spectrum request in the even chain (the wait sits in front of the
    // product instead, behind the whole forward transform); 2 | 4: the odd chain's loads as in k_seq_rows
    constexpr int PF = 2 | 4 | 8;
    // the spectrum counter is requested here and looked at behind the forward transform: in the usual case
    // (spectrum ready by then) the wait costs nothing, not even the round trip of one poll; otherwise every
    // wave polls for itself (no barrier inside the chain)
    bool okh = true;
    const unsigned h_early = S.prep_blocks ? __hip_atomic_load(&S.sync[64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const SeqIo<T> io{Lc, twB, L};
    seq_chain<T, 0, PF>(e, a, io, rs_sig, rs_spec, pb, th, ph + 32, tw, L, [&]() {
        if (__builtin_amdgcn_readfirstlane(h_early) < S.prep_blocks) okh = surf_wait(&S.sync[64], S.prep_blocks);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    });
    seq_chain<T, 1, PF>(o, a, io, rs_sig, rs_spec, cmul(pb, cfac), th, ph + 48, tw, L);

    // last radix-2 stage + |.|^2 + argmax + 16-byte write-through stores (k_seq_rows epilogue, one row)
    T bv_lo = T(0), bv_hi = T(0);
    int bi_lo = 0, bi_hi = 0;
    T *const out = A.surface ? A.surface + (size_t)r * F_L : nullptr;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? F_L * (int)sizeof(T) : 0, 0x00020000);
    T mlo[16], mhi[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        C lo, hi;
        axpy_w32(i, e[i], o[i], lo, hi);
        mlo[i] = norm_sqr(lo);  // mod.rs:147
        mhi[i] = norm_sqr(hi);
        bi_lo = mlo[i] > bv_lo ? i : bi_lo;  // first strictly greater (mod.rs:148-151)
        bv_lo = vmax(bv_lo, mlo[i]);
        bi_hi = mhi[i] > bv_hi ? i : bi_hi;
        bv_hi = vmax(bv_hi, mhi[i]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        typename pair_vec<T>::type dlo, dhi;
        pair_xor1(mlo[2 * j], mlo[2 * j + 1], EVEN_LANES, ~EVEN_LANES, dlo);
        pair_xor1(mhi[2 * j], mhi[2 * j + 1], EVEN_LANES, ~EVEN_LANES, dhi);
        const int m = mpair + 256 * (2 * j + (odd ? 1 : 0));
        store_vec_aux<CAF_AUX_SC1>(rs, (unsigned)(m * sizeof(T)), dlo);
        store_vec_aux<CAF_AUX_SC1>(rs, (unsigned)((m + F_N) * sizeof(T)), dhi);
    }
    const __amdgpu_buffer_rsrc_t rs_ri =
        __builtin_amdgcn_make_buffer_rsrc((void *)A.row_idx, 0, A.rows * (int)sizeof(uint64_t), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rv =
        __builtin_amdgcn_make_buffer_rsrc((void *)A.row_val, 0, A.rows * (int)sizeof(T), 0x00020000);
    T bv = bv_lo;
    uint32_t bi = bv_lo > T(0) ? (uint32_t)(L.t + 256 * bi_lo) : 0u;
    if (bv_hi > bv) { bv = bv_hi; bi = (uint32_t)(L.t + 256 * bi_hi + F_N); }
    wave_arg_reduce_maxmin(bv, bi);
    {
        T *sv = reinterpret_cast<T *>(scratch);
        uint32_t *si = reinterpret_cast<uint32_t *>(scratch + 32);
        if (L.lane == 63) { sv[L.wave] = bv; si[L.wave] = bi; }
    }
    // EVERY wave's surface stores are in memory before the row is counted: the last row publishes the launch
    // (h_seq) on the strength of that count, and a host that polls h_seq may read the surface right away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (L.tid == 0) {
        const T *sv = reinterpret_cast<const T *>(scratch);
        const uint32_t *si = reinterpret_cast<const uint32_t *>(scratch + 32);
        T rb = sv[0];
        uint32_t ri = si[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) arg_merge(rb, ri, sv[w], si[w]);
        store_vec_aux<CAF_AUX_SC1>(rs_ri, (unsigned)(r * sizeof(uint64_t)), caf_v2u{ri, 0u});
        store_one_aux<CAF_AUX_SC1>(rs_rv, (unsigned)(r * sizeof(T)), 0u, rb);
        if (S.h_peak) {
            // SYSTEM-scope stores (sc0 sc1): only for those does the s_waitcnt below mean "visible to the host".
            // Plain stores to the pinned buffers are acknowledged early; a soak of 10^5 surfaces then showed
            // 1-3 surfaces whose row words reached the host AFTER the sequence word of the last row
            // (tools/stream_soak.py).
            __hip_atomic_store(&S.h_ridx[r], (uint64_t)ri, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&S.h_rval[r], rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // (system scope like the row words: the flag must not reach the host behind the sequence word)
        if (!(ok && okh)) __hip_atomic_store(S.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this row's peak is in memory before it is counted
        sh[1] = __hip_atomic_fetch_add(&S.sync[96], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if ((int)__builtin_amdgcn_readfirstlane(sh[1]) != A.rows - 1) return;

    // ---- the last row to finish: find_peak over all rows (k_peak's arithmetic), re-arm the counters ----
    asm volatile("" ::: "memory");
    double pv = 0.0;
    uint32_t pr = 0xffffffffu;  // "no row" sorts last among equal (zero) values
    for (int q = L.tid; q < A.rows; q += S_THREADS) {
        const double v = (double)load_real_sc1(rs_rv, (unsigned)(q * sizeof(T)), (T *)nullptr);
        if (v > 0.0) arg_merge(pv, pr, v, (uint32_t)q);
    }
    wave_arg_reduce(pv, pr);
    double *const pvs = reinterpret_cast<double *>(scratch);       // [4]
    uint32_t *const prs = reinterpret_cast<uint32_t *>(scratch + 32);  // [4]
    __syncthreads();  // thread 0 has read sv / si
    if (L.lane == 0) { pvs[L.wave] = pv; prs[L.wave] = pr; }
    __syncthreads();
    if (L.tid == 0) {
        for (int w = 1; w < 4; ++w) arg_merge(pv, pr, pvs[w], prs[w]);
        caf_peak pk;
        if (pr == 0xffffffffu) {
            pk.val = 0.0; pk.freq = 0.0; pk.idx = 0; pk.row = -1;
        } else {
            pk.val = pv; pk.freq = S.freqs[pr];
            const caf_v2u iv = __builtin_amdgcn_raw_buffer_load_b64(rs_ri, (unsigned)(pr * sizeof(uint64_t)), 0, CAF_AUX_SC1);
            pk.idx = iv.x;
            pk.row = S.row_base + (int64_t)pr;
        }
        S.peak[0] = pk;
        if (S.h_peak) S.h_peak[0] = pk;
        if (S.h_seq) {
            // sync[127] counts this slot's launches and is never re-armed; the system-scope release puts the
            // row peaks (written by the other rows before they were counted) and the record in front of it
            const unsigned long long seq = (unsigned long long)(++S.sync[127]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(S.h_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // every other workgroup of this launch is past its last counter access
        __hip_atomic_store(&S.sync[32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&S.sync[64], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&S.sync[96], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&S.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace caf
