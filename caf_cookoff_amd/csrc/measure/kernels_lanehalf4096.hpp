// kernels_lanehalf4096.hpp -- MEASUREMENT LIBRARY ONLY (-DCAF_MEASURE, CAF_ROW_KERNEL=1): the first tuned n = 4096 row
// kernel of round 1, k_fused_rows -- 512 threads per row, the even- and odd-bin chains of a row in the two lane halves
// of every wave, joined by v_permlane32_swap.  It runs at the speed of the product kernel k_seq_rows (kernels_seq4096.hpp;
// HISTORY.md section 5) and carries the only stamped (s_memtime) build left, tools/stamps.py.  The haystack spectrum it
// reads is the product's (k_seq_prepare).  Moved out of kernels_fused4096.hpp in round 3 so that the product headers
// hold product code only.
//
// One 512-thread workgroup (8 waves, 2 per SIMD) computes one whole CAF row in VGPRs + LDS: lanes 0-31 of every wave
// run the EVEN-bin chain, lanes 32-63 the ODD-bin chain of the same 32 butterflies (8 waves x 32 = 256 butterfly
// columns x 16 points = 4096); the last radix-2 stage c[m], c[m+4096] = E[m] +- T^m O[m] pairs lane l with lane l+32
// of the same wave (v_permlane32_swap: no LDS, no barrier); |.|^2, the first-max argmax (mod.rs:143-151) and the
// coalesced surface store are the epilogue of the last butterfly.  LDS: (2 chains x 4352 (padded) + 256) x
// sizeof(complex) + 256 B = 140.25 KiB (f64) / 70.25 KiB (f32); 3 workgroup barriers per row.
#pragma once
#include "../kernels_fused4096.hpp"

namespace caf {

constexpr int F_NSTAMP = 17;
// In-kernel stamp (cdna_hip_programming.md section 7): s_memtime + lgkmcnt(0) in ONE asm
// statement, fenced by sched_barriers.  Only the DIAG instantiation executes any.
#define CAF_STAMP(i)                                                                        \
    if constexpr (DIAG) {                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st[i])::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    }


// Per-thread geometry shared by the prepare and the row kernel.
struct FusedLane {
    int tid, lane, wave, chain, t, hi4, lo4;
    int pA, pB, pC;  // LDS element offsets of the three access patterns (within the chain)
    __device__ __forceinline__ FusedLane()
    {
        tid = threadIdx.x;
        lane = tid & 63;
        wave = tid >> 6;
        chain = lane >> 5;             // 0: even bins (E), 1: odd bins (O)
        t = wave * 32 + (lane & 31);   // butterfly column 0..255
        hi4 = t >> 4;
        lo4 = t & 15;
        pA = t + hi4;
        pB = hi4 * F_BLK + lo4;
        pC = hi4 * F_BLK + 17 * lo4;
    }
};

// Forward (DIF) chain after the mixer: v[q] = u[t + 256q]  ->  v[k2] = G[k0 + 16*k1 + 256*k2]
// with (k0,k1) = (hi4, lo4).  One workgroup barrier (exchange 1); exchange 2 is wave-local.
template <typename T>
__device__ __forceinline__ void fwd_chain(cpx<T> (&v)[16], const cpx<T> (&twA)[16], const cpx<T> *twB,
                                          cpx<T> *Lc, const FusedLane &L)
{
    // pass 1: over n2, twiddle W_4096^(t*k0)
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twA[k]);
    // exchange 1: [k0][t]  ->  thread (k0'=hi4, n0'=lo4) gathers over n1
#pragma unroll
    for (int k = 0; k < 16; ++k) Lc[L.pA + k * F_BLK] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = Lc[L.pB + 17 * k];
    // pass 2: over n1, twiddle W_256^(n0'*k1)
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twB[16 * k]);
    // exchange 2 (wave-local): write [k0'][k1][n0'], read transposed
#pragma unroll
    for (int k = 0; k < 16; ++k) Lc[L.pB + 17 * k] = v[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = Lc[L.pC + k];
    // pass 3: over n0
    dft16(v);
}

// mixer (mod.rs:46-65) + conjugation: v[q] = conj(a[q] * pb * ps[q])
template <typename T>
__device__ __forceinline__ cpx<T> row_phasor_base(const cpx<T> *ph, const FusedLane &L, cpx<T> cfac)
{
    return cmul(cmul(ph[L.lo4], ph[16 + L.hi4]), cfac);
}

// ---- the row kernel ------------------------------------------------------------------------
// (f32 at __launch_bounds__(512, 4) -- two workgroups = 4 waves per SIMD -- was measured:
// 128 VGPRs cost 37 spills and 12 % of throughput.)
template <typename T, bool DIAG = false>
__global__ __launch_bounds__(F_THREADS) void k_fused_rows(const FusedArgs<T> A)
{
    using C = cpx<T>;
    unsigned long long st[F_NSTAMP] = {};
    int iter = 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[fused_lds_bytes<T>()];
    C *const lds = reinterpret_cast<C *>(smem);
    C *const twb = lds + 2 * F_CHAIN;  // twb[k*16 + lo] = W_256^(lo*k)
    unsigned char *const scratch = smem + (2 * F_CHAIN + 256) * sizeof(C);
    const FusedLane L;
    C *const Lc = lds + L.chain * F_CHAIN;

    // ---- twiddles: six W_4096^(t*k) in registers (rest derived), W_256^(lo4*k) in LDS ---------
    TwSet<T> tw;
    tw.w1 = A.tab.tw4096[L.t * 1];
    tw.w2 = A.tab.tw4096[L.t * 2];
    tw.w3 = A.tab.tw4096[L.t * 3];
    tw.w4 = A.tab.tw4096[L.t * 4];
    tw.w8 = A.tab.tw4096[L.t * 8];
    tw.w12 = A.tab.tw4096[L.t * 12];
    if (L.tid < 256) twb[L.tid] = A.tab.tw4096[16 * (L.tid & 15) * (L.tid >> 4)];
    const C *const twB = twb + L.lo4;  // twB[16*k]
    // last-stage twiddle base T^(t + 256*m2): lanes 0-31 end up owning m2 = 8+i -> extra *i
    const C th = A.tab.th[L.t];
    const C tbase = L.chain == 0 ? muli(th) : th;
    const C cfac = L.chain ? conj(th) : C{T(1), T(0)};  // odd chain: e^{-2*pi*i*t/8192}
    const int mbase = L.t + (L.chain == 0 ? 2048 : 0);
    const int mpair = (L.t & ~1) + (L.chain == 0 ? 2048 : 0);
    const bool odd = L.lane & 1;

    // ---- software pipeline prologue: needle samples of the first row -----------------
    int g = blockIdx.x;
    C a[16];
    const unsigned voff_sig = (unsigned)(L.t * sizeof(C));
    const unsigned voff_spec = (unsigned)((L.chain * 4096 + L.t) * sizeof(C));
    {
        const int gc = g < A.total ? g : A.total - 1;  // total >= 1 (host guarantees)
        const __amdgpu_buffer_rsrc_t rs_sig =
            __builtin_amdgcn_make_buffer_rsrc((void *)(A.sig + (size_t)(gc / A.rows) * F_N), 0, F_N * (int)sizeof(C), 0x00020000);
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = bload(rs_sig, voff_sig, (unsigned)(256 * q * sizeof(C)), (C *)nullptr);
    }
    __syncthreads();  // twb table visible

    // (A static s_setprio(1) for waves 4-7 was measured: it only swaps which wave of a SIMD
    // pair starves -- barrier wait moves from waves 0-3 to waves 4-7, throughput -2 %.)
    int parity = 0, prev_g = -1;
    for (; g < A.total; parity ^= 1, ++iter) {
        CAF_STAMP(0);
        const int gn = g + gridDim.x;
        const int b = g / A.rows, r = g - b * A.rows;
        C v[16];
        // ---- mixer (mod.rs:46-65) fused into the first butterfly's operands ---------
        {
            const C *ph = A.phasor + (size_t)r * 64;
            const C pb = row_phasor_base(ph, L, cfac);
            const C *ps = ph + 32 + L.chain * 16;
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = conj(cmul(cmul(a[q], pb), ps[q]));
        }
        CAF_STAMP(1);
        // ---- forward chain -------------------------------------------------------------
        dft16(v);
        apply_twA(v, tw);
        CAF_STAMP(2);
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[L.pA + k * F_BLK] = v[k];
        CAF_STAMP(3);
        __syncthreads();
        CAF_STAMP(4);
        // row result of the previous iteration (its scratch was written before this barrier)
        if (L.tid == 0 && prev_g >= 0) {
            const T *sv = reinterpret_cast<const T *>(scratch + (parity ^ 1) * 128);
            const uint32_t *si = reinterpret_cast<const uint32_t *>(scratch + (parity ^ 1) * 128 + 64);
            T bv = sv[0];
            uint32_t bi = si[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) arg_merge(bv, bi, sv[w], si[w]);
            A.row_idx[prev_g] = bi;
            A.row_val[prev_g] = bv;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pB + 17 * k];
        CAF_STAMP(5);
        dft16(v);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twB[16 * k]);
        CAF_STAMP(6);
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[L.pB + 17 * k] = v[k];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pC + k];
        CAF_STAMP(7);
        // haystack-spectrum loads issued here land under the pass-3 butterfly
        C h[16];
        {
            const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(A.spec + (size_t)b * (2 * 16 * 256)), 0, 2 * 16 * 256 * (int)sizeof(C), 0x00020000);
#pragma unroll
            for (int k = 0; k < 16; ++k) h[k] = bload(rs_spec, voff_spec, (unsigned)(256 * k * sizeof(C)), (C *)nullptr);
        }
        dft16(v);  // -> G[k0 + 16*k1 + 256*k2], k2 = register
        CAF_STAMP(8);
        // ---- spectrum product (xcor_rustfft.rs:64-73) ---------------------------------
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = cmul(v[k], h[k]);
        // ---- inverse (DIT) pass I: over k2 -> m0 -----------------------------------------
        dft16(v);
        CAF_STAMP(9);
        // exchange 3 (wave-local): thread (k0,k1) writes transposed, thread (k0,m0) gathers k1
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[L.pC + k] = v[k];
        wave_lds_fence();
        // twiddle W_256^(k1*m0)
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pB + 17 * k];
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twB[16 * k]);
        CAF_STAMP(10);
        // ---- pass II: over k1 -> m1 ------------------------------------------------------
        dft16(v);
        // exchange 4: [k0][16*m1 + m0] -> thread j=t gathers over k0
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[L.pB + 17 * k] = v[k];
        CAF_STAMP(11);
        __syncthreads();
        CAF_STAMP(12);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pA + k * F_BLK];
        CAF_STAMP(13);
        // All LDS reads of this row are done: release the next row's exchange-1 writes here,
        // so the epilogue, the stores and the next mixer run barrier-free.
        __syncthreads();
        CAF_STAMP(14);
        apply_twA(v, tw);
        // ---- pass III: over k0 -> y[t + 256*m2], m2 = register ---------------------------
        dft16(v);

        // ---- last radix-2 stage across the lane halves + epilogue ------------------------
        // after the swap: lanes 0-31 hold (E,O)[t+256*(8+i)], lanes 32-63 (E,O)[t+256*i].
        // Register rows are retired two at a time: combine, |.|^2, argmax, 16-B stores; the
        // freed registers immediately receive the NEXT row's needle samples, which land
        // under the remaining stores, the argmax reduction and the next row's phasor loads.
        // per-lane running maxima over increasing lag index: strict '>' keeps the first
        T bv_lo = T(0), bv_hi = T(0);
        int bi_lo = 0, bi_hi = 0;
        T *const out = A.surface ? A.surface + (size_t)g * F_L : nullptr;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? F_L * (int)sizeof(T) : 0, 0x00020000);
        const int gc = gn < A.total ? gn : A.total - 1;  // clamped: a[] is always redefined
        const __amdgpu_buffer_rsrc_t rs_sig =
            __builtin_amdgcn_make_buffer_rsrc((void *)(A.sig + (size_t)(gc / A.rows) * F_N), 0, F_N * (int)sizeof(C), 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T mlo[2], mhi[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = 2 * j + u;
                C o = v[i], e = v[8 + i];
                swap32(e, o);
                const C w32 = {(T)W32_COS[i], (T)W32_SIN[i]};  // W_32^i
                const C z = cmul(cmul(o, tbase), w32);
                mlo[u] = norm_sqr(e + z);  // mod.rs:147
                mhi[u] = norm_sqr(e - z);
                if (mlo[u] > bv_lo) { bv_lo = mlo[u]; bi_lo = i; }
                if (mhi[u] > bv_hi) { bv_hi = mhi[u]; bi_hi = i; }
                a[i] = bload(rs_sig, voff_sig, (unsigned)(256 * i * sizeof(C)), (C *)nullptr);
                a[8 + i] = bload(rs_sig, voff_sig, (unsigned)(256 * (8 + i) * sizeof(C)), (C *)nullptr);
            }
            // even lane keeps register row 2j, odd lane row 2j+1; each gets the partner's value
            const T slo = dpp_xor1<T>(odd ? mlo[0] : mlo[1]);
            const T shi = dpp_xor1<T>(odd ? mhi[0] : mhi[1]);
            const int m = mpair + 256 * (2 * j + (odd ? 1 : 0));
            store_pair_wt(rs, (unsigned)(m * sizeof(T)), odd ? slo : mlo[0], odd ? mlo[1] : slo);
            store_pair_wt(rs, (unsigned)((m + F_N) * sizeof(T)), odd ? shi : mhi[0], odd ? mhi[1] : shi);
        }
        CAF_STAMP(15);
        // lags m (lo part) all precede lags m + 4096 (hi part): init (0.0, lag 0) like mod.rs:143
        T bv = bv_lo;
        uint32_t bi = bv_lo > T(0) ? (uint32_t)(mbase + 256 * bi_lo) : 0u;
        if (bv_hi > bv) { bv = bv_hi; bi = (uint32_t)(mbase + 256 * bi_hi + F_N); }
        wave_arg_reduce_dpp(bv, bi);
        {
            T *sv = reinterpret_cast<T *>(scratch + parity * 128);
            uint32_t *si = reinterpret_cast<uint32_t *>(scratch + parity * 128 + 64);
            if (L.lane == 63) { sv[L.wave] = bv; si[L.wave] = bi; }
        }
        CAF_STAMP(16);
        if constexpr (DIAG) {
            if (blockIdx.x == 0 && L.lane == 0 && iter < 32) {
#pragma unroll
                for (int i = 0; i < F_NSTAMP; ++i) A.dbg[((size_t)iter * 8 + L.wave) * F_NSTAMP + i] = st[i];
            }
        }
        prev_g = g;
        g = gn;
    }
    // last row's result
    __syncthreads();
    if (L.tid == 0 && prev_g >= 0) {
        const T *sv = reinterpret_cast<const T *>(scratch + (parity ^ 1) * 128);
        const uint32_t *si = reinterpret_cast<const uint32_t *>(scratch + (parity ^ 1) * 128 + 64);
        T bv = sv[0];
        uint32_t bi = si[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) arg_merge(bv, bi, sv[w], si[w]);
        A.row_idx[prev_g] = bi;
        A.row_val[prev_g] = bv;
    }
}

}  // namespace caf
