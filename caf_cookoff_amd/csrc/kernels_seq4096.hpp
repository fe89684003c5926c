// kernels_seq4096.hpp -- "sequential-chain" Doppler-row kernel for n = 4096 (L = 8192).
//
// Same mathematics and the same LDS exchange geometry as kernels_fused4096.hpp, but one
// 256-thread workgroup (4 waves, one per SIMD) owns a row and runs the EVEN-bin chain and
// then the ODD-bin chain in the same threads:
//   * LDS per workgroup = ONE padded chain (4352 complex) + the 256-entry W_256 table
//     = 72.3 KiB (f64) / 36.2 KiB (f32), so TWO (f64) or THREE (f32, 168 VGPRs) workgroups are
//     resident per CU.  They share nothing and hit their barriers independently: while one
//     sits in an LDS exchange or waits on L2, the other has the VALU.  In the 512-thread
//     kernel all 8 waves are in the same phase and the two pipes run serially (profiles/r01_*).
//   * both chain outputs E[m], O[m] of a thread are in its own registers, so the last
//     radix-2 stage needs no cross-lane traffic at all.
//   * register budget: 64 (v) + 64 (E stash) + twiddles.  Only W^(t), W^(2t), W^(3t), W^(4t),
//     W^(8t), W^(12t) are held (24 VGPRs); the other nine W^(t(4a+b)) = W^(4at) W^(bt) cost one
//     extra complex multiply each per use (+4 % VALU).
//   * lane-wide common factors (the mixer's w^t, the odd chain's T^t) ride on the stage
//     twiddles (TwFold); compile-time twiddles are folded into 6-FMA butterflies (bfly_w).
//   * input loads are software-pipelined (PF bits below), LDS writes of each exchange are
//     issued group by group from the last butterfly stage (dft16_sink), rows are handed out
//     by a device-scope ticket counter (static stride for launches of <= 4 rows/workgroup).
//   * 5 workgroup barriers per row (2 per chain + the argmax publication, which also re-aligns
//     the waves); the other four exchanges are wave-local.
#pragma once
#include "kernels_fused4096.hpp"

namespace caf {

constexpr int S_THREADS = 256;

template <typename T>
constexpr size_t seq_lds_bytes() { return (F_CHAIN + 256) * sizeof(cpx<T>) + 128; }  // +128: two argmax slots, next-row word

struct SeqLane {
    int tid, lane, wave, t, hi4, lo4, pA, pB, pC;
    __device__ __forceinline__ SeqLane()
    {
        tid = threadIdx.x;
        lane = tid & 63;
        wave = tid >> 6;
        t = tid;
        hi4 = t >> 4;
        lo4 = t & 15;
        pA = t + hi4;
        pB = hi4 * F_BLK + lo4;
        pC = hi4 * F_BLK + 17 * lo4;
    }
};

// One chain of one row: needle samples -> y[m2] = IDFT_4096(C_chain)[t + 256*m2].
// 2 workgroup barriers; exchanges 2 and 3 are wave-local.
template <typename T>
__device__ __forceinline__ void keep(cpx<T> &x)
{
    asm volatile("" : "+v"(x.x), "+v"(x.y));
}

// PF (software pipelining of the input loads; each bit moves one group of loads earlier; the product rows run
// PF = 15, the one-launch surface kernel 14):
//   1: even chain - haystack-spectrum loads issued right after the mixer
//   2: odd chain  - needle samples loaded during the even chain's last pass
//   4: odd chain  - haystack-spectrum loads issued before pass 3
//   8: even chain - the NEXT row's needle samples loaded while the epilogue retires registers
template <typename T>
__device__ __forceinline__ void load_samples(cpx<T> (&a)[16], const __amdgpu_buffer_rsrc_t rs_sig, const SeqLane &L)
{
    using C = cpx<T>;
#pragma unroll
    for (int q = 0; q < 16; ++q)
        a[q] = bload(rs_sig, (unsigned)(L.t * sizeof(C)), (unsigned)(256 * q * sizeof(C)), (C *)nullptr);
}

// MEMORY POLICY of the sequential-chain kernels: every LDS access, barrier, global load and surface store a row makes
// goes through these hooks.  The product instantiates SeqIo<T> and nothing else; the measurement library instantiates
// the SAME kernel bodies over policies that leave some of the accesses out (measure/kernels_ablate.hpp: VALU only,
// no LDS, no loads, no stores -- wrong results, timing only) for the issue ceiling bench.py reports.
template <typename T>
struct SeqIo {
    using C = cpx<T>;
    C *Lc;           // the workgroup's padded chain
    const C *twB;    // LDS table W_256^(lo4 k) at [16 k]
    const SeqLane &L;
    // exchange patterns (kernels_fused4096.hpp, "LDS geometry"): A by column, B gather, C transposed
    __device__ __forceinline__ void sinkA(int k, C x) const { Lc[L.pA + k * F_BLK] = x; }
    __device__ __forceinline__ void sinkB(int k, C x) const { Lc[L.pB + 17 * k] = x; }
    __device__ __forceinline__ void sinkC(int k, C x) const { Lc[L.pC + k] = x; }
    __device__ __forceinline__ void readA(C (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pA + k * F_BLK];
    }
    __device__ __forceinline__ void readB(C (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pB + 17 * k];
    }
    __device__ __forceinline__ void readC(C (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pC + k];
    }
    __device__ __forceinline__ C twb(int k) const { return twB[16 * k]; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ void fence() const { wave_lds_fence(); }
    // global memory: needle samples, haystack-spectrum values, surface stores
    __device__ __forceinline__ void samples(C (&a)[16], const __amdgpu_buffer_rsrc_t rs_sig) const { load_samples(a, rs_sig, L); }
    __device__ __forceinline__ C sample(const __amdgpu_buffer_rsrc_t rs_sig, int i) const
    {
        return bload(rs_sig, (unsigned)(L.t * sizeof(C)), (unsigned)(256 * i * sizeof(C)), (C *)nullptr);
    }
    __device__ __forceinline__ C spec(const __amdgpu_buffer_rsrc_t rs_spec, unsigned voff, int k) const
    {
        return bload(rs_spec, voff, (unsigned)(256 * k * sizeof(C)), (C *)nullptr);
    }
    template <typename V>
    __device__ __forceinline__ void store(const __amdgpu_buffer_rsrc_t rs, unsigned byte_off, V d) const
    {
        store_vec_aux<CAF_AUX_SC1>(rs, byte_off, d);  // 16-byte write-through
    }
};

// HW: called once before the haystack-spectrum values are requested (k_seq_surface waits there for the
// workgroups that compute them; a no-op everywhere else)
struct SeqNoWait {
    __device__ __forceinline__ void operator()() const {}
};
template <typename T, int CH, int PF, typename IO, typename HW = SeqNoWait>
__device__ __forceinline__ void seq_chain(cpx<T> (&v)[16], cpx<T> (&a)[16], const IO &io, const __amdgpu_buffer_rsrc_t rs_sig,
                                          const __amdgpu_buffer_rsrc_t rs_spec, const cpx<T> pb, const cpx<T> post,
                                          const cpx<T> *__restrict__ ps, const TwSet<T> &tw, const SeqLane &L, HW hwait = HW{})
{
    using C = cpx<T>;
    const unsigned voff_spec = (unsigned)((CH * 4096 + L.t) * sizeof(C));
    constexpr bool A_PRELOADED = (CH == 0 && (PF & 8)) || (CH == 1 && (PF & 2));
    constexpr bool H_EARLY = (CH == 0 && (PF & 1));
    constexpr bool H_MID = (CH == 1 && (PF & 4));
    C h[16];
    // ---- mixer (mod.rs:46-65) fused into the first butterfly's operands -----------------
    if constexpr (!A_PRELOADED) io.samples(a, rs_sig);
#pragma unroll
    // conj(a * w^t * step[q]) = conj(a * step[q]) * conj(w^t): the lane factor conj(w^t)
    // commutes with the first butterfly and is folded into its output twiddles (TwFold)
    for (int q = 0; q < 16; ++q) v[q] = cmul_conj(a[q], ps[q]);
    const TwFold<T> fmix(tw, conj(pb));
    if constexpr (H_EARLY) {
        hwait();
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = io.spec(rs_spec, voff_spec, k);
    }
    // ---- forward (DIF) ------------------------------------------------------------------
    dft16_sink(v, [&](int k, C x) { io.sinkA(k, twA_k(x, k, tw, fmix)); });
    io.sync();
    io.readB(v);
    dft16_sink(v, [&](int k, C x) { io.sinkB(k, k ? cmul(x, io.twb(k)) : x); });
    io.fence();
    io.readC(v);
    if constexpr (H_MID) {
        hwait();
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = io.spec(rs_spec, voff_spec, k);
    }
    dft16(v);
    if constexpr (!(H_EARLY || H_MID)) hwait();
    // ---- spectrum product (xcor_rustfft.rs:64-73) ------------------------------------------
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        C hk;
        if constexpr (H_EARLY || H_MID) hk = h[k];
        else hk = io.spec(rs_spec, voff_spec, k);
        v[k] = cmul(v[k], hk);
    }
    // ---- inverse (DIT) ----------------------------------------------------------------------
    io.fence();
    dft16_sink(v, [&](int k, C x) { io.sinkC(k, x); });
    io.fence();
    io.readB(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], io.twb(k));
    io.fence();
    dft16_sink(v, [&](int k, C x) { io.sinkB(k, x); });
    io.sync();
    io.readA(v);
    // No barrier here: the next exchange-1 write (pattern A) of this thread overwrites exactly
    // the sixteen addresses it has just read, and nobody else touches them before the barrier
    // that follows that write.
    // the odd chain reads the same needle samples: fetch them under the last butterfly
    if constexpr (CH == 0 && (PF & 2)) io.samples(a, rs_sig);
    if constexpr (CH == 1) {  // odd chain: T^t of the last radix-2 stage folded into the twiddles
        const TwFold<T> fpost(tw, post);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = twA_k(v[k], k, tw, fpost);
    } else {
        apply_twA(v, tw);
    }
    dft16(v);
}

// ---- haystack spectrum, one 256-thread workgroup per (surface, chain) ------------------------
// Hs = FFT_8192(haystack ++ 0)/8192 = conj(IDFT(conj h))/L in the register layout the row
// kernels multiply in: spec[b][chain][k2][t].  Also zeroes the row-ticket counter of the row
// kernel that follows on the stream.
template <typename T>
__global__ __launch_bounds__(S_THREADS) void k_seq_prepare(const FusedArgs<T> A, const cpx<T> *__restrict__ phasor)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[seq_lds_bytes<T>()];
    // streaming slots: workgroups past the transforming ones only stage the needles in (one 16-byte
    // read of pinned host memory per thread and step, all in flight together) and leave
    if (blockIdx.x >= A.fft_blocks) {
        const unsigned nthr = (gridDim.x - A.fft_blocks) * S_THREADS;
        for (unsigned i = (blockIdx.x - A.fft_blocks) * S_THREADS + threadIdx.x; i < A.stage_n16; i += nthr)
            A.stage_dst[i] = A.stage_src[i];
        return;
    }
    C *const Lc = reinterpret_cast<C *>(smem);
    C *const twb = Lc + F_CHAIN;
    const SeqLane L;
    TwSet<T> tw;
    tw.w1 = A.tab.tw4096[L.t * 1];
    tw.w2 = A.tab.tw4096[L.t * 2];
    tw.w3 = A.tab.tw4096[L.t * 3];
    tw.w4 = A.tab.tw4096[L.t * 4];
    tw.w8 = A.tab.tw4096[L.t * 8];
    tw.w12 = A.tab.tw4096[L.t * 12];
    twb[L.tid] = A.tab.tw4096[16 * (L.tid & 15) * (L.tid >> 4)];
    const C *const twB = twb + L.lo4;
    if (blockIdx.x == 0 && L.tid == 0 && A.work) *A.work = 0u;
    const C *__restrict__ ph = phasor + (size_t)A.rows * 64;  // the f = 0 row
    const T inv = T(1.0 / 8192.0);
    __syncthreads();
    for (int w = blockIdx.x; w < 2 * A.total; w += (int)A.fft_blocks) {
        const int b = w >> 1, chain = w & 1;
        const C *sig = A.sig + (size_t)b * F_N;
        C pb = cmul(ph[L.lo4], ph[16 + L.hi4]);
        if (chain) pb = cmulc(pb, A.tab.th[L.t]);
        const C *ps = ph + 32 + 16 * chain;
        C v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = conj(cmul(cmul(sig[L.t + 256 * q], pb), ps[q]));
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
        C *spec = A.spec + (size_t)b * (2 * 16 * 256) + chain * (16 * 256);
#pragma unroll
        for (int k = 0; k < 16; ++k) spec[k * 256 + L.t] = {v[k].x * inv, -v[k].y * inv};
        __syncthreads();  // the next iteration's pattern-A writes vs this one's pattern-B/C reads
    }
}

// W_32^m2, m2 < 16
__device__ constexpr double W32C16[16] = {1.0, 0.98078528040323044912618223613424, 0.92387953251128675612818318939679,
                                          0.83146961230254523707878837761791, 0.70710678118654752440084436210485,
                                          0.55557023301960222474283081394853, 0.38268343236508977172845998403040,
                                          0.19509032201612826784828486847702, 0.0, -0.19509032201612826784828486847702,
                                          -0.38268343236508977172845998403040, -0.55557023301960222474283081394853,
                                          -0.70710678118654752440084436210485, -0.83146961230254523707878837761791,
                                          -0.92387953251128675612818318939679, -0.98078528040323044912618223613424};
__device__ constexpr double W32S16[16] = {0.0, 0.19509032201612826784828486847702, 0.38268343236508977172845998403040,
                                          0.55557023301960222474283081394853, 0.70710678118654752440084436210485,
                                          0.83146961230254523707878837761791, 0.92387953251128675612818318939679,
                                          0.98078528040323044912618223613424, 1.0, 0.98078528040323044912618223613424,
                                          0.92387953251128675612818318939679, 0.83146961230254523707878837761791,
                                          0.70710678118654752440084436210485, 0.55557023301960222474283081394853,
                                          0.38268343236508977172845998403040, 0.19509032201612826784828486847702};

// lo = e + w z, hi = e - w z for the compile-time constant w = W_32^i = c + i s, in 6 FMAs:
// w z = c (z + i tau z), tau = s/c   (|c| >= |s|)   or   s (kappa z + i z), kappa = c/s.
template <typename T>
__device__ __forceinline__ void axpy_w32(int i, cpx<T> e, cpx<T> z, cpx<T> &lo, cpx<T> &hi)
{
    bfly_w(e, z, W32C16[i], W32S16[i], lo, hi);
}

// waves per SIMD the register allocator must leave room for: f64 rows need 256 VGPRs (2; LDS
// allows no more anyway); the packed-f32 rows fit 168 without spills (3 workgroups of 36 KiB LDS
// per CU: 108.6 k surfaces/s vs 103.5 k at 2 x 216 VGPRs and 102.8 k at 4 x 128 with 31 spills)
template <typename T>
constexpr int seq_waves_per_simd() { return sizeof(T) == 8 ? 2 : 3; }

// `phasor` is passed as its own __restrict__ parameter (not inside FusedArgs) so that the
// wave-uniform step entries become scalar loads: they cost no vector-memory issue slot and,
// unlike vector loads, are not ordered behind the epilogue's stores by the in-order vmcnt.
// IO: the memory policy (SeqIo<T> in the product; see there).
template <typename T, int PF = 15, typename IO = SeqIo<T>>
__global__ __launch_bounds__(S_THREADS, seq_waves_per_simd<T>()) void k_seq_rows(const FusedArgs<T> A,
                                                                             const cpx<T> *__restrict__ phasor)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[seq_lds_bytes<T>()];
    C *const Lc = reinterpret_cast<C *>(smem);
    C *const twb = Lc + F_CHAIN;  // twb[k*16 + lo] = W_256^(lo*k)
    unsigned char *const scratch = smem + (F_CHAIN + 256) * sizeof(C);
    const SeqLane L;

    TwSet<T> tw;
    tw.w1 = A.tab.tw4096[L.t * 1];
    tw.w2 = A.tab.tw4096[L.t * 2];
    tw.w3 = A.tab.tw4096[L.t * 3];
    tw.w4 = A.tab.tw4096[L.t * 4];
    tw.w8 = A.tab.tw4096[L.t * 8];
    tw.w12 = A.tab.tw4096[L.t * 12];
    twb[L.tid] = A.tab.tw4096[16 * (L.tid & 15) * (L.tid >> 4)];
    const IO io{Lc, twb + L.lo4, L};
    const C th = A.tab.th[L.t];       // T^t = e^{2*pi*i*t/8192}
    const C cfac = conj(th);          // odd chain input rotation e^{-2*pi*i*t/8192}
    const int mpair = L.t & ~1;
    const bool odd = L.lane & 1;
    constexpr unsigned long long EVEN_LANES = 0x5555555555555555ull;
    __syncthreads();

    C a[16];
    if constexpr (PF & 8) {
        const int gc = (int)blockIdx.x < A.total ? (int)blockIdx.x : A.total - 1;
        io.samples(a, __builtin_amdgcn_make_buffer_rsrc((void *)(A.sig + (size_t)(gc / A.rows) * F_N), 0,
                                                        F_N * (int)sizeof(C), 0x00020000));
    }
    // Rows are handed out dynamically: workgroup i starts with row i, then draws tickets
    // gridDim + 0, 1, 2 ... from one device-scope counter (zeroed by the prepare kernel that
    // precedes this launch on the stream).  Identical workgroups finish 25 identical rows up
    // to 30 % apart (per-CU clock/L2 effects, profiles/r01_v3 notes); with tickets they all end
    // within one row time.  Tickets are in surface-major order, so at any moment the whole chip
    // works on 1-2 surfaces and their inputs stay L2-resident, as with static striding.
    volatile int *const next_row = reinterpret_cast<volatile int *>(scratch + 112);
    // phasor base w^t of the first row (later rows: fetched in the previous row's epilogue)
    C pb;
    {
        const int g0 = (int)blockIdx.x < A.total ? (int)blockIdx.x : A.total - 1;
        const C *ph0 = phasor + (size_t)(g0 % A.rows) * 64;
        pb = cmul(ph0[L.lo4], ph0[16 + L.hi4]);
    }
    for (int g = blockIdx.x; g < A.total;) {
        if (L.tid == 0)
            *next_row = A.work ? (int)gridDim.x + (int)atomicAdd(A.work, 1u) : g + (int)gridDim.x;
        const int b = g / A.rows, r = g - b * A.rows;
        const C *__restrict__ ph = phasor + (size_t)r * 64;
        const __amdgpu_buffer_rsrc_t rs_sig =
            __builtin_amdgcn_make_buffer_rsrc((void *)(A.sig + (size_t)b * F_N), 0, F_N * (int)sizeof(C), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.spec + (size_t)b * (2 * 16 * 256)), 0, 2 * 16 * 256 * (int)sizeof(C), 0x00020000);
        C e[16], o[16];
        seq_chain<T, 0, PF>(e, a, io, rs_sig, rs_spec, pb, th, ph + 32, tw, L);
        // the ticket was stored before the chain's barriers: visible to every wave by now
        const int gn = __builtin_amdgcn_readfirstlane(*next_row);
        const int gc = gn < A.total ? gn : A.total - 1;  // clamped: a[] is always redefined
        const __amdgpu_buffer_rsrc_t rs_sig_next = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.sig + (size_t)(gc / A.rows) * F_N), 0, F_N * (int)sizeof(C), 0x00020000);
        seq_chain<T, 1, PF>(o, a, io, rs_sig, rs_spec, cmul(pb, cfac), th, ph + 48, tw, L);

        // ---- last radix-2 stage (in registers) + |.|^2 + argmax + 16-B write-through stores --
        T bv_lo = T(0), bv_hi = T(0);
        int bi_lo = 0, bi_hi = 0;
        T *const out = A.surface ? A.surface + (size_t)g * F_L : nullptr;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? F_L * (int)sizeof(T) : 0, 0x00020000);
        // All magnitudes first; each retired (e[i], o[i]) pair frees the registers that receive the
        // next row's needle sample.  Every load of the next row is issued BEFORE the first store:
        // vmcnt retires in order, so a load issued after the 16 write-through stores could not be
        // consumed until those stores had reached memory.
        T mlo[16], mhi[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // m = t + 256*i
            C lo, hi;  // E[m] +- W_32^i * (T^t O[m]); T^t came folded into the odd chain's last twiddles
            axpy_w32(i, e[i], o[i], lo, hi);
            mlo[i] = norm_sqr(lo);  // mod.rs:147
            mhi[i] = norm_sqr(hi);
            bi_lo = mlo[i] > bv_lo ? i : bi_lo;  // first strictly greater (mod.rs:148-151)
            bv_lo = vmax(bv_lo, mlo[i]);         // one v_max instead of a 64-bit select
            bi_hi = mhi[i] > bv_hi ? i : bi_hi;
            bv_hi = vmax(bv_hi, mhi[i]);
            if constexpr (PF & 8) a[i] = io.sample(rs_sig_next, i);
        }
        {   // phasor base of the next row
            const C *phn = phasor + (size_t)(gc % A.rows) * 64;
            pb = cmul(phn[L.lo4], phn[16 + L.hi4]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            typename pair_vec<T>::type dlo, dhi;
            pair_xor1(mlo[2 * j], mlo[2 * j + 1], EVEN_LANES, ~EVEN_LANES, dlo);
            pair_xor1(mhi[2 * j], mhi[2 * j + 1], EVEN_LANES, ~EVEN_LANES, dhi);
            const int m = mpair + 256 * (2 * j + (odd ? 1 : 0));
            io.store(rs, (unsigned)(m * sizeof(T)), dlo);
            io.store(rs, (unsigned)((m + F_N) * sizeof(T)), dhi);
        }
        T bv = bv_lo;
        uint32_t bi = bv_lo > T(0) ? (uint32_t)(L.t + 256 * bi_lo) : 0u;
        if (bv_hi > bv) { bv = bv_hi; bi = (uint32_t)(L.t + 256 * bi_hi + F_N); }
        wave_arg_reduce_maxmin(bv, bi);
        T *const sv = reinterpret_cast<T *>(scratch);
        uint32_t *const si = reinterpret_cast<uint32_t *>(scratch + 32);
        if (L.lane == 63) { sv[L.wave] = bv; si[L.wave] = bi; }
        __syncthreads();  // the argmax publication: one more barrier per row, which also re-aligns the four waves
        if (L.tid == 0) {
            bv = sv[0];
            bi = si[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) arg_merge(bv, bi, sv[w], si[w]);
            A.row_idx[g] = bi;
            A.row_val[g] = bv;
        }
        g = gn;
    }
}

}  // namespace caf
