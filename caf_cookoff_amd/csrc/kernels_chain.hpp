// kernels_chain.hpp -- size-generic LDS-resident Doppler-row kernel ("chain" path).
//
// Same mathematics as kernels_seq4096.hpp (mod.rs:121-166 with FFT(haystack) hoisted and every
// transform a positive-exponent one), for any padded length L = 2n = R * M that splits into
// R in {2, 4, 8, 16} "chains" of M = 2^LOGM points, M * sizeof(complex) <= ~128 KiB so ONE chain lives
// in a workgroup's LDS:
//
//   forward, decimation in frequency over the first radix-R stage: bins k = R k' + r,
//     G[R k' + r] = IDFT_M(u_r)[k'],  u_r[n'] = W_L^(n' r) * sum_{j < R/2} conj(s[n' + M j]) W_R^(j r)
//     (the upper half of s = needle * w^n ++ 0 is the zero padding of mod.rs:130, so only R/2
//     of the R terms exist: R = 2 -> one, R = 4 -> two, R = 8 -> four);
//   product with the pre-permuted haystack spectrum Hs = FFT_L(haystack ++ 0)/L in registers
//     (xcor_rustfft.rs:64-73);
//   inverse, decimation in time: y_r = IDFT_M(Hs G restricted to chain r), natural order, and the
//     last radix-R stage  c[m' + M j] = sum_r W_R^(j r) W_L^(m' r) y_r[m']  in registers;
//   |.|^2 (mod.rs:147), first-max argmax (mod.rs:143-151), surface store.
//
// One workgroup of W = M/16 threads owns a row; a thread holds 16 points.  An M-point transform is
// NS = LOGM/4 radix-16 stages plus one radix-2/4/8 stage when LOGM is not a multiple of 4, done
// IN PLACE in the LDS chain: stage s works on blocks of B_s = M >> 4s elements with stride
// S_s = B_s/16; thread t owns elements (t / S_s) B_s + (t % S_s) + j S_s.  The forward is DIF
// (natural in, digit-reversed out), the inverse the mirrored DIT (digit-reversed in, natural out),
// so no reordering pass exists and Hs is stored by the same forward code in the register layout
// the rows multiply in.  Element e sits at LDS position e + (e >> 4) (one pad per 16): every
// access of every stage is then "per-thread base + compile-time offset" and at most 2-way
// bank-conflicted.  The exchange between stages s and s+1 stays inside one wave when
// B_s <= 1024 (the wave's 64 x 16 elements are whole blocks): only the first exchange of a
// transform needs a workgroup barrier.
//
// R = 2: both chain outputs stay in registers (as in kernels_seq4096.hpp).  R = 4 (n = 32768
// complex64 = BASELINE configs[3]; n = 16384 complex128): the last radix-4 stage needs all four
// chain outputs of a lag, 64 complex per thread -- more than the register file holds at this
// occupancy -- so a = y0 + W y2 and b = y0 - W y2 go to a per-workgroup scratch slab in global
// memory (32 complex per thread, written and read back by the SAME thread: no synchronisation,
// served by L2 / the Infinity Cache) while chains 1 and 3 run.  Traffic per row: the surface once +
// that slab once each way, instead of three passes over a work row (kernels_big65536.hpp).
// R = 8 (n = 65536 complex64, n = 32768 complex128): the same with chain pairs (r', r'+4) and two
// radix-4 combinations at the end; six slab arrays.
// R = 16 (n = 131072 complex64, n = 65536 complex128): all sixteen chain outputs go to the workgroup's slab
// (2 MiB per workgroup) and the SAME workgroup then streams them back lag by lag through one radix-16
// butterfly -- a compute phase and a memory phase per row, which different workgroups run at different times.
#pragma once
#include "kernels_seq4096.hpp"

namespace caf {

template <int LOGM>
struct ChainGeo {
    static constexpr int M = 1 << LOGM;
    static constexpr int W = M / 16;                 // threads per workgroup
    static constexpr int NS = LOGM / 4;              // radix-16 stages
    static constexpr int RL = 1 << (LOGM % 4);       // last small radix (1: none)
    static constexpr int NST = NS + (RL > 1 ? 1 : 0);
    static constexpr int CHAIN = M + M / 16;         // padded chain, elements
    static constexpr int blk(int s) { return M >> (4 * s); }
    static constexpr int str(int s) { return (M >> (4 * s)) >> 4; }
    // exchange after stage s (radix-16 stage index) is wave-local?
    static constexpr bool local_after(int s) { return (M >> (4 * s)) <= 1024 || W <= 64; }
    static constexpr int off(int s, int j) { return j * str(s) + ((j * str(s)) >> 4); }
};

// LDS bytes of one workgroup: padded chain + stage-1/2 twiddle tables + argmax scratch
constexpr size_t chain_lds_bytes_v(int logm, size_t csize)
{
    const size_t M = (size_t)1 << logm;
    return (M + M / 16 + M / 16 + M / 256 + 16) * csize + 256;
}
// waves per SIMD the register allocator must leave room for (csize = sizeof(complex))
constexpr int chain_wps_v(int logm, size_t csize, int nb = 1)
{
    const int W = (1 << logm) / 16 / nb;  // threads per workgroup
    const int waves_wg = (W + 63) / 64;
    const int by_lds = (int)(160 * 1024 / chain_lds_bytes_v(logm, csize));
    const int per_simd = (waves_wg * (by_lds < 1 ? 1 : by_lds) + 3) / 4;  // what LDS lets reside
    const int cap = (csize == 16 || nb > 1) ? 2 : 4;                      // 256 / 128 VGPRs
    const int floor_ = (waves_wg + 3) / 4;                                // one workgroup must fit
    return per_simd < cap ? (per_simd < floor_ ? floor_ : per_simd) : (cap < floor_ ? floor_ : cap);
}
// resident workgroups per CU
constexpr size_t chain_wg_per_cu_v(int logm, size_t csize, int nb = 1)
{
    const size_t W = ((size_t)1 << logm) / 16 / nb, waves_wg = (W + 63) / 64;
    size_t per_cu = 160 * 1024 / chain_lds_bytes_v(logm, csize);
    if (per_cu * waves_wg > (size_t)chain_wps_v(logm, csize, nb) * 4) per_cu = (size_t)chain_wps_v(logm, csize, nb) * 4 / waves_wg;
    return per_cu < 1 ? 1 : per_cu;
}
// scratch-slab arrays per workgroup: R = 4: a, b;  R = 8: P0/aP, Q0/bP, aQ, bQ, P1, Q1
constexpr int chain_slab_arrays_v(int R) { return R == 16 ? 16 : R == 8 ? 6 : 2; }  // R = 16: every chain output
// butterflies per thread of the row kernel (see ChainLane: two was measured and lost)
constexpr int chain_nb_v(int, size_t) { return 1; }
template <typename T, int LOGM>
constexpr size_t chain_lds_bytes() { return chain_lds_bytes_v(LOGM, sizeof(cpx<T>)); }
template <typename T, int LOGM>
constexpr int chain_waves_per_simd() { return chain_wps_v(LOGM, sizeof(cpx<T>)); }

// Row-independent tables of a chain plan, built once per (LOGM, R, dtype) in the context:
//   twM[m] = e^{2 pi i m / M}, m < M;   th[(r-1) W + t] = e^{2 pi i t r / (R M)}, t < W, r = 1..R-1
template <typename T>
__global__ void k_chain_tables(cpx<T> *__restrict__ twM, cpx<T> *__restrict__ th, int M, int R)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int W = M / 16;
    if (i < M) twM[i] = cispi_f64<T>(2.0 * (double)i / (double)M);
    if (i < (R - 1) * W) {
        const int r = i / W + 1, t = i % W;
        th[i] = cispi_f64<T>(2.0 * (double)t * (double)r / ((double)R * (double)M));
    }
}

// Per-row phasor table, chain_ph_v(R) entries per row (row `nrows` = the f = 0 row for the haystack):
//   [0..15] w^j   [16..31] w^(16 j)   [32..47] w^(256 j)
//   [48 + 16 r + q] step_r[q] = w^(W q) * e^{-2 pi i q r / (16 R)}      (r < R, q < 16)
//   R = 4: [112] w^M                              (second half of the needle)
//   R = 8: [176 + 3 r' + (j-1)] kappa_{r',j} = w^(M j) * e^{-2 pi i j r' / 8}, r' < 4, j = 1..3
//          (quarters 1..3 of the needle, with the pruned radix-8 input stage's constants folded in)
//   R = 16: [304 + 7 r' + (j-1)] kappa_{r',j} = w^(M j) * e^{-2 pi i j r' / 16}, r' < 8, j = 1..7
// every entry from one f64 sincos of the exact phase product (never an f32
// recurrence), w = e^{j ph}, ph = ((2 PI) f)(1/fs) as mod.rs:54-56.
constexpr int chain_ph_v(int R) { return R == 16 ? 512 : R == 8 ? 256 : 128; }
template <typename T>
__global__ void k_chain_phasors(const double *__restrict__ ph, int nrows, int M, int R, cpx<T> *__restrict__ tab)
{
    const int PH = chain_ph_v(R);
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g / PH, e = g % PH;
    if (row > nrows) return;
    const double p = row < nrows ? ph[row] : 0.0;
    const int W = M / 16;
    double mult = 0.0, rot = 0.0;  // phase p*mult, extra rotation e^{-2 pi i rot}
    if (e < 16) mult = (double)e;
    else if (e < 32) mult = 16.0 * (double)(e - 16);
    else if (e < 48) mult = 256.0 * (double)(e - 32);
    else if (e < 48 + 16 * R) {
        const int r = (e - 48) >> 4, q = (e - 48) & 15;
        mult = (double)W * (double)q;
        rot = (double)(q * r) / (16.0 * (double)R);
    } else if (R == 4 && e == 112) mult = (double)M;
    else if (R == 8 && e >= 176 && e < 188) {
        const int rp = (e - 176) / 3, j = (e - 176) % 3 + 1;
        mult = (double)M * (double)j;
        rot = (double)(j * rp) / 8.0;
    } else if (R == 16 && e >= 304 && e < 360) {
        const int rp = (e - 304) / 7, j = (e - 304) % 7 + 1;
        mult = (double)M * (double)j;
        rot = (double)(j * rp) / 16.0;
    }
    double s, c, s2, c2;
    sincos(p * mult, &s, &c);
    sincospi(-2.0 * rot, &s2, &c2);
    tab[(size_t)row * PH + e] = {(T)(c * c2 - s * s2), (T)(c * s2 + s * c2)};
}

template <typename T>
struct ChainArgs {
    const cpx<T> *sig;      // prepare: haystack [batch][n]; rows: needle [batch][n]
    cpx<T> *spec;           // Hs [batch][R][8 register pairs][W][2]: prepare writes, rows read (one 16/32-byte access per pair)
    const cpx<T> *twM;      // [M]
    const cpx<T> *th;       // [(R-1) W]
    T *surface;             // [batch][rows][L] or nullptr
    uint64_t *row_idx;      // [batch][rows]
    T *row_val;             // [batch][rows]
    cpx<T> *slab;           // R >= 4: [gridDim.x][chain_slab_arrays_v(R)][16][W] scratch of the last radix-R stage
    int rows;               // rows per surface handled by this plan
    int total;              // batch*rows (prepare: batch)
};

// ---- small last stages on 16 contiguous elements (natural order in and out) --------------------
template <typename T>
__device__ __forceinline__ void dft8(cpx<T> &x0, cpx<T> &x1, cpx<T> &x2, cpx<T> &x3, cpx<T> &x4, cpx<T> &x5, cpx<T> &x6,
                                     cpx<T> &x7)
{
    dft4(x0, x2, x4, x6);  // E[0..3] in x0, x2, x4, x6
    dft4(x1, x3, x5, x7);  // O[0..3] in x1, x3, x5, x7
    const cpx<T> o1 = mul_w8(x3), o2 = muli(x5), o3 = mul_w8_3(x7);
    const cpx<T> e0 = x0, e1 = x2, e2 = x4, e3 = x6, o0 = x1;
    x0 = e0 + o0; x4 = e0 - o0;
    x1 = e1 + o1; x5 = e1 - o1;
    x2 = e2 + o2; x6 = e2 - o2;
    x3 = e3 + o3; x7 = e3 - o3;
}
template <typename T, int RL>
__device__ __forceinline__ void small_stage(cpx<T> (&v)[16])
{
    if constexpr (RL == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const cpx<T> a = v[2 * i], b = v[2 * i + 1];
            v[2 * i] = a + b;
            v[2 * i + 1] = a - b;
        }
    } else if constexpr (RL == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) dft4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    } else if constexpr (RL == 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            dft8(v[8 * i], v[8 * i + 1], v[8 * i + 2], v[8 * i + 3], v[8 * i + 4], v[8 * i + 5], v[8 * i + 6], v[8 * i + 7]);
    }
}

// Per-thread geometry + tables of one workgroup.  A thread owns NB butterflies per stage
// ("slots" b = 0 .. NB-1: butterfly ids beta_b = t + b * W/NB, W = M/16).  The product uses NB = 1.
// With NB = 2 a workgroup has half the waves, each with twice the registers, and a wave's two slots
// are software-pipelined against each other through the wave-local exchanges (slot A's LDS writes
// and reads in flight while the wave computes slot B's butterfly).  Motivation: all waves of a
// workgroup run the same phase at the same time, so the LDS phases and the VALU phases of a CU do
// not overlap (n = 32768 complex64: VALU 35 % busy + LDS 31 % busy, waves parked 56 % --
// profiles/r02_c3; without any global memory the kernel still takes 0.91 of 1.50 ms).  Measured
// (parity-green, CAF_CHAIN_ABL=200 of the measurement build): NB = 2 is SLOWER, 1.60 vs 1.39 ms, and
// 1.02 vs 0.91 ms without global memory: two waves per SIMD with 116 spilled registers, and a
// 4-bit lgkmcnt cannot wait for "all but the other slot's 32 operations".
//
// Every memory access of a row -- chain elements in LDS, the barrier after a non-local exchange, needle samples,
// haystack-spectrum values, the scratch slab, surface stores -- goes through a policy class.  The product instantiates
// ChainIo<T> only (ABL = 0 in every kernel name of libcaf_hip.so); the measurement library derives ablations from it
// (measure/kernels_ablate.hpp: ChainIoCut<T, mask>, WRONG results, timing only) and instantiates the same bodies.
template <typename T>
struct ChainIo {
    using C = cpx<T>;
    static constexpr bool wg_barriers = true;  // non-local exchanges end in a workgroup barrier
    __device__ __forceinline__ static void lds_st(C *Lc, int pos, C x) { Lc[pos] = x; }
    __device__ __forceinline__ static C lds_ld(const C *Lc, int pos) { return Lc[pos]; }
    // needle sample of register row q, nonzero block j, butterfly beta (the indices only name the value)
    __device__ __forceinline__ static C sample(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int, int, int)
    {
        return bload(rs, voff, soff, (C *)nullptr);
    }
    // haystack-spectrum values k, k + 1 of a chain
    __device__ __forceinline__ static void spec2(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int, C &h0, C &h1)
    {
        bload2(rs, voff, soff, h0, h1);
    }
    // this workgroup's scratch slab, array arr, register row i
    __device__ __forceinline__ static void slab_st(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, C x)
    {
        bstore(rs, voff, soff, x);
    }
    __device__ __forceinline__ static C slab_ld(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int, int)
    {
        return bload(rs, voff, soff, (C *)nullptr);
    }
    __device__ __forceinline__ static void surf_st(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, T m)
    {
        store_one_aux<CAF_AUX_SC1>(rs, voff, soff, m);
    }
};
template <typename T, int ABL>
struct ChainIoFor;  // ABL != 0: measure/kernels_ablate.hpp
template <typename T>
struct ChainIoFor<T, 0> {
    using type = ChainIo<T>;
};

template <typename T, int LOGM, int NB = 1, int ABL = 0>
struct ChainLane {
    using G = ChainGeo<LOGM>;
    using C = cpx<T>;
    using IO = typename ChainIoFor<T, ABL>::type;
    static constexpr int TH = G::W / NB;  // threads per workgroup
    static_assert(G::W % NB == 0 && (TH % 64 == 0 || NB == 1), "slots must be whole waves apart");
    static_assert(G::NST < 3 || G::local_after(1), "exchanges after stage 1 are assumed wave-local");
    int t;
    int beta[NB];     // butterfly id of slot b
    int base[NB][3];  // LDS base position of radix-16 stages 0..2
    int o[NB][3];     // beta % S_s
    int basef[NB];    // last small stage / S = 1 stage: 17 beta
    C *Lc;            // chain
    const C *tw1;     // LDS [16][S_1]
    const C *tw2;     // LDS [16][S_2]
    TwSet<T> tw[NB];  // stage-0 twiddles W_M^(beta k), k in {1,2,3,4,8,12}

    __device__ __forceinline__ ChainLane(unsigned char *smem, const C *__restrict__ twM)
    {
        t = threadIdx.x;
        Lc = reinterpret_cast<C *>(smem);
        C *tab1 = Lc + G::CHAIN;
        C *tab2 = tab1 + G::M / 16;
        tw1 = tab1;
        tw2 = tab2;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int be = t + b * TH;
            beta[b] = be;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if (s < G::NS) {
                    const int S = G::str(s), B = G::blk(s);
                    const int g = be / S, oo = be % S;
                    o[b][s] = oo;
                    base[b][s] = g * (B + B / 16) + oo + (S >= 16 ? (oo >> 4) : 0);
                } else {
                    o[b][s] = 0;
                    base[b][s] = 0;
                }
            }
            basef[b] = 17 * be;
            tw[b].w1 = twM[be * 1];
            tw[b].w2 = twM[be * 2];
            tw[b].w3 = twM[be * 3];
            tw[b].w4 = twM[be * 4];
            tw[b].w8 = twM[be * 8];
            tw[b].w12 = twM[be * 12];
        }
        // stage tables [k][o] = W_{B_s}^(o k) = twM[o k M / B_s]
        if constexpr (G::NS >= 2 && G::str(1) > 1) {
            constexpr int S = G::str(1), B = G::blk(1);
            for (int i = t; i < 16 * S; i += TH) tab1[i] = twM[(i % S) * (i / S) * (G::M / B)];
        }
        if constexpr (G::NS >= 3 && G::str(2) > 1) {
            constexpr int S = G::str(2), B = G::blk(2);
            for (int i = t; i < 16 * S; i += TH) tab2[i] = twM[(i % S) * (i / S) * (G::M / B)];
        }
    }

    // chain element accessors
    __device__ __forceinline__ void st(int pos, C x) const { IO::lds_st(Lc, pos, x); }
    __device__ __forceinline__ C ld(int pos) const { return IO::lds_ld(Lc, pos); }
    // synchronise the exchange that follows radix-16 stage s
    template <int S>
    __device__ __forceinline__ void sync_after() const
    {
        if constexpr (G::local_after(S) || !IO::wg_barriers)
            wave_lds_fence();
        else
            __syncthreads();
    }
    template <int S>
    static constexpr bool strided() { return S < G::NS && G::str(S < G::NS ? S : 0) > 1; }
    template <int S>
    __device__ __forceinline__ void rd(C (&v)[16], int b) const
    {
        if constexpr (strided<S>()) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld(base[b][S < 3 ? S : 0] + G::off(S, j));
        } else {  // S = 1 radix-16 stage or the small last stage: 16 contiguous elements
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = ld(basef[b] + j);
        }
    }
    template <int S>
    __device__ __forceinline__ void wr(const C (&v)[16], int b) const
    {
        if constexpr (strided<S>()) {
#pragma unroll
            for (int j = 0; j < 16; ++j) st(base[b][S < 3 ? S : 0] + G::off(S, j), v[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) st(basef[b] + j, v[j]);
        }
    }
    // twiddle W_{B_S}^(o k) of radix-16 stage S >= 1 from its LDS table
    template <int S>
    __device__ __forceinline__ C twk(int k, int b) const
    {
        constexpr int St = G::str(S < G::NS ? S : 0);
        return (S == 1 ? tw1 : tw2)[k * St + o[b][S < 3 ? S : 0]];
    }
    // the last stage of the forward / first of the inverse: no twiddle, no memory
    __device__ __forceinline__ void c_last(C (&v)[16]) const
    {
        if constexpr (G::RL > 1)
            small_stage<T, G::RL>(v);
        else
            dft16(v);
    }

    // Forward DIF chain over the NB slots.  In: v[b][q] = u[beta_b + W q] WITHOUT the lane-common factor
    // lane[b], which is folded into the stage-0 output twiddles.  Out: v[b][k] in this path's layout.
    __device__ __forceinline__ void forward(C (&v)[NB][16], const C (&lane)[NB]) const
    {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const TwFold<T> f0(tw[b], lane[b]);
            dft16_sink(v[b], [&](int k, C x) { st(base[b][0] + G::off(0, k), twA_k(x, k, tw[b], f0)); });
        }
        sync_after<0>();
#pragma unroll
        for (int b = 0; b < NB; ++b) rd<1>(v[b], b);
        fwd_steps<1>(v);
    }
    // stage S of every slot; slot b's write + next read are in flight while slot b+1 computes
    template <int S>
    __device__ __forceinline__ void fwd_steps(C (&v)[NB][16]) const
    {
        if constexpr (S == G::NST - 1) {
#pragma unroll
            for (int b = 0; b < NB; ++b) c_last(v[b]);
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                dft16_sink(v[b], [&](int k, C x) { st(base[b][S < 3 ? S : 0] + G::off(S, k), k ? cmul(x, twk<S>(k, b)) : x); });
                sync_after<S>();  // wave-local (static_assert above)
                rd<S + 1>(v[b], b);
                if constexpr (NB > 1) __builtin_amdgcn_sched_barrier(0);
            }
            fwd_steps<S + 1>(v);
        }
    }

    // Inverse DIT chain (mirror).  In: v[b][k] in the forward's output layout.  Out: v[b][i] =
    // y[beta_b + W i]; the stage-0 butterfly's input twiddle is applied by the caller-supplied functor
    // tw0(b, k, x) (so a per-chain lane factor can ride on it).
    template <typename F>
    __device__ __forceinline__ void inverse(C (&v)[NB][16], F &&tw0) const
    {
        inv_steps<G::NST - 1>(v);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[b][k] = tw0(b, k, v[b][k]);
            dft16(v[b]);
        }
    }
    template <int S>
    __device__ __forceinline__ void inv_steps(C (&v)[NB][16]) const
    {
        static_assert(S >= 1, "stage 0 is finished by inverse()");
        constexpr bool local = G::local_after(S - 1) || !IO::wg_barriers;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if constexpr (S == G::NST - 1) {
                c_last(v[b]);
                wave_lds_fence();  // this thread's forward reads of the same 16 positions are done (program order)
                wr<S>(v[b], b);
            } else {
#pragma unroll
                for (int k = 1; k < 16; ++k) v[b][k] = cmul(v[b][k], twk<S>(k, b));
                wave_lds_fence();
                dft16_sink(v[b], [&](int k, C x) { st(base[b][S < 3 ? S : 0] + G::off(S, k), x); });
            }
            if constexpr (local) {
                wave_lds_fence();
                rd<S - 1>(v[b], b);
                if constexpr (NB > 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (!local) {
            __syncthreads();
#pragma unroll
            for (int b = 0; b < NB; ++b) rd<S - 1>(v[b], b);
        }
        if constexpr (S - 1 >= 1) inv_steps<S - 1>(v);
    }
};

// lane-common phasor w^beta from the three-level table of a row
template <typename T, int LOGM>
__device__ __forceinline__ cpx<T> chain_pb(const cpx<T> *__restrict__ ph, int t)
{
    cpx<T> pb = cmul(ph[t & 15], ph[16 + ((t >> 4) & 15)]);
    if constexpr (ChainGeo<LOGM>::W > 256) pb = cmul(pb, ph[32 + (t >> 8)]);
    return pb;
}

// Inputs of TWO chains from one pass over the needle samples: chains (rA, rB) = (0, 1) for R = 2,
// (0, 2) and (1, 3) for R = 4 read the same a0 (and b = wM a1):
//   v_r[q] = conj(x_r[q] * step_r[q]),  x_rA = a0 + s b, x_rB = a0 - s b, s = (-i)^rA   (R = 2: x = a0).
// The second chain's input waits in registers while the first chain runs -- the same registers that
// afterwards hold the first chain's result while the second runs -- so sharing the loads costs no
// register and halves the needle traffic (L2 -> CU).  The samples are fetched in groups, one group
// ahead of its use: hoisting all of them (what the scheduler does on its own) costs 32-64 VGPRs.
template <typename T, int LOGM, int R, int ABL = 0>
__device__ __forceinline__ void chain_input_pair(cpx<T> (&vA)[16], cpx<T> (&vB)[16], const __amdgpu_buffer_rsrc_t rs_sig,
                                                 int rA, int rB, int beta, const cpx<T> *__restrict__ ph)
{
    using C = cpx<T>;
    constexpr int W = ChainGeo<LOGM>::W, M = ChainGeo<LOGM>::M;
    constexpr int NQ = R / 2;                                   // nonzero quarters / halves of the padded needle
    constexpr int GQ = (R == 2) ? 4 : (R == 16) ? 1 : (sizeof(T) == 4 ? 2 : 4);  // register rows per fetch group (two groups in flight)
    const C *psA = ph + 48 + 16 * rA, *psB = ph + 48 + 16 * rB;
    const unsigned voff = (unsigned)(beta * sizeof(C));
    C k1 = C{T(1), T(0)}, k2 = k1, k3 = k1, k4 = k1, k5 = k1, k6 = k1, k7 = k1;
    using IO = typename ChainIoFor<T, ABL>::type;
    if constexpr (R == 4) k1 = ph[112];  // w^M
    if constexpr (R == 8) {              // kappa_{rA, 1..3}
        k1 = ph[176 + 3 * rA];
        k2 = ph[176 + 3 * rA + 1];
        k3 = ph[176 + 3 * rA + 2];
    }
    if constexpr (R == 16) {             // kappa_{rA, 1..7}
        const C *kp = ph + 304 + 7 * rA;
        k1 = kp[0]; k2 = kp[1]; k3 = kp[2]; k4 = kp[3]; k5 = kp[4]; k6 = kp[5]; k7 = kp[6];
    }
    C a[2][GQ][NQ];
    auto fetch = [&](int grp) {
#pragma unroll
        for (int u = 0; u < GQ; ++u) {
            const int q = GQ * grp + u;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                a[grp & 1][u][j] = IO::sample(rs_sig, voff, (unsigned)((M * j + W * q) * sizeof(C)), q, j, beta);
            }
        }
    };
    fetch(0);
#pragma unroll
    for (int grp = 0; grp < 16 / GQ; ++grp) {
        if (grp < 16 / GQ - 1) fetch(grp + 1);
#pragma unroll
        for (int u = 0; u < GQ; ++u) {
            const int q = GQ * grp + u;
            const C x = a[grp & 1][u][0];
            if constexpr (R == 2) {
                vA[q] = cmul_conj(x, psA[q]);
                vB[q] = cmul_conj(x, psB[q]);
            } else if constexpr (R == 4) {
                const C b = cmul(a[grp & 1][u][1], k1);  // rA = 0: x +- b;  rA = 1: x -+ i b
                vA[q] = cmul_conj(rA == 0 ? x + b : sub_i(x, b), psA[q]);
                vB[q] = cmul_conj(rA == 0 ? x - b : add_i(x, b), psB[q]);
            } else if constexpr (R == 8) {
                // x_rA = E + O, x_(rA+4) = E - O;  E = a0 + kappa_2 a2,  O = kappa_1 a1 + kappa_3 a3
                const C E = cfma(x, a[grp & 1][u][2], k2);
                const C O = cfma(cmul(a[grp & 1][u][1], k1), a[grp & 1][u][3], k3);
                vA[q] = cmul_conj(E + O, psA[q]);
                vB[q] = cmul_conj(E - O, psB[q]);
            } else {
                // R = 16: x_rA = E + O, x_(rA+8) = E - O;  E = sum of the even, O of the odd eighths kappa_j a_j
                const C(&aa)[NQ] = a[grp & 1][u];
                const C E = cfma(cfma(cfma(x, aa[2], k2), aa[4], k4), aa[6], k6);
                const C O = cfma(cfma(cfma(cmul(aa[1], k1), aa[3], k3), aa[5], k5), aa[7], k7);
                vA[q] = cmul_conj(E + O, psA[q]);
                vB[q] = cmul_conj(E - O, psB[q]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// e^{2 pi i k / 64}, k < 32 (compile-time constants of the last radix-R stage)
__device__ constexpr double W64C[32] = {
    1.0, 0.99518472667219688624, 0.98078528040323044913, 0.95694033573220886494, 0.92387953251128675613,
    0.88192126434835502971, 0.83146961230254523708, 0.77301045336273696081, 0.70710678118654752440,
    0.63439328416364549822, 0.55557023301960222474, 0.47139673682599764856, 0.38268343236508977173,
    0.29028467725446236764, 0.19509032201612826785, 0.09801714032956060199, 0.0, -0.09801714032956060199,
    -0.19509032201612826785, -0.29028467725446236764, -0.38268343236508977173, -0.47139673682599764856,
    -0.55557023301960222474, -0.63439328416364549822, -0.70710678118654752440, -0.77301045336273696081,
    -0.83146961230254523708, -0.88192126434835502971, -0.92387953251128675613, -0.95694033573220886494,
    -0.98078528040323044913, -0.99518472667219688624};
__device__ constexpr double W64S[32] = {
    0.0, 0.09801714032956060199, 0.19509032201612826785, 0.29028467725446236764, 0.38268343236508977173,
    0.47139673682599764856, 0.55557023301960222474, 0.63439328416364549822, 0.70710678118654752440,
    0.77301045336273696081, 0.83146961230254523708, 0.88192126434835502971, 0.92387953251128675613,
    0.95694033573220886494, 0.98078528040323044913, 0.99518472667219688624, 1.0, 0.99518472667219688624,
    0.98078528040323044913, 0.95694033573220886494, 0.92387953251128675613, 0.88192126434835502971,
    0.83146961230254523708, 0.77301045336273696081, 0.70710678118654752440, 0.63439328416364549822,
    0.55557023301960222474, 0.47139673682599764856, 0.38268343236508977173, 0.29028467725446236764,
    0.19509032201612826785, 0.09801714032956060199};
// e^{2 pi i k / 128}, k < 64 (last radix-8 stage: L / W = 128)
__device__ constexpr double W128C[64] = {1.00000000000000000000, 0.99879545620517240501, 0.99518472667219692873, 0.98917650996478101444, 0.98078528040323043058, 0.97003125319454397424, 0.95694033573220882438, 0.94154406518302080631, 0.92387953251128673848, 0.90398929312344333820, 0.88192126434835504956, 0.85772861000027211809, 0.83146961230254523567, 0.80320753148064494287, 0.77301045336273699338, 0.74095112535495910588, 0.70710678118654757274, 0.67155895484701833009, 0.63439328416364548779, 0.59569930449243346793, 0.55557023301960228867, 0.51410274419322166128, 0.47139673682599780857, 0.42755509343028219593, 0.38268343236508983729, 0.33688985339222005111, 0.29028467725446233105, 0.24298017990326398197, 0.19509032201612833135, 0.14673047445536174793, 0.09801714032956077016, 0.04906767432741812596, 0.0, -0.04906767432741800800, -0.09801714032956064526, -0.14673047445536163691, -0.19509032201612819257, -0.24298017990326387094, -0.29028467725446216452, -0.33688985339221994009, -0.38268343236508972627, -0.42755509343028186287, -0.47139673682599769755, -0.51410274419322166128, -0.55557023301960195560, -0.59569930449243335691, -0.63439328416364537677, -0.67155895484701844111, -0.70710678118654746172, -0.74095112535495888384, -0.77301045336273699338, -0.80320753148064483184, -0.83146961230254534669, -0.85772861000027200706, -0.88192126434835493853, -0.90398929312344333820, -0.92387953251128673848, -0.94154406518302069529, -0.95694033573220882438, -0.97003125319454397424, -0.98078528040323043058, -0.98917650996478101444, -0.99518472667219681771, -0.99879545620517240501};
__device__ constexpr double W128S[64] = {0.00000000000000000000, 0.04906767432741801493, 0.09801714032956060363, 0.14673047445536174793, 0.19509032201612824808, 0.24298017990326387094, 0.29028467725446233105, 0.33688985339222005111, 0.38268343236508978178, 0.42755509343028208491, 0.47139673682599764204, 0.51410274419322166128, 0.55557023301960217765, 0.59569930449243335691, 0.63439328416364548779, 0.67155895484701833009, 0.70710678118654746172, 0.74095112535495910588, 0.77301045336273699338, 0.80320753148064483184, 0.83146961230254523567, 0.85772861000027211809, 0.88192126434835493853, 0.90398929312344333820, 0.92387953251128673848, 0.94154406518302080631, 0.95694033573220893540, 0.97003125319454397424, 0.98078528040323043058, 0.98917650996478101444, 0.99518472667219681771, 0.99879545620517240501, 1.00000000000000000000, 0.99879545620517240501, 0.99518472667219692873, 0.98917650996478101444, 0.98078528040323043058, 0.97003125319454397424, 0.95694033573220893540, 0.94154406518302080631, 0.92387953251128673848, 0.90398929312344344922, 0.88192126434835504956, 0.85772861000027211809, 0.83146961230254545772, 0.80320753148064494287, 0.77301045336273710440, 0.74095112535495899486, 0.70710678118654757274, 0.67155895484701855214, 0.63439328416364548779, 0.59569930449243346793, 0.55557023301960217765, 0.51410274419322177231, 0.47139673682599786408, 0.42755509343028202940, 0.38268343236508989280, 0.33688985339222032867, 0.29028467725446238656, 0.24298017990326406523, 0.19509032201612860891, 0.14673047445536180344, 0.09801714032956082567, 0.04906767432741796636};


// chain input of ONE chain (the haystack transform; not on the row path): w = 1
template <typename T, int LOGM, int R>
__device__ __forceinline__ void chain_input_one(cpx<T> (&v)[16], const __amdgpu_buffer_rsrc_t rs_sig, int r, int beta,
                                                const cpx<T> *__restrict__ ph)
{
    using C = cpx<T>;
    constexpr int W = ChainGeo<LOGM>::W, M = ChainGeo<LOGM>::M;
    const C *ps = ph + 48 + 16 * r;
    const unsigned voff = (unsigned)(beta * sizeof(C));
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        C x = bload(rs_sig, voff, (unsigned)(W * q * sizeof(C)), (C *)nullptr);
        if constexpr (R == 4) {
            const C b = bload(rs_sig, voff, (unsigned)((M + W * q) * sizeof(C)), (C *)nullptr);  // w = 1: wM = 1
            x = r == 0 ? x + b : r == 1 ? sub_i(x, b) : r == 2 ? x - b : add_i(x, b);
        }
        if constexpr (R == 8) {  // the f = 0 row's kappa_{r', j} = e^{-2 pi i j r'/8}; chains r'+4 take E - O
            const int rp = r & 3;
            const C a1 = bload(rs_sig, voff, (unsigned)((M + W * q) * sizeof(C)), (C *)nullptr);
            const C a2 = bload(rs_sig, voff, (unsigned)((2 * M + W * q) * sizeof(C)), (C *)nullptr);
            const C a3 = bload(rs_sig, voff, (unsigned)((3 * M + W * q) * sizeof(C)), (C *)nullptr);
            const C E = cfma(x, a2, ph[176 + 3 * rp + 1]);
            const C O = cfma(cmul(a1, ph[176 + 3 * rp]), a3, ph[176 + 3 * rp + 2]);
            x = r < 4 ? E + O : E - O;
        }
        if constexpr (R == 16) {  // the f = 0 row's kappa_{r', j} = e^{-2 pi i j r'/16}; chains r'+8 take E - O
            const C *kp = ph + 304 + 7 * (r & 7);
            C E = x, O = C{T(0), T(0)};
#pragma unroll
            for (int j = 1; j < 8; ++j) {
                const C aj = bload(rs_sig, voff, (unsigned)((j * M + W * q) * sizeof(C)), (C *)nullptr);
                if (j & 1) O = cfma(O, aj, kp[j - 1]);
                else E = cfma(E, aj, kp[j - 1]);
            }
            x = r < 8 ? E + O : E - O;
        }
        v[q] = cmul_conj(x, ps[q]);
    }
}

// ---- haystack spectrum: one workgroup per (surface, chain) ------------------------------------
// Hs = FFT_L(haystack ++ 0)/L = conj(IDFT_L(conj h))/L, register layout spec[b][r][k][beta].
template <typename T, int LOGM, int R>
__global__ __launch_bounds__(ChainGeo<LOGM>::W) void k_chain_prepare(const ChainArgs<T> A, const cpx<T> *__restrict__ phasor)
{
    using G = ChainGeo<LOGM>;
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[chain_lds_bytes<T, LOGM>()];
    const ChainLane<T, LOGM, 1> L(smem, A.twM);
    constexpr int NS_IN = R * G::M / 2;  // samples per input
    const C *__restrict__ ph = phasor + (size_t)A.rows * chain_ph_v(R);  // the f = 0 row
    const T inv = T(1.0 / (double)(R * G::M));
    __syncthreads();
    for (int w = blockIdx.x; w < R * A.total; w += gridDim.x) {
        const int b = __builtin_amdgcn_readfirstlane(w / R), r = __builtin_amdgcn_readfirstlane(w % R);
        const __amdgpu_buffer_rsrc_t rs_sig = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.sig + (size_t)b * NS_IN), 0, NS_IN * (int)sizeof(C), 0x00020000);
        C lane[1] = {chain_pb<T, LOGM>(ph, L.t)};
        if (r) lane[0] = cmulc(lane[0], A.th[(r - 1) * G::W + L.t]);
        lane[0] = conj(lane[0]);
        C v[1][16];
        chain_input_one<T, LOGM, R>(v[0], rs_sig, r, L.t, ph);
        L.forward(v, lane);
        C *spec = A.spec + ((size_t)b * R + r) * (16 * G::W);
#pragma unroll
        for (int k = 0; k < 16; ++k) spec[((k >> 1) * G::W + L.t) * 2 + (k & 1)] = {v[0][k].x * inv, -v[0][k].y * inv};
        __syncthreads();  // the next iteration's stage-0 writes vs this one's reads by other waves
    }
}

// one chain of one row over the NB slots: v = chain input (chain_input_pair)
//   -> y'_r[b][i] = th_r(beta_b) * IDFT_M(Hs G_r)[beta_b + W i]
template <typename T, int LOGM, int R, int NB, int ABL>
__device__ __forceinline__ void chain_run(cpx<T> (&v)[NB][16], const ChainLane<T, LOGM, NB, ABL> &L, const ChainArgs<T> &A,
                                          const __amdgpu_buffer_rsrc_t rs_spec, int r, const cpx<T> (&pb)[NB])
{
    using G = ChainGeo<LOGM>;
    using C = cpx<T>;
    C lane[NB], post[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        post[b] = C{T(1), T(0)};
        lane[b] = pb[b];
        if (r) {
            post[b] = A.th[(r - 1) * G::W + L.beta[b]];  // W_L^(beta r)
            lane[b] = cmulc(lane[b], post[b]);            // w^beta e^{-2 pi i beta r / L}
        }
        lane[b] = conj(lane[b]);
    }
    L.forward(v, lane);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const unsigned voff = (unsigned)((r * 16 * G::W + 2 * L.beta[b]) * sizeof(C));
#pragma unroll
        for (int half = 0; half < 2; ++half) {  // two groups of eight values (register pressure, see chain_input_pair)
            C h[8];
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                ChainLane<T, LOGM, NB, ABL>::IO::spec2(rs_spec, voff, (unsigned)(2 * G::W * (4 * half + k / 2) * sizeof(C)), k, h[k], h[k + 1]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[b][8 * half + k] = cmul(v[b][8 * half + k], h[k]);  // xcor_rustfft.rs:64-73
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (R < 8) {  // r is a compile-time constant after inlining: chain 0 keeps its cheaper twiddle form
        if (r) {
            L.inverse(v, [&](int b, int k, C x) { const TwFold<T> fpost(L.tw[b], post[b]); return twA_k(x, k, L.tw[b], fpost); });
        } else {
            L.inverse(v, [&](int b, int k, C x) { return twA_k(x, k, L.tw[b]); });
        }
    } else {  // r is a run-time value in the looped rows: post = 1 for r = 0 is exact
        L.inverse(v, [&](int b, int k, C x) { const TwFold<T> fpost(L.tw[b], post[b]); return twA_k(x, k, L.tw[b], fpost); });
    }
}

// ---- the row kernel --------------------------------------------------------------------------------
// target("no-load-store-opt"): the compiler's load/store optimizer pairs the complex64 LDS reads of an exchange
// into ds_read2_b64, which gfx950 serves at HALF the rate of two ds_read_b64 (8 vs 2 + 2 LDS cycles,
// MI355X_MICROARCH.md section LDS), and its wider live ranges cost the 128-VGPR instantiations spills:
// without it k_chain_rows<float, 14, 4> has 562 ds_read_b64 + 32 ds_read2_b64 instead of 162 + 232 and 54
// scratch accesses instead of ~150, and runs 1.30 instead of 1.51-1.57 ms per 4096 x 65536 surface;
// the complex128 instantiations (ds_read_b128 either way) are unchanged.
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass does not know the AMDGPU feature and would warn)
#define CAF_NO_LOAD_STORE_OPT __attribute__((target("no-load-store-opt")))
#else
#define CAF_NO_LOAD_STORE_OPT
#endif
template <typename T, int LOGM, int R, int NB = 1, int ABL = 0>
__global__ CAF_NO_LOAD_STORE_OPT
__launch_bounds__(ChainGeo<LOGM>::W / NB, chain_wps_v(LOGM, sizeof(cpx<T>), NB)) void k_chain_rows(
    const ChainArgs<T> A, const cpx<T> *__restrict__ phasor)
{
    using G = ChainGeo<LOGM>;
    using C = cpx<T>;
    constexpr int W = G::W, M = G::M, Lp = R * M, NS_IN = R * M / 2;
    constexpr int NWV = (W / NB + 63) / 64;  // waves per workgroup
    __shared__ __attribute__((aligned(16))) unsigned char smem[chain_lds_bytes<T, LOGM>()];
    using IO = typename ChainIoFor<T, ABL>::type;
    const ChainLane<T, LOGM, NB, ABL> L(smem, A.twM);
    unsigned char *const scratch = smem + chain_lds_bytes<T, LOGM>() - 256;  // per-wave argmax partials
    T *const sv = reinterpret_cast<T *>(scratch);
    uint32_t *const si = reinterpret_cast<uint32_t *>(scratch + 128);
    const int lane = L.t & 63, wave = L.t >> 6;
    __syncthreads();

    for (int g = blockIdx.x; g < A.total; g += gridDim.x) {
        // g is workgroup-uniform; the division runs on the VALU, so pin the results to SGPRs (a buffer
        // descriptor built from a VGPR value costs a waterfall loop per load)
        const int bs = __builtin_amdgcn_readfirstlane(g / A.rows);
        const int r_row = __builtin_amdgcn_readfirstlane(g - bs * A.rows);
        const C *__restrict__ ph = phasor + (size_t)r_row * chain_ph_v(R);
        const __amdgpu_buffer_rsrc_t rs_sig = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.sig + (size_t)bs * NS_IN), 0, NS_IN * (int)sizeof(C), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.spec + (size_t)bs * R * 16 * W), 0, R * 16 * W * (int)sizeof(C), 0x00020000);
        T *const out = A.surface ? A.surface + (size_t)g * Lp : nullptr;
        const __amdgpu_buffer_rsrc_t rs_out =
            __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? Lp * (int)sizeof(T) : 0, 0x00020000);
        C pb[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) pb[b] = chain_pb<T, LOGM>(ph, L.beta[b]);

        T bv[R];
        int bi[R];
#pragma unroll
        for (int j = 0; j < R; ++j) { bv[j] = T(0); bi[j] = 0; }
        // mag of block j (lags m' + M j, m' = beta_b + W i): store + first-strictly-greater running
        // max; called with i outer, b inner, so the code i*NB + b increases with the lag inside a block
        auto emit = [&](int j, int i, int b, C c) {
            const T m = norm_sqr(c);  // mod.rs:147
            bi[j] = m > bv[j] ? i * NB + b : bi[j];
            bv[j] = vmax(bv[j], m);
            IO::surf_st(rs_out, (unsigned)(L.beta[b] * sizeof(T)), (unsigned)((M * j + W * i) * sizeof(T)), m);
        };

        if constexpr (R == 2) {
            C e[NB][16], o[NB][16];
#pragma unroll
            for (int b = 0; b < NB; ++b) chain_input_pair<T, LOGM, R, ABL>(e[b], o[b], rs_sig, 0, 1, L.beta[b], ph);
            chain_run<T, LOGM, R, NB, ABL>(e, L, A, rs_spec, 0, pb);
            chain_run<T, LOGM, R, NB, ABL>(o, L, A, rs_spec, 1, pb);
            // c[m'] , c[m' + M] = E +- W_32^i (th O): th came folded into the odd chain's last twiddles
#pragma unroll
            for (int i = 0; i < 16; ++i) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    C lo, hi;
                    bfly_w(e[b][i], o[b][i], W32C16[i], W32S16[i], lo, hi);
                    emit(0, i, b, lo);
                    emit(1, i, b, hi);
                }
            }
        } else {
            // this workgroup's scratch slab through a buffer descriptor: per-lane byte offset in one VGPR,
            // the register-row offset in an SGPR (32 separate 64-bit global addresses would cost 64 VGPRs).
            // One access per value, on purpose.  Packing two values into one 16-byte store per lane was
            // tried in complex64: a buffer_store_dwordx4 followed within two wait states by a VALU write of
            // its data registers needs an s_nop (CDNA3 ISA 4.5), which the compiler inserts for its own
            // instructions but NOT in front of inline asm -- and the packed-f32 arithmetic here is inline
            // asm (cplx.hpp): 2 % of a surface's lags came out as the previous row's values.
            constexpr int NARR = chain_slab_arrays_v(R);
            const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(A.slab + (size_t)blockIdx.x * (NARR * 16 * W)), 0, NARR * 16 * W * (int)sizeof(C), 0x00020000);
            auto slab_st = [&](int arr, int i, int b, C x) {
                IO::slab_st(rs_slab, (unsigned)(L.beta[b] * sizeof(C)), (unsigned)((arr * 16 + i) * W * sizeof(C)), x);
            };
            auto slab_ld = [&](int arr, int i, int b) -> C {
                return IO::slab_ld(rs_slab, (unsigned)(L.beta[b] * sizeof(C)), (unsigned)((arr * 16 + i) * W * sizeof(C)), arr, i);
            };
            if constexpr (R == 16) {
                // COMPUTE PHASE: the eight chain pairs (r', r'+8) as iterations of a run-time loop (the chain code exists
                // once, as for R = 8); every chain output y'_r goes to slab array r.  MEMORY PHASE: lag by lag,
                // c[m' + M j] = sum_r W_16^(j r) W_256^(i r) y'_r[m'],  m' = beta + W i: sixteen coalesced slab reads,
                // the W_256 twiddles from a table that takes the (now idle) chain's first 256 LDS positions, one
                // radix-16 butterfly, |.|^2, argmax, sixteen coalesced surface stores.
                static_assert(NB == 1, "looped rows are written for one butterfly per thread");
                C cur[1][16], oth[1][16];
#pragma clang loop unroll(disable)
                for (int it = 0; it < 8; ++it) {
                    const int rp = __builtin_amdgcn_readfirstlane(it);
                    chain_input_pair<T, LOGM, R, ABL>(cur[0], oth[0], rs_sig, rp, rp + 8, L.beta[0], ph);
#pragma clang loop unroll(disable)
                    for (int c = 0; c < 2; ++c) {
                        const int r = __builtin_amdgcn_readfirstlane(rp + 8 * c);
                        chain_run<T, LOGM, R, NB, ABL>(cur, L, A, rs_spec, r, pb);
                        const unsigned so = (unsigned)(r * 16 * W * sizeof(C));
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            IO::slab_st(rs_slab, (unsigned)(L.beta[0] * sizeof(C)), so + (unsigned)(i * W * sizeof(C)), cur[0][i]);
#pragma unroll
                        for (int i = 0; i < 16; ++i) { const C t = cur[0][i]; cur[0][i] = oth[0][i]; oth[0][i] = t; }
                    }
                }
                __syncthreads();  // every wave is done with the chain: its LDS becomes the W_256 table
                if (L.t < 256) L.Lc[L.t] = A.twM[L.t * (M / 256)];
                __syncthreads();
#pragma clang loop unroll(disable)
                for (int i = 0; i < 16; ++i) {
                    C z[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        z[r] = IO::slab_ld(rs_slab, (unsigned)(L.beta[0] * sizeof(C)), (unsigned)((r * 16 + i) * W * sizeof(C)), r, i);
#pragma unroll
                    for (int r = 1; r < 16; ++r) z[r] = cmul(z[r], L.Lc[(i * r) & 255]);
                    dft16(z);
#pragma unroll
                    for (int j = 0; j < 16; ++j) emit(j, i, 0, z[j]);
                }
                __syncthreads();  // the table's positions belong to the next row's chain again
            } else if constexpr (R == 8) {
                // LOOPED form (R = 4 in this form ran 1-2 % slower than the unrolled form below -- measured in rounds 2-3,
                // no longer in the source; R = 8 gains 14 %: 2.10 vs 2.44 ms per 2048 x 131072 complex64 rows, 74 vs 206
                // spills): the R/2 chain pairs are iterations of a run-time loop and the two chains of a pair
                // iterations of an inner one, so the chain code (input stage, forward, spectrum product, inverse:
                // ~2.8 k instructions) exists once instead of R times.  Fully inlined the R = 4 row is ~90 KB of
                // code and the R = 8 row ~180 KB, against a 64 KB instruction cache shared by two CUs.
                static_assert(NB == 1, "looped rows are written for one butterfly per thread");
                constexpr int NP = R / 2;
                C cur[1][16], oth[1][16];
#pragma clang loop unroll(disable)
                for (int it = 0; it < NP; ++it) {
                    // R = 8: z_r = W_128^(i r) y'_r.  Pairs (r', r'+4): P_r' = y'_r' + W_32^i y'_(r'+4), Q_r' = y'_r' - ...;
                    //   lags m' + M j:  j = 2j'   : sum_r' (i)^(j' r') thP^r' P_r',  thP = W_128^i
                    //                   j = 2j'+1 : sum_r' (i)^(j' r') thQ^r' Q_r',  thQ = W_128^(i+16)
                    //   i.e. two radix-4 combinations like the R = 4 one.  Pair order 0, 2, 1, 3; slab arrays:
                    //   [0],[1] = P0, Q0 -> aP, bP;  [2],[3] = aQ, bQ;  [4],[5] = P1, Q1;  P3, Q3 stay in registers.
                    const int rp = __builtin_amdgcn_readfirstlane(((it & 1) << 1) | (it >> 1));
                    chain_input_pair<T, LOGM, R, ABL>(cur[0], oth[0], rs_sig, rp, rp + NP, L.beta[0], ph);
#pragma clang loop unroll(disable)
                    for (int c = 0; c < 2; ++c) {
                        chain_run<T, LOGM, R, NB, ABL>(cur, L, A, rs_spec, __builtin_amdgcn_readfirstlane(rp + NP * c), pb);
#pragma unroll
                        for (int i = 0; i < 16; ++i) { const C t = cur[0][i]; cur[0][i] = oth[0][i]; oth[0][i] = t; }
                    }
                    // cur = y'_rp, oth = y'_(rp + 4)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {  // P, Q in place of y'_r', y'_(r'+4)
                        C P, Q;
                        bfly_w(cur[0][i], oth[0][i], W32C16[i], W32S16[i], P, Q);
                        cur[0][i] = P;
                        oth[0][i] = Q;
                    }
                    if (it == 0) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) { slab_st(0, i, 0, cur[0][i]); slab_st(1, i, 0, oth[0][i]); }
                    } else if (it == 1) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            C aP, bP, aQ, bQ;
                            bfly_w(slab_ld(0, i, 0), cur[0][i], W128C[2 * i], W128S[2 * i], aP, bP);
                            bfly_w(slab_ld(1, i, 0), oth[0][i], W128C[2 * i + 32], W128S[2 * i + 32], aQ, bQ);
                            slab_st(0, i, 0, aP);
                            slab_st(1, i, 0, bP);
                            slab_st(2, i, 0, aQ);
                            slab_st(3, i, 0, bQ);
                            if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                        }
                    } else if (it == 2) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) { slab_st(4, i, 0, cur[0][i]); slab_st(5, i, 0, oth[0][i]); }
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            C cc, dd, o0, o1, o2, o3;
                            bfly_w(slab_ld(4, i, 0), cur[0][i], W128C[2 * i], W128S[2 * i], cc, dd);
                            bfly_w(slab_ld(0, i, 0), cc, W128C[i], W128S[i], o0, o2);
                            bfly_w(slab_ld(1, i, 0), dd, -W128S[i], W128C[i], o1, o3);
                            emit(0, i, 0, o0);
                            emit(2, i, 0, o1);
                            emit(4, i, 0, o2);
                            emit(6, i, 0, o3);
                            bfly_w(slab_ld(5, i, 0), oth[0][i], W128C[2 * i + 32], W128S[2 * i + 32], cc, dd);
                            bfly_w(slab_ld(2, i, 0), cc, W128C[i + 16], W128S[i + 16], o0, o2);
                            bfly_w(slab_ld(3, i, 0), dd, -W128S[i + 16], W128C[i + 16], o1, o3);
                            emit(1, i, 0, o0);
                            emit(3, i, 0, o1);
                            emit(5, i, 0, o2);
                            emit(7, i, 0, o3);
                            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            } else if constexpr (R == 4) {
                // z_r = W_64^(i r) y'_r;  a, b = z0 +- z2;  c, d = z1 +- z3 = W_64^i (y1' +- W_64^(2i) y3');
                // c[m' + M j]: j = 0, 2: a +- c;  j = 1, 3: b +- i d
                {
                    C y0[NB][16], y2[NB][16];
#pragma unroll
                    for (int b = 0; b < NB; ++b) chain_input_pair<T, LOGM, R, ABL>(y0[b], y2[b], rs_sig, 0, 2, L.beta[b], ph);
                    chain_run<T, LOGM, R, NB, ABL>(y0, L, A, rs_spec, 0, pb);
                    chain_run<T, LOGM, R, NB, ABL>(y2, L, A, rs_spec, 2, pb);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            C a, bb;
                            bfly_w(y0[b][i], y2[b][i], W64C[2 * i], W64S[2 * i], a, bb);
                            slab_st(0, i, b, a);
                            slab_st(1, i, b, bb);
                        }
                    }
                }
                C y1[NB][16], y3[NB][16];
#pragma unroll
                for (int b = 0; b < NB; ++b) chain_input_pair<T, LOGM, R, ABL>(y1[b], y3[b], rs_sig, 1, 3, L.beta[b], ph);
                chain_run<T, LOGM, R, NB, ABL>(y1, L, A, rs_spec, 1, pb);
                chain_run<T, LOGM, R, NB, ABL>(y3, L, A, rs_spec, 3, pb);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        C cc, dd;
                        bfly_w(y1[b][i], y3[b][i], W64C[2 * i], W64S[2 * i], cc, dd);
                        const C a = slab_ld(0, i, b), bb = slab_ld(1, i, b);
                        C c0, c2, c1, c3;
                        bfly_w(a, cc, W64C[i], W64S[i], c0, c2);     // a +- W_64^i c'
                        bfly_w(bb, dd, -W64S[i], W64C[i], c1, c3);  // b +- i W_64^i d'
                        emit(0, i, b, c0);
                        emit(1, i, b, c1);
                        emit(2, i, b, c2);
                        emit(3, i, b, c3);
                    }
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most four (a, b) pairs per slot in flight
                }
            }
        }
        // lags of block j all precede those of block j+1: init (0.0, lag 0) like mod.rs:143
        T best = T(0);
        uint32_t besti = 0u;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (bv[j] > best) {
                best = bv[j];
                besti = (uint32_t)(L.t + (W / NB) * (bi[j] % NB) + W * (bi[j] / NB) + M * j);
            }
        wave_arg_reduce_maxmin(best, besti);
        if (lane == 63) { sv[wave] = best; si[wave] = besti; }
        __syncthreads();
        if (L.t == 0) {
            T rb = sv[0];
            uint32_t ri = si[0];
            for (int w = 1; w < NWV; ++w) arg_merge(rb, ri, sv[w], si[w]);
            A.row_idx[g] = ri;
            A.row_val[g] = rb;
        }
        __syncthreads();  // sv/si are rewritten by the next row
    }
}

}  // namespace caf
