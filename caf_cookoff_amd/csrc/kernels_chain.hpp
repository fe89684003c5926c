// kernels_chain.hpp -- size-generic LDS-resident Doppler-row kernel ("chain" path).
//
// Same mathematics as kernels_seq4096.hpp (mod.rs:121-166 with FFT(haystack) hoisted and every
// transform a positive-exponent one), for any padded length L = 2n = R * M that splits into
// R in {2, 4} "chains" of M = 2^LOGM points, M * sizeof(complex) <= ~128 KiB so ONE chain lives
// in a workgroup's LDS:
//
//   forward, decimation in frequency over the first radix-R stage: bins k = R k' + r,
//     G[R k' + r] = IDFT_M(u_r)[k'],  u_r[n'] = W_L^(n' r) * sum_{j < R/2} conj(s[n' + M j]) W_R^(j r)
//     (the upper half of s = needle * w^n ++ 0 is the zero padding of mod.rs:130, so only R/2
//     of the R terms exist: R = 2 -> one, R = 4 -> two);
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
// L2-resident) while chains 1 and 3 run.  Traffic per row: the surface once + that slab once
// each way, instead of three passes over a work row (kernels_big65536.hpp).
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
constexpr int chain_wps_v(int logm, size_t csize)
{
    const int W = (1 << logm) / 16;
    const int waves_wg = (W + 63) / 64;
    const int by_lds = (int)(160 * 1024 / chain_lds_bytes_v(logm, csize));
    const int per_simd = (waves_wg * (by_lds < 1 ? 1 : by_lds) + 3) / 4;  // what LDS lets reside
    const int cap = csize == 16 ? 2 : 4;                                  // 256 / 128 VGPRs
    const int floor_ = (waves_wg + 3) / 4;                                // one workgroup must fit
    return per_simd < cap ? (per_simd < floor_ ? floor_ : per_simd) : (cap < floor_ ? floor_ : cap);
}
// resident workgroups per CU
constexpr size_t chain_wg_per_cu_v(int logm, size_t csize)
{
    const size_t W = ((size_t)1 << logm) / 16, waves_wg = (W + 63) / 64;
    size_t per_cu = 160 * 1024 / chain_lds_bytes_v(logm, csize);
    if (per_cu * waves_wg > (size_t)chain_wps_v(logm, csize) * 4) per_cu = (size_t)chain_wps_v(logm, csize) * 4 / waves_wg;
    return per_cu < 1 ? 1 : per_cu;
}
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

// Per-row phasor table, CH_PH entries per row (row `nrows` = the f = 0 row for the haystack):
//   [0..15] w^j   [16..31] w^(16 j)   [32..47] w^(256 j)
//   [48 + 16 r + q] step_r[q] = w^(W q) * e^{-2 pi i q r / (16 R)}      (r < R, q < 16)
//   [112] w^M                                                          (R = 4: second half of the needle)
// every entry from one f64 sincos of the exact phase product (never an f32
// recurrence), w = e^{j ph}, ph = ((2 PI) f)(1/fs) as mod.rs:54-56.
constexpr int CH_PH = 128;
template <typename T>
__global__ void k_chain_phasors(const double *__restrict__ ph, int nrows, int M, int R, cpx<T> *__restrict__ tab)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g / CH_PH, e = g % CH_PH;
    if (row > nrows) return;
    const double p = row < nrows ? ph[row] : 0.0;
    const int W = M / 16;
    double mult = 0.0, rot = 0.0;  // phase p*mult, extra rotation e^{-2 pi i rot}
    if (e < 16) mult = (double)e;
    else if (e < 32) mult = 16.0 * (double)(e - 16);
    else if (e < 48) mult = 256.0 * (double)(e - 32);
    else if (e < 112) {
        const int r = (e - 48) >> 4, q = (e - 48) & 15;
        mult = (double)W * (double)q;
        rot = (double)(q * r) / (16.0 * (double)R);
    } else if (e == 112) mult = (double)M;
    double s, c, s2, c2;
    sincos(p * mult, &s, &c);
    sincospi(-2.0 * rot, &s2, &c2);
    tab[(size_t)row * CH_PH + e] = {(T)(c * c2 - s * s2), (T)(c * s2 + s * c2)};
}

template <typename T>
struct ChainArgs {
    const cpx<T> *sig;      // prepare: haystack [batch][n]; rows: needle [batch][n]
    cpx<T> *spec;           // Hs [batch][R][16][W]: prepare writes, rows read
    const cpx<T> *twM;      // [M]
    const cpx<T> *th;       // [(R-1) W]
    T *surface;             // [batch][rows][L] or nullptr
    uint64_t *row_idx;      // [batch][rows]
    T *row_val;             // [batch][rows]
    cpx<T> *slab;           // R = 4: [gridDim.x][2][16][W] scratch (a, b of the last radix-4 stage)
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

// Per-thread geometry + tables of one workgroup
template <typename T, int LOGM>
struct ChainLane {
    using G = ChainGeo<LOGM>;
    using C = cpx<T>;
    int t;
    int base[3];   // LDS base position of radix-16 stages 0..2
    int o[3];      // t % S_s
    int basef;     // last small stage / S = 1 stage: 17 t
    C *Lc;         // chain
    const C *tw1;  // LDS [16][S_1]
    const C *tw2;  // LDS [16][S_2]
    TwSet<T> tw;   // stage-0 twiddles W_M^(t k), k in {1,2,3,4,8,12}

    __device__ __forceinline__ ChainLane(unsigned char *smem, const C *__restrict__ twM)
    {
        t = threadIdx.x;
        Lc = reinterpret_cast<C *>(smem);
        C *tab1 = Lc + G::CHAIN;
        C *tab2 = tab1 + G::M / 16;
        tw1 = tab1;
        tw2 = tab2;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (s < G::NS) {
                const int S = G::str(s), B = G::blk(s);
                const int g = t / S, oo = t % S;
                o[s] = oo;
                base[s] = g * (B + B / 16) + oo + (S >= 16 ? (oo >> 4) : 0);
            } else {
                o[s] = 0;
                base[s] = 0;
            }
        }
        basef = 17 * t;
        tw.w1 = twM[t * 1];
        tw.w2 = twM[t * 2];
        tw.w3 = twM[t * 3];
        tw.w4 = twM[t * 4];
        tw.w8 = twM[t * 8];
        tw.w12 = twM[t * 12];
        // stage tables [k][o] = W_{B_s}^(o k) = twM[o k M / B_s]
        if constexpr (G::NS >= 2 && G::str(1) > 1) {
            constexpr int S = G::str(1), B = G::blk(1);
            for (int i = t; i < 16 * S; i += G::W) tab1[i] = twM[(i % S) * (i / S) * (G::M / B)];
        }
        if constexpr (G::NS >= 3 && G::str(2) > 1) {
            constexpr int S = G::str(2), B = G::blk(2);
            for (int i = t; i < 16 * S; i += G::W) tab2[i] = twM[(i % S) * (i / S) * (G::M / B)];
        }
    }

    // synchronise the exchange that follows radix-16 stage s
    template <int S>
    __device__ __forceinline__ void sync_after() const
    {
        if constexpr (G::local_after(S))
            wave_lds_fence();
        else
            __syncthreads();
    }
    template <int S>
    __device__ __forceinline__ void read_stage(C (&v)[16]) const
    {
        if constexpr (S < G::NS && G::str(S < G::NS ? S : 0) > 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = Lc[base[S] + G::off(S, j)];
        } else {  // S = 1 radix-16 stage or the small last stage: 16 contiguous elements
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = Lc[basef + j];
        }
    }
    template <int S>
    __device__ __forceinline__ void write_stage(const C (&v)[16]) const
    {
        if constexpr (S < G::NS && G::str(S < G::NS ? S : 0) > 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) Lc[base[S] + G::off(S, j)] = v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) Lc[basef + j] = v[j];
        }
    }
    // twiddle W_{B_S}^(o k) of radix-16 stage S >= 1 from its LDS table
    template <int S>
    __device__ __forceinline__ C twk(int k) const
    {
        constexpr int St = G::str(S < G::NS ? S : 0);
        return (S == 1 ? tw1 : tw2)[k * St + o[S]];
    }

    // Forward DIF chain.  In: v[q] = u[t + W q] WITHOUT the lane-common factor `lane`, which is
    // folded into the stage-0 output twiddles.  Out: v[k] = G[...] in this path's register layout.
    __device__ __forceinline__ void forward(C (&v)[16], const C lane) const
    {
        const TwFold<T> f0(tw, lane);
        if constexpr (G::NST == 1) {  // (not instantiated: LOGM >= 8)
            dft16(v);
            return;
        }
        dft16_sink(v, [&](int k, C x) { Lc[base[0] + G::off(0, k)] = twA_k(x, k, tw, f0); });
        sync_after<0>();
        stage_fwd<1>(v);
    }
    template <int S>
    __device__ __forceinline__ void stage_fwd(C (&v)[16]) const
    {
        read_stage<S>(v);
        if constexpr (S >= G::NS) {  // the small last stage
            small_stage<T, G::RL>(v);
        } else if constexpr (S == G::NST - 1) {  // last stage is a radix-16 one (S_s = 1): no twiddle, stays in registers
            dft16(v);
        } else {
            dft16_sink(v, [&](int k, C x) { Lc[base[S] + G::off(S, k)] = k ? cmul(x, twk<S>(k)) : x; });
            sync_after<S>();
            stage_fwd<S + 1>(v);
        }
    }

    // Inverse DIT chain (mirror).  In: v[k] in the forward's output layout.  Out: v[i] = y[t + W i]
    // BEFORE the stage-0 butterfly's input twiddle is applied by the caller-supplied functor
    // `tw0(k, x)` (so a per-chain lane factor can ride on it), i.e. this function ends with
    // dft16 of tw0-twiddled inputs.
    template <typename F>
    __device__ __forceinline__ void inverse(C (&v)[16], F &&tw0) const
    {
        stage_inv<G::NST - 1>(v);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = tw0(k, v[k]);
        dft16(v);
    }
    template <int S>
    __device__ __forceinline__ void stage_inv(C (&v)[16]) const
    {
        if constexpr (S == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = Lc[base[0] + G::off(0, j)];
            return;
        } else {
            if constexpr (S >= G::NS) {
                small_stage<T, G::RL>(v);
                wave_lds_fence();  // this thread's forward reads of the same 16 positions are done (program order)
                write_stage<S>(v);
            } else if constexpr (S == G::NST - 1) {
                dft16(v);
                wave_lds_fence();
                write_stage<S>(v);
            } else {
                read_stage<S>(v);
#pragma unroll
                for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twk<S>(k));
                wave_lds_fence();
                dft16_sink(v, [&](int k, C x) { Lc[base[S] + G::off(S, k)] = x; });
            }
            sync_after<S - 1>();
            stage_inv<S - 1>(v);
        }
    }
};

// lane-common phasor w^t from the three-level table of a row
template <typename T, int LOGM>
__device__ __forceinline__ cpx<T> chain_pb(const cpx<T> *__restrict__ ph, int t)
{
    cpx<T> pb = cmul(ph[t & 15], ph[16 + ((t >> 4) & 15)]);
    if constexpr (ChainGeo<LOGM>::W > 256) pb = cmul(pb, ph[32 + (t >> 8)]);
    return pb;
}

// chain input: v[q] = conj(x[q] * step_r[q]),  x = a0 (R = 2) or a0 + (-i)^r wM a1 (R = 4).
// The samples are fetched in groups of four register rows, one group ahead of its use: at most
// 2 x 4 (R = 2) / 2 x 8 (R = 4) loads are in flight.  Hoisting all 16 / 32 of them (what the
// scheduler does on its own) costs 32 / 64 VGPRs on top of the 64 the two chains' data need and
// spills at four waves per SIMD.
template <typename T, int LOGM, int R>
__device__ __forceinline__ void chain_input(cpx<T> (&v)[16], const __amdgpu_buffer_rsrc_t rs_sig, int r, int t,
                                            const cpx<T> *__restrict__ ph)
{
    using C = cpx<T>;
    constexpr int W = ChainGeo<LOGM>::W, M = ChainGeo<LOGM>::M;
    const C *ps = ph + 48 + 16 * r;
    const unsigned voff = (unsigned)(t * sizeof(C));
    C wM = C{T(1), T(0)};
    if constexpr (R == 4) wM = ph[112];
    C a0[2][4], a1[2][4];
    auto fetch = [&](int grp) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = 4 * grp + u;
            a0[grp & 1][u] = bload(rs_sig, voff, (unsigned)(W * q * sizeof(C)), (C *)nullptr);
            if constexpr (R == 4) a1[grp & 1][u] = bload(rs_sig, voff, (unsigned)((M + W * q) * sizeof(C)), (C *)nullptr);
        }
    };
    fetch(0);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        if (grp < 3) fetch(grp + 1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = 4 * grp + u;
            C x = a0[grp & 1][u];
            if constexpr (R == 4) {
                const C b = cmul(a1[grp & 1][u], wM);
                x = r == 0 ? x + b : r == 1 ? sub_i(x, b) : r == 2 ? x - b : add_i(x, b);
            }
            v[q] = cmul_conj(x, ps[q]);
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

// ---- haystack spectrum: one workgroup per (surface, chain) ------------------------------------
// Hs = FFT_L(haystack ++ 0)/L = conj(IDFT_L(conj h))/L, register layout spec[b][r][k][t].
template <typename T, int LOGM, int R>
__global__ __launch_bounds__(ChainGeo<LOGM>::W) void k_chain_prepare(const ChainArgs<T> A, const cpx<T> *__restrict__ phasor)
{
    using G = ChainGeo<LOGM>;
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[chain_lds_bytes<T, LOGM>()];
    const ChainLane<T, LOGM> L(smem, A.twM);
    constexpr int NS_IN = R * G::M / 2;  // samples per input
    const C *__restrict__ ph = phasor + (size_t)A.rows * CH_PH;  // the f = 0 row
    const T inv = T(1.0 / (double)(R * G::M));
    __syncthreads();
    for (int w = blockIdx.x; w < R * A.total; w += gridDim.x) {
        const int b = __builtin_amdgcn_readfirstlane(w / R), r = __builtin_amdgcn_readfirstlane(w % R);
        const __amdgpu_buffer_rsrc_t rs_sig = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.sig + (size_t)b * NS_IN), 0, NS_IN * (int)sizeof(C), 0x00020000);
        C lane = chain_pb<T, LOGM>(ph, L.t);
        if (r) lane = cmulc(lane, A.th[(r - 1) * G::W + L.t]);
        C v[16];
        chain_input<T, LOGM, R>(v, rs_sig, r, L.t, ph);
        L.forward(v, conj(lane));
        C *spec = A.spec + ((size_t)b * R + r) * (16 * G::W);
#pragma unroll
        for (int k = 0; k < 16; ++k) spec[k * G::W + L.t] = {v[k].x * inv, -v[k].y * inv};
        __syncthreads();  // the next iteration's stage-0 writes vs this one's reads by other waves
    }
}

// one chain of one row: needle -> y'_r[i] = th_r(t) * IDFT_M(Hs G_r)[t + W i]
template <typename T, int LOGM, int R>
__device__ __forceinline__ void chain_run(cpx<T> (&v)[16], const ChainLane<T, LOGM> &L, const ChainArgs<T> &A,
                                          const __amdgpu_buffer_rsrc_t rs_sig, const __amdgpu_buffer_rsrc_t rs_spec, int r,
                                          const cpx<T> pb, const cpx<T> *__restrict__ ph)
{
    using G = ChainGeo<LOGM>;
    using C = cpx<T>;
    C lane = pb;
    C post = C{T(1), T(0)};
    if (r) {
        post = A.th[(r - 1) * G::W + L.t];  // W_L^(t r)
        lane = cmulc(lane, post);           // w^t e^{-2 pi i t r / L}
    }
    chain_input<T, LOGM, R>(v, rs_sig, r, L.t, ph);
    L.forward(v, conj(lane));
    const unsigned voff = (unsigned)((r * 16 * G::W + L.t) * sizeof(C));
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // two groups of eight loads (register pressure, see chain_input)
        C h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = bload(rs_spec, voff, (unsigned)(G::W * (8 * half + k) * sizeof(C)), (C *)nullptr);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[8 * half + k] = cmul(v[8 * half + k], h[k]);  // xcor_rustfft.rs:64-73
        __builtin_amdgcn_sched_barrier(0);
    }
    if (r) {
        const TwFold<T> fpost(L.tw, post);
        L.inverse(v, [&](int k, C x) { return twA_k(x, k, L.tw, fpost); });
    } else {
        L.inverse(v, [&](int k, C x) { return twA_k(x, k, L.tw); });
    }
}

// ---- the row kernel --------------------------------------------------------------------------------
template <typename T, int LOGM, int R>
__global__ __launch_bounds__(ChainGeo<LOGM>::W, chain_wps_v(LOGM, sizeof(cpx<T>))) void k_chain_rows(
    const ChainArgs<T> A, const cpx<T> *__restrict__ phasor)
{
    using G = ChainGeo<LOGM>;
    using C = cpx<T>;
    constexpr int W = G::W, M = G::M, Lp = R * M, NS_IN = R * M / 2;
    constexpr int NW = (W + 63) / 64;  // waves per workgroup
    __shared__ __attribute__((aligned(16))) unsigned char smem[chain_lds_bytes<T, LOGM>()];
    const ChainLane<T, LOGM> L(smem, A.twM);
    unsigned char *const scratch = smem + chain_lds_bytes<T, LOGM>() - 256;  // per-wave argmax partials
    T *const sv = reinterpret_cast<T *>(scratch);
    uint32_t *const si = reinterpret_cast<uint32_t *>(scratch + 128);
    const int lane = L.t & 63, wave = L.t >> 6;
    __syncthreads();

    for (int g = blockIdx.x; g < A.total; g += gridDim.x) {
        // g is workgroup-uniform; the division runs on the VALU, so pin the results to SGPRs (a buffer
        // descriptor built from a VGPR value costs a waterfall loop per load)
        const int b = __builtin_amdgcn_readfirstlane(g / A.rows);
        const int r_row = __builtin_amdgcn_readfirstlane(g - b * A.rows);
        const C *__restrict__ ph = phasor + (size_t)r_row * CH_PH;
        const __amdgpu_buffer_rsrc_t rs_sig = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.sig + (size_t)b * NS_IN), 0, NS_IN * (int)sizeof(C), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.spec + (size_t)b * R * 16 * W), 0, R * 16 * W * (int)sizeof(C), 0x00020000);
        T *const out = A.surface ? A.surface + (size_t)g * Lp : nullptr;
        const __amdgpu_buffer_rsrc_t rs_out =
            __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? Lp * (int)sizeof(T) : 0, 0x00020000);
        const C pb = chain_pb<T, LOGM>(ph, L.t);
        const unsigned voff_out = (unsigned)(L.t * sizeof(T));

        T bv[R];
        int bi[R];
#pragma unroll
        for (int j = 0; j < R; ++j) { bv[j] = T(0); bi[j] = 0; }
        // mag of block j (lags m' + M j), register row i: store + first-strictly-greater running max
        auto emit = [&](int j, int i, C c) {
            const T m = norm_sqr(c);  // mod.rs:147
            bi[j] = m > bv[j] ? i : bi[j];
            bv[j] = vmax(bv[j], m);
            store_one_aux<CAF_AUX_SC1>(rs_out, voff_out, (unsigned)((M * j + W * i) * sizeof(T)), m);
        };

        if constexpr (R == 2) {
            C e[16], o[16];
            chain_run<T, LOGM, R>(e, L, A, rs_sig, rs_spec, 0, pb, ph);
            chain_run<T, LOGM, R>(o, L, A, rs_sig, rs_spec, 1, pb, ph);
            // c[m'] , c[m' + M] = E +- W_32^i (th O): th came folded into the odd chain's last twiddles
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                C lo, hi;
                bfly_w(e[i], o[i], W32C16[i], W32S16[i], lo, hi);
                emit(0, i, lo);
                emit(1, i, hi);
            }
        } else {
            // this workgroup's scratch slab through a buffer descriptor: per-lane byte offset in one VGPR,
            // the register-row offset in an SGPR (32 separate 64-bit global addresses would cost 64 VGPRs)
            const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(A.slab + (size_t)blockIdx.x * (2 * 16 * W)), 0, 2 * 16 * W * (int)sizeof(C), 0x00020000);
            const unsigned voff_slab = (unsigned)(L.t * sizeof(C));
            {
                C y0[16], y2[16];
                chain_run<T, LOGM, R>(y0, L, A, rs_sig, rs_spec, 0, pb, ph);
                chain_run<T, LOGM, R>(y2, L, A, rs_sig, rs_spec, 2, pb, ph);
#pragma unroll
                for (int i = 0; i < 16; ++i) {  // a, b = y0 +- W_64^(2 i) y2'
                    C a, bb;
                    bfly_w(y0[i], y2[i], W64C[2 * i], W64S[2 * i], a, bb);
                    bstore(rs_slab, voff_slab, (unsigned)(i * W * sizeof(C)), a);
                    bstore(rs_slab, voff_slab, (unsigned)((16 + i) * W * sizeof(C)), bb);
                }
            }
            C y1[16], y3[16];
            chain_run<T, LOGM, R>(y1, L, A, rs_sig, rs_spec, 1, pb, ph);
            chain_run<T, LOGM, R>(y3, L, A, rs_sig, rs_spec, 3, pb, ph);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                // c' , d' = y1' +- W_64^(3i - i) y3' = y1' +- W_64^(2i) y3'   (z_r = W_64^(i r) y'_r)
                C cc, dd;
                bfly_w(y1[i], y3[i], W64C[2 * i], W64S[2 * i], cc, dd);
                const C a = bload(rs_slab, voff_slab, (unsigned)(i * W * sizeof(C)), (C *)nullptr);
                const C bb = bload(rs_slab, voff_slab, (unsigned)((16 + i) * W * sizeof(C)), (C *)nullptr);
                C c0, c2, c1, c3;
                bfly_w(a, cc, W64C[i], W64S[i], c0, c2);              // a +- W_64^i c'
                bfly_w(bb, dd, -W64S[i], W64C[i], c1, c3);           // b +- i W_64^i d'
                emit(0, i, c0);
                emit(1, i, c1);
                emit(2, i, c2);
                emit(3, i, c3);
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most four (a, b) pairs in flight
            }
        }
        // lags of block j all precede those of block j+1: init (0.0, lag 0) like mod.rs:143
        T best = T(0);
        uint32_t besti = 0u;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (bv[j] > best) { best = bv[j]; besti = (uint32_t)(L.t + W * bi[j] + M * j); }
        wave_arg_reduce_maxmin(best, besti);
        if (lane == 63) { sv[wave] = best; si[wave] = besti; }
        __syncthreads();
        if (L.t == 0) {
            T rb = sv[0];
            uint32_t ri = si[0];
            for (int w = 1; w < NW; ++w) arg_merge(rb, ri, sv[w], si[w]);
            A.row_idx[g] = ri;
            A.row_val[g] = rb;
        }
        __syncthreads();  // sv/si are rewritten by the next row
    }
}

}  // namespace caf
