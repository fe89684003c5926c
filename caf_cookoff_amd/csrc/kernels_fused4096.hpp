// kernels_fused4096.hpp -- what the n = 4096 (L = 2n = 8192) row kernels share: the table kernels of a plan
// (W_4096 twiddles, per-row Doppler phasors), the radix-16 butterfly in its FMA-optimised forms (dft4 / dft16 /
// dft16_sink, Linzer-Feig twiddle folding: TwSet, TwFold, twA_k), wave-local LDS fences and lane swaps, the argument
// block FusedArgs, the padded LDS geometry (F_BLK, F_CHAIN) and the buffer-descriptor loads / write-through stores.
//
// The decomposition every n = 4096 kernel uses (mod.rs:138-151 with FFT(haystack) hoisted; conj(FFT(s)) ==
// IDFT(conj s), so every transform is a positive-exponent one):
//   * The zero padding (mod.rs:130) makes the first radix-2 stage of the 8192-point transform free:
//       even bins = IDFT_4096(u[n]),  odd bins = IDFT_4096(u[n] e^{+2 pi i n / 8192});
//     the half-bin rotation is folded into the per-row Doppler phasor table, so the mixer costs nothing extra.
//   * each 4096-point transform is radix-16 x 16 x 16 with 16 points per lane in registers: forward DIF (natural in,
//     digit-reversed out), the product with the pre-permuted haystack spectrum in registers, the inverse the mirrored
//     DIT; of the four LDS exchanges per chain the two inner ones stay inside one wave (no barrier).
//   * inter-pass twiddles W_4096^(t k) live in registers for the lifetime of the persistent workgroup,
//     W_256^((t & 15) k) in a 256-entry LDS table.
// The row kernels themselves: kernels_seq4096.hpp (complex128 product), kernels_duo4096.hpp (complex64 product),
// kernels_surf4096.hpp (one launch per surface); the lane-half kernel this file was first written for lives in
// measure/kernels_lanehalf4096.hpp.
#pragma once
#include "cplx.hpp"

namespace caf {

constexpr int F_N = 4096;       // samples per input
constexpr int F_L = 8192;       // padded transform length
constexpr int F_THREADS = 512;  // 8 waves
constexpr int F_COLS = 256;     // butterfly columns per chain

// Row-independent tables (built once per context by k_fused_tables):
//   tw4096[m] = e^{+2*pi*i*m/4096}, m < 4096
//   th[t]     = e^{+2*pi*i*t/8192}, t < 512
template <typename T>
struct FusedTables {
    const cpx<T> *tw4096;
    const cpx<T> *th;
};

template <typename T>
__global__ void k_fused_tables(cpx<T> *__restrict__ tw4096, cpx<T> *__restrict__ th)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4096) tw4096[i] = cispi_f64<T>(2.0 * (double)i / 4096.0);
    if (i < 512) th[i] = cispi_f64<T>(2.0 * (double)i / 8192.0);
}

// Per-row phasor table of a plan: 64 entries per row (1 KiB f64),
//   [ 0..15] lo[j]    = e^{j*ph*j}
//   [16..31] hi[j]    = e^{j*ph*16*j}
//   [32..47] step0[q] = e^{j*ph*256*q}
//   [48..63] step1[q] = e^{j*ph*256*q} * e^{-2*pi*i*256*q/8192}      (odd chain)
// needle[t+256q]*lo[t&15]*hi[t>>4]*step_c[q]*(c ? e^{-2*pi*i*t/8192} : 1) is the mixer
// output (mod.rs:46-65) times the odd chain's half-bin rotation; the chain input is its
// conjugate.  Row `nrows` (one past the end) is the f = 0 row used to transform the
// haystack.  Every entry comes from one f64 sincos of the exact phase product.
template <typename T>
__global__ void k_fused_phasors(const double *__restrict__ ph, int nrows, cpx<T> *__restrict__ tab)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g >> 6, e = g & 63, j = e & 15, kind = e >> 4;
    if (row > nrows) return;
    const double p = row < nrows ? ph[row] : 0.0;
    const double mult = kind == 0 ? 1.0 : kind == 1 ? 16.0 : 256.0;
    double s, co, s2 = 0.0, c2 = 1.0;
    sincos(p * (mult * (double)j), &s, &co);
    if (kind == 3) sincospi(-2.0 * (double)(256 * j) / 8192.0, &s2, &c2);
    tab[(size_t)row * 64 + e] = {(T)(co * c2 - s * s2), (T)(co * s2 + s * c2)};
}

// ---- radix-16 butterfly, positive exponent, natural-order in and out -----------
template <typename T>
__device__ __forceinline__ void dft4(cpx<T> &a, cpx<T> &b, cpx<T> &c, cpx<T> &d)
{
    const cpx<T> apc = a + c, amc = a - c, bpd = b + d, bmd = b - d;
    a = apc + bpd;        // X0
    b = add_i(amc, bmd);  // X1 = x0 + i x1 - x2 - i x3
    c = apc - bpd;        // X2
    d = sub_i(amc, bmd);  // X3
}

template <typename T>
__device__ __forceinline__ cpx<T> mul_w8(cpx<T> a)  // * e^{i*pi/4}
{
    const T r = T(0.70710678118654752440084436210485);
    return {r * (a.x - a.y), r * (a.x + a.y)};
}
template <typename T>
__device__ __forceinline__ cpx<T> mul_w8_3(cpx<T> a)  // * e^{i*3*pi/4}
{
    const T r = T(0.70710678118654752440084436210485);
    return {-r * (a.x + a.y), r * (a.x - a.y)};
}

template <typename T>
__device__ __forceinline__ void swp(cpx<T> &a, cpx<T> &b) { const cpx<T> t = a; a = b; b = t; }

// v + r (i v) and r v + i v
template <typename T>
__device__ __forceinline__ cpx<T> lf_tan(cpx<T> v, T r) { return {vfma(-r, v.y, v.x), vfma(r, v.x, v.y)}; }
template <typename T>
__device__ __forceinline__ cpx<T> lf_cot(cpx<T> v, T r) { return {vfma(r, v.x, -v.y), vfma(r, v.y, v.x)}; }
// p = u + g b, m = u - g b for a real g
template <typename T>
__device__ __forceinline__ void axpy_pm(T g, cpx<T> b, cpx<T> u, cpx<T> &p, cpx<T> &m)
{
    p = {vfma(g, b.x, u.x), vfma(g, b.y, u.y)};
    m = {vfma(-g, b.x, u.x), vfma(-g, b.y, u.y)};
}
__device__ __forceinline__ void axpy_pm(float g, cpx<float> b, cpx<float> u, cpx<float> &p, cpx<float> &m)
{
    axpy_pm_const(g, b, u, p, m);  // g is a compile-time constant at every call site: SGPR operand (cplx.hpp)
}

// p = u + w v, m = u - w v for a unit constant w = c + i s known at compile time (after
// inlining), in six FMAs instead of a complex multiply + two complex adds (Linzer-Feig):
//   w v = g b,  g = c, b = v + i (s/c) v   if |c| >= |s|;   g = s, b = (c/s) v + i v   otherwise.
template <typename T>
__device__ __forceinline__ void bfly_w(cpx<T> u, cpx<T> v, double c, double s, cpx<T> &p, cpx<T> &m)
{
    if (s == 0.0) {  // w = +-1
        p = c < 0.0 ? u - v : u + v;
        m = c < 0.0 ? u + v : u - v;
        return;
    }
    if (c == 0.0) {  // w = +-i
        p = s < 0.0 ? sub_i(u, v) : add_i(u, v);
        m = s < 0.0 ? add_i(u, v) : sub_i(u, v);
        return;
    }
    cpx<T> b;
    T g;
    if (c * c >= s * s) {
        b = lf_tan(v, (T)(s / c));
        g = (T)c;
    } else {
        b = lf_cot(v, (T)(c / s));
        g = (T)s;
    }
    axpy_pm(g, b, u, p, m);
}

// dft4 of (a, w1 b, w2 c, w3 d) for unit constants with w3 = w1 * wr:  24 FMAs (20 when w2 and
// wr are +-i) instead of three complex multiplies + 16 adds.
template <typename T>
__device__ __forceinline__ void dft4_w(cpx<T> &a, cpx<T> &b, cpx<T> &c, cpx<T> &d, double c1, double s1, double c2,
                                       double s2, double cr, double sr)
{
    cpx<T> p, m, r, q;
    bfly_w(a, c, c2, s2, p, m);
    bfly_w(b, d, cr, sr, r, q);
    bfly_w(p, r, c1, s1, a, c);   // X0, X2
    bfly_w(m, q, -s1, c1, b, d);  // X1, X3 = m +- i w1 q
}

// second radix-4 stage of the 16-point butterfly for residue r0: inputs v[4 r0 + j] carry the
// constant twiddles W_16^(r0 j), folded into the butterflies
template <typename T>
__device__ __forceinline__ void dft16_stage2(cpx<T> (&v)[16], int r0)
{
    constexpr double C1 = 0.92387953251128675612818318939679;  // cos(pi/8)
    constexpr double S1 = 0.38268343236508977172845998403040;  // sin(pi/8)
    constexpr double R = 0.70710678118654752440084436210485;
    cpx<T> &a = v[4 * r0], &b = v[4 * r0 + 1], &c = v[4 * r0 + 2], &d = v[4 * r0 + 3];
    if (r0 == 0) dft4(a, b, c, d);
    if (r0 == 1) dft4_w(a, b, c, d, C1, S1, R, R, R, R);          // W16^1, W16^2, W16^3 = W16^1 W16^2
    if (r0 == 2) dft4_w(a, b, c, d, R, R, 0.0, 1.0, 0.0, 1.0);    // W16^2, W16^4, W16^6 = W16^2 W16^4
    if (r0 == 3) dft4_w(a, b, c, d, S1, C1, -R, R, -R, R);        // W16^3, W16^6, W16^9 = W16^3 W16^6
}

template <typename T>
__device__ __forceinline__ void dft16(cpx<T> (&v)[16])
{
    // stage 1: over q1 for each q0 (inputs v[q0 + 4*q1]) -> a[r0;q0] at v[q0 + 4*r0]
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) dft4(v[q0], v[q0 + 4], v[q0 + 8], v[q0 + 12]);
    // stage 2 (with the W16^(q0*r0) twiddles): over q0 for each r0 -> X[r0 + 4*r1] at v[4*r0 + r1]
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0) dft16_stage2(v, r0);
    // 4x4 transpose of the register names -> X[k] at v[k]
    swp(v[1], v[4]); swp(v[2], v[8]); swp(v[3], v[12]);
    swp(v[6], v[9]); swp(v[7], v[13]); swp(v[11], v[14]);
}

// dft16 whose outputs are handed to `sink(k, X[k])` as soon as each radix-4 group of the
// second stage retires: the caller's twiddle multiply + ds_write of four outputs can then sit
// between butterfly groups instead of all sixteen ds_write_b128 piling up behind the math
// (SQ_WAIT_INST_LDS in profiles/).  Measured effect: within noise, with or without a
// scheduling fence per group.
template <typename T, typename F>
__device__ __forceinline__ void dft16_sink(cpx<T> (&v)[16], F &&sink)
{
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) dft4(v[q0], v[q0 + 4], v[q0 + 8], v[q0 + 12]);
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0) {
        dft16_stage2(v, r0);  // X[r0 + 4*r1] at v[4*r0 + r1]
#pragma unroll
        for (int r1 = 0; r1 < 4; ++r1) sink(r0 + 4 * r1, v[4 * r0 + r1]);
    }
}

// the six held twiddles W_4096^(t*k), k in {1,2,3,4,8,12}
template <typename T>
struct TwSet {
    cpx<T> w1, w2, w3, w4, w8, w12;
};

// x * W_4096^(t*k) for a compile-time k (after unrolling): k = 4a + b -> W^(4a t) * W^(b t)
template <typename T>
__device__ __forceinline__ cpx<T> twA_k(cpx<T> x, int k, const TwSet<T> &w)
{
    const int a = k >> 2, b = k & 3;
    if (a == 1) x = cmul(x, w.w4);
    if (a == 2) x = cmul(x, w.w8);
    if (a == 3) x = cmul(x, w.w12);
    if (b == 1) x = cmul(x, w.w1);
    if (b == 2) x = cmul(x, w.w2);
    if (b == 3) x = cmul(x, w.w3);
    return x;
}

// Same twiddles with a lane-wide common factor c folded in: q[b] = W^(b t) * c (q[0] = c), so
// x * W^(t k) * c costs one multiply by W^(4a t) (a != 0) and one by q[b] -- the factor is free
// for 12 of the 16 elements and costs 3 + 4 multiplies per use instead of 16.
template <typename T>
struct TwFold {
    cpx<T> q[4];
    __device__ __forceinline__ TwFold(const TwSet<T> &w, cpx<T> c)
    {
        q[0] = c;
        q[1] = cmul(w.w1, c);
        q[2] = cmul(w.w2, c);
        q[3] = cmul(w.w3, c);
    }
};
template <typename T>
__device__ __forceinline__ cpx<T> twA_k(cpx<T> x, int k, const TwSet<T> &w, const TwFold<T> &f)
{
    const int a = k >> 2, b = k & 3;
    if (a == 1) x = cmul(x, w.w4);
    if (a == 2) x = cmul(x, w.w8);
    if (a == 3) x = cmul(x, w.w12);
    return cmul(x, f.q[b]);
}

template <typename T>
__device__ __forceinline__ void apply_twA(cpx<T> (&v)[16], const TwSet<T> &w)
{
    v[1] = cmul(v[1], w.w1);
    v[2] = cmul(v[2], w.w2);
    v[3] = cmul(v[3], w.w3);
    v[4] = cmul(v[4], w.w4);
    v[5] = cmul(cmul(v[5], w.w4), w.w1);
    v[6] = cmul(cmul(v[6], w.w4), w.w2);
    v[7] = cmul(cmul(v[7], w.w4), w.w3);
    v[8] = cmul(v[8], w.w8);
    v[9] = cmul(cmul(v[9], w.w8), w.w1);
    v[10] = cmul(cmul(v[10], w.w8), w.w2);
    v[11] = cmul(cmul(v[11], w.w8), w.w3);
    v[12] = cmul(v[12], w.w12);
    v[13] = cmul(cmul(v[13], w.w12), w.w1);
    v[14] = cmul(cmul(v[14], w.w12), w.w2);
    v[15] = cmul(cmul(v[15], w.w12), w.w3);
}

// Orders this wave's LDS accesses without a workgroup barrier: LDS operations of
// one wave execute in issue order, so a compiler-level fence is all that is needed
// for the wave-local exchanges (the 256-element block a wave owns is read and
// written by that wave only).
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lanes 32-63 of `a` <-> lanes 0-31 of `b`
__device__ __forceinline__ void swap32_u(unsigned &a, unsigned &b)
{
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swap32(double &a, double &b)
{
    unsigned alo = (unsigned)__double_as_longlong(a), ahi = (unsigned)(__double_as_longlong(a) >> 32);
    unsigned blo = (unsigned)__double_as_longlong(b), bhi = (unsigned)(__double_as_longlong(b) >> 32);
    swap32_u(alo, blo);
    swap32_u(ahi, bhi);
    a = __longlong_as_double(((long long)ahi << 32) | alo);
    b = __longlong_as_double(((long long)bhi << 32) | blo);
}
__device__ __forceinline__ void swap32(float &a, float &b)
{
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    swap32_u(ua, ub);
    a = __uint_as_float(ua);
    b = __uint_as_float(ub);
}
template <typename T>
__device__ __forceinline__ void swap32(cpx<T> &a, cpx<T> &b)
{
    swap32(a.x, b.x);
    swap32(a.y, b.y);
}

// W_32^i = e^{2*pi*i*i/32}, i < 8
__device__ constexpr double W32_COS[8] = {1.0, 0.98078528040323044912618223613424, 0.92387953251128675612818318939679,
                                          0.83146961230254523707878837761791, 0.70710678118654752440084436210485,
                                          0.55557023301960222474283081394853, 0.38268343236508977172845998403040,
                                          0.19509032201612826784828486847702};
__device__ constexpr double W32_SIN[8] = {0.0, 0.19509032201612826784828486847702, 0.38268343236508977172845998403040,
                                          0.55557023301960222474283081394853, 0.70710678118654752440084436210485,
                                          0.83146961230254523707878837761791, 0.92387953251128675612818318939679,
                                          0.98078528040323044912618223613424};

template <typename T>
struct FusedArgs {
    const cpx<T> *sig;      // prepare: haystack [batch][4096]; rows: needle [batch][4096]
    cpx<T> *spec;           // Hs [batch][2][16][256]: prepare writes, rows read
    const cpx<T> *phasor;   // [rows+1][64]  (k_fused_phasors)
    FusedTables<T> tab;
    T *surface;             // [batch][rows][8192] or nullptr
    uint64_t *row_idx;      // [batch][rows]
    T *row_val;             // [batch][rows]
    int rows;               // rows per surface handled by this plan
    int total;              // batch*rows (prepare: batch)
    unsigned long long *dbg;  // DIAG builds only: [iter][wave][F_NSTAMP] s_memtime stamps of workgroup 0
    unsigned *work;           // row-ticket counter of this launch (zeroed by the prepare kernel)
    // streaming slots: the haystack-spectrum kernel also stages the needles from the slot's pinned host
    // buffer (device mapping) into device memory, 16 bytes per thread and step (no separate copy node)
    const uint4 *stage_src;
    uint4 *stage_dst;
    unsigned stage_n16;
    unsigned fft_blocks;      // prepare: workgroups [0, fft_blocks) transform, the rest only stage needles in
};

// LDS geometry.  One chain = 16 blocks (one per high digit) of 256 elements, element j of
// a block stored at j + (j >> 4): one pad element per 16.  Every exchange pattern is then
// "one per-thread base + a compile-time offset" (ds_* 16-bit immediates, 3 base VGPRs in
// all) and every wave-level access is bank-conflict-free:
//   pattern A (by column):  pos(k, t)            = k*272 + t + (t>>4)          offset k*272
//   pattern B (gather n1):  pos(hi4, 16*k + lo4) = hi4*272 + lo4 + 17*k        offset k*17
//   pattern C (transposed): pos(hi4, 16*lo4 + k) = hi4*272 + 17*lo4 + k        offset k
constexpr int F_BLK = 272;             // padded block stride (elements)
constexpr int F_CHAIN = 16 * F_BLK;    // 4352 elements per chain
// LDS bytes: 2 padded chains + 256-entry W_256 table + argmax scratch
template <typename T>
constexpr size_t fused_lds_bytes() { return (2 * F_CHAIN + 256) * sizeof(cpx<T>) + 256; }

// ---- surface store: 16 bytes per lane, write-through ------------------------------------
// Adjacent lanes own adjacent lags; a quad_perm DPP swap gives every lane two consecutive
// lags of one register row, so a row leaves as 16-B-per-lane `sc1` stores: they go straight
// through L2 without keeping the line, which would otherwise evict the L2-resident inputs
// (needle, haystack spectrum, phasors) that every row re-reads (MI355X_MICROARCH.md, store
// flavours).
template <typename T>
__device__ __forceinline__ T dpp_xor1(T v);
template <>
__device__ __forceinline__ double dpp_xor1<double>(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0xB1, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <>
__device__ __forceinline__ float dpp_xor1<float>(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

typedef unsigned caf_v4u __attribute__((ext_vector_type(4)));
typedef unsigned caf_v2u __attribute__((ext_vector_type(2)));
constexpr int CAF_AUX_SC1 = 16;  // gfx940+ cache-policy bit 4 = sc1 (write-through, line not kept)

template <int AUX>
__device__ __forceinline__ void store_pair_aux(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, double x0, double x1)
{
    const long long a = __double_as_longlong(x0), b = __double_as_longlong(x1);
    caf_v4u d = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void store_pair_aux(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, float x0, float x1)
{
    caf_v2u d = {__float_as_uint(x0), __float_as_uint(x1)};
    __builtin_amdgcn_raw_buffer_store_b64(d, rs, byte_off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void store_one_aux(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, double x)
{
    const long long a = __double_as_longlong(x);
    caf_v2u d = {(unsigned)a, (unsigned)(a >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(d, rs, voff, soff, AUX);
}
template <int AUX>
__device__ __forceinline__ void store_one_aux(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float x)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), rs, voff, soff, AUX);
}
// Lane pairing for the 16-B (f64) / 8-B (f32) surface stores.  Every lane holds x0 = element of
// register row 2j and x1 = element of row 2j+1 at lag position t; lanes (2p, 2p+1) hold
// consecutive lags.  The even lane stores row 2j's pair (own x0, partner's x0), the odd lane row
// 2j+1's pair (partner's x1, own x1).  v_cndmask_b32_dpp does the neighbour fetch and the
// select in ONE instruction per dword: D = vcc ? src1 : dpp(src0).
// (s_nop 1: a DPP operand may have been written by the preceding VALU instruction; the
// compiler's hazard recogniser does not look inside inline asm.)
#define CAF_DPP_X1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
__device__ __forceinline__ void pair_xor1(double x0, double x1, unsigned long long even_mask,
                                          unsigned long long odd_mask, caf_v4u &d)
{
    const unsigned x0l = (unsigned)__double_as_longlong(x0), x0h = (unsigned)(__double_as_longlong(x0) >> 32);
    const unsigned x1l = (unsigned)__double_as_longlong(x1), x1h = (unsigned)(__double_as_longlong(x1) >> 32);
    unsigned a, b, c, e;
    asm("s_nop 1\n\t"
        "s_mov_b64 vcc, %[ev]\n\t"
        "v_cndmask_b32_dpp %[a], %[x1l], %[x0l], vcc " CAF_DPP_X1 "\n\t"
        "v_cndmask_b32_dpp %[b], %[x1h], %[x0h], vcc " CAF_DPP_X1 "\n\t"
        "s_mov_b64 vcc, %[od]\n\t"
        "v_cndmask_b32_dpp %[c], %[x0l], %[x1l], vcc " CAF_DPP_X1 "\n\t"
        "v_cndmask_b32_dpp %[e], %[x0h], %[x1h], vcc " CAF_DPP_X1
        : [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c), [e] "=&v"(e)
        : [x0l] "v"(x0l), [x0h] "v"(x0h), [x1l] "v"(x1l), [x1h] "v"(x1h), [ev] "s"(even_mask), [od] "s"(odd_mask)
        : "vcc");
    d = caf_v4u{a, b, c, e};
}
__device__ __forceinline__ void pair_xor1(float x0, float x1, unsigned long long even_mask,
                                          unsigned long long odd_mask, caf_v2u &d)
{
    unsigned a, c;
    asm("s_nop 1\n\t"
        "s_mov_b64 vcc, %[ev]\n\t"
        "v_cndmask_b32_dpp %[a], %[x1], %[x0], vcc " CAF_DPP_X1 "\n\t"
        "s_mov_b64 vcc, %[od]\n\t"
        "v_cndmask_b32_dpp %[c], %[x0], %[x1], vcc " CAF_DPP_X1
        : [a] "=&v"(a), [c] "=&v"(c)
        : [x0] "v"(__float_as_uint(x0)), [x1] "v"(__float_as_uint(x1)), [ev] "s"(even_mask), [od] "s"(odd_mask)
        : "vcc");
    d = caf_v2u{a, c};
}
template <int AUX>
__device__ __forceinline__ void store_vec_aux(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, caf_v4u d)
{
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, byte_off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void store_vec_aux(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, caf_v2u d)
{
    __builtin_amdgcn_raw_buffer_store_b64(d, rs, byte_off, 0, AUX);
}
template <typename T> struct pair_vec;
template <> struct pair_vec<double> { using type = caf_v4u; };
template <> struct pair_vec<float> { using type = caf_v2u; };

template <typename T>
__device__ __forceinline__ void store_pair_wt(__amdgpu_buffer_rsrc_t rs, unsigned byte_off, T x0, T x1)
{
    store_pair_aux<CAF_AUX_SC1>(rs, byte_off, x0, x1);
}

// 16-B (f64) / 8-B (f32) complex load through a buffer descriptor: per-lane byte offset in a
// VGPR, the 4096-byte-strided register-row offset in an SGPR -> one address VGPR per table.
__device__ __forceinline__ cpx<double> bload(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, cpx<double> *)
{
    const caf_v4u r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return {__longlong_as_double(((long long)r.y << 32) | r.x), __longlong_as_double(((long long)r.w << 32) | r.z)};
}
__device__ __forceinline__ cpx<float> bload(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, cpx<float> *)
{
    const caf_v2u r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return {__uint_as_float(r.x), __uint_as_float(r.y)};
}

// complex store through a buffer descriptor (default cache policy), twin of bload
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, cpx<double> x)
{
    const long long a = __double_as_longlong(x.x), b = __double_as_longlong(x.y);
    __builtin_amdgcn_raw_buffer_store_b128(caf_v4u{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)}, rs,
                                           voff, soff, 0);
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, cpx<float> x)
{
    __builtin_amdgcn_raw_buffer_store_b64(caf_v2u{__float_as_uint(x.x), __float_as_uint(x.y)}, rs, voff, soff, 0);
}

// two ADJACENT complex values per lane: one 16-byte load in complex64, two in complex128.
// (No store twin on purpose: a 16-byte buffer store whose data registers are rewritten by inline-asm
// VALU code within two wait states stores garbage -- the compiler's hazard recogniser does not see into
// inline asm; see the slab stores of kernels_chain.hpp.)
template <int AUX = 0>
__device__ __forceinline__ void bload2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, cpx<float> &x0, cpx<float> &x1)
{
    const caf_v4u r = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX);
    x0 = {__uint_as_float(r.x), __uint_as_float(r.y)};
    x1 = {__uint_as_float(r.z), __uint_as_float(r.w)};
}
template <int AUX = 0>
__device__ __forceinline__ void bload2(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, cpx<double> &x0, cpx<double> &x1)
{
    const caf_v4u r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, AUX);
    const caf_v4u r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16, soff, AUX);
    x0 = {__longlong_as_double(((long long)r0.y << 32) | r0.x), __longlong_as_double(((long long)r0.w << 32) | r0.z)};
    x1 = {__longlong_as_double(((long long)r1.y << 32) | r1.x), __longlong_as_double(((long long)r1.w << 32) | r1.z)};
}

}  // namespace caf
