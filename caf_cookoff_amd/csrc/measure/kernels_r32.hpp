// kernels_r32.hpp -- BASELINE configs[3] (n = 32768 complex64, L = 65536 = 4 chains of M = 16384 points) with
// THIRTY-TWO points per thread: the structural alternative to k_chain_rows<float, 14, 4> that VERDICT r02 item 4
// asked for.  Same mathematics and chain decomposition as kernels_chain.hpp; what changes is the stage plan
//
//     16384 = 32 x 32 x 16      (k_chain_rows: 16 x 16 x 16 x 4)
//
// run by 512 threads (8 waves, 2 per SIMD, 256 VGPRs) instead of 1024 (16 waves, 128 VGPRs): TWO LDS exchanges
// per 16384-point transform instead of three (-33 % LDS bytes, the resource that does not overlap with the
// butterflies because all waves of the one resident workgroup run the same phase), two twiddle passes instead
// of three, and with 64 VGPRs per 32-point complex64 array there is room to keep `a = y0 + W y2` in registers
// across chains 1 and 3, so that only `b = y0 - W y2` makes the round trip through the per-workgroup slab
// (128 KiB per row each way instead of 256).
//
// Stage plan of one transform (forward, decimation in frequency; the inverse is the mirrored DIT):
//   A  thread t < 512 holds u[t + 512 q], q < 32: radix-32 over q, twiddle W_M^(t k), write block k
//   -- workgroup barrier --
//   B  thread (g, o) = (t >> 4, t & 15) holds block g's elements o + 16 j, j < 32: radix-32 over j, twiddle
//      W_512^(o k'), in place
//   -- wave-local (a wave owns blocks 4 w ... 4 w + 3) --
//   C  two groups of 16 contiguous elements per thread: radix-16, no twiddle
// LDS geometry: block g at 560 g, element x of a block at x + (x >> 4): pattern A = 560 k + t + (t >> 4),
// pattern B = 560 g + o + 17 j, pattern C = 560 g + 17 k' + o'.  560 = 512 + 32 pads + 16: 560 mod 32 = 16
// puts the two blocks a half-wave touches in patterns B and C on disjoint bank halves.
#pragma once
#include "../kernels_chain.hpp"

namespace caf {

constexpr int W_LOGM = 14, W_M = 1 << W_LOGM, W_R = 4, W_L = W_R * W_M;
constexpr int W_T = 512;            // threads
constexpr int W_BLK = 560;          // padded block stride (elements)
constexpr int W_CHAIN = 32 * W_BLK; // 17920 elements = 140 KiB in complex64
constexpr int W_PH = 192;           // phasor-table entries per row
constexpr size_t r32_lds_bytes() { return (size_t)(W_CHAIN + 512) * sizeof(cpx<float>) + 256; }

// twM[m] = e^{2 pi i m / M};  th[(r - 1) 512 + t] = e^{2 pi i t r / L}, r = 1 .. 3
template <typename T>
__global__ void k_r32_tables(cpx<T> *__restrict__ twM, cpx<T> *__restrict__ th)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W_M) twM[i] = cispi_f64<T>(2.0 * (double)i / (double)W_M);
    if (i < 3 * W_T) {
        const int r = i / W_T + 1, t = i % W_T;
        th[i] = cispi_f64<T>(2.0 * (double)t * (double)r / (double)W_L);
    }
}

// per-row phasors (row `nrows` = the f = 0 row of the haystack transform), every entry one f64 sincos of the exact
// phase (mod.rs:54-56):
//   [0..15] w^j   [16..31] w^(16 j)   [32..47] w^(256 j)   [48 + 32 r + q] w^(512 q) e^{-2 pi i q r / 128}   [176] w^M
template <typename T>
__global__ void k_r32_phasors(const double *__restrict__ ph, int nrows, cpx<T> *__restrict__ tab)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g / W_PH, e = g % W_PH;
    if (row > nrows) return;
    const double p = row < nrows ? ph[row] : 0.0;
    double mult = 0.0, rot = 0.0;
    if (e < 16) mult = (double)e;
    else if (e < 32) mult = 16.0 * (double)(e - 16);
    else if (e < 48) mult = 256.0 * (double)(e - 32);
    else if (e < 176) {
        const int r = (e - 48) >> 5, q = (e - 48) & 31;
        mult = 512.0 * (double)q;
        rot = (double)(q * r) / 128.0;
    } else if (e == 176) mult = (double)W_M;
    double s, c, s2, c2;
    sincos(p * mult, &s, &c);
    sincospi(-2.0 * rot, &s2, &c2);
    tab[(size_t)row * W_PH + e] = {(T)(c * c2 - s * s2), (T)(c * s2 + s * c2)};
}

// radix-32 butterfly, positive exponent, natural order in and out: two radix-16 butterflies + W_32 combination
template <typename T>
__device__ __forceinline__ void dft32(cpx<T> (&v)[32])
{
    cpx<T> e[16], o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { e[k] = v[2 * k]; o[k] = v[2 * k + 1]; }
    dft16(e);
    dft16(o);
#pragma unroll
    for (int k = 0; k < 16; ++k) bfly_w(e[k], o[k], W32C16[k], W32S16[k], v[k], v[k + 16]);
}

// the ten held twiddles W_M^(t k), k in {1, 2, 3, 4, 8, ..., 28}; x * W^(t k) * c for k = 4 a + b is one multiply by
// W^(4 a t) (a != 0) and one by q[b] = W^(b t) c
template <typename T>
struct Tw10 {
    cpx<T> w1, w2, w3, wa[8];  // wa[a] = W^(4 a t), a = 1 .. 7
};
template <typename T>
struct TwFold10 {
    cpx<T> q[4];
    __device__ __forceinline__ TwFold10(const Tw10<T> &w, cpx<T> c)
    {
        q[0] = c;
        q[1] = cmul(w.w1, c);
        q[2] = cmul(w.w2, c);
        q[3] = cmul(w.w3, c);
    }
};
template <typename T>
__device__ __forceinline__ cpx<T> tw32_k(cpx<T> x, int k, const Tw10<T> &w, const TwFold10<T> &f)
{
    const int a = k >> 2, b = k & 3;
    if (a) x = cmul(x, w.wa[a]);
    return cmul(x, f.q[b]);
}

template <typename T>
struct R32Args {
    const cpx<T> *sig;   // prepare: haystack [batch][n]; rows: needle [batch][n]   (n = 32768)
    cpx<T> *spec;        // Hs [batch][4][16 register pairs][512][2]
    const cpx<T> *twM;   // [M]
    const cpx<T> *th;    // [3][512]
    T *surface;          // [batch][rows][L] or nullptr
    uint64_t *row_idx;
    T *row_val;
    cpx<T> *slab;        // [gridDim.x][2][32][512]: b = y0 - W y2 and a = y0 + W y2 of the row in flight
    int rows, total;
};

template <typename T, int ABL = 0>
struct R32Lane {
    using C = cpx<T>;
    int t, g, o, pA, pB, pC;
    C *Lc;
    const C *twB;  // LDS [k'][o] = W_512^(o k')
    Tw10<T> tw;
    __device__ __forceinline__ R32Lane(unsigned char *smem, const C *__restrict__ twM)
    {
        t = threadIdx.x;
        g = t >> 4;
        o = t & 15;
        Lc = reinterpret_cast<C *>(smem);
        C *tab = Lc + W_CHAIN;
        twB = tab + o;
        pA = t + (t >> 4);
        pB = g * W_BLK + o;
        pC = ((t >> 6) * 4 + ((t & 63) >> 4)) * W_BLK + 17 * (t & 15);  // block 4 w + (lane >> 4), group k' = lane & 15 (and + 16)
        tw.w1 = twM[t];
        tw.w2 = twM[2 * t];
        tw.w3 = twM[3 * t];
#pragma unroll
        for (int a = 1; a < 8; ++a) tw.wa[a] = twM[4 * a * t];
        tw.wa[0] = C{T(1), T(0)};
        for (int i = t; i < 512; i += W_T) tab[i] = twM[(i & 15) * (i >> 4) * (W_M / 512)];  // [k' = i >> 4][o = i & 15]
    }
    // forward DIF: v[q] = u[t + 512 q] without the lane-common factor `lane` (folded into the stage-A twiddles)
    __device__ __forceinline__ void forward(C (&v)[32], C lane) const
    {
        {
            const TwFold10<T> f(tw, lane);
            dft32(v);
#pragma unroll
            for (int k = 0; k < 32; ++k) Lc[k * W_BLK + pA] = tw32_k(v[k], k, tw, f);
        }
        if constexpr (ABL & 16) wave_lds_fence(); else __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = Lc[pB + 17 * j];
        dft32(v);
#pragma unroll
        for (int k = 0; k < 32; ++k) Lc[pB + 17 * k] = k ? cmul(v[k], twB[16 * k]) : v[k];
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 16; ++j) v[16 * c + j] = Lc[pC + 272 * c + j];
        dft16(reinterpret_cast<C(&)[16]>(v[0]));
        dft16(reinterpret_cast<C(&)[16]>(v[16]));
    }
    // inverse DIT (mirror): v in the forward's output layout -> v[i] = post-twiddled y[t + 512 i]
    __device__ __forceinline__ void inverse(C (&v)[32], C post) const
    {
        dft16(reinterpret_cast<C(&)[16]>(v[0]));
        dft16(reinterpret_cast<C(&)[16]>(v[16]));
        wave_lds_fence();  // the forward's reads of these positions (program order inside the wave)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 16; ++j) Lc[pC + 272 * c + j] = v[16 * c + j];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const C x = Lc[pB + 17 * k];
            v[k] = k ? cmul(x, twB[16 * k]) : x;
        }
        dft32(v);
#pragma unroll
        for (int j = 0; j < 32; ++j) Lc[pB + 17 * j] = v[j];
        if constexpr (ABL & 16) wave_lds_fence(); else __syncthreads();
        {
            const TwFold10<T> f(tw, post);
#pragma unroll
            for (int k = 0; k < 32; ++k) v[k] = tw32_k(Lc[k * W_BLK + pA], k, tw, f);
        }
        dft32(v);
    }
};

template <typename T>
__device__ __forceinline__ cpx<T> r32_pb(const cpx<T> *__restrict__ ph, int t)
{
    return cmul(cmul(ph[t & 15], ph[16 + ((t >> 4) & 15)]), ph[32 + (t >> 8)]);
}

// chain inputs of the pair (rA, rA + 2) from one pass over the needle: x_rA = a0 + s b, x_(rA+2) = a0 - s b,
// b = w^M a1, s = (-i)^rA; v_r[q] = conj(x_r[q] step_r[q])
// ABL (timing only, wrong results): 1 no slab traffic, 2 no surface stores, 4 no spectrum loads, 8 no needle loads, 16 no barriers
// PF (software pipelining; 2 waves per SIMD hide little latency, but there are registers to spare):
//   1 the chain's 32 haystack-spectrum values are requested BEFORE its forward transform
//   2 no scheduling fences in the epilogue (slab loads hoisted as far as the registers allow)
//   4 needle samples in groups of eight rows instead of four (sixteen loads in flight ahead of their use)
// PARK: vA / vB still hold a = y0 + W y2 and b = y0 - W y2 of the first chain pair; each register row is stored to the
// workgroup's slab right before the second pair's input overwrites it, BEHIND the needle loads of the rows that
// follow -- the stores then drain under this input stage and the next forward transform instead of in front of them
// (vector memory operations retire in order: a load issued after 64 stores waits for all of them).
template <typename T, int ABL = 0, int PF = 0, bool PARK = false>
__device__ __forceinline__ void r32_input_pair(cpx<T> (&vA)[32], cpx<T> (&vB)[32], const __amdgpu_buffer_rsrc_t rs_sig, int rA,
                                               int t, const cpx<T> *__restrict__ ph, const __amdgpu_buffer_rsrc_t rs_slab)
{
    using C = cpx<T>;
    const C *psA = ph + 48 + 32 * rA, *psB = ph + 48 + 32 * (rA + 2);
    C k1 = ph[176];
    if (rA) k1 = C{k1.y, -k1.x};  // b = (-i)^rA w^M a1  ->  x_rA = a0 + b, x_(rA+2) = a0 - b   (rA is wave-uniform)
    const unsigned voff = (unsigned)(t * sizeof(C));
    constexpr int GQ = (PF & 4) ? 8 : 4;
    C a[2][GQ][2];
    auto fetch = [&](int grp) {
#pragma unroll
        for (int u = 0; u < GQ; ++u) {
            const int q = GQ * grp + u;
            if constexpr (ABL & 8) {
                a[grp & 1][u][0] = C{T(q + 1), T(t)};
                a[grp & 1][u][1] = C{T(t), T(q + 2)};
                keep(a[grp & 1][u][0]);
                keep(a[grp & 1][u][1]);
                continue;
            }
            a[grp & 1][u][0] = bload(rs_sig, voff, (unsigned)((W_T * q) * sizeof(C)), (C *)nullptr);
            a[grp & 1][u][1] = bload(rs_sig, voff, (unsigned)((W_M + W_T * q) * sizeof(C)), (C *)nullptr);
        }
    };
    fetch(0);
#pragma unroll
    for (int grp = 0; grp < 32 / GQ; ++grp) {
        if (grp < 32 / GQ - 1) fetch(grp + 1);
#pragma unroll
        for (int u = 0; u < GQ; ++u) {
            const int q = GQ * grp + u;
            const C x = a[grp & 1][u][0], b = cmul(a[grp & 1][u][1], k1);
            if constexpr (PARK && !(ABL & 1)) {
                if constexpr (!(ABL & 32)) bstore(rs_slab, voff, (unsigned)((32 + q) * W_T * sizeof(C)), vA[q]);
                bstore(rs_slab, voff, (unsigned)(q * W_T * sizeof(C)), vB[q]);
            }
            vA[q] = cmul_conj(x + b, psA[q]);
            vB[q] = cmul_conj(x - b, psB[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one chain: input v -> y'_r[i] = th_r(t) IDFT_M(Hs G_r)[t + 512 i]
template <typename T, int ABL = 0, int PF = 0>
__device__ __forceinline__ void r32_chain(cpx<T> (&v)[32], const R32Lane<T, ABL> &L, const R32Args<T> &A,
                                          const __amdgpu_buffer_rsrc_t rs_spec, int r, cpx<T> pb)
{
    using C = cpx<T>;
    C post = C{T(1), T(0)}, lane = pb;
    if (r) {
        post = A.th[(r - 1) * W_T + L.t];
        lane = cmulc(lane, post);
    }
    const unsigned voff = (unsigned)((r * 32 * W_T + 2 * L.t) * sizeof(C));
    if constexpr (PF & 1) {
        C h[32];
#pragma unroll
        for (int k = 0; k < 32; k += 2) {
            if constexpr (ABL & 4) { h[k] = C{T(1), T(k)}; h[k + 1] = C{T(k), T(1)}; keep(h[k]); keep(h[k + 1]); continue; }
            bload2(rs_spec, voff, (unsigned)(2 * W_T * (k / 2) * sizeof(C)), h[k], h[k + 1]);
        }
        L.forward(v, conj(lane));
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = cmul(v[k], h[k]);  // xcor_rustfft.rs:64-73
    } else {
        L.forward(v, conj(lane));
#pragma unroll
        for (int grp = 0; grp < 4; ++grp) {  // four groups of eight spectrum values (xcor_rustfft.rs:64-73)
            C h[8];
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                if constexpr (ABL & 4) { h[k] = C{T(1), T(k)}; h[k + 1] = C{T(k), T(1)}; keep(h[k]); keep(h[k + 1]); continue; }
                bload2(rs_spec, voff, (unsigned)(2 * W_T * (4 * grp + k / 2) * sizeof(C)), h[k], h[k + 1]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[8 * grp + k] = cmul(v[8 * grp + k], h[k]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    L.inverse(v, post);
}

// haystack spectrum: one workgroup per (surface, chain)
template <typename T>
__global__ __launch_bounds__(W_T) void k_r32_prepare(const R32Args<T> A, const cpx<T> *__restrict__ phasor)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[r32_lds_bytes()];
    const R32Lane<T> L(smem, A.twM);
    const C *__restrict__ ph = phasor + (size_t)A.rows * W_PH;  // the f = 0 row
    const T inv = T(1.0 / (double)W_L);
    __syncthreads();
    for (int w = blockIdx.x; w < W_R * A.total; w += gridDim.x) {
        const int b = __builtin_amdgcn_readfirstlane(w / W_R), r = __builtin_amdgcn_readfirstlane(w % W_R);
        const C *__restrict__ sig = A.sig + (size_t)b * (W_L / 2);
        const C *ps = ph + 48 + 32 * r;
        C lane = r32_pb(ph, L.t);
        if (r) lane = cmulc(lane, A.th[(r - 1) * W_T + L.t]);
        C v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const C x = sig[L.t + W_T * q], bb = sig[W_M + L.t + W_T * q];  // w = 1: w^M = 1
            const C xr = r == 0 ? x + bb : r == 1 ? sub_i(x, bb) : r == 2 ? x - bb : add_i(x, bb);
            v[q] = cmul_conj(xr, ps[q]);
        }
        L.forward(v, conj(lane));
        C *spec = A.spec + ((size_t)b * W_R + r) * (32 * W_T);
#pragma unroll
        for (int k = 0; k < 32; ++k) spec[((k >> 1) * W_T + L.t) * 2 + (k & 1)] = {v[k].x * inv, -v[k].y * inv};
        __syncthreads();
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
#define CAF_R32_ATTR __attribute__((target("no-load-store-opt")))
#else
#define CAF_R32_ATTR
#endif
template <typename T, int ABL = 0, int PF = 0>
__global__ CAF_R32_ATTR __launch_bounds__(W_T, 2) void k_r32_rows(const R32Args<T> A, const cpx<T> *__restrict__ phasor)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[r32_lds_bytes()];
    const R32Lane<T, ABL> L(smem, A.twM);
    unsigned char *const scratch = smem + r32_lds_bytes() - 256;
    T *const sv = reinterpret_cast<T *>(scratch);
    uint32_t *const si = reinterpret_cast<uint32_t *>(scratch + 128);
    const int lane = L.t & 63, wave = L.t >> 6;
    __syncthreads();
    for (int g = blockIdx.x; g < A.total; g += gridDim.x) {
        const int bs = __builtin_amdgcn_readfirstlane(g / A.rows);
        const int r_row = __builtin_amdgcn_readfirstlane(g - bs * A.rows);
        const C *__restrict__ ph = phasor + (size_t)r_row * W_PH;
        const __amdgpu_buffer_rsrc_t rs_sig = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.sig + (size_t)bs * (W_L / 2)), 0, (W_L / 2) * (int)sizeof(C), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.spec + (size_t)bs * W_R * 32 * W_T), 0, W_R * 32 * W_T * (int)sizeof(C), 0x00020000);
        T *const out = A.surface ? A.surface + (size_t)g * W_L : nullptr;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? W_L * (int)sizeof(T) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.slab + (size_t)blockIdx.x * (64 * W_T)), 0, 64 * W_T * (int)sizeof(C), 0x00020000);
        const C pb = r32_pb(ph, L.t);
        const unsigned tvo = (unsigned)(L.t * sizeof(C));

        // The two chain pairs (0, 2) and (1, 3) are iterations of a run-time loop and the two chains of a pair
        // iterations of an inner one (registers swapped between them): the chain code exists once (the fully
        // unrolled row is ~20 k instructions = 160 KB against a 64 KB instruction cache shared by two CUs).
        C cur[32], oth[32];
        r32_input_pair<T, ABL, PF>(cur, oth, rs_sig, 0, L.t, ph, rs_slab);
#pragma clang loop unroll(disable)
        for (int it = 0; it < 2; ++it) {
            const int rp = __builtin_amdgcn_readfirstlane(it);
#pragma clang loop unroll(disable)
            for (int c = 0; c < 2; ++c) {
                r32_chain<T, ABL, PF>(cur, L, A, rs_spec, __builtin_amdgcn_readfirstlane(rp + 2 * c), pb);
#pragma unroll
                for (int i = 0; i < 32; ++i) { const C x = cur[i]; cur[i] = oth[i]; oth[i] = x; }
            }
            if (it == 0) {  // cur = y'_0, oth = y'_2  ->  a = y0 + W y2, b = y0 - W y2 in place, parked by the next input stage
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    C s, d;
                    bfly_w(cur[i], oth[i], W128C[2 * i], W128S[2 * i], s, d);
                    cur[i] = s;
                    oth[i] = d;
                }
                r32_input_pair<T, ABL, PF, true>(cur, oth, rs_sig, 1, L.t, ph, rs_slab);
            }
        }
        C (&y1)[32] = cur, (&y3)[32] = oth;
        T bv[4] = {T(0), T(0), T(0), T(0)};
        int bi[4] = {0, 0, 0, 0};
        auto emit = [&](int j, int i, C c) {
            const T m = norm_sqr(c);  // mod.rs:147
            bi[j] = m > bv[j] ? i : bi[j];  // first strictly greater (mod.rs:148-151): i ascends inside a block
            bv[j] = vmax(bv[j], m);
            if constexpr (ABL & 2) { asm volatile("" ::"v"(m)); return; }
            store_one_aux<CAF_AUX_SC1>(rs_out, (unsigned)(L.t * sizeof(T)), (unsigned)((W_M * j + W_T * i) * sizeof(T)), m);
        };
        // the parked a, b come back in groups of four lags, TWO groups ahead of their use (two waves per SIMD do not hide
        // a round trip to L2 / the Infinity Cache per group)
        constexpr int EG = 4, NG = 32 / EG;
        C pa[3][EG], pbv[3][EG];
        auto eload = [&](int grp) {
#pragma unroll
            for (int u = 0; u < EG; ++u) {
                const int i = EG * grp + u;
                if constexpr (ABL & 1) {
                    pa[grp % 3][u] = C{T(1), T(i)};
                    pbv[grp % 3][u] = C{T(i), T(1)};
                    keep(pa[grp % 3][u]);
                    keep(pbv[grp % 3][u]);
                } else {
                    if constexpr (ABL & 32) { pa[grp % 3][u] = C{T(1), T(i)}; keep(pa[grp % 3][u]); }  // timing only: half the slab traffic
                    else pa[grp % 3][u] = bload(rs_slab, tvo, (unsigned)((32 + i) * W_T * sizeof(C)), (C *)nullptr);
                    pbv[grp % 3][u] = bload(rs_slab, tvo, (unsigned)(i * W_T * sizeof(C)), (C *)nullptr);
                }
            }
        };
        eload(0);
        eload(1);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (grp + 2 < NG) eload(grp + 2);
#pragma unroll
            for (int u = 0; u < EG; ++u) {
                const int i = EG * grp + u;
                C cc, dd, c0, c1, c2, c3;
                bfly_w(y1[i], y3[i], W128C[2 * i], W128S[2 * i], cc, dd);
                bfly_w(pa[grp % 3][u], cc, W128C[i], W128S[i], c0, c2);
                bfly_w(pbv[grp % 3][u], dd, -W128S[i], W128C[i], c1, c3);
                emit(0, i, c0);
                emit(1, i, c1);
                emit(2, i, c2);
                emit(3, i, c3);
            }
            if (!(PF & 2)) __builtin_amdgcn_sched_barrier(0);
        }
        T best = T(0);
        uint32_t besti = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (bv[j] > best) { best = bv[j]; besti = (uint32_t)(L.t + W_T * bi[j] + W_M * j); }
        wave_arg_reduce_maxmin(best, besti);
        if (lane == 63) { sv[wave] = best; si[wave] = besti; }
        __syncthreads();
        if (L.t == 0) {
            T rb = sv[0];
            uint32_t ri = si[0];
            for (int w = 1; w < W_T / 64; ++w) arg_merge(rb, ri, sv[w], si[w]);
            A.row_idx[g] = ri;
            A.row_val[g] = rb;
        }
        __syncthreads();
    }
}

}  // namespace caf
