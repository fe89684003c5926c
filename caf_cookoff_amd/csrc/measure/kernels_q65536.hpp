// kernels_q65536.hpp -- 16 x 4096 decomposition of the n = 32768 (L = 65536) CAF row.
//
// BASELINE configs[3] again, with TWO passes over memory instead of the three of
// kernels_big65536.hpp.  L = 16 M, M = 4096, positive-exponent transforms throughout
// (conj(FFT(s)) = IDFT(conj s), as in the 4096-sample kernels), W_N = e^{+2 pi i / N}:
//
//   u[n]  = conj(needle[n] w^n), n < 32768 = 8 M; zero beyond (mod.rs:130)
//   n = M n1 + n2,  k = k1 + 16 k2,  m = m2 + M m1
//   y_k1[n2] = sigma^n2 * sum_{n1<8} conj(needle[M n1 + n2]) rho^n1,
//              sigma = W_L^k1 conj(w),  rho = conj(w)^M W_16^k1           (8 terms: the padding is free)
//   G[k1 + 16 k2] = DFT_M(y_k1)[k2]                                        (one 4096-point chain in LDS)
//   z_k1[m2]      = DFT_M(G[k1 + 16 .] Hs[k1 + 16 .])[m2],  Hs = conj(DFT_L(conj haystack))/L
//   c[m2 + M m1]  = sum_k1 W_16^(m1 k1) * ( W_L^(m2 k1) z_k1[m2] )         (one 16-point butterfly per lag m2)
//
//   k_q_rows : per (row, group of RES residues k1): the eight needle segments are read once and
//              accumulated into RES chain inputs (scalar rho^n1 coefficients), then each residue runs
//              the same 4096-point forward / spectrum product / inverse chain in LDS as the
//              400 x 8192 row kernels (kernels_duo4096.hpp DuoIo) and stores W_L^(m2 k1) z_k1[m2]
//              -> work[row][k1][m2]                                   (512 KiB written per row in c64)
//   k_q_cols : per lag pair: 16-point butterfly over k1, |.|^2 (mod.rs:147), surface store, argmax
//              partials                                              (512 KiB read, 256 KiB written)
// The haystack spectrum is the same k_q_rows front half with w = 1 (prepare = 1).
//
// Status: measurement variant (CAF_BIG_PATH=1), parity-green in both types.  k_q_cols (32 us per 256
// rows) beats the three-pass form's last kernel (46 us), but k_q_rows needs 140 us where the first two
// passes of kernels_big65536.hpp take 84 us: sixteen 4096-point LDS chains per row cost more than
// two passes of 256-point transforms, and each task re-reads the eight needle segments
// (RES = 2 residues per task: no spills; RES = 4: 73 spills, 180 us).  2.63 vs 2.19 ms per
// 4096 x 65536 surface, so the three-pass form stays the product path.
#pragma once
#include "kernels_big65536.hpp"
#include "../kernels_duo4096.hpp"

namespace caf {

constexpr int Q_M = 4096;

template <typename T>
struct QArgs {
    const cpx<T> *sig;     // [batch][32768] needle (prepare: haystack)
    const cpx<T> *tw4096;  // W_4096^m, m < 4096
    const cpx<T> *outw;    // [16 k1][256 t]: W_L^(k1 t)
    cpx<T> *work;          // [rows of one launch][16 k1][4096 m2]
    cpx<T> *spec;          // Hs in the chain's register layout: [batch][16 k1][16 k][256 t]
    T *surface;            // [batch*rows][65536] or nullptr
    T *part_val;           // [batch*rows][8] argmax partials of k_q_cols
    uint32_t *part_idx;
    int rows;              // Doppler rows per surface in this plan
    int prepare;           // 1: haystack transform (phasor row = rows, one task row per surface)
    unsigned wr0;          // first work row of this launch
    unsigned nw;           // work rows of this launch
};

// Per (row, residue k1) table of 64 entries:
//   [0..15]  sigma^j          [16..31] sigma^(16 j)        [32..47] sigma^(256 q)
//   [48..55] rho^n1           [56..63] unused (1)
// Row `nrows` (one past the end) is the w = 1 row of the haystack transform.  Every entry is one
// f64 sincos / sincospi pair of the exact phase, rounded once to T.
template <typename T>
__global__ void k_q_phasors(const double *__restrict__ ph, int nrows, cpx<T> *__restrict__ tab)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t row = gid >> 10;
    const int k1 = (int)(gid >> 6) & 15, e = (int)gid & 63;
    if (row > (size_t)nrows) return;
    const double p = row < (size_t)nrows ? ph[row] : 0.0;
    double mult = 0.0, rat = 0.0, wmul = 0.0;  // phase = 2 pi rat - p wmul
    if (e < 48) {
        mult = e < 16 ? (double)e : e < 32 ? 16.0 * (double)(e - 16) : 256.0 * (double)(e - 32);
        rat = mult * (double)k1 / 65536.0;
        wmul = mult;
    } else if (e < 56) {
        mult = (double)(e - 48);
        rat = mult * (double)k1 / 16.0;
        wmul = 4096.0 * mult;
    }
    double s1, c1, s2, c2;
    sincospi(2.0 * rat, &s1, &c1);
    sincos(-p * wmul, &s2, &c2);
    tab[gid] = {(T)(c1 * c2 - s1 * s2), (T)(c1 * s2 + s1 * c2)};
}

template <typename T>
__global__ void k_q_tables(cpx<T> *__restrict__ outw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // k1*256 + t
    if (i < 4096) outw[i] = cispi_f64<T>(2.0 * (double)((i >> 8) * (i & 255)) / 65536.0);
}

template <typename T>
__device__ __forceinline__ void load_tw(TwSet<T> &tw, const cpx<T> *__restrict__ tw4096, const SeqLane &L)
{
    tw.w1 = tw4096[L.t * 1];
    tw.w2 = tw4096[L.t * 2];
    tw.w3 = tw4096[L.t * 3];
    tw.w4 = tw4096[L.t * 4];
    tw.w8 = tw4096[L.t * 8];
    tw.w12 = tw4096[L.t * 12];
}

// residues per workgroup task: the eight needle segments are read once per task
template <typename T>
constexpr int q_res() { return sizeof(T) == 4 ? 2 : 1; }

template <typename T, int RES>
__global__ __launch_bounds__(S_THREADS, seq_waves_per_simd<T>()) void k_q_rows(const QArgs<T> A,
                                                                           const cpx<T> *__restrict__ phasor)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[seq_lds_bytes<T>()];
    C *const Lc = reinterpret_cast<C *>(smem);
    C *const twb = Lc + F_CHAIN;
    const SeqLane L;
    const cpx<T> *__restrict__ const tw4096 = A.tw4096;
    TwSet<T> tw;
    load_tw(tw, tw4096, L);
    twb[L.tid] = tw4096[16 * (L.tid & 15) * (L.tid >> 4)];
    const DuoIo<T> io{Lc, twb + L.lo4, L};
    __syncthreads();
    constexpr unsigned G = 16 / RES;
    const unsigned ntask = A.nw * G;
    for (unsigned task = blockIdx.x; task < ntask; task += gridDim.x) {
        const unsigned y = task / G, grp = task % G;
        const size_t wr = (size_t)y + A.wr0;
        const size_t b = A.prepare ? wr : wr / (size_t)A.rows;
        const int r = A.prepare ? A.rows : (int)(wr % (size_t)A.rows);
        const C *__restrict__ sig = A.sig + b * B_N;
        const C *__restrict__ tab = phasor + ((size_t)r * 16 + grp * RES) * 64;
        // ---- chain inputs of RES residues from one pass over the needle (mixer, mod.rs:46-65) --
        C acc[RES][16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const C x = conj(sig[L.t + 256 * q]);
#pragma unroll
            for (int i = 0; i < RES; ++i) acc[i][q] = x;
        }
#pragma unroll
        for (int n1 = 1; n1 < 8; ++n1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const C x = conj(sig[Q_M * n1 + L.t + 256 * q]);
#pragma unroll
                for (int i = 0; i < RES; ++i) acc[i][q] = cfma(acc[i][q], x, tab[i * 64 + 48 + n1]);
            }
        }
#pragma unroll
        for (int i = 0; i < RES; ++i) {
            const int k1 = (int)grp * RES + i;
            const C *__restrict__ ti = tab + i * 64;
            C v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = cmul(acc[i][q], ti[32 + q]);
            {
                const TwFold<T> f(tw, cmul(ti[L.lo4], ti[16 + L.hi4]));  // sigma^t on the stage twiddles
                dft16(v);
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = twA_k(v[k], k, tw, f);
            }
            io.write_A(v);
            __syncthreads();
            io.read_B(v);
            dft16(v);
            io.mul_twB(v);
            io.write_B(v);
            wave_lds_fence();
            io.read_C(v);
            dft16(v);
            C *spec = A.spec + (b * 16 + (size_t)k1) * (16 * 256);
            if (A.prepare) {  // Hs = conj(.)/L in this register layout
                const T inv = T(1.0 / 65536.0);
#pragma unroll
                for (int k = 0; k < 16; ++k) spec[k * 256 + L.t] = {v[k].x * inv, -v[k].y * inv};
                __syncthreads();  // other waves still read their blocks; the next residue writes into them
                continue;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = cmul(v[k], spec[k * 256 + L.t]);  // xcor_rustfft.rs:64-73
            dft16(v);
            wave_lds_fence();
            io.write_C(v);
            wave_lds_fence();
            io.read_B(v);
            io.mul_twB(v);
            dft16(v);
            wave_lds_fence();
            io.write_B(v);
            __syncthreads();
            io.read_A(v);
            {
                const TwFold<T> f(tw, A.outw[k1 * 256 + L.t]);  // W_L^(k1 t) of the output rotation
#pragma unroll
                for (int k = 0; k < 16; ++k) v[k] = twA_k(v[k], k, tw, f);
            }
            dft16(v);
            C *out = A.work + (size_t)y * B_L + (size_t)k1 * Q_M + L.t;
            out[0] = v[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) out[256 * k] = cmul(v[k], tw4096[(16 * k1 * k) & 4095]);  // W_256^(k1 k)
            // the next residue's pattern-A write touches only the addresses this thread has just read
        }
    }
}

// ---- 16-point butterfly over the residues + |.|^2 + argmax partials: 8 tiles of 512 lags per row ----
template <typename T>
__global__ __launch_bounds__(256) void k_q_cols(const QArgs<T> A)
{
    using C = cpx<T>;
    __shared__ T s_v[4];
    __shared__ uint32_t s_i[4];
    const unsigned ntask = A.nw * 8;
    for (unsigned task = blockIdx.x; task < ntask; task += gridDim.x) {
        const unsigned y = task >> 3, tile = task & 7;
        const size_t wr = (size_t)y + A.wr0;
        const uint32_t m2 = tile * 512 + 2 * threadIdx.x;
        const C *in = A.work + (size_t)y * B_L + m2;
        C v0[16], v1[16];
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            v0[k1] = in[k1 * Q_M];
            v1[k1] = in[k1 * Q_M + 1];
        }
        dft16(v0);
        dft16(v1);
        T bv = T(0);
        uint32_t bi = 0;
        T *out = A.surface ? A.surface + wr * B_L + m2 : nullptr;
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) {  // lag m = m2 + 4096 m1: increasing with m1, then within the pair
            const T a0 = norm_sqr(v0[m1]), a1 = norm_sqr(v1[m1]);  // mod.rs:147
            if (out) {
                out[Q_M * m1] = a0;
                out[Q_M * m1 + 1] = a1;
            }
            if (a0 > bv) { bv = a0; bi = m2 + (uint32_t)(Q_M * m1); }
            if (a1 > bv) { bv = a1; bi = m2 + 1 + (uint32_t)(Q_M * m1); }
        }
        wave_arg_reduce_dpp(bv, bi);
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 63) { s_v[wave] = bv; s_i[wave] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            bv = s_v[0];
            bi = s_i[0];
            for (int w = 1; w < 4; ++w) arg_merge(bv, bi, s_v[w], s_i[w]);
            A.part_val[wr * 8 + tile] = bv;
            A.part_idx[wr * 8 + tile] = bi;
        }
        __syncthreads();  // s_v / s_i are rewritten by the next task
    }
}

// fold the 8 lag-tile partials of each row (first maximum wins, mod.rs:148-151): one thread per row
template <typename T>
__global__ void k_q_rowpeak(const T *__restrict__ part_val, const uint32_t *__restrict__ part_idx, size_t nrows,
                            uint64_t *__restrict__ row_idx, T *__restrict__ row_val)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    T bv = T(0);
    uint32_t bi = 0;
    for (int t = 0; t < 8; ++t) arg_merge(bv, bi, part_val[r * 8 + t], part_idx[r * 8 + t]);
    row_idx[r] = bi;
    row_val[r] = bv;
}

}  // namespace caf
