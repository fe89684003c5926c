// kernels_small.hpp -- Doppler rows of SHORT inputs: n = 1 ... 512 samples (L = 2n = 2 ... 1024).
//
// xcor_rustfft.rs:2 promises "any power of two"; below n = 1024 a row is too small for a workgroup (the chain
// kernels of kernels_chain.hpp start at n = 1024) and used to take 2 log2(L) + 3 launches over HBM.  Here a row
// belongs to a GROUP OF LANES of one wave -- TPR = max(1, L / 16) lanes, 16 points per lane (L points when
// L < 16: one lane owns a whole row) -- and a workgroup carries many rows at once, everything in one launch:
//
//   u[i]  = conj(needle[i] * w^i), i < n; 0 for i >= n      (mixer mod.rs:46-65 + zero padding mod.rs:130;
//                                                            phasors in f64: w^tl (w^TPR)^i)
//   G     = IDFT_L(u) = conj(FFT_L(s))                      (positive exponent, unnormalised)
//   P[k]  = Hs[k] * G[k],  Hs = FFT_L(haystack ++ 0) / L    (xcor_rustfft.rs:64-73; Hs once per surface: k_small<.., true>)
//   c     = IDFT_L(P);  mag[k] = |c[k]|^2                   (xcor_rustfft.rs:76, mod.rs:147)
//   first-strictly-greater argmax over the row (mod.rs:143-151), surface store, row peak.
//
// Two kernels share this arithmetic (Stockham autosort passes, radix 16 while at least 16 points remain, then one
// radix-8 / 4 / 2 pass; all lanes of a row sit in one wave, so no exchange needs a workgroup barrier):
//   k_small       the first form: the row lives in two LDS buffers, every pass reads and writes LDS.  Still the
//                 kernel of L < 16 (n = 1, 2, 4) and of the haystack spectra of every n <= 512;
//   k_small_rows  (second half of this file) L = 16 ... 1024: the row lives in registers, LDS is only the
//                 exchange between two passes of a transform.  2-3.4x the first form's rate.
#pragma once
#include "kernels_chain.hpp"

namespace caf {

template <int LOGL>
struct SmallGeo {
    static constexpr int L = 1 << LOGL;
    static constexpr int N = L / 2;
    static constexpr int TPR = L >= 16 ? L / 16 : 1;  // lanes per row
    static constexpr int PT = L / TPR;                // points per lane (16, or L when L < 16)
    static constexpr int THREADS = 256;
    static constexpr int RPW = THREADS / TPR;         // rows per workgroup
    // LDS elements per row: two buffers of L points + one pad element, so that the lanes of a wave that own
    // DIFFERENT rows (TPR < 64) do not all hit the same bank (an unpadded stride of 2 L complex is a multiple of
    // the 256-byte bank period for every L >= 16: a 64-way conflict when one lane owns a row)
    static constexpr int ROWSTRIDE = 2 * L + 1;
};

template <typename T, int LOGL>
constexpr size_t small_lds_bytes()
{
    using G = SmallGeo<LOGL>;
    return ((size_t)G::L + (size_t)G::RPW * G::ROWSTRIDE) * sizeof(cpx<T>);
}

// in-place DFT_R (positive exponent, natural order) of v[0 .. R-1]
template <typename T, int R>
__device__ __forceinline__ void small_dft(cpx<T> (&v)[16])
{
    if constexpr (R == 16) {
        dft16(v);
    } else if constexpr (R == 8) {
        dft8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    } else if constexpr (R == 4) {
        dft4(v[0], v[1], v[2], v[3]);
    } else {
        const cpx<T> a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
}

// One Stockham pass of radix R over a row of L points: x -> y.  n_cur = current sub-transform length, s = L / n_cur.
//   butterfly (p, q), p < n_cur / R, q < s:  in_j = x[q + s (p + (n_cur / R) j)],  out_k = DFT_R(in)_k * W_ncur^(p k)
//   y[q + s (R p + k)] = out_k
template <typename T, int LOGL, int R, int NCUR>
__device__ __forceinline__ void small_pass(const cpx<T> *x, cpx<T> *y, const cpx<T> *twl, int tl)
{
    using G = SmallGeo<LOGL>;
    constexpr int L = G::L, S = L / NCUR, M = NCUR / R;  // M butterflies per sub-transform
    constexpr int NBF = (L / R) / G::TPR;                // butterflies per lane
#pragma unroll
    for (int b = 0; b < NBF; ++b) {
        const int bid = tl + b * G::TPR;
        const int p = bid / S, q = bid % S;
        cpx<T> v[16];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = x[q + S * (p + M * j)];
        small_dft<T, R>(v);
#pragma unroll
        for (int k = 0; k < R; ++k) {
            cpx<T> o = v[k];
            if (k && M > 1) o = cmul(o, twl[((p * k) * S) & (L - 1)]);  // W_ncur^(p k) = W_L^(p k s)
            y[q + S * (R * p + k)] = o;
        }
    }
}

template <int NCUR>
constexpr int small_radix() { return NCUR >= 16 ? 16 : NCUR; }

// all passes: returns the buffer that holds the result (x or y)
template <typename T, int LOGL, int NCUR = (1 << LOGL)>
__device__ __forceinline__ cpx<T> *small_idft(cpx<T> *x, cpx<T> *y, const cpx<T> *twl, int tl)
{
    if constexpr (NCUR == 1) {
        return x;
    } else {
        constexpr int R = small_radix<NCUR>();
        small_pass<T, LOGL, R, NCUR>(x, y, twl, tl);
        wave_lds_fence();
        return small_idft<T, LOGL, NCUR / R>(y, x, twl, tl);
    }
}

template <typename T>
struct SmallArgs {
    const cpx<T> *sig;   // prepare: haystack [batch][n]; rows: needle [batch][n]
    cpx<T> *spec;        // Hs [batch][L], natural order
    const cpx<T> *twL;   // [L]: e^{2 pi i m / L}
    const double *ph;    // [rows]: ((2 PI) f)(1 / fs) of this plan's rows (mod.rs:54-56)
    T *surface;          // [batch][rows][L] or nullptr
    uint64_t *row_idx;   // [batch][rows]
    T *row_val;          // [batch][rows]
    int rows;            // rows per surface handled by this plan
    int total;           // batch * rows (prepare: batch)
};

// PREP = true: Hs of `total` haystacks (w = 1, output conj(.)/L to spec); PREP = false: `total` Doppler rows
template <typename T, int LOGL, bool PREP>
__global__ __launch_bounds__(SmallGeo<LOGL>::THREADS) void k_small(const SmallArgs<T> A)
{
    using G = SmallGeo<LOGL>;
    using C = cpx<T>;
    constexpr int L = G::L, N = G::N, TPR = G::TPR, PT = G::PT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[small_lds_bytes<T, LOGL>()];
    C *const twl = reinterpret_cast<C *>(smem);
    const int tid = threadIdx.x, rw = tid / TPR, tl = tid % TPR;
    C *const bx = twl + L + (size_t)rw * G::ROWSTRIDE, *const by = bx + L;
    for (int i = tid; i < L; i += G::THREADS) twl[i] = A.twL[i];
    __syncthreads();
    const int ngroups = (A.total + G::RPW - 1) / G::RPW;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int g = grp * G::RPW + rw;  // this lane group's row (or haystack)
        const bool live = g < A.total;
        const int gc = live ? g : A.total - 1;  // idle groups redo the last row and store nothing
        const int bs = PREP ? gc : gc / A.rows, r = PREP ? 0 : gc - bs * A.rows;
        const C *__restrict__ sig = A.sig + (size_t)bs * N;
        const double ph = PREP ? 0.0 : A.ph[r];
        // ---- mixer + conjugation + zero padding -> bx
        // phasor of sample m = tl + TPR i:  w^m = w^tl (w^TPR)^i -- two f64 sincos per lane and row (base and step),
        // then i successive f64 multiplications (i < 16: error <= ~2e-15, an order below the reference's own
        // recurrence); rounded to T once per sample (never run the phasor in f32).  One sincos
        // per SAMPLE cost as much as both transforms of the row.
        double wr = 1.0, wi = 0.0, sr = 1.0, si = 0.0;
        if constexpr (!PREP) {
            sincos(ph * (double)tl, &wi, &wr);
            sincos(ph * (double)TPR, &si, &sr);
        }
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int m = tl + TPR * i;
            C u = C{T(0), T(0)};
            if (m < N) {
                const C a = sig[m];
                if constexpr (PREP) u = conj(a);
                else u = cmul_conj(a, C{(T)wr, (T)wi});
            }
            bx[m] = u;
            if constexpr (!PREP) {
                const double nr = wr * sr - wi * si, ni = wr * si + wi * sr;
                wr = nr;
                wi = ni;
            }
        }
        wave_lds_fence();
        C *res = small_idft<T, LOGL>(bx, by, twl, tl);
        C *oth = res == bx ? by : bx;
        if constexpr (PREP) {
            const T inv = T(1.0 / (double)L);
            if (live) {
#pragma unroll
                for (int i = 0; i < PT; ++i) {
                    const int k = tl + TPR * i;
                    A.spec[(size_t)bs * L + k] = C{res[k].x * inv, -res[k].y * inv};
                }
            }
            wave_lds_fence();
        } else {
            const C *__restrict__ hs = A.spec + (size_t)bs * L;
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int k = tl + TPR * i;
                res[k] = cmul(res[k], hs[k]);  // same lane wrote and reads position k: no fence needed before
            }
            wave_lds_fence();
            C *c = small_idft<T, LOGL>(res, oth, twl, tl);
            T bv = T(0);
            uint32_t bi = 0u;
            // the magnitudes go to the row's other (now free) buffer first and leave from there wave-wide: the
            // 64 / TPR rows of a wave are adjacent in the surface, so the wave stores one contiguous run with
            // consecutive lanes on consecutive lags (a lane storing its own row's values directly would write 4 or
            // 8 bytes at a stride of a whole row)
            C *const free_buf = c == bx ? by : bx;
            T *const mg = reinterpret_cast<T *>(free_buf);
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int k = tl + TPR * i;
                const T m = norm_sqr(c[k]);  // mod.rs:147
                mg[k] = m;
                if (m > bv) { bv = m; bi = (uint32_t)k; }  // first strictly greater (mod.rs:148-151); k ascends with i
            }
            if (A.surface) {
                wave_lds_fence();
                constexpr int RPWV = 64 / TPR;                     // rows per wave
                const int wrow0 = (tid >> 6) * RPWV;               // first row of this wave inside the workgroup
                const int g0 = grp * G::RPW + wrow0;               // ... and in the launch
                const T *const mg0 = reinterpret_cast<const T *>(twl + L + (size_t)wrow0 * G::ROWSTRIDE + (free_buf - bx));
                T *const out0 = A.surface + (size_t)g0 * L;
                const int lane = tid & 63;
#pragma unroll
                for (int e0 = 0; e0 < RPWV * L; e0 += 64) {
                    const int e = e0 + lane, rr = e / L, k = e % L;
                    if (g0 + rr < A.total) out0[e] = mg0[(size_t)rr * (G::ROWSTRIDE * (sizeof(C) / sizeof(T))) + k];
                }
            }
            // reduce over the row's TPR lanes (a power of two, aligned inside the wave); equal values keep the lower lag
#pragma unroll
            for (int msk = TPR >> 1; msk >= 1; msk >>= 1) {
                const T ov = shfl_xor_t<T>(bv, msk);
                const uint32_t oi = (uint32_t)__shfl_xor((int)bi, msk, 64);
                arg_merge(bv, bi, ov, oi);
            }
            if (live && tl == 0) {
                A.row_idx[g] = bi;
                A.row_val[g] = bv;
            }
            wave_lds_fence();  // the next group's mixer writes vs this group's reads of c
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Rows of L = 16 ... 1024 points with the transforms held in REGISTERS (round 3, second form).  k_small above
// (still the kernel of L < 16 and of the haystack spectra) moves a row through LDS eight times per row at
// L = 1024 -- mixer, three passes, spectrum product, three passes -- against an LDS whose STORES run at ~80 B/clk
// per CU (MI355X_MICROARCH.md section LDS), with a 64-lane-to-one-bank store pattern in the first pass of every
// transform (y[16 tl + k]) and 148 KiB of LDS per 256 threads: one wave per SIMD.  Here a lane keeps its 16
// points in registers from the needle samples to |.|^2:
//   * the mixer's own index pattern (sample tl + TPR j in slot j) IS the first pass' input pattern;
//   * the last pass of a Stockham transform is in place (butterfly q reads and writes q + S k), and the 16 / R
//     butterflies of a lane cover exactly tl + TPR j: the spectrum product and the second transform's first pass
//     take them from the registers they are in, and the second transform's outputs are the lags tl + TPR j;
//   * LDS is only the exchange between two passes of one transform (one for L = 32 ... 256, two for 512 / 1024;
//     none for L = 16): a single buffer per row, every element e at e + (e >> 4) and rows L + L / 16 apart, which
//     tools/lds_banks_small.py shows free of bank conflicts for every L and both dtypes;
//   * twiddles sit in LDS per pass as [k - 1][p], so that consecutive lanes read consecutive entries;
//   * the per-lane start phasor w^tl and the row's step w^TPR come from a table built with the plan by the same
//     f64 sincos the kernel used to run twice per lane and row.
// 512 threads: complex128 152 KiB of LDS (two waves per SIMD), complex64 76 KiB and <= 128 VGPRs (four).
// The arithmetic (butterflies, twiddle products, their order) is that of k_small: bit-identical surfaces.
template <int LOGL>
struct SmallRowGeo {
    static constexpr int L = 1 << LOGL, N = L / 2;
    static constexpr int TPR = L / 16;          // lanes per row
    static constexpr int THREADS = 512, WAVES = THREADS / 64;
    static constexpr int RPW = THREADS / TPR;   // rows per workgroup
    static constexpr int RPWV = 64 / TPR;       // rows per wave
    static constexpr int ROWX = L + L / 16;     // exchange row stride (elements)
    // |.|^2 staging rows (real elements) of the wave-wide surface store, tools/lds_banks_small.py
    template <typename T>
    static constexpr int ROWT = L >= 256 ? L : L + (sizeof(T) == 8 ? L / 16 : (L >= 32 ? L / 32 : 1));
    // twiddle entries of the pass with sub-transform length ncur and of all passes after it
    static constexpr int tw_entries(int ncur)
    {
        int n = 0;
        while (ncur > 1) {
            const int r = ncur >= 16 ? 16 : ncur, m = ncur / r;
            if (m > 1) n += (r - 1) * m;
            ncur /= r;
        }
        return n;
    }
    static constexpr int TWN = (tw_entries(L) + 15) & ~15;
    // the surface leaves straight from the registers when a row's lanes cover >= 128 contiguous bytes per store
    template <typename T>
    static constexpr bool DIRECT = TPR * sizeof(T) >= 128;
};

template <typename T, int LOGL>
constexpr size_t small_rows_lds_bytes()
{
    using G = SmallRowGeo<LOGL>;
    return ((size_t)G::TWN + (size_t)G::WAVES * G::RPWV * G::ROWX) * sizeof(cpx<T>);
}

__device__ __forceinline__ int small_pad(int e) { return e + (e >> 4); }

template <typename T, int R>
__device__ __forceinline__ void small_dft_at(cpx<T> (&v)[16], int o)
{
    if constexpr (R == 16) {
        dft16(v);
    } else if constexpr (R == 8) {
        dft8(v[o], v[o + 1], v[o + 2], v[o + 3], v[o + 4], v[o + 5], v[o + 6], v[o + 7]);
    } else if constexpr (R == 4) {
        dft4(v[o], v[o + 1], v[o + 2], v[o + 3]);
    } else {
        const cpx<T> a = v[o], b = v[o + 1];
        v[o] = a + b;
        v[o + 1] = a - b;
    }
}

// per-pass twiddle tables [k - 1][p] = W_L^(p k S), p < M, from the natural-order W_L table
template <typename T, int LOGL, int NCUR, int TWOFF>
__device__ __forceinline__ void small_rows_tw_fill(cpx<T> *tw, const cpx<T> *__restrict__ twL, int tid)
{
    using G = SmallRowGeo<LOGL>;
    constexpr int L = G::L, R = NCUR >= 16 ? 16 : NCUR, S = L / NCUR, M = NCUR / R;
    if constexpr (M > 1) {
        for (int i = tid; i < (R - 1) * M; i += G::THREADS) {
            const int k = i / M + 1, p = i % M;
            tw[TWOFF + i] = twL[(p * k * S) & (L - 1)];
        }
        small_rows_tw_fill<T, LOGL, NCUR / R, TWOFF + (R - 1) * M>(tw, twL, tid);
    }
}

// The passes of one transform from sub-transform length NCUR on.  In: v[b R + j] = x[q + S (p + M j)] of butterfly
// b (bid = tl + b TPR, p = bid / S, q = bid % S) -- for the first pass that is x[tl + TPR j] in slot j.  Out:
// slot j = X[tl + TPR j].  Same butterflies, twiddles and order of operations as small_pass.
template <typename T, int LOGL, int NCUR, int TWOFF>
__device__ __forceinline__ void small_rows_tf(cpx<T> (&v)[16], cpx<T> *xb, const cpx<T> *tw, int tl)
{
    using G = SmallRowGeo<LOGL>;
    using C = cpx<T>;
    constexpr int L = G::L, TPR = G::TPR, R = NCUR >= 16 ? 16 : NCUR, S = L / NCUR, M = NCUR / R, NBF = 16 / R;
#pragma unroll
    for (int b = 0; b < NBF; ++b) small_dft_at<T, R>(v, b * R);
    if constexpr (M > 1) {
#pragma unroll
        for (int b = 0; b < NBF; ++b) {
            const int p = (tl + b * TPR) / S;
#pragma unroll
            for (int k = 1; k < R; ++k) v[b * R + k] = cmul(v[b * R + k], tw[TWOFF + (k - 1) * M + p]);
        }
#pragma unroll
        for (int b = 0; b < NBF; ++b) {
            const int bid = tl + b * TPR, p = bid / S, q = bid % S;
#pragma unroll
            for (int k = 0; k < R; ++k) xb[small_pad(q + S * (R * p + k))] = v[b * R + k];
        }
        wave_lds_fence();
        constexpr int NC2 = NCUR / R, R2 = NC2 >= 16 ? 16 : NC2, S2 = L / NC2, M2 = NC2 / R2, NBF2 = 16 / R2;
#pragma unroll
        for (int b = 0; b < NBF2; ++b) {
            const int bid = tl + b * TPR, p = bid / S2, q = bid % S2;
#pragma unroll
            for (int j = 0; j < R2; ++j) v[b * R2 + j] = xb[small_pad(q + S2 * (p + M2 * j))];
        }
        wave_lds_fence();
        small_rows_tf<T, LOGL, NC2, TWOFF + (R - 1) * M>(v, xb, tw, tl);
    } else {
        // last pass (p = 0, q = bid, S = L / R): output k of butterfly b is X[tl + TPR (b + k NBF)]
        C t[16];
#pragma unroll
        for (int b = 0; b < NBF; ++b)
#pragma unroll
            for (int k = 0; k < R; ++k) t[b + k * NBF] = v[b * R + k];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = t[j];
    }
}

// w^tl (tl < TPR) and w^TPR of every row of a plan: [rows][TPR + 1]
__global__ __launch_bounds__(256) void k_small_phasors(const double *__restrict__ ph, int rows, int tpr, cpx<double> *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * (tpr + 1)) return;
    const int r = (int)(i / (tpr + 1)), t = (int)(i % (tpr + 1));
    double s, c;
    sincos(ph[r] * (double)t, &s, &c);
    out[i] = cpx<double>{c, s};
}

template <typename T, int LOGL>
__global__ __launch_bounds__(SmallRowGeo<LOGL>::THREADS, sizeof(T) == 8 ? 2 : 4) void k_small_rows(
    const SmallArgs<T> A, const cpx<double> *__restrict__ phz)
{
    using G = SmallRowGeo<LOGL>;
    using C = cpx<T>;
    constexpr int L = G::L, TPR = G::TPR, RPWV = G::RPWV, ROWT = G::template ROWT<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[small_rows_lds_bytes<T, LOGL>()];
    C *const tw = reinterpret_cast<C *>(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rwv = lane / TPR, tl = lane % TPR;
    C *const xw = tw + G::TWN + (size_t)wave * (RPWV * G::ROWX);  // this wave's rows
    C *const xb = xw + (size_t)rwv * G::ROWX;
    small_rows_tw_fill<T, LOGL, L, 0>(tw, A.twL, tid);
    __syncthreads();
    const int ngroups = (A.total + G::RPW - 1) / G::RPW;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int g0 = grp * G::RPW + wave * RPWV;  // first row of this wave
        const int g = g0 + rwv;
        const bool live = g < A.total;
        const int gc = live ? g : A.total - 1;  // idle groups redo the last row and store nothing
        const int bs = gc / A.rows, r = gc - bs * A.rows;
        const C *__restrict__ sig = A.sig + (size_t)bs * G::N;
        const C *__restrict__ hs = A.spec + (size_t)bs * L;
        // ---- mixer + conjugation + zero padding (mod.rs:46-65,130): slot j = conj(needle[m] w^m), m = tl + TPR j
        double wr, wi, sr, si;
        if (phz) {
            const cpx<double> w0 = phz[(size_t)r * (TPR + 1) + tl], st = phz[(size_t)r * (TPR + 1) + TPR];
            wr = w0.x; wi = w0.y; sr = st.x; si = st.y;
        } else {
            const double ph = A.ph[r];
            sincos(ph * (double)tl, &wi, &wr);
            sincos(ph * (double)TPR, &si, &sr);
        }
        C v[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // m < N <=> j < 8
            v[j] = cmul_conj(sig[tl + TPR * j], C{(T)wr, (T)wi});
            const double nr = wr * sr - wi * si, ni = wr * si + wi * sr;
            wr = nr;
            wi = ni;
        }
#pragma unroll
        for (int j = 8; j < 16; ++j) v[j] = C{T(0), T(0)};
        small_rows_tf<T, LOGL, L, 0>(v, xb, tw, tl);  // G = IDFT_L(u) = conj(FFT_L(s))
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = cmul(v[j], hs[tl + TPR * j]);  // xcor_rustfft.rs:64-73
        small_rows_tf<T, LOGL, L, 0>(v, xb, tw, tl);  // xcor_rustfft.rs:76
        T mg[16];
        T bv = T(0);
        uint32_t bi = 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            mg[j] = norm_sqr(v[j]);  // mod.rs:147
            if (mg[j] > bv) { bv = mg[j]; bi = (uint32_t)(tl + TPR * j); }  // first strictly greater (mod.rs:148-151): lags ascend with j
        }
        if (A.surface) {
            if constexpr (G::template DIRECT<T>) {
                if (live) {
                    T *const out = A.surface + (size_t)g * L + tl;
#pragma unroll
                    for (int j = 0; j < 16; ++j) out[TPR * j] = mg[j];
                }
            } else {
                // through the wave's (now idle) exchange rows, so that the wave stores contiguous runs: its RPWV rows
                // are adjacent in the surface
                T *const st = reinterpret_cast<T *>(xw);
#pragma unroll
                for (int j = 0; j < 16; ++j) st[rwv * ROWT + tl + TPR * j] = mg[j];
                wave_lds_fence();
                T *const out0 = A.surface + (size_t)g0 * L;
#pragma unroll
                for (int e0 = 0; e0 < RPWV * L; e0 += 64) {
                    const int e = e0 + lane, rr = e / L, k = e % L;
                    if (g0 + rr < A.total) out0[e] = st[rr * ROWT + k];
                }
                wave_lds_fence();  // the next group's exchange writes vs these reads
            }
        }
        // reduce over the row's TPR lanes (a power of two, aligned inside the wave); equal values keep the lower lag
#pragma unroll
        for (int msk = TPR >> 1; msk >= 1; msk >>= 1) {
            const T ov = shfl_xor_t<T>(bv, msk);
            const uint32_t oi = (uint32_t)__shfl_xor((int)bi, msk, 64);
            arg_merge(bv, bi, ov, oi);
        }
        if (live && tl == 0) {
            A.row_idx[g] = bi;
            A.row_val[g] = bv;
        }
    }
}

}  // namespace caf
