// kernels_small.hpp -- Doppler rows of SHORT inputs: n = 1 ... 512 samples (L = 2n = 2 ... 1024).
//
// xcor_rustfft.rs:2 promises "any power of two"; below n = 1024 a row is too small for a workgroup (the chain
// kernels of kernels_chain.hpp start at n = 1024) and used to take 2 log2(L) + 3 launches over HBM.  Here a row
// belongs to a GROUP OF LANES of one wave -- TPR = max(1, L / 16) lanes, 16 points per lane (L points when
// L < 16: one lane owns a whole row) -- and a 256-thread workgroup carries 256 / TPR rows at once.  Everything
// between the needle samples and the |.|^2 values happens in LDS and registers, in one launch:
//
//   u[i]  = conj(needle[i] * w^i), i < n; 0 for i >= n      (mixer mod.rs:46-65 + zero padding mod.rs:130;
//                                                            phasors in f64: w^tl (w^TPR)^i, two sincos per lane and row)
//   G     = IDFT_L(u) = conj(FFT_L(s))                      (positive exponent, unnormalised)
//   P[k]  = Hs[k] * G[k],  Hs = FFT_L(haystack ++ 0) / L    (xcor_rustfft.rs:64-73; Hs once per surface: k_small_prepare)
//   c     = IDFT_L(P);  mag[k] = |c[k]|^2                   (xcor_rustfft.rs:76, mod.rs:147)
//   first-strictly-greater argmax over the row (mod.rs:143-151), surface store, row peak.
//
// The transform is a Stockham autosort FFT (natural order in and out) between two LDS buffers of the row, radix
// 16 while at least 16 points remain, then one radix-8 / 4 / 2 pass; a lane does 16 / R butterflies per pass.
// All lanes of a row sit in one wave, so the exchanges between passes need no workgroup barrier (LDS operations
// of a wave execute in order).  Twiddles come from a W_L table in LDS (one per workgroup).
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

}  // namespace caf
