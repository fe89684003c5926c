// kernels_big65536.hpp -- four-step CAF rows for n = 32768 (L = 2n = 65536 = 256 x 256).
//
// BASELINE configs[3] (4096 Doppler rows x 65536 lags, complex64).  One padded row
// (512 KiB c64 / 1 MiB c128) does not fit a CU's 160 KiB of LDS, so the L-point transforms
// are done as 256 x 256 "four-step" transforms over an HBM/L2-resident work row
// X[r][c], index = 256*r + c, with three kernels per launch, each touching the row once:
//
//   k_big_cols_fwd : mixer (mod.rs:46-65) + conj + zero padding fused into the load;
//                    256-point transforms down the COLUMNS (over r), twiddle W_L^(k1*c)
//   k_big_rows     : 256-point transforms along the ROWS (over c), spectrum product with
//                    the pre-permuted H/L (xcor_rustfft.rs:64-73), and the inverse row
//                    transforms straight away (the data is still in registers)
//   k_big_cols_inv : twiddle, inverse column transforms, |.|^2 (mod.rs:147), surface store,
//                    per-tile argmax partials
// then k_big_rowpeak folds the 16 column-tile partials of a row (first-max, mod.rs:148-151).
// As in the 4096-sample kernels every transform is a positive-exponent one
// (conj(FFT(s)) = IDFT(conj s)), the forward is DIF and the inverse the mirrored DIT, so no
// reordering pass exists: H is stored in the register layout k_big_rows multiplies in.
//
// A workgroup (256 threads) handles a tile of 16 columns (cols kernels) or 16 rows (rows
// kernel) = 4096 points, 16 per lane; each 256-point transform is radix-16 x 16 with ONE
// LDS transpose (cross-wave for column tiles, inside 16 consecutive lanes for row tiles).
// Global accesses are 16 lanes x one complex = 128 B (c64) / 256 B (c128) segments.
// HBM traffic per row (c64): 512 KiB written + 512 KiB read/written + 512 KiB read
// + 256 KiB surface = 9x the algorithmic 256 KiB -- the price of L > LDS.
#pragma once
#include "../kernels_fused4096.hpp"

namespace caf {

constexpr int B_N = 32768;
constexpr int B_L = 65536;
constexpr int B_THREADS = 256;
constexpr int B_P = 272;  // LDS plane stride (elements): 16*16 + 16 pad

template <typename T>
struct BigArgs {
    const cpx<T> *sig;      // [batch][n] needle (prepare: haystack)
    const cpx<T> *phasor;   // [rows+1][384]: A[r] = e^{j*ph*256*r} (r<128), B[c] = e^{j*ph*c} (c<256); last row = 1
    const cpx<T> *w256;     // e^{2*pi*i*j/256}, j < 256
    const cpx<T> *wL;       // e^{2*pi*i*j/65536}, j < 256
    cpx<T> *work;           // [rows of one launch][65536] work rows, indexed by blockIdx.y (reused by every chunk)
    cpx<T> *spec;           // H/L in k_big_rows register layout: [batch][256 k1][16 kb][16 s]
    T *surface;             // [batch*rows][65536] or nullptr
    T *part_val;            // [batch*rows][16] per column-tile argmax partials
    uint32_t *part_idx;
    int rows;               // Doppler rows per surface in this plan
    int prepare;            // 1: haystack transform (phasor row = rows, one work row per surface)
    unsigned wr0;           // first work row of this launch
    unsigned nw;            // work rows of this launch; a workgroup loops over rows y = blockIdx.y, +gridDim.y, ...
};

template <typename T>
__global__ void k_big_tables(cpx<T> *__restrict__ w256, cpx<T> *__restrict__ wL)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < 256) {
        w256[j] = cispi_f64<T>(2.0 * (double)j / 256.0);
        wL[j] = cispi_f64<T>(2.0 * (double)j / 65536.0);
    }
}

template <typename T>
__global__ void k_big_phasors(const double *__restrict__ ph, int nrows, cpx<T> *__restrict__ tab)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g / 384, e = g % 384;
    if (row > nrows) return;
    const double p = row < nrows ? ph[row] : 0.0;
    const double idx = e < 128 ? 256.0 * (double)e : (double)(e - 128);
    tab[(size_t)row * 384 + e] = cis_f64<T>(p * idx);
}

// W_L^(p), p < 65536, from two 256-entry tables: W_256^(p>>8) * W_L^(p&255)
template <typename T>
__device__ __forceinline__ cpx<T> twiddle_L(const cpx<T> *w256, const cpx<T> *wL, unsigned p)
{
    return cmul(w256[p >> 8], wL[p & 255]);
}

// 256-point DIF: element index = s + 16*q (q = register) -> k = s + 16*kb (kb = register).
template <typename T, typename Ex>
__device__ __forceinline__ void dif256(cpx<T> (&v)[16], int s, const cpx<T> *w256, Ex &&transpose)
{
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], w256[(s * k) & 255]);
    transpose(v);
    dft16(v);
}
// Per-lane twiddles W_256^(s*k), k = 1..15: fetched ONCE per workgroup (a table gather per use
// made all three kernels TA-bound: ~47-60 gather instructions per lane and tile).
template <typename T>
struct Tw256 {
    cpx<T> w[16];
    __device__ __forceinline__ Tw256(const cpx<T> *w256, int s)
    {
#pragma unroll
        for (int k = 1; k < 16; ++k) w[k] = w256[(s * k) & 255];
    }
};
template <typename T, typename Ex>
__device__ __forceinline__ void dif256(cpx<T> (&v)[16], const Tw256<T> &tw, Ex &&transpose)
{
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw.w[k]);
    transpose(v);
    dft16(v);
}
template <typename T, typename Ex>
__device__ __forceinline__ void dit256(cpx<T> (&v)[16], const Tw256<T> &tw, Ex &&transpose)
{
    dft16(v);
    transpose(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tw.w[k]);
    dft16(v);
}
// mirrored DIT: k = s + 16*kb (kb = register) -> m = s + 16*q (q = register)
template <typename T, typename Ex>
__device__ __forceinline__ void dit256(cpx<T> (&v)[16], int s, const cpx<T> *w256, Ex &&transpose)
{
    dft16(v);
    transpose(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], w256[(s * k) & 255]);
    dft16(v);
}

// resident workgroups per CU the register allocator must leave room for (one wave of each per SIMD)
template <typename T>
constexpr int big_waves_per_simd() { return sizeof(T) == 4 ? 3 : 2; }

template <typename T>
constexpr size_t big_lds_bytes() { return 16 * B_P * sizeof(cpx<T>); }

// column tiles: lane = g + 16*s (g = column in tile, s = sub-index).  Transpose across waves:
// plane [k][16*s + g], stride 272.
template <typename T>
__device__ __forceinline__ void transpose_cols(cpx<T> (&v)[16], cpx<T> *lds, int g, int s)
{
    __syncthreads();  // previous readers of the planes are done
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[k * B_P + 16 * s + g] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds[s * B_P + 16 * k + g];
}
// row tiles: lane = s + 16*g (16 consecutive lanes share a row): wave-local, [g][k][17]
template <typename T>
__device__ __forceinline__ void transpose_rows(cpx<T> (&v)[16], cpx<T> *lds, int g, int s)
{
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 16; ++k) lds[g * B_P + k * 17 + s] = v[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = lds[g * B_P + s * 17 + k];
}

// ---- columns, forward: grid (16 column tiles, rows*batch) ------------------------------------
template <typename T>
__global__ __launch_bounds__(B_THREADS, big_waves_per_simd<T>()) void k_big_cols_fwd(const BigArgs<T> A)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[big_lds_bytes<T>()];
    C *const lds = reinterpret_cast<C *>(smem);
    const int g = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + g;
    // Persistent over work rows (same column tile): the per-lane twiddles of the 256-point
    // transform and of the four-step rotation W_L^(k1*c), k1 = s + 16*kb, live in registers.
    const Tw256<T> tws(A.w256, s);
    C twl[16];
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) twl[kb] = twiddle_L(A.w256, A.wL, (unsigned)((s + 16 * kb) * c));
    for (unsigned y = blockIdx.y; y < A.nw; y += gridDim.y) {
    const size_t wr = (size_t)y + A.wr0;  // work row = b*rows + r  (prepare: b)
    const size_t b = A.prepare ? wr : wr / A.rows;
    const int r = A.prepare ? A.rows : (int)(wr % A.rows);
    const C *sig = A.sig + b * B_N;
    const C *ph = A.phasor + (size_t)r * 384;
    const C pc = ph[128 + c];
    C v[16];
    // rows s + 16*q of the 256 x 256 image; rows >= 128 are the zero padding (mod.rs:130)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = s + 16 * q;
        if (q < 8)
            v[q] = conj(cmul(cmul(sig[256 * row + c], ph[row]), pc));
        else
            v[q] = C{T(0), T(0)};
    }
    dif256(v, tws, [&](C(&x)[16]) { transpose_cols(x, lds, g, s); });
    // four-step twiddle W_L^(k1*c), k1 = s + 16*kb, and store Y[k1][c]
    C *out = A.work + (size_t)y * B_L;  // work rows are chunk-local: the same 128 MiB is reused by every chunk
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        const int k1 = s + 16 * kb;
        out[256 * k1 + c] = cmul(v[kb], twl[kb]);
    }
    }
}

// ---- rows: forward, x H/L, inverse: grid (16 row tiles, rows*batch) -----------------------------
template <typename T>
__global__ __launch_bounds__(B_THREADS, big_waves_per_simd<T>()) void k_big_rows(const BigArgs<T> A)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[big_lds_bytes<T>()];
    C *const lds = reinterpret_cast<C *>(smem);
    const int s = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int k1 = blockIdx.x * 16 + g;
    const Tw256<T> tws(A.w256, s);
    for (unsigned y = blockIdx.y; y < A.nw; y += gridDim.y) {
    const size_t wr = (size_t)y + A.wr0;
    const size_t b = A.prepare ? wr : wr / A.rows;
    C *row = A.work + (size_t)y * B_L + 256 * k1;
    C v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = row[s + 16 * q];
    dif256(v, tws, [&](C(&x)[16]) { transpose_rows(x, lds, g, s); });
    // now v[kb] = Z[k1][k2 = s + 16*kb]; spectrum layout [k1][kb][s]
    C *spec = A.spec + b * B_L + 256 * k1;
    if (A.prepare) {
        const T inv = T(1.0 / 65536.0);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) spec[16 * kb + s] = {v[kb].x * inv, -v[kb].y * inv};  // conj(.)/L
        continue;
    }
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) v[kb] = cmul(v[kb], spec[16 * kb + s]);
    dit256(v, tws, [&](C(&x)[16]) { transpose_rows(x, lds, g, s); });
#pragma unroll
    for (int q = 0; q < 16; ++q) row[s + 16 * q] = v[q];
    }
}

// ---- columns, inverse + |.|^2 + argmax partials: grid (16 column tiles, rows*batch) ---------------
template <typename T>
__global__ __launch_bounds__(B_THREADS, big_waves_per_simd<T>() - 1) void k_big_cols_inv(const BigArgs<T> A)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[big_lds_bytes<T>()];
    __shared__ T s_v[4];
    __shared__ uint32_t s_i[4];
    C *const lds = reinterpret_cast<C *>(smem);
    const int g = threadIdx.x & 15, s = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + g;
    const Tw256<T> tws(A.w256, s);
    C twl[16];
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) twl[kb] = twiddle_L(A.w256, A.wL, (unsigned)((s + 16 * kb) * c));
    for (unsigned y = blockIdx.y; y < A.nw; y += gridDim.y) {
    const size_t wr = (size_t)y + A.wr0;
    const C *in = A.work + (size_t)y * B_L;
    C v[16];
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
        const int k1 = s + 16 * kb;
        v[kb] = cmul(in[256 * k1 + c], twl[kb]);
    }
    dit256(v, tws, [&](C(&x)[16]) { transpose_cols(x, lds, g, s); });
    // v[q] = y[256*(s + 16*q) + c]
    T bv = T(0);
    uint32_t bi = 0;
    T *out = A.surface ? A.surface + wr * B_L : nullptr;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t m = 256u * (uint32_t)(s + 16 * q) + (uint32_t)c;
        const T mag = norm_sqr(v[q]);  // mod.rs:147
        if (out) out[m] = mag;
        if (mag > bv) { bv = mag; bi = m; }  // m increases with q: strict '>' keeps the first
    }
    wave_arg_reduce_dpp(bv, bi);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) { s_v[wave] = bv; s_i[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bv = s_v[0];
        bi = s_i[0];
        for (int w = 1; w < 4; ++w) arg_merge(bv, bi, s_v[w], s_i[w]);
        A.part_val[wr * 16 + blockIdx.x] = bv;
        A.part_idx[wr * 16 + blockIdx.x] = bi;
    }
    __syncthreads();  // s_v / s_i are rewritten by the next row
    }
}

// fold the 16 column-tile partials of each row: one thread per row
template <typename T>
__global__ void k_big_rowpeak(const T *__restrict__ part_val, const uint32_t *__restrict__ part_idx, size_t nrows,
                              uint64_t *__restrict__ row_idx, T *__restrict__ row_val)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    T bv = T(0);
    uint32_t bi = 0;
    for (int t = 0; t < 16; ++t) arg_merge(bv, bi, part_val[r * 16 + t], part_idx[r * 16 + t]);
    row_idx[r] = bi;
    row_val[r] = bv;
}

}  // namespace caf
