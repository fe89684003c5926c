// kernels_xcor.hpp -- xcor_rustfft::Xcor::run (xcor_rustfft.rs:51-78) as ONE launch for n = 2 ... 16384:
//     out[k] = sum_m a[(m + k) mod n] conj(b[m]) = IFFT_unnormalised( FFT(a) conj(FFT(b)) / n )
// FFT(x) = conj(IDFT(conj x)), so with X = IDFT_n(conj a), Y = IDFT_n(conj b):  out = IDFT_n( conj(X) Y / n )
// -- three positive-exponent transforms, the same butterflies the row kernels use:
//   n <= 1024          k_xcor_small: the Stockham passes of kernels_small.hpp, one lane group (n / 16 lanes of one wave)
//   2048 ... 16384     k_xcor_chain: the LDS-resident chain of kernels_chain.hpp (n / 16 threads, 16 points per thread;
//                      complex128 up to n = 8192: one chain must fit in LDS)
// a, b and out are read / written once each -- the host-pointer entry points hand in the device mappings of pinned
// staging buffers, so a call is one launch and one stream synchronise instead of 2 log2(n) + 3 launches over HBM.
#pragma once
#include "kernels_small.hpp"

namespace caf {

template <typename T, int LOGN>
__global__ __launch_bounds__(64) void k_xcor_small(const cpx<T> *__restrict__ a, const cpx<T> *__restrict__ b,
                                                   const cpx<T> *__restrict__ twN, cpx<T> *__restrict__ out)
{
    using G = SmallGeo<LOGN>;  // L of that geometry = n here: TPR = max(1, n / 16) lanes, PT points per lane
    using C = cpx<T>;
    constexpr int N = G::L, TPR = G::TPR, PT = G::PT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[5 * (size_t)N * sizeof(C)];
    C *const twl = reinterpret_cast<C *>(smem);
    C *const ax = twl + N, *const ay = ax + N, *const bx = ay + N, *const by = bx + N;
    const int tl = threadIdx.x;
    for (int i = tl; i < N; i += 64) twl[i] = twN[i];
    if (tl < TPR) {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int m = tl + TPR * i;
            ax[m] = conj(a[m]);
            bx[m] = conj(b[m]);
        }
    }
    __syncthreads();
    if (tl >= TPR) return;  // one lane group does the work (all of it inside one wave: no further barrier)
    const C *X = small_idft<T, LOGN>(ax, ay, twl, tl);
    C *Y = small_idft<T, LOGN>(bx, by, twl, tl);
    C *const free_b = Y == bx ? by : bx;
    const T inv = T(1.0 / (double)N);
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int k = tl + TPR * i;  // (this lane wrote position k of both results: no fence needed before)
        const C p = cmulc(Y[k], X[k]);  // Y conj(X)   (xcor_rustfft.rs:64-73: conj, multiply, divide by n)
        Y[k] = C{p.x * inv, p.y * inv};
    }
    wave_lds_fence();
    const C *c = small_idft<T, LOGN>(Y, free_b, twl, tl);  // xcor_rustfft.rs:76 (unnormalised inverse)
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        const int k = tl + TPR * i;
        out[k] = c[k];
    }
}

template <typename T, int LOGM>
__global__ __launch_bounds__(ChainGeo<LOGM>::W) void k_xcor_chain(const cpx<T> *__restrict__ a, const cpx<T> *__restrict__ b,
                                                                 const cpx<T> *__restrict__ twM, cpx<T> *__restrict__ out)
{
    using G = ChainGeo<LOGM>;
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[chain_lds_bytes<T, LOGM>()];
    const ChainLane<T, LOGM, 1> L(smem, twM);
    __syncthreads();
    const C one[1] = {C{T(1), T(0)}};
    C x[1][16], y[1][16];
#pragma unroll
    for (int q = 0; q < 16; ++q) x[0][q] = conj(a[L.t + G::W * q]);
    L.forward(x, one);  // X = IDFT(conj a), in the chain's register layout
    __syncthreads();    // the second transform's first exchange vs other waves' last reads of the first
#pragma unroll
    for (int q = 0; q < 16; ++q) y[0][q] = conj(b[L.t + G::W * q]);
    L.forward(y, one);  // Y = IDFT(conj b), same layout
    const T inv = T(1.0 / (double)G::M);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const C p = cmulc(y[0][k], x[0][k]);  // FFT(a) conj(FFT(b)) = conj(X) Y
        y[0][k] = C{p.x * inv, p.y * inv};
    }
    L.inverse(y, [&](int bb, int k, C v) { return twA_k(v, k, L.tw[bb]); });  // natural order out
#pragma unroll
    for (int i = 0; i < 16; ++i) out[L.t + G::W * i] = y[0][i];
}

}  // namespace caf
