// cplx.hpp -- minimal complex helpers for the gfx950 CAF kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace caf {

template <typename T>
struct alignas(2 * sizeof(T)) cpx {
    T x, y;
};

using cd = cpx<double>;
using cf = cpx<float>;

template <typename T>
__host__ __device__ __forceinline__ cpx<T> mk(T x, T y) { return cpx<T>{x, y}; }

template <typename T>
__device__ __forceinline__ cpx<T> operator+(cpx<T> a, cpx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T>
__device__ __forceinline__ cpx<T> operator-(cpx<T> a, cpx<T> b) { return {a.x - b.x, a.y - b.y}; }

// a*b : 2 mul + 2 fma
template <typename T>
__device__ __forceinline__ cpx<T> cmul(cpx<T> a, cpx<T> b)
{
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// a*conj(b)
template <typename T>
__device__ __forceinline__ cpx<T> cmulc(cpx<T> a, cpx<T> b)
{
    return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
template <typename T>
__device__ __forceinline__ cpx<T> conj(cpx<T> a) { return {a.x, -a.y}; }
// i*a
template <typename T>
__device__ __forceinline__ cpx<T> muli(cpx<T> a) { return {-a.y, a.x}; }
template <typename T>
__device__ __forceinline__ T norm_sqr(cpx<T> a) { return a.x * a.x + a.y * a.y; }  // mod.rs:147

// e^{j*ang}, angle in f64 radians, rounded once to T (never
// run the phasor in f32).
template <typename T>
__device__ __forceinline__ cpx<T> cis_f64(double ang)
{
    double s, c;
    sincos(ang, &s, &c);
    return {(T)c, (T)s};
}
// e^{j*pi*x}
template <typename T>
__device__ __forceinline__ cpx<T> cispi_f64(double x)
{
    double s, c;
    sincospi(x, &s, &c);
    return {(T)c, (T)s};
}

// "first strictly greater" argmax merge: larger value wins, equal values keep the
// LOWER index (mod.rs:148-151 scanned left to right from (0.0, idx 0)).
template <typename T>
__device__ __forceinline__ void arg_merge(T &bv, uint32_t &bi, T v, uint32_t i)
{
    const bool take = (v > bv) || (v == bv && i < bi);
    bv = take ? v : bv;
    bi = take ? i : bi;
}

template <typename T>
__device__ __forceinline__ T shfl_xor_t(T v, int m);
template <>
__device__ __forceinline__ double shfl_xor_t<double>(double v, int m) { return __shfl_xor(v, m, 64); }
template <>
__device__ __forceinline__ float shfl_xor_t<float>(float v, int m) { return __shfl_xor(v, m, 64); }

template <typename T>
__device__ __forceinline__ void wave_arg_reduce(T &bv, uint32_t &bi)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const T ov = shfl_xor_t<T>(bv, m);
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, m, 64);
        arg_merge(bv, bi, ov, oi);
    }
}

}  // namespace caf
