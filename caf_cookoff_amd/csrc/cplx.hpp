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

__device__ __forceinline__ double vfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double vmax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_fmaxf(a, b); }

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

// DPP flavour of the same reduction (no LDS-pipe traffic): afterwards lane 63 holds the
// wave's (max value, lowest index among equals); other lanes hold partial results.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_t(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL, ROW_MASK>((int)b), hi = dpp_i<CTRL, ROW_MASK>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_t(float v)
{
    return __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(v)));
}
template <typename T, int CTRL, int ROW_MASK>
__device__ __forceinline__ void arg_step(T &bv, uint32_t &bi)
{
    const T ov = dpp_t<CTRL, ROW_MASK>(bv);
    const uint32_t oi = (uint32_t)dpp_i<CTRL, ROW_MASK>((int)bi);
    arg_merge(bv, bi, ov, oi);  // lanes not written by the DPP keep (bv, bi): merge with self
}
template <typename T>
__device__ __forceinline__ void wave_arg_reduce_dpp(T &bv, uint32_t &bi)
{
    arg_step<T, 0xB1, 0xF>(bv, bi);   // quad_perm [1,0,3,2]
    arg_step<T, 0x4E, 0xF>(bv, bi);   // quad_perm [2,3,0,1]
    arg_step<T, 0x141, 0xF>(bv, bi);  // row_half_mirror
    arg_step<T, 0x140, 0xF>(bv, bi);  // row_mirror      -> every lane of a 16-row holds the row result
    arg_step<T, 0x142, 0xA>(bv, bi);  // row_bcast15 into rows 1,3
    arg_step<T, 0x143, 0xC>(bv, bi);  // row_bcast31 into rows 2,3 -> lane 63 has the wave result
}

}  // namespace caf
