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
__device__ __forceinline__ cpx<T> cmul_conj(cpx<T> a, cpx<T> b) { return {a.x * b.x - a.y * b.y, -(a.x * b.y + a.y * b.x)}; }
// acc + a*b
template <typename T>
__device__ __forceinline__ cpx<T> cfma(cpx<T> acc, cpx<T> a, cpx<T> b)
{
    return {acc.x + (a.x * b.x - a.y * b.y), acc.y + (a.x * b.y + a.y * b.x)};
}
template <typename T>
__device__ __forceinline__ cpx<T> conj(cpx<T> a) { return {a.x, -a.y}; }
// a + i b, a - i b
template <typename T>
__device__ __forceinline__ cpx<T> add_i(cpx<T> a, cpx<T> b) { return {a.x - b.y, a.y + b.x}; }
template <typename T>
__device__ __forceinline__ cpx<T> sub_i(cpx<T> a, cpx<T> b) { return {a.x + b.y, a.y - b.x}; }
// i*a
template <typename T>
__device__ __forceinline__ cpx<T> muli(cpx<T> a) { return {-a.y, a.x}; }
template <typename T>
__device__ __forceinline__ T norm_sqr(cpx<T> a) { return a.x * a.x + a.y * a.y; }  // mod.rs:147

// ---- packed-f32 forms (complex64 path) -------------------------------------------------------
// A cpx<float> lives in an even-aligned VGPR pair, and gfx950's v_pk_{add,mul,fma}_f32 process
// both halves in one issue slot.  The compiler selects them for plain element-wise arithmetic and
// for broadcasts, but not for operands whose halves are SWAPPED (the cross terms of a complex
// multiply, multiplication by +-i): it builds the swapped pair with v_xor + v_mov instead.  The
// overloads below spell those with op_sel/op_sel_hi/neg_lo/neg_hi:
//   op_sel[i]    = which half of source i feeds the LOW  result (0 = .x, 1 = .y), default 0
//   op_sel_hi[i] = which half of source i feeds the HIGH result,                   default 1
typedef float caf_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ caf_v2f pk(cpx<float> a) { return caf_v2f{a.x, a.y}; }
__device__ __forceinline__ cpx<float> unpk(caf_v2f v) { return {v.x, v.y}; }

__device__ __forceinline__ cpx<float> operator+(cpx<float> a, cpx<float> b) { return unpk(pk(a) + pk(b)); }
__device__ __forceinline__ cpx<float> operator-(cpx<float> a, cpx<float> b) { return unpk(pk(a) - pk(b)); }

// a + i b  and  a - i b: one v_pk_add_f32 each
__device__ __forceinline__ cpx<float> add_i(cpx<float> a, cpx<float> b)
{
    caf_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(pk(a)), "v"(pk(b)));
    return unpk(r);
}
__device__ __forceinline__ cpx<float> sub_i(cpx<float> a, cpx<float> b)
{
    caf_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(pk(a)), "v"(pk(b)));
    return unpk(r);
}
// a*b: (a.x b.x, a.x b.y) then + a.y * (-b.y, b.x)
__device__ __forceinline__ cpx<float> cmul(cpx<float> a, cpx<float> b)
{
    const caf_v2f av = pk(a), bv = pk(b), t = av.xx * bv;
    caf_v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(av), "v"(bv), "v"(t));
    return unpk(r);
}
// acc + a*b: acc + a.x (b.x, b.y), then + a.y (-b.y, b.x)
__device__ __forceinline__ cpx<float> cfma(cpx<float> acc, cpx<float> a, cpx<float> b)
{
    const caf_v2f av = pk(a), bv = pk(b), t = __builtin_elementwise_fma(av.xx, bv, pk(acc));
    caf_v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=v"(r) : "v"(av), "v"(bv), "v"(t));
    return unpk(r);
}
// a*conj(b): (a.x b.x, -a.x b.y) then + a.y * (b.y, b.x)
__device__ __forceinline__ cpx<float> cmulc(cpx<float> a, cpx<float> b)
{
    const caf_v2f av = pk(a), bv = pk(b), t = av.xx * caf_v2f{bv.x, -bv.y};
    caf_v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(av), "v"(bv), "v"(t));
    return unpk(r);
}
// conj(a*b): (a.x b.x, -a.x b.y) then + a.y * (-b.y, -b.x)
__device__ __forceinline__ cpx<float> cmul_conj(cpx<float> a, cpx<float> b)
{
    const caf_v2f av = pk(a), bv = pk(b), t = av.xx * caf_v2f{bv.x, -bv.y};
    caf_v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]"
        : "=v"(r) : "v"(av), "v"(bv), "v"(t));
    return unpk(r);
}
// v + r (i v)  and  r v + i v  (the Linzer-Feig pre-rotations of bfly_w).  r is a compile-time
// constant after inlining: it is passed in an SGPR pair ("s"), not a VGPR pair -- a kernel's
// butterfly constants (dozens of distinct ones) then cost scalar registers and s_mov, not vector
// registers the row data needs (the chain kernels run at 128 VGPRs).
__device__ __forceinline__ cpx<float> lf_tan(cpx<float> v, float r)
{
    caf_v2f o;
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]"
        : "=v"(o) : "s"(caf_v2f{r, r}), "v"(pk(v)));
    return unpk(o);
}
__device__ __forceinline__ cpx<float> lf_cot(cpx<float> v, float r)
{
    caf_v2f o;
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[0,1,0] neg_lo:[0,0,1]"
        : "=v"(o) : "s"(caf_v2f{r, r}), "v"(pk(v)));
    return unpk(o);
}
// u + g b and u - g b for a compile-time real g (SGPR operand, low half broadcast to both lanes)
__device__ __forceinline__ void axpy_pm_const(float g, cpx<float> b, cpx<float> u, cpx<float> &p, cpx<float> &m)
{
    caf_v2f pp, mm;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(pp) : "s"(caf_v2f{g, g}), "v"(pk(b)), "v"(pk(u)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
        : "=v"(mm) : "s"(caf_v2f{g, g}), "v"(pk(b)), "v"(pk(u)));
    p = unpk(pp);
    m = unpk(mm);
}

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

// Cheaper form of the same reduction: wave maximum of the values first (v_max through the DPP
// network), then the minimum index among the lanes that hold it (v_min_u32 with a DPP operand).
// Equal values keep the LOWER index, like arg_merge.  Lane 63 holds the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void max_step(double &v)
{   // (asm: llvm.maxnum would first canonicalise the bit-cast DPP result with a second v_max)
    const double o = dpp_t<CTRL, ROW_MASK>(v);
    asm("v_max_f64 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void max_step(float &v)
{
    const float o = dpp_t<CTRL, ROW_MASK>(v);
    asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(o));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void min_step(uint32_t &i)
{
    const uint32_t o = (uint32_t)dpp_i<CTRL, ROW_MASK>((int)i);
    i = o < i ? o : i;
}
template <typename T>
__device__ __forceinline__ T lane63(T v);
template <>
__device__ __forceinline__ double lane63<double>(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <>
__device__ __forceinline__ float lane63<float>(float v)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <typename T>
__device__ __forceinline__ void wave_arg_reduce_maxmin(T &bv, uint32_t &bi)
{
    T m = bv;
    max_step<0xB1, 0xF>(m);
    max_step<0x4E, 0xF>(m);
    max_step<0x141, 0xF>(m);
    max_step<0x140, 0xF>(m);
    max_step<0x142, 0xA>(m);
    max_step<0x143, 0xC>(m);
    m = lane63(m);  // wave-uniform
    uint32_t c = bv == m ? bi : 0xffffffffu;
    min_step<0xB1, 0xF>(c);
    min_step<0x4E, 0xF>(c);
    min_step<0x141, 0xF>(c);
    min_step<0x140, 0xF>(c);
    min_step<0x142, 0xA>(c);
    min_step<0x143, 0xC>(c);
    bv = m;
    bi = c;
}

}  // namespace caf
