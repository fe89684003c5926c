// kernels_ablate.hpp -- MEASUREMENT LIBRARY ONLY (-DCAF_MEASURE): memory policies that turn a product row
// kernel into its arithmetic-only skeleton.  The kernel body is the product's, unchanged; only the policy
// object its memory hooks go through differs.  Results are WRONG by construction -- timing only: what remains is
// the VALU instruction stream of the row, i.e. the issue ceiling bench.py reports as `secondary`.
#pragma once
#include "../kernels_duo4096.hpp"

namespace caf {

// k_duo_rows<T, 0, DuoIoNull<T>>: no LDS traffic, no barriers, no global loads, no surface stores
template <typename T>
struct DuoIoNull {
    cpx<T> *Lc;
    const cpx<T> *twB;
    const SeqLane &L;
    __device__ __forceinline__ void hold(cpx<T> (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) keep(v[k]);
    }
    __device__ __forceinline__ void write_A(cpx<T> (&v)[16]) const { hold(v); }
    __device__ __forceinline__ void read_A(cpx<T> (&v)[16]) const { hold(v); }
    __device__ __forceinline__ void write_B(cpx<T> (&v)[16]) const { hold(v); }
    __device__ __forceinline__ void read_B(cpx<T> (&v)[16]) const { hold(v); }
    __device__ __forceinline__ void write_C(cpx<T> (&v)[16]) const { hold(v); }
    __device__ __forceinline__ void read_C(cpx<T> (&v)[16]) const { hold(v); }
    __device__ __forceinline__ void mul_twB(cpx<T> (&v)[16]) const
    {
        cpx<T> w = {T(0.6), T(0.8)};  // a register operand instead of the LDS table: same multiply count
        keep(w);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], w);
    }
    __device__ __forceinline__ void sink_A(int, cpx<T> x) const { keep(x); }
    __device__ __forceinline__ void sync() const {}
    __device__ __forceinline__ void fence() const {}
    __device__ __forceinline__ void samples(cpx<T> (&a)[16], const __amdgpu_buffer_rsrc_t) const
    {
#pragma unroll
        for (int q = 0; q < 16; ++q) { a[q] = cpx<T>{T(q + 1), T(L.t)}; keep(a[q]); }
    }
    __device__ __forceinline__ cpx<T> sample(const __amdgpu_buffer_rsrc_t, int i) const
    {
        cpx<T> x = {T(i + 1), T(L.t)};
        keep(x);
        return x;
    }
    __device__ __forceinline__ void spectrum(cpx<T> (&h)[16], const __amdgpu_buffer_rsrc_t, int chain) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) { h[k] = cpx<T>{T(1 + chain), T(k)}; keep(h[k]); }
    }
    template <typename V>
    __device__ __forceinline__ void store(const __amdgpu_buffer_rsrc_t, unsigned, V d) const
    {
        asm volatile("" ::"v"(d));
    }
};

}  // namespace caf
