// kernels_ablate.hpp -- MEASUREMENT LIBRARY ONLY (-DCAF_MEASURE): memory policies that turn a product row
// kernel (k_duo_rows, k_seq_rows) into its arithmetic-only skeleton, or leave out one kind of memory access.  The kernel body is the product's, unchanged; only the policy
// object its memory hooks go through differs.  Results are WRONG by construction -- timing only: what remains is
// the VALU instruction stream of the row, i.e. the issue ceiling bench.py reports as `secondary`.
#pragma once
#include "../kernels_duo4096.hpp"
#include "../kernels_chain.hpp"

namespace caf {

// an optimisation barrier on a value WITHOUT ordering constraints (not volatile: it may move freely with its operands,
// it only hides the value's origin so that a synthesised operand is not constant-folded).  Operands that stand in for
// per-row LOADS use the volatile keep() instead: like a load they must be produced once per row, not hoisted out of
// the row loop (sixteen samples + thirty-two spectrum values held across the loop would spill)
template <typename T>
__device__ __forceinline__ void opaque(cpx<T> &x)
{
    asm("" : "+v"(x.x), "+v"(x.y));
}

// k_duo_rows<T, DuoIoNull<T>>: no LDS traffic, no barriers, no global loads, no surface stores.  An LDS exchange
// becomes a hand-over through sixteen registers of the policy object (identity "permutation"): every value a stage
// produces still feeds the next stage, so nothing is dead and no scheduling fence is needed -- the compiler keeps the
// product's arithmetic and is as free to schedule it as in the product.
template <typename T>
struct DuoIoNull {
    using C = cpx<T>;
    C *Lc;
    const C *twB;
    const SeqLane &L;
    mutable C r[16];
    __device__ __forceinline__ void put(const C (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) r[k] = v[k];
    }
    __device__ __forceinline__ void get(C (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = r[k];
    }
    __device__ __forceinline__ void write_A(const C (&v)[16]) const { put(v); }
    __device__ __forceinline__ void read_A(C (&v)[16]) const { get(v); }
    __device__ __forceinline__ void write_B(const C (&v)[16]) const { put(v); }
    __device__ __forceinline__ void read_B(C (&v)[16]) const { get(v); }
    __device__ __forceinline__ void write_C(const C (&v)[16]) const { put(v); }
    __device__ __forceinline__ void read_C(C (&v)[16]) const { get(v); }
    __device__ __forceinline__ void mul_twB(C (&v)[16]) const
    {
        C w = {T(0.6), T(0.8)};  // a register operand instead of the LDS table: same multiply count
        opaque(w);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], w);
    }
    __device__ __forceinline__ void sink_A(int k, C x) const { r[k] = x; }
    // where the product has a barrier or a wave-level LDS fence the schedule keeps its phase boundary (a compile-time
    // fence, no instruction): without them the compiler interleaves the stages of the whole row and spills
    __device__ __forceinline__ void sync() const { __builtin_amdgcn_sched_barrier(0); }
    __device__ __forceinline__ void fence() const { __builtin_amdgcn_sched_barrier(0); }
    // stand-in for every loaded value: one opaque register pair (see SeqIoCut)
    __device__ __forceinline__ C fake() const
    {
        C w = {T(0.8), T(0.6)};
        opaque(w);
        return w;
    }
    __device__ __forceinline__ void samples(C (&a)[16], const __amdgpu_buffer_rsrc_t) const
    {
        const C w = fake();
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = w;
    }
    __device__ __forceinline__ C sample(const __amdgpu_buffer_rsrc_t, int) const { return fake(); }
    __device__ __forceinline__ void spectrum(C (&h)[16], const __amdgpu_buffer_rsrc_t, int) const
    {
        const C w = fake();
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = w;
    }
    template <typename V>
    __device__ __forceinline__ void store(const __amdgpu_buffer_rsrc_t, unsigned, V d) const
    {
        asm volatile("" ::"v"(d));  // the row's results must stay live
    }
};

// k_duo_rows<T, DuoIoNoStore<T>>: the product row without its surface stores (CAF_STORE_MODE=3 with CAF_ROW_KERNEL=3)
template <typename T>
struct DuoIoNoStore : DuoIo<T> {
    template <typename V>
    __device__ __forceinline__ void store(const __amdgpu_buffer_rsrc_t, unsigned, V d) const
    {
        asm volatile("" ::"v"(d));
    }
};

// ---- k_seq_rows<T, PF, IO>: policies derived from the product's SeqIo<T> ------------------------------------------
// bits of WHAT: 1 = no LDS traffic / barriers (hand-over through registers, as above), 2 = no global loads,
// 4 = no surface stores (7 = VALU only)
template <typename T, int WHAT>
struct SeqIoCut : SeqIo<T> {
    using C = cpx<T>;
    using B = SeqIo<T>;
    mutable C r[(WHAT & 1) ? 16 : 1];
    __device__ __forceinline__ SeqIoCut(C *lc, const C *twb, const SeqLane &l) : B{lc, twb, l}
    {
        wfake = C{T(0.8), T(0.6)};
        opaque(wfake);
    }
    // (the hand-over pins each value with a volatile asm, in order, like the original in-line ablation of rounds 1-2:
    // the complex128 kernel sits at exactly 256 VGPRs, and with the stages free to interleave across the phase
    // boundary the allocator spills 55-160 registers -- no ceiling of the product's instruction stream)
    __device__ __forceinline__ void get(C (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            keep(r[(WHAT & 1) ? k : 0]);
            v[k] = r[(WHAT & 1) ? k : 0];
        }
    }
    __device__ __forceinline__ void sinkA(int k, C x) const { if constexpr (WHAT & 1) r[k] = x; else B::sinkA(k, x); }
    __device__ __forceinline__ void sinkB(int k, C x) const { if constexpr (WHAT & 1) r[k] = x; else B::sinkB(k, x); }
    __device__ __forceinline__ void sinkC(int k, C x) const { if constexpr (WHAT & 1) r[k] = x; else B::sinkC(k, x); }
    __device__ __forceinline__ void readA(C (&v)[16]) const { if constexpr (WHAT & 1) get(v); else B::readA(v); }
    __device__ __forceinline__ void readB(C (&v)[16]) const { if constexpr (WHAT & 1) get(v); else B::readB(v); }
    __device__ __forceinline__ void readC(C (&v)[16]) const { if constexpr (WHAT & 1) get(v); else B::readC(v); }
    __device__ __forceinline__ C twb(int k) const
    {
        if constexpr (WHAT & 1) { C w = {T(0.6), T(0.8)}; opaque(w); return w; }
        else return B::twb(k);
    }
    // (cut: the phase boundary stays as a compile-time scheduling fence, see DuoIoNull)
    __device__ __forceinline__ void sync() const { if constexpr (WHAT & 1) __builtin_amdgcn_sched_barrier(0); else B::sync(); }
    __device__ __forceinline__ void fence() const { if constexpr (WHAT & 1) __builtin_amdgcn_sched_barrier(0); else B::fence(); }
    // stand-in for every loaded value: ONE opaque register pair made at construction.  The multiplies the loaded values
    // feed stay (their other operands differ), the sixteen-value arrays the product keeps in flight collapse to one
    // pair -- fewer registers than the product, never more (per-call stand-ins made the allocator spill 37-60 VGPRs)
    mutable C wfake;
    __device__ __forceinline__ C fake(int) const { return wfake; }
    __device__ __forceinline__ void samples(C (&a)[16], const __amdgpu_buffer_rsrc_t rs) const
    {
        if constexpr (WHAT & 2) {
            const C x = fake(0);
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = x;
        } else B::samples(a, rs);
    }
    __device__ __forceinline__ C sample(const __amdgpu_buffer_rsrc_t rs, int i) const
    {
        if constexpr (WHAT & 2) return fake(i);
        else return B::sample(rs, i);
    }
    __device__ __forceinline__ C spec(const __amdgpu_buffer_rsrc_t rs, unsigned voff, int k) const
    {
        if constexpr (WHAT & 2) return fake(k);
        else return B::spec(rs, voff, k);
    }
    template <typename V>
    __device__ __forceinline__ void store(const __amdgpu_buffer_rsrc_t rs, unsigned off, V d) const
    {
        if constexpr (WHAT & 4) asm volatile("" ::"v"(d));
        else B::store(rs, off, d);
    }
};

// k_chain_rows<T, LOGM, R, NB, MASK>: the chain row kernel with one or several kinds of memory access cut out
// (CAF_CHAIN_ABL of the measurement library; bench.py's configs[3] ceiling is MASK = 31).  MASK bits: 1 no workgroup
// barriers (every exchange ends in a wave-local fence), 2 no haystack-spectrum loads, 4 no slab traffic, 8 no needle
// loads, 16 no surface stores, 128 no LDS chain traffic.  A cut load returns a synthesised value behind the volatile
// keep() (produced once per use, like the load it stands for); a cut store only keeps its operand alive.
template <typename T, int MASK>
struct ChainIoCut {
    using C = cpx<T>;
    using P = ChainIo<T>;
    static constexpr bool wg_barriers = !(MASK & 1);
    __device__ __forceinline__ static void lds_st(C *Lc, int pos, C x)
    {
        if constexpr (MASK & 128) keep(x);
        else P::lds_st(Lc, pos, x);
    }
    __device__ __forceinline__ static C lds_ld(const C *Lc, int pos)
    {
        if constexpr (MASK & 128) { C x = C{T(pos), T(1)}; keep(x); return x; }
        else return P::lds_ld(Lc, pos);
    }
    __device__ __forceinline__ static C sample(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int q, int j, int beta)
    {
        if constexpr (MASK & 8) { C x = C{T(q + 1 + j), T(beta)}; keep(x); return x; }
        else return P::sample(rs, voff, soff, q, j, beta);
    }
    __device__ __forceinline__ static void spec2(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int k, C &h0, C &h1)
    {
        if constexpr (MASK & 2) { h0 = C{T(1), T(k)}; h1 = C{T(k), T(1)}; keep(h0); keep(h1); }
        else P::spec2(rs, voff, soff, k, h0, h1);
    }
    __device__ __forceinline__ static void slab_st(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, C x)
    {
        if constexpr (MASK & 4) keep(x);
        else P::slab_st(rs, voff, soff, x);
    }
    __device__ __forceinline__ static C slab_ld(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, int arr, int i)
    {
        if constexpr (MASK & 4) { C x = C{T(i), T(arr)}; keep(x); return x; }
        else return P::slab_ld(rs, voff, soff, arr, i);
    }
    __device__ __forceinline__ static void surf_st(const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, T m)
    {
        if constexpr (MASK & 16) asm volatile("" ::"v"(m));
        else P::surf_st(rs, voff, soff, m);
    }
};
template <typename T, int ABL>
struct ChainIoFor {
    using type = ChainIoCut<T, ABL>;
};

}  // namespace caf
