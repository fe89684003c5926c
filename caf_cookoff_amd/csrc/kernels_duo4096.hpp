// kernels_duo4096.hpp -- "two chains in flight" Doppler-row kernel for n = 4096 (L = 8192).
//
// Same mathematics, LDS geometry, tables and epilogue as kernels_seq4096.hpp (one 256-thread
// workgroup per row, one padded chain of LDS, 2 f64 / 3 f32 workgroups per CU), but the even-bin
// and the odd-bin chain of a row are software-pipelined against each other instead of running
// one after the other: while one chain's exchange drains into LDS (a 64 KiB exchange occupies
// the CU's LDS store path for 400-800 cycles) the same wave computes the other chain's
// butterflies.  Both chains' sixteen points stay in registers (2 x 64 VGPRs in f64); the single
// LDS area is handed back and forth:
//
//   E.S1 -> W_A | O.S1 (regs) | B1 | E.R_B | B2 | O.W_A | E.S2 (regs) | B3 | O.R_B |
//   E.W_B E.R_C | O.S2 | O.W_B O.R_C | E.S3 | E.W_C E.R_B | O.S3 | O.W_C O.R_B | E.S4 | E.W_B |
//   O.S4 (regs) | B4 | E.R_A | B5 | O.W_B | E.S5 | B6 | O.R_A | O.S5 | last stage, stores
//
// (S = butterfly stage in registers, W/R = LDS write/read in pattern A/B/C, B = workgroup
// barrier.)  Pattern-A accesses cross waves, patterns B and C stay inside a wave's own four
// 256-blocks, where LDS operations execute in program order: only the A <-> B hand-overs need
// barriers (six per row + the argmax publication; the sequential kernel has four + one).  The
// needle samples are loaded once per row (the sequential kernel loads them once per chain).
//
// Measured (batch 128): complex64 119.4 k vs 112.6 k surfaces/s for the sequential kernel -> this
// is the complex64 product kernel.  complex128: 65.2 k vs 66.9 k -- with 2 x 64 data VGPRs held
// the haystack spectrum cannot be prefetched early (35 spills if it is), so the f64 product
// kernel stays the sequential one; CAF_ROW_KERNEL=3 / 0 select either for both types.
#pragma once
#include "kernels_seq4096.hpp"

namespace caf {

template <typename T>
struct DuoIo {
    cpx<T> *Lc;
    const cpx<T> *twB;
    const SeqLane &L;
    __device__ __forceinline__ void write_A(const cpx<T> (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[L.pA + k * F_BLK] = v[k];
    }
    __device__ __forceinline__ void read_A(cpx<T> (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pA + k * F_BLK];
    }
    __device__ __forceinline__ void write_B(const cpx<T> (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[L.pB + 17 * k] = v[k];
    }
    __device__ __forceinline__ void read_B(cpx<T> (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pB + 17 * k];
    }
    __device__ __forceinline__ void write_C(const cpx<T> (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[L.pC + k] = v[k];
    }
    __device__ __forceinline__ void read_C(cpx<T> (&v)[16]) const
    {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[L.pC + k];
    }
    __device__ __forceinline__ void mul_twB(cpx<T> (&v)[16]) const
    {
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twB[16 * k]);
    }
    // every other access to memory the row makes goes through these hooks too, so that the measurement
    // library can instantiate the SAME kernel body with a policy that keeps everything in registers
    // (measure/kernels_ablate.hpp, DuoIoNull: the packed-f32 / FP64 issue ceiling bench.py reports); the product
    // library only ever instantiates this policy
    __device__ __forceinline__ void sink_A(int k, cpx<T> x) const { Lc[L.pA + k * F_BLK] = x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ void fence() const { wave_lds_fence(); }
    __device__ __forceinline__ void samples(cpx<T> (&a)[16], const __amdgpu_buffer_rsrc_t rs) const { load_samples(a, rs, L); }
    __device__ __forceinline__ cpx<T> sample(const __amdgpu_buffer_rsrc_t rs, int i) const
    {
        return bload(rs, (unsigned)(L.t * sizeof(cpx<T>)), (unsigned)(256 * i * sizeof(cpx<T>)), (cpx<T> *)nullptr);
    }
    __device__ __forceinline__ void spectrum(cpx<T> (&h)[16], const __amdgpu_buffer_rsrc_t rs_spec, int chain) const
    {
        const unsigned voff = (unsigned)((chain * 4096 + L.t) * sizeof(cpx<T>));
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = bload(rs_spec, voff, (unsigned)(256 * k * sizeof(cpx<T>)), (cpx<T> *)nullptr);
    }
    template <typename V>
    __device__ __forceinline__ void store(const __amdgpu_buffer_rsrc_t rs, unsigned byte_off, V d) const
    {
        store_vec_aux<CAF_AUX_SC1>(rs, byte_off, d);
    }
};

template <typename T, typename IO = DuoIo<T>>
__global__ __launch_bounds__(S_THREADS, seq_waves_per_simd<T>()) void k_duo_rows(const FusedArgs<T> A,
                                                                             const cpx<T> *__restrict__ phasor)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[seq_lds_bytes<T>()];
    C *const Lc = reinterpret_cast<C *>(smem);
    C *const twb = Lc + F_CHAIN;  // twb[k*16 + lo] = W_256^(lo*k)
    unsigned char *const scratch = smem + (F_CHAIN + 256) * sizeof(C);
    const SeqLane L;

    TwSet<T> tw;
    tw.w1 = A.tab.tw4096[L.t * 1];
    tw.w2 = A.tab.tw4096[L.t * 2];
    tw.w3 = A.tab.tw4096[L.t * 3];
    tw.w4 = A.tab.tw4096[L.t * 4];
    tw.w8 = A.tab.tw4096[L.t * 8];
    tw.w12 = A.tab.tw4096[L.t * 12];
    twb[L.tid] = A.tab.tw4096[16 * (L.tid & 15) * (L.tid >> 4)];
    const IO io{Lc, twb + L.lo4, L};
    const C th = A.tab.th[L.t];  // T^t = e^{2*pi*i*t/8192}
    const C cfac = conj(th);     // odd chain input rotation e^{-2*pi*i*t/8192}
    const int mpair = L.t & ~1;
    const bool odd = L.lane & 1;
    constexpr unsigned long long EVEN_LANES = 0x5555555555555555ull;
    __syncthreads();

    C a[16];
    {
        const int gc = (int)blockIdx.x < A.total ? (int)blockIdx.x : A.total - 1;
        io.samples(a, __builtin_amdgcn_make_buffer_rsrc((void *)(A.sig + (size_t)(gc / A.rows) * F_N), 0,
                                                        F_N * (int)sizeof(C), 0x00020000));
    }
    volatile int *const next_row = reinterpret_cast<volatile int *>(scratch + 112);
    C pb;
    {
        const int g0 = (int)blockIdx.x < A.total ? (int)blockIdx.x : A.total - 1;
        const C *ph0 = phasor + (size_t)(g0 % A.rows) * 64;
        pb = cmul(ph0[L.lo4], ph0[16 + L.hi4]);
    }
    for (int g = blockIdx.x; g < A.total;) {
        if (L.tid == 0)
            *next_row = A.work ? (int)gridDim.x + (int)atomicAdd(A.work, 1u) : g + (int)gridDim.x;
        const int b = g / A.rows, r = g - b * A.rows;
        const C *__restrict__ ph = phasor + (size_t)r * 64;
        const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.spec + (size_t)b * (2 * 16 * 256)), 0, 2 * 16 * 256 * (int)sizeof(C), 0x00020000);
        C e[16], o[16], h[16];

        // ---- S1: mixer (mod.rs:46-65) + first forward butterfly of both chains ----------------
#pragma unroll
        for (int q = 0; q < 16; ++q) e[q] = cmul_conj(a[q], ph[32 + q]);
        {
            const TwFold<T> fe(tw, conj(pb));
            dft16_sink(e, [&](int k, C x) { io.sink_A(k, twA_k(x, k, tw, fe)); });  // E.W_A
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] = cmul_conj(a[q], ph[48 + q]);
        {
            const TwFold<T> fo(tw, conj(cmul(pb, cfac)));
            dft16(o);                                                                         // under E.W_A
#pragma unroll
            for (int k = 0; k < 16; ++k) o[k] = twA_k(o[k], k, tw, fo);
        }
        io.sync();  // B1
        io.read_B(e);
        // the ticket was stored before B1: visible to every wave by now
        const int gn = __builtin_amdgcn_readfirstlane(*next_row);
        const int gc = gn < A.total ? gn : A.total - 1;  // clamped: a[] is always redefined
        io.sync();  // B2: every wave holds its E data, the area is free
        io.write_A(o);    // O.W_A
        // ---- E.S2 under O.W_A ----
        dft16(e);
        io.mul_twB(e);
        io.sync();  // B3
        io.read_B(o);
        io.write_B(e);
        io.fence();
        io.read_C(e);
        // ---- O.S2 under E's wave-local exchange ----
        dft16(o);
        io.mul_twB(o);
        io.fence();  // E.R_C (other lanes' pattern-B slots) before they are overwritten
        io.write_B(o);
        io.fence();
        io.read_C(o);
        // ---- E.S3: last forward butterfly, spectrum product (xcor_rustfft.rs:64-73), first inverse one
        io.spectrum(h, rs_spec, 0);  // (earlier costs 33 f64 spills; the butterfly below covers the L2 latency)
        dft16(e);
#pragma unroll
        for (int k = 0; k < 16; ++k) e[k] = cmul(e[k], h[k]);
        dft16(e);
        io.spectrum(h, rs_spec, 1);
        io.fence();
        io.write_C(e);
        io.fence();
        io.read_B(e);
        // ---- O.S3 ----
        dft16(o);
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = cmul(o[k], h[k]);
        dft16(o);
        io.fence();
        io.write_C(o);
        io.fence();
        io.read_B(o);
        // ---- E.S4 ----
        io.mul_twB(e);
        dft16(e);
        io.fence();
        io.write_B(e);
        // ---- O.S4 under E.W_B ----
        io.mul_twB(o);
        dft16(o);
        io.sync();  // B4
        io.read_A(e);
        io.sync();  // B5: every wave holds its E data
        io.write_B(o);
        // ---- E.S5 under O.W_B ----
        apply_twA(e, tw);
        dft16(e);
        io.sync();  // B6
        io.read_A(o);
        {
            const TwFold<T> fpost(tw, th);  // T^t of the last radix-2 stage folded into the twiddles
#pragma unroll
            for (int k = 0; k < 16; ++k) o[k] = twA_k(o[k], k, tw, fpost);
        }
        dft16(o);

        // ---- last radix-2 stage (in registers) + |.|^2 + argmax + write-through stores ----------
        const __amdgpu_buffer_rsrc_t rs_sig_next = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.sig + (size_t)(gc / A.rows) * F_N), 0, F_N * (int)sizeof(C), 0x00020000);
        T bv_lo = T(0), bv_hi = T(0);
        int bi_lo = 0, bi_hi = 0;
        T *const out = A.surface ? A.surface + (size_t)g * F_L : nullptr;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? F_L * (int)sizeof(T) : 0, 0x00020000);
        T mlo[16], mhi[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // m = t + 256*i
            C lo, hi;
            axpy_w32(i, e[i], o[i], lo, hi);
            mlo[i] = norm_sqr(lo);  // mod.rs:147
            mhi[i] = norm_sqr(hi);
            bi_lo = mlo[i] > bv_lo ? i : bi_lo;  // first strictly greater (mod.rs:148-151)
            bv_lo = vmax(bv_lo, mlo[i]);
            bi_hi = mhi[i] > bv_hi ? i : bi_hi;
            bv_hi = vmax(bv_hi, mhi[i]);
            a[i] = io.sample(rs_sig_next, i);
        }
        {   // phasor base of the next row
            const C *phn = phasor + (size_t)(gc % A.rows) * 64;
            pb = cmul(phn[L.lo4], phn[16 + L.hi4]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            typename pair_vec<T>::type dlo, dhi;
            pair_xor1(mlo[2 * j], mlo[2 * j + 1], EVEN_LANES, ~EVEN_LANES, dlo);
            pair_xor1(mhi[2 * j], mhi[2 * j + 1], EVEN_LANES, ~EVEN_LANES, dhi);
            const int m = mpair + 256 * (2 * j + (odd ? 1 : 0));
            io.store(rs, (unsigned)(m * sizeof(T)), dlo);
            io.store(rs, (unsigned)((m + F_N) * sizeof(T)), dhi);
        }
        T bv = bv_lo;
        uint32_t bi = bv_lo > T(0) ? (uint32_t)(L.t + 256 * bi_lo) : 0u;
        if (bv_hi > bv) { bv = bv_hi; bi = (uint32_t)(L.t + 256 * bi_hi + F_N); }
        wave_arg_reduce_maxmin(bv, bi);
        T *sv = reinterpret_cast<T *>(scratch);
        uint32_t *si = reinterpret_cast<uint32_t *>(scratch + 32);
        if (L.lane == 63) { sv[L.wave] = bv; si[L.wave] = bi; }
        __syncthreads();
        if (L.tid == 0) {
            bv = sv[0];
            bi = si[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) arg_merge(bv, bi, sv[w], si[w]);
            A.row_idx[g] = bi;
            A.row_val[g] = bv;
        }
        g = gn;
    }
}

}  // namespace caf
