// kernels_r8_4096.hpp -- radix-8 Doppler-row kernel for n = 4096 (L = 8192): 4 waves per SIMD.
//
// Why: on gfx950 a wave-level FP64 VALU instruction issues every ~2.7 / 2.2 / 2.0 ns per
// SIMD with 1 / 2 / 4 resident waves (tools/ubench/f64_rates.hip), and the 16-point-per-lane
// kernels (kernels_seq4096.hpp, kernels_fused4096.hpp) are capped at 2 waves per SIMD by
// their 64-register butterfly + 16 KiB of LDS per wave, so their LDS exchanges, barriers and
// L2 latencies are only partly hidden (profiles/r01_v2: ablation is almost additive).
// Here every lane holds 8 points (32 VGPRs f64), one 512-thread workgroup owns a row and
// runs the even-bin and the odd-bin chain back to back, LDS is 8 KiB per wave and the
// register budget 128: TWO workgroups = 16 waves = 4 per SIMD are resident per CU.
//
// Each 4096-point transform is 8x8x8x8.  Forward DIF: pass 1 over n = 512q + t, then wave
// k0 owns the whole 512-point sub-transform of output digit k0, so only the FIRST exchange
// crosses waves (barrier); exchanges 2 and 3 are wave-local.  The product with the
// pre-permuted haystack spectrum happens in registers; the inverse is the mirrored DIT.
// Per chain: 6 exchanges, 2 of them with a barrier (+1 barrier before the next chain writes).
//
// LDS chain image: 8 wave-blocks of 576 elements (512 used by exchange 1; exchanges 2/3 use
// the padded forms below).  Every access is base VGPR + immediate and bank-conflict free:
//   A  (ex1 write / ex6 read): 576*k + t                      (k = register index)
//   B1 (ex1 read  / ex6 write): 576*w + l + 64*k
//   B2 (ex2 write / ex5 read):  576*w + l + 72*k
//   C  (ex2 read  / ex5 write): 576*w + 72*(l>>3) + (l&7) + 8*k
//   C3 (ex3 write / ex4 read):  576*w + 72*(l>>3) + (l&7) + 9*k
//   D  (ex3 read  / ex4 write): 576*w + 72*(l>>3) + 9*(l&7) + k
// Inter-pass twiddles: W_4096^(t*k) from three held values (w1, w2, w4; the other four are
// one extra complex multiply each), W_512^(l*k) and W_64^((l&7)*k) from LDS tables.
// LDS per workgroup: 4608*16 + 7*64*16 + 7*8*16 + 128 = 81 920 B = exactly half a CU's 160 KiB.
#pragma once
#include "../kernels_fused4096.hpp"

namespace caf {

constexpr int R_THREADS = 512;
constexpr int R_BLK = 576;            // wave-block stride (elements)
constexpr int R_CHAIN = 8 * R_BLK;    // 4608 elements

template <typename T>
constexpr size_t r8_lds_bytes() { return (R_CHAIN + 7 * 64 + 7 * 8) * sizeof(cpx<T>) + 128; }

// Per-row phasor table for this kernel (64 entries):
//   [ 0..15] lo[j]    = e^{j*ph*j}
//   [16..47] hi[j]    = e^{j*ph*16*j},  j < 32
//   [48..55] step0[q] = e^{j*ph*512*q}
//   [56..63] step1[q] = e^{j*ph*512*q} * e^{-2*pi*i*512*q/8192}      (odd chain)
template <typename T>
__global__ void k_r8_phasors(const double *__restrict__ ph, int nrows, cpx<T> *__restrict__ tab)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g >> 6, e = g & 63;
    if (row > nrows) return;
    const double p = row < nrows ? ph[row] : 0.0;
    double mult, s2 = 0.0, c2 = 1.0;
    int j;
    if (e < 16) { j = e; mult = 1.0; }
    else if (e < 48) { j = e - 16; mult = 16.0; }
    else { j = (e - 48) & 7; mult = 512.0; }
    double s, co;
    sincos(p * (mult * (double)j), &s, &co);
    if (e >= 56) sincospi(-2.0 * (double)(512 * j) / 8192.0, &s2, &c2);
    tab[(size_t)row * 64 + e] = {(T)(co * c2 - s * s2), (T)(co * s2 + s * c2)};
}

// ---- radix-8 butterfly, positive exponent, natural order in and out --------------------
template <typename T>
__device__ __forceinline__ void dft8(cpx<T> (&v)[8])
{
    // stage 1: pairs (q0, q0+4)
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) {
        const cpx<T> a = v[q0], b = v[q0 + 4];
        v[q0] = a + b;
        v[q0 + 4] = a - b;
    }
    // W8^(q0) on the difference branch
    v[5] = mul_w8(v[5]);
    v[6] = muli(v[6]);
    v[7] = mul_w8_3(v[7]);
    // stage 2: radix-4 over q0 for r0 = 0 (v[0..3]) and r0 = 1 (v[4..7]) -> X[r0 + 2*r1] at v[4*r0 + r1]
    dft4(v[0], v[1], v[2], v[3]);
    dft4(v[4], v[5], v[6], v[7]);
    // natural order: out[r0 + 2*r1] = v[4*r0 + r1]
    const cpx<T> t1 = v[1], t2 = v[2], t3 = v[3], t4 = v[4], t5 = v[5], t6 = v[6];
    v[1] = t4; v[2] = t1; v[3] = t5; v[4] = t2; v[5] = t6; v[6] = t3;
}

struct R8Lane {
    int tid, lane, wave, pA, pB, pC, pD;
    __device__ __forceinline__ R8Lane()
    {
        tid = threadIdx.x;
        lane = tid & 63;
        wave = tid >> 6;
        pA = tid;
        pB = R_BLK * wave + lane;
        pC = R_BLK * wave + 72 * (lane >> 3) + (lane & 7);
        pD = R_BLK * wave + 72 * (lane >> 3) + 9 * (lane & 7);
    }
};

template <typename T>
struct TwSet8 {
    cpx<T> w1, w2, w4;  // W_4096^(t*k), k = 1, 2, 4
};

// x[k] *= W_4096^(t*k), k = 1..7, from the three held values
template <typename T>
__device__ __forceinline__ void apply_twA8(cpx<T> (&v)[8], const TwSet8<T> &w)
{
    const cpx<T> w3 = cmul(w.w1, w.w2);
    v[1] = cmul(v[1], w.w1);
    v[2] = cmul(v[2], w.w2);
    v[3] = cmul(v[3], w3);
    v[4] = cmul(v[4], w.w4);
    v[5] = cmul(cmul(v[5], w.w4), w.w1);
    v[6] = cmul(cmul(v[6], w.w4), w.w2);
    v[7] = cmul(cmul(v[7], w.w4), w3);
}

// Forward chain: v[q] = u[t + 512q]  ->  v[k3] = G[w + 8*(l>>3) + 64*(l&7) + 512*k3]
template <typename T>
__device__ __forceinline__ void r8_forward(cpx<T> (&v)[8], const TwSet8<T> &tw, const cpx<T> *twB,
                                           const cpx<T> *twC, cpx<T> *Lc, const R8Lane &L)
{
    dft8(v);
    apply_twA8(v, tw);
#pragma unroll
    for (int k = 0; k < 8; ++k) Lc[L.pA + R_BLK * k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = Lc[L.pB + 64 * k];
    dft8(v);
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], twB[64 * (k - 1)]);
#pragma unroll
    for (int k = 0; k < 8; ++k) Lc[L.pB + 72 * k] = v[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = Lc[L.pC + 8 * k];
    dft8(v);
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], twC[8 * (k - 1)]);
#pragma unroll
    for (int k = 0; k < 8; ++k) Lc[L.pC + 9 * k] = v[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = Lc[L.pD + k];
    dft8(v);
}

// Inverse chain (mirror): v[k3] = C[...]  ->  v[m3] = y[t + 512*m3].  Ends with the barrier
// after which the chain image may be overwritten.
template <typename T>
__device__ __forceinline__ void r8_inverse(cpx<T> (&v)[8], const TwSet8<T> &tw, const cpx<T> *twB,
                                           const cpx<T> *twC, cpx<T> *Lc, const R8Lane &L)
{
    dft8(v);
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) Lc[L.pD + k] = v[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = Lc[L.pC + 9 * k];
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], twC[8 * (k - 1)]);
    dft8(v);
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) Lc[L.pC + 8 * k] = v[k];
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = Lc[L.pB + 72 * k];
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], twB[64 * (k - 1)]);
    dft8(v);
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) Lc[L.pB + 64 * k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = Lc[L.pA + R_BLK * k];
    __syncthreads();  // every LDS read of this chain is done
    apply_twA8(v, tw);
    dft8(v);
}

// Shared set-up: twiddle tables into LDS, per-thread constants.
template <typename T>
struct R8Setup {
    cpx<T> *Lc, *twB, *twC;
    unsigned char *scratch;
    TwSet8<T> tw;
    cpx<T> th;
    __device__ __forceinline__ R8Setup(unsigned char *smem, const FusedArgs<T> &A, const R8Lane &L)
    {
        using C = cpx<T>;
        Lc = reinterpret_cast<C *>(smem);
        C *tb = Lc + R_CHAIN;          // tb[(k-1)*64 + l] = W_512^(l*k)
        C *tc = tb + 7 * 64;           // tc[(k-1)*8 + b]  = W_64^(b*k)
        scratch = smem + (R_CHAIN + 7 * 64 + 7 * 8) * sizeof(C);
        if (L.tid < 7 * 64) tb[L.tid] = A.tab.tw4096[8 * (L.tid & 63) * ((L.tid >> 6) + 1)];
        if (L.tid < 7 * 8) tc[L.tid] = A.tab.tw4096[64 * (L.tid & 7) * ((L.tid >> 3) + 1)];
        twB = tb + L.lane;
        twC = tc + (L.lane & 7);
        tw.w1 = A.tab.tw4096[L.tid * 1];
        tw.w2 = A.tab.tw4096[L.tid * 2];
        tw.w4 = A.tab.tw4096[L.tid * 4];
        th = A.tab.th[L.tid];  // th512: e^{2*pi*i*t/8192}
    }
};

// mixer (mod.rs:46-65) + conjugation for one chain: v[q] = conj(a[t+512q] * w^(t+512q) * (CH ? conj(T^n) : 1))
template <typename T, int CH>
__device__ __forceinline__ void r8_mixer(cpx<T> (&v)[8], const __amdgpu_buffer_rsrc_t rs_sig, const cpx<T> *ph,
                                         const cpx<T> th, const R8Lane &L)
{
    using C = cpx<T>;
    C a[8];
#pragma unroll
    for (int q = 0; q < 8; ++q)
        a[q] = bload(rs_sig, (unsigned)(L.tid * sizeof(C)), (unsigned)(512 * q * sizeof(C)), (C *)nullptr);
    C pb = cmul(ph[L.tid & 15], ph[16 + (L.tid >> 4)]);
    if (CH) pb = cmulc(pb, th);  // * e^{-2*pi*i*t/8192}
    const C *ps = ph + 48 + CH * 8;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = conj(cmul(cmul(a[q], pb), ps[q]));
}

// ---- haystack spectrum in this kernel's register layout: spec[chain][k3][t] ----------------
template <typename T>
__global__ __launch_bounds__(R_THREADS, 4) void k_r8_prepare(const FusedArgs<T> A)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[r8_lds_bytes<T>()];
    const R8Lane L;
    const R8Setup<T> S(smem, A, L);
    __syncthreads();
    const C *ph = A.phasor + (size_t)A.rows * 64;  // the f = 0 row
    if (blockIdx.x == 0 && L.tid == 0 && A.work) *A.work = 0u;
    const T inv = T(1.0 / 8192.0);
    for (int b = blockIdx.x; b < A.total; b += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs_sig =
            __builtin_amdgcn_make_buffer_rsrc((void *)(A.sig + (size_t)b * F_N), 0, F_N * (int)sizeof(C), 0x00020000);
        C v[8];
        r8_mixer<T, 0>(v, rs_sig, ph, S.th, L);
        r8_forward(v, S.tw, S.twB, S.twC, S.Lc, L);
        C *spec = A.spec + (size_t)b * 8192;
#pragma unroll
        for (int k = 0; k < 8; ++k) spec[k * 512 + L.tid] = {v[k].x * inv, -v[k].y * inv};
        __syncthreads();
        r8_mixer<T, 1>(v, rs_sig, ph, S.th, L);
        r8_forward(v, S.tw, S.twB, S.twC, S.Lc, L);
#pragma unroll
        for (int k = 0; k < 8; ++k) spec[4096 + k * 512 + L.tid] = {v[k].x * inv, -v[k].y * inv};
        __syncthreads();
    }
}

// W_16^m3 = e^{2*pi*i*m3/16}, m3 < 8
__device__ constexpr double W16C8[8] = {1.0, 0.92387953251128675612818318939679, 0.70710678118654752440084436210485,
                                        0.38268343236508977172845998403040, 0.0, -0.38268343236508977172845998403040,
                                        -0.70710678118654752440084436210485, -0.92387953251128675612818318939679};
__device__ constexpr double W16S8[8] = {0.0, 0.38268343236508977172845998403040, 0.70710678118654752440084436210485,
                                        0.92387953251128675612818318939679, 1.0, 0.92387953251128675612818318939679,
                                        0.70710678118654752440084436210485, 0.38268343236508977172845998403040};

template <typename T, int STORE = 0>
__global__ __launch_bounds__(R_THREADS, 4) void k_r8_rows(const FusedArgs<T> A)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[r8_lds_bytes<T>()];
    const R8Lane L;
    const R8Setup<T> S(smem, A, L);
    const int mpair = L.tid & ~1;
    const bool odd = L.lane & 1;
    __syncthreads();

    for (int g = blockIdx.x; g < A.total; g += gridDim.x) {
        const int b = g / A.rows, r = g - b * A.rows;
        const C *ph = A.phasor + (size_t)r * 64;
        const __amdgpu_buffer_rsrc_t rs_sig =
            __builtin_amdgcn_make_buffer_rsrc((void *)(A.sig + (size_t)b * F_N), 0, F_N * (int)sizeof(C), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(A.spec + (size_t)b * 8192), 0, 8192 * (int)sizeof(C), 0x00020000);
        C e[8], o[8];
        // ---- even-bin chain ----------------------------------------------------------------
        r8_mixer<T, 0>(e, rs_sig, ph, S.th, L);
        r8_forward(e, S.tw, S.twB, S.twC, S.Lc, L);
#pragma unroll
        for (int k = 0; k < 8; ++k)  // spectrum product (xcor_rustfft.rs:64-73)
            e[k] = cmul(e[k], bload(rs_spec, (unsigned)(L.tid * sizeof(C)), (unsigned)(512 * k * sizeof(C)), (C *)nullptr));
        r8_inverse(e, S.tw, S.twB, S.twC, S.Lc, L);
        // ---- odd-bin chain -----------------------------------------------------------------
        r8_mixer<T, 1>(o, rs_sig, ph, S.th, L);
        r8_forward(o, S.tw, S.twB, S.twC, S.Lc, L);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            o[k] = cmul(o[k], bload(rs_spec, (unsigned)((4096 + L.tid) * sizeof(C)), (unsigned)(512 * k * sizeof(C)), (C *)nullptr));
        r8_inverse(o, S.tw, S.twB, S.twC, S.Lc, L);

        // ---- last radix-2 stage (registers) + |.|^2 + argmax + 16-B write-through stores ------
        T bv_lo = T(0), bv_hi = T(0);
        int bi_lo = 0, bi_hi = 0;
        T *const out = A.surface ? A.surface + (size_t)g * F_L : nullptr;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? F_L * (int)sizeof(T) : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T mlo[2], mhi[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = 2 * j + u;  // m = t + 512*i
                const C w16 = {(T)W16C8[i], (T)W16S8[i]};
                const C z = cmul(cmul(o[i], S.th), w16);  // T^m * O[m]
                mlo[u] = norm_sqr(e[i] + z);              // mod.rs:147
                mhi[u] = norm_sqr(e[i] - z);
                if (mlo[u] > bv_lo) { bv_lo = mlo[u]; bi_lo = i; }
                if (mhi[u] > bv_hi) { bv_hi = mhi[u]; bi_hi = i; }
            }
            const T slo = dpp_xor1<T>(odd ? mlo[0] : mlo[1]);
            const T shi = dpp_xor1<T>(odd ? mhi[0] : mhi[1]);
            const int m = mpair + 512 * (2 * j + (odd ? 1 : 0));
            if constexpr (STORE != 3) {
                store_pair_aux<CAF_AUX_SC1>(rs, (unsigned)(m * sizeof(T)), odd ? slo : mlo[0], odd ? mlo[1] : slo);
                store_pair_aux<CAF_AUX_SC1>(rs, (unsigned)((m + F_N) * sizeof(T)), odd ? shi : mhi[0], odd ? mhi[1] : shi);
            } else {
                asm volatile("" ::"v"(slo), "v"(shi));
            }
        }
        T bv = bv_lo;
        uint32_t bi = bv_lo > T(0) ? (uint32_t)(L.tid + 512 * bi_lo) : 0u;
        if (bv_hi > bv) { bv = bv_hi; bi = (uint32_t)(L.tid + 512 * bi_hi + F_N); }
        wave_arg_reduce_dpp(bv, bi);
        T *sv = reinterpret_cast<T *>(S.scratch);
        uint32_t *si = reinterpret_cast<uint32_t *>(S.scratch + 64);
        if (L.lane == 63) { sv[L.wave] = bv; si[L.wave] = bi; }
        __syncthreads();
        if (L.tid == 0) {
            bv = sv[0];
            bi = si[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) arg_merge(bv, bi, sv[w], si[w]);
            A.row_idx[g] = bi;
            A.row_val[g] = bv;
        }
        // scratch is rewritten only after the next row's barriers
    }
}

}  // namespace caf
