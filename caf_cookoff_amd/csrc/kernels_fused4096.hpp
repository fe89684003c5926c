// kernels_fused4096.hpp -- fused Doppler-row kernel for n = 4096 (L = 2n = 8192).
//
// One 512-thread workgroup (8 waves, 2 per SIMD) computes one whole CAF row
//   mag[k] = | IDFT_L( Hs[.] * IDFT_L(conj(needle*w^n ++ 0))[.] )[k] |^2
// (= mod.rs:138-151 with FFT(haystack) hoisted; conj(FFT(s)) == IDFT(conj s), so
// every transform here is a positive-exponent one) entirely in VGPRs + LDS:
//
//   * The zero padding (mod.rs:130) makes the first radix-2 stage of the
//     8192-point transform free: even bins  = IDFT_4096(u[n]),
//                                 odd bins   = IDFT_4096(u[n]*e^{+2*pi*i*n/8192}).
//     The extra half-bin rotation is folded into the per-row Doppler phasor table,
//     so the mixer (a1) costs nothing extra for the odd chain.
//   * lanes 0-31 of every wave run the EVEN chain, lanes 32-63 the ODD chain of the
//     same 32 butterflies; 8 waves x 32 = 256 butterfly columns x 16 points = 4096.
//   * each 4096-point transform is radix-16 x 16 x 16, 16 points per lane in
//     registers.  Forward is DIF (natural in, digit-reversed out), the product with
//     the pre-permuted haystack spectrum happens in registers, the inverse is the
//     mirrored DIT (digit-reversed in, natural out): 4 LDS exchanges per row and
//     chain, of which the two inner ones stay inside one wave (no barrier).
//   * the last radix-2 stage  c[m], c[m+4096] = E[m] +- T^m O[m]  pairs lane l with
//     lane l+32 of the same wave: v_permlane32_swap, no LDS, no barrier.
//   * inter-pass twiddles W_4096^(t*k) live in registers for the lifetime of the
//     (persistent) workgroup, W_256^((t&15)*k) in a 256-entry LDS table: no
//     twiddle traffic to L2/HBM per row.
//   * |.|^2, the first-max argmax (mod.rs:143-151) and the coalesced surface store
//     are the epilogue of the last butterfly.
//
// LDS: (2 chains x 4096 + 256) x sizeof(complex) + 256 B = 132.25 KiB (f64) / 66.25 KiB (f32).
// Barriers: 3 per row.  HBM traffic per row: the 2n-real output row once; inputs,
// spectrum and phasor tables are L2-resident.
#pragma once
#include "cplx.hpp"

namespace caf {

constexpr int F_N = 4096;       // samples per input
constexpr int F_L = 8192;       // padded transform length
constexpr int F_THREADS = 512;  // 8 waves
constexpr int F_COLS = 256;     // butterfly columns per chain

// Row-independent tables (built once per context by k_fused_tables):
//   tw4096[m] = e^{+2*pi*i*m/4096}, m < 4096
//   th[t]     = e^{+2*pi*i*t/8192}, t < 256
template <typename T>
struct FusedTables {
    const cpx<T> *tw4096;
    const cpx<T> *th;
};

template <typename T>
__global__ void k_fused_tables(cpx<T> *__restrict__ tw4096, cpx<T> *__restrict__ th)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 4096) tw4096[i] = cispi_f64<T>(2.0 * (double)i / 4096.0);
    if (i < 256) th[i] = cispi_f64<T>(2.0 * (double)i / 8192.0);
}

// Per-row phasor tables of a plan (one thread per (row, chain, t)):
//   base[row][c][t] = e^{j*ph*t}        * (c ? e^{-2*pi*i*t/8192}       : 1),  t < 256
//   step[row][c][q] = e^{j*ph*256*q}    * (c ? e^{-2*pi*i*256*q/8192}   : 1),  q < 16
// so that needle[t+256q]*base*step = needle[n]*w^n*(c ? conj(T^n) : 1) and the chain
// input is its conjugate.  Row `nrows` (one past the end) is the f = 0 row used to
// transform the haystack.
template <typename T>
__global__ void k_fused_phasors(const double *__restrict__ ph, int nrows,
                                cpx<T> *__restrict__ base, cpx<T> *__restrict__ step)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g / 512, rem = g % 512, c = rem / 256, t = rem % 256;
    if (row > nrows) return;
    const double p = row < nrows ? ph[row] : 0.0;
    {
        double s, co, s2, c2;
        sincos(p * (double)t, &s, &co);
        sincospi(c ? -2.0 * (double)t / 8192.0 : 0.0, &s2, &c2);
        base[(size_t)row * 512 + c * 256 + t] = {(T)(co * c2 - s * s2), (T)(co * s2 + s * c2)};
    }
    if (t < 16) {
        const int q = t;
        double s, co, s2, c2;
        sincos(p * (double)(256 * q), &s, &co);
        sincospi(c ? -2.0 * (double)(256 * q) / 8192.0 : 0.0, &s2, &c2);
        step[(size_t)row * 32 + c * 16 + q] = {(T)(co * c2 - s * s2), (T)(co * s2 + s * c2)};
    }
}

// ---- radix-16 butterfly, positive exponent, natural-order in and out -----------
template <typename T>
__device__ __forceinline__ void dft4(cpx<T> &a, cpx<T> &b, cpx<T> &c, cpx<T> &d)
{
    const cpx<T> apc = a + c, amc = a - c, bpd = b + d, jbmd = muli(b - d);
    a = apc + bpd;   // X0
    b = amc + jbmd;  // X1 = x0 + i x1 - x2 - i x3
    c = apc - bpd;   // X2
    d = amc - jbmd;  // X3
}

template <typename T>
__device__ __forceinline__ cpx<T> mul_w8(cpx<T> a)  // * e^{i*pi/4}
{
    const T r = T(0.70710678118654752440084436210485);
    return {r * (a.x - a.y), r * (a.x + a.y)};
}
template <typename T>
__device__ __forceinline__ cpx<T> mul_w8_3(cpx<T> a)  // * e^{i*3*pi/4}
{
    const T r = T(0.70710678118654752440084436210485);
    return {-r * (a.x + a.y), r * (a.x - a.y)};
}

template <typename T>
__device__ __forceinline__ void swp(cpx<T> &a, cpx<T> &b) { const cpx<T> t = a; a = b; b = t; }

template <typename T>
__device__ __forceinline__ void dft16(cpx<T> (&v)[16])
{
    const T c1 = T(0.92387953251128675612818318939679);  // cos(pi/8)
    const T s1 = T(0.38268343236508977172845998403040);  // sin(pi/8)
    // stage 1: over q1 for each q0 (inputs v[q0 + 4*q1]) -> a[r0;q0] at v[q0 + 4*r0]
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) dft4(v[q0], v[q0 + 4], v[q0 + 8], v[q0 + 12]);
    // W16^(q0*r0)
    v[5] = cmul(v[5], cpx<T>{c1, s1});     // e=1
    v[9] = mul_w8(v[9]);                   // e=2
    v[13] = cmul(v[13], cpx<T>{s1, c1});   // e=3
    v[6] = mul_w8(v[6]);                   // e=2
    v[10] = muli(v[10]);                   // e=4
    v[14] = mul_w8_3(v[14]);               // e=6
    v[7] = cmul(v[7], cpx<T>{s1, c1});     // e=3
    v[11] = mul_w8_3(v[11]);               // e=6
    v[15] = cmul(v[15], cpx<T>{-c1, -s1}); // e=9
    // stage 2: over q0 for each r0 -> X[r0 + 4*r1] at v[4*r0 + r1]
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0) dft4(v[4 * r0], v[4 * r0 + 1], v[4 * r0 + 2], v[4 * r0 + 3]);
    // 4x4 transpose of the register names -> X[k] at v[k]
    swp(v[1], v[4]); swp(v[2], v[8]); swp(v[3], v[12]);
    swp(v[6], v[9]); swp(v[7], v[13]); swp(v[11], v[14]);
}

// Orders this wave's LDS accesses without a workgroup barrier: LDS operations of
// one wave execute in issue order, so a compiler-level fence is all that is needed
// for the wave-local exchanges (the 256-element block a wave owns is read and
// written by that wave only).
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// lanes 32-63 of `a` <-> lanes 0-31 of `b`
__device__ __forceinline__ void swap32_u(unsigned &a, unsigned &b)
{
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
__device__ __forceinline__ void swap32(double &a, double &b)
{
    unsigned alo = (unsigned)__double_as_longlong(a), ahi = (unsigned)(__double_as_longlong(a) >> 32);
    unsigned blo = (unsigned)__double_as_longlong(b), bhi = (unsigned)(__double_as_longlong(b) >> 32);
    swap32_u(alo, blo);
    swap32_u(ahi, bhi);
    a = __longlong_as_double(((long long)ahi << 32) | alo);
    b = __longlong_as_double(((long long)bhi << 32) | blo);
}
__device__ __forceinline__ void swap32(float &a, float &b)
{
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    swap32_u(ua, ub);
    a = __uint_as_float(ua);
    b = __uint_as_float(ub);
}
template <typename T>
__device__ __forceinline__ void swap32(cpx<T> &a, cpx<T> &b)
{
    swap32(a.x, b.x);
    swap32(a.y, b.y);
}

// W_32^i = e^{2*pi*i*i/32}, i < 8
__device__ constexpr double W32_COS[8] = {1.0, 0.98078528040323044912618223613424, 0.92387953251128675612818318939679,
                                          0.83146961230254523707878837761791, 0.70710678118654752440084436210485,
                                          0.55557023301960222474283081394853, 0.38268343236508977172845998403040,
                                          0.19509032201612826784828486847702};
__device__ constexpr double W32_SIN[8] = {0.0, 0.19509032201612826784828486847702, 0.38268343236508977172845998403040,
                                          0.55557023301960222474283081394853, 0.70710678118654752440084436210485,
                                          0.83146961230254523707878837761791, 0.92387953251128675612818318939679,
                                          0.98078528040323044912618223613424};

template <typename T>
struct FusedArgs {
    const cpx<T> *sig;      // PREPARE: haystack [batch][4096]; else needle [batch][4096]
    cpx<T> *spec;           // Hs [batch][2][16][256]: PREPARE writes, else reads
    const cpx<T> *ph_base;  // [rows+1][2][256]
    const cpx<T> *ph_step;  // [rows+1][2][16]
    FusedTables<T> tab;
    T *surface;             // [batch][rows][8192] or nullptr
    uint64_t *row_idx;      // [batch][rows]
    T *row_val;             // [batch][rows]
    int rows;               // rows per surface handled by this plan
    int total;              // batch*rows (PREPARE: batch)
};

// LDS bytes: 2 chains x 4096 complex + 256-entry W_256 table + argmax scratch
template <typename T>
constexpr size_t fused_lds_bytes() { return (2 * 4096 + 256) * sizeof(cpx<T>) + 256; }

template <typename T, bool PREPARE>
__global__ __launch_bounds__(F_THREADS) void k_fused_rows(const FusedArgs<T> A)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[fused_lds_bytes<T>()];
    C *const lds = reinterpret_cast<C *>(smem);
    C *const twb = lds + 2 * 4096;  // twb[k*16 + lo] = W_256^(lo*k)
    unsigned char *const scratch = smem + (2 * 4096 + 256) * sizeof(C);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int chain = lane >> 5;              // 0: even bins (E), 1: odd bins (O)
    const int t = wave * 32 + (lane & 31);    // butterfly column 0..255
    const int hi4 = t >> 4, lo4 = t & 15;
    C *const Lc = lds + chain * 4096;

    // ---- register-resident twiddles --------------------------------------------
    // ---- twiddles: W_4096^(t*k) in registers, W_256^(lo4*k) in LDS ---------------
    C twA[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) twA[k] = A.tab.tw4096[t * k];
    if (tid < 256) twb[tid] = A.tab.tw4096[16 * (tid & 15) * (tid >> 4)];
    __syncthreads();
    const C *const twB = twb + lo4;  // twB[16*k]
    // last-stage twiddle base: T^(t + 256*m2), lower lanes own m2 = 8+i -> extra *i
    C tbase = A.tab.th[t];
    if (chain == 0) tbase = muli(tbase);

    C v[16];
    int parity = 0;
    for (int g = blockIdx.x; g < A.total; g += gridDim.x, parity ^= 1) {
        const int b = PREPARE ? g : g / A.rows;
        const int r = PREPARE ? A.rows : g % A.rows;  // PREPARE uses the f=0 row

        // ---- mixer (mod.rs:46-65) fused into the first butterfly load -----------
        {
            const C *sig = A.sig + (size_t)b * F_N;
            const C pb = A.ph_base[(size_t)r * 512 + chain * 256 + t];
            const C *ps = A.ph_step + (size_t)r * 32 + chain * 16;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const C x = sig[t + 256 * q];
                v[q] = conj(cmul(cmul(x, pb), ps[q]));
            }
        }
        // ---- forward (DIF) pass 1: over n2, twiddle W_4096^(t*k0) ---------------
        dft16(v);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twA[k]);
        // exchange 1: [k0][t]  ->  thread (k0'=hi4, n0'=lo4) gathers over n1
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[k * 256 + t] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[hi4 * 256 + 16 * k + lo4];
        // ---- pass 2: over n1, twiddle W_256^(n0'*k1) ----------------------------
        dft16(v);
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twB[16 * k]);
        // exchange 2 (wave-local, XOR-swizzled): [k0'][k1][n0'^k1]
#pragma unroll
        for (int k = 0; k < 16; ++k) Lc[hi4 * 256 + k * 16 + (lo4 ^ k)] = v[k];
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = Lc[hi4 * 256 + lo4 * 16 + (k ^ lo4)];
        // ---- pass 3: over n0 -> G[k0 + 16*k1 + 256*k2], k2 = register ----------
        dft16(v);

        if constexpr (PREPARE) {
            // Hs = conj(IDFT(conj h)) / L = FFT(h)/L, stored in register layout
            C *spec = A.spec + (size_t)b * (2 * 16 * 256) + chain * (16 * 256);
            const T inv = T(1.0 / 8192.0);
#pragma unroll
            for (int k = 0; k < 16; ++k) spec[k * 256 + t] = {v[k].x * inv, -v[k].y * inv};
            wave_lds_fence();
            __syncthreads();  // next iteration's exchange-1 writes vs this one's reads
            continue;
        } else {
            // ---- spectrum product (xcor_rustfft.rs:64-73) -----------------------
            const C *spec = A.spec + (size_t)b * (2 * 16 * 256) + chain * (16 * 256);
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = cmul(v[k], spec[k * 256 + t]);
            // ---- inverse (DIT) pass I: over k2 -> m0 ----------------------------
            dft16(v);
            // exchange 3 (wave-local): thread (k0,k1) -> [k0][k1][m0^k1]
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 16; ++k) Lc[hi4 * 256 + lo4 * 16 + (k ^ lo4)] = v[k];
            wave_lds_fence();
            // thread (k0, m0=lo4) gathers over k1, twiddle W_256^(k1*m0)
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = Lc[hi4 * 256 + k * 16 + (lo4 ^ k)];
#pragma unroll
            for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twB[16 * k]);
            // ---- pass II: over k1 -> m1 -----------------------------------------
            dft16(v);
            // exchange 4: [k0][16*m1 + m0] -> thread j=t gathers over k0
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 16; ++k) Lc[hi4 * 256 + 16 * k + lo4] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = Lc[k * 256 + t];
#pragma unroll
            for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], twA[k]);
            // ---- pass III: over k0 -> y[t + 256*m2], m2 = register ---------------
            dft16(v);

            // ---- last radix-2 stage across the lane halves + epilogue -----------
            // after the swap: lanes 0-31 hold (E,O)[t+256*(8+i)], lanes 32-63 (E,O)[t+256*i]
            T bv = T(0);
            uint32_t bi = 0;
            T *out = A.surface ? A.surface + (size_t)g * F_L : nullptr;
            const int mbase = t + (chain == 0 ? 2048 : 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                C o = v[i], e = v[8 + i];
                swap32(e, o);
                const C w32 = {(T)W32_COS[i], (T)W32_SIN[i]};  // W_32^i
                const C z = cmul(cmul(o, tbase), w32);
                const T m_lo = norm_sqr(e + z);   // mod.rs:147
                const T m_hi = norm_sqr(e - z);
                const int m = mbase + 256 * i;
                if (out) {
                    out[m] = m_lo;
                    out[m + F_N] = m_hi;
                }
                arg_merge(bv, bi, m_lo, (uint32_t)m);
                arg_merge(bv, bi, m_hi, (uint32_t)(m + F_N));
            }
            wave_arg_reduce(bv, bi);
            T *sv = reinterpret_cast<T *>(scratch + parity * 128);
            uint32_t *si = reinterpret_cast<uint32_t *>(scratch + parity * 128 + 64);
            if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
            __syncthreads();  // also orders exchange-4 reads before the next exchange-1 writes
            if (tid == 0) {
#pragma unroll
                for (int w = 1; w < 8; ++w) arg_merge(bv, bi, sv[w], si[w]);
                A.row_idx[g] = bi;
                A.row_val[g] = bv;
            }
        }
    }
}

}  // namespace caf
