// kernels_generic.hpp -- size-generic (any power-of-two n) CAF kernels.
//
// This is the correctness path for every shape the fused row kernel
// (kernels_fused4096.hpp) does not cover: small/odd test sizes, the standalone
// apply_freq_shift / xcor entry points, and L = 2n > 8192.  It is still a HIP
// path (the product has no CPU fallback), organised as plain batched passes over
// HBM-resident buffers:
//   mix+pad -> log2(L) Stockham radix-2 stages -> spectrum product ->
//   log2(L) stages -> |.|^2 + per-row argmax -> peak
// Every load/store is lane-contiguous (coalesced); no LDS tiling is attempted
// here -- the hot 400x8192 shape never takes this path.
#pragma once
#include "cplx.hpp"
#include "kernels_fused4096.hpp"  // dft4 / dft16 butterflies for the radix-16 passes
#include "../../include/caf_hip.h"

namespace caf {

// ph[r] = ((2*PI)*f)*dt, dt = 1.0/fs -- mod.rs:54-56 evaluated left to right in
// f64 with explicit round-to-nearest multiplies (no contraction possible).
__global__ void k_phase(const double *__restrict__ freqs, int nfreq, uint32_t fs,
                        double *__restrict__ ph)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nfreq) return;
    const double dt = __ddiv_rn(1.0, (double)fs);
    const double two_pi = __dmul_rn(2.0, 3.14159265358979323846264338327950288);
    ph[r] = __dmul_rn(__dmul_rn(two_pi, freqs[r]), dt);
}

// W_L^k = e^{+2*pi*i*k/L}, k < L/2 (inverse direction; forward uses conj).
template <typename T>
__global__ void k_twiddle(cpx<T> *__restrict__ tw, size_t half_len, size_t L)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= half_len) return;
    tw[k] = cispi_f64<T>(2.0 * (double)k / (double)L);
}

// a1 (mod.rs:46-65): out[i] = in[i]*e^{j*ph*i}
template <typename T>
__global__ void k_apply_shift(const cpx<T> *__restrict__ in, size_t n, double ph,
                              cpx<T> *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // sample 0 is multiplied by the recurrence's initial 1 + 0j with a full complex multiply (mod.rs:57-60) whatever ph is (fs == 0
    // makes it inf / NaN): a finite sample comes out unchanged, an inf / NaN component spreads exactly as it does in the reference
    out[i] = i == 0 ? cmul(in[0], cpx<T>{T(1), T(0)}) : cmul(in[i], cis_f64<T>(ph * (double)i));
}

// Row r of batch b: dst[(b*rows + r)*L + i] = i<n ? needle[b*n+i]*e^{j*ph[r]*i} : 0
// (mod.rs:130 zero-pad at the END, :138 mixer).  rows==1 && ph==nullptr: plain pad
// (used for the haystack, mod.rs:131).
template <typename T>
__global__ void k_mix_pad(const cpx<T> *__restrict__ src, size_t n, size_t L,
                          const double *__restrict__ ph, size_t rows,
                          cpx<T> *__restrict__ dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t br = blockIdx.y;  // b*rows + r
    if (i >= L) return;
    const size_t b = br / rows, r = br % rows;
    cpx<T> v = {T(0), T(0)};
    if (i < n) {
        v = src[b * n + i];
        if (ph) v = cmul(v, cis_f64<T>(ph[r] * (double)i));
    }
    dst[br * L + i] = v;
}

// One Stockham radix-2 stage over `nbatch` length-L rows: x -> y.
//   n_cur: current sub-transform length, s = L / n_cur (stride), m = n_cur/2
//   y[q + s*2p]     = a + b
//   y[q + s*(2p+1)] = (a - b) * w^p,  w = e^{dir*2*pi*i/n_cur}
template <typename T>
__global__ void k_fft_stage(const cpx<T> *__restrict__ x, cpx<T> *__restrict__ y,
                            const cpx<T> *__restrict__ tw, size_t L, size_t n_cur,
                            int inverse)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // < L/2
    const size_t row = blockIdx.y;
    const size_t half = L >> 1;
    if (g >= half) return;
    const size_t s = L / n_cur;
    const size_t q = g % s, p = g / s;
    const cpx<T> a = x[row * L + g];
    const cpx<T> b = x[row * L + g + half];
    cpx<T> w = tw[p * s];  // W_L^(p*L/n_cur) = W_ncur^p
    if (!inverse) w.y = -w.y;
    y[row * L + q + s * (2 * p)] = a + b;
    y[row * L + q + s * (2 * p + 1)] = cmul(a - b, w);
}

// One Stockham pass of radix R (16 while at least 16 points remain, then 8 / 4 / 2) over `nbatch` length-L rows: x -> y,
// the same recursion as the radix-2 stage above with R-point butterflies (dft16 / dft8 / dft4 in registers):
//   butterfly (p, q), p < n_cur / R, q < s = L / n_cur:  in_j = x[q + s (p + (n_cur / R) j)],
//   y[q + s (R p + k)] = DFT_R(in)_k * w^(p k),  w = e^{dir 2 pi i / n_cur}
// log16(L) passes over HBM instead of log2(L): the path of every shape no LDS-resident kernel covers.
// tw holds W_L^m for m < L/2 (W^(m + L/2) = -W^m); the negative-exponent direction conjugates in and out.
template <typename T, int R>
__global__ __launch_bounds__(256) void k_fft_pass(const cpx<T> *__restrict__ x, cpx<T> *__restrict__ y,
                                                  const cpx<T> *__restrict__ tw, size_t L, size_t n_cur, int inverse)
{
    const size_t bid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // < L / R
    const size_t row = blockIdx.y;
    if (bid >= L / R) return;
    const size_t s = L / n_cur, m = n_cur / R, half = L >> 1;
    const size_t p = bid / s, q = bid % s;
    const cpx<T> *xr = x + row * L;
    cpx<T> *yr = y + row * L;
    cpx<T> v[16];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        v[j] = xr[q + s * (p + m * j)];
        if (!inverse) v[j].y = -v[j].y;
    }
    if constexpr (R == 16) {
        dft16(v);
    } else if constexpr (R == 8) {
        dft4(v[0], v[2], v[4], v[6]);
        dft4(v[1], v[3], v[5], v[7]);
        const cpx<T> o1 = mul_w8(v[3]), o2 = muli(v[5]), o3 = mul_w8_3(v[7]);
        const cpx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
        v[0] = e0 + o0; v[4] = e0 - o0;
        v[1] = e1 + o1; v[5] = e1 - o1;
        v[2] = e2 + o2; v[6] = e2 - o2;
        v[3] = e3 + o3; v[7] = e3 - o3;
    } else if constexpr (R == 4) {
        dft4(v[0], v[1], v[2], v[3]);
    } else {
        const cpx<T> a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        cpx<T> o = v[k];
        if (k && m > 1) {
            const size_t e = (p * (size_t)k * s) % L;  // W_ncur^(p k) = W_L^(p k s)
            cpx<T> w = tw[e < half ? e : e - half];
            if (e >= half) { w.x = -w.x; w.y = -w.y; }
            o = cmul(o, w);
        }
        if (!inverse) o.y = -o.y;
        yr[q + s * ((size_t)R * p + k)] = o;
    }
}

// C[row][k] = (H[b][k] * conj(S[row][k])) / L   (xcor_rustfft.rs:64-73; conj then
// multiply then divide by n, in that order)
template <typename T>
__global__ void k_mul_conj(const cpx<T> *__restrict__ H, cpx<T> *__restrict__ S,
                           size_t L, size_t rows)
{
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t br = blockIdx.y;
    if (k >= L) return;
    const size_t b = br / rows;
    const cpx<T> h = H[b * L + k];
    const cpx<T> p = cmulc(h, S[br * L + k]);
    const T nn = (T)L;
    S[br * L + k] = {p.x / nn, p.y / nn};
}

// |.|^2 + first-max argmax of one row per workgroup (mod.rs:143-151).
template <typename T>
__global__ __launch_bounds__(256) void k_mag_argmax(const cpx<T> *__restrict__ c, size_t L,
                                                    T *__restrict__ surface,
                                                    uint64_t *__restrict__ row_idx,
                                                    T *__restrict__ row_val)
{
    __shared__ T s_v[4];
    __shared__ uint32_t s_i[4];
    const size_t row = blockIdx.x;
    T bv = T(0);
    uint32_t bi = 0;
    for (size_t i = threadIdx.x; i < L; i += blockDim.x) {
        const T m = norm_sqr(c[row * L + i]);
        if (surface) surface[row * L + i] = m;
        arg_merge(bv, bi, m, (uint32_t)i);
    }
    wave_arg_reduce(bv, bi);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_v[wave] = bv; s_i[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) arg_merge(bv, bi, s_v[w], s_i[w]);
        row_idx[row] = bi;
        row_val[row] = bv;
    }
}

// find_peak (mod.rs:31-42): first row (list order) whose peak is strictly greater
// than the running best, starting from (freq 0.0, idx 0, val 0.0).
// One workgroup per batch entry.  Streaming slots (caf_stream_*) pass `host`: the device mappings
// of the slot's pinned result buffers; the workgroup then also stages its surface's row peaks and
// its caf_peak record out to host memory (the graph needs no separate stage-out node).
struct PeakStageOut {
    caf_peak *peak;      // [batch] or nullptr
    uint64_t *row_idx;   // [batch][rows]
    void *row_val;       // [batch][rows] of the plan's real type
};
template <typename T>
__global__ __launch_bounds__(256) void k_peak(const double *__restrict__ freqs,
                                              const uint64_t *__restrict__ row_idx,
                                              const T *__restrict__ row_val, int rows,
                                              int64_t row_base, caf_peak *__restrict__ out, const PeakStageOut host)
{
    __shared__ double s_v[4];
    __shared__ uint32_t s_i[4];
    const size_t b = blockIdx.x;
    double bv = 0.0;
    uint32_t br = 0xffffffffu;  // "no row" sorts last among equal (zero) values
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const double v = (double)row_val[b * rows + r];
        if (v > 0.0) arg_merge(bv, br, v, (uint32_t)r);
        if (host.peak) {
            host.row_idx[b * rows + r] = row_idx[b * rows + r];
            ((T *)host.row_val)[b * rows + r] = row_val[b * rows + r];
        }
    }
    wave_arg_reduce(bv, br);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_v[wave] = bv; s_i[wave] = br; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) arg_merge(bv, br, s_v[w], s_i[w]);
        caf_peak p;
        if (br == 0xffffffffu) {
            p.val = 0.0; p.freq = 0.0; p.idx = 0; p.row = -1;
        } else {
            p.val = bv; p.freq = freqs[br]; p.idx = row_idx[b * rows + br];
            p.row = row_base + (int64_t)br;
        }
        out[b] = p;
        if (host.peak) host.peak[b] = p;
    }
}

// Go / Python flavoured view of a |.|^2 surface: out[r][i] = sqrt(surf[r][(off - i) mod L]),
// width = L (Go, off = n) or n (Python 'same', off = n/2).
template <typename T>
__global__ void k_view(const T *__restrict__ surf, size_t L, size_t width, size_t off, T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t r = blockIdx.y;
    if (i >= width) return;
    const size_t src = (off + L - (i % L)) % L;
    out[r * width + i] = sqrt(surf[r * L + src]);
}

// Probe pair for "do these two HIP streams run concurrently?" (caf_api.hip: streams_overlap): a bounded
// ~0.2 ms idle loop on one stream, an empty kernel on the other.
__global__ void k_idle(unsigned iters)
{
    for (unsigned i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
}
__global__ void k_empty() {}

// Streaming stage-in / stage-out (caf_stream_*): up to three (src, dst, bytes) jobs copied by ONE
// kernel node of the slot's graph.  One side of every job is PINNED HOST memory mapped into the
// device address space (hipHostMalloc is coherent: device accesses are uncached), so the two
// hipMemcpyAsync H2D nodes and the three D2H nodes of a slot become two kernel nodes, which replay
// several microseconds faster than copy-engine nodes.  Sizes are multiples of 4 bytes; buffers are
// 16-byte aligned.
struct CopyJobs {
    const void *src[3];
    void *dst[3];
    unsigned long long bytes[3];
};
__global__ __launch_bounds__(256) void k_stage_copy(const CopyJobs J)
{
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const size_t n16 = J.bytes[j] / 16, n4 = (J.bytes[j] % 16) / 4;
        const uint4 *s = (const uint4 *)J.src[j];
        uint4 *d = (uint4 *)J.dst[j];
        for (size_t i = tid; i < n16; i += nthr) d[i] = s[i];
        if (tid < n4) ((uint32_t *)(d + n16))[tid] = ((const uint32_t *)(s + n16))[tid];
    }
}

}  // namespace caf
