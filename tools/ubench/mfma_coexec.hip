// Micro-benchmark: does the FP64 matrix pipe of gfx950 run BESIDE the FP64 vector pipe of the same SIMD, and at
// what price to the vector stream?  (the matrix pipe is 100 % idle in the issue-bound
// complex128 row kernel; before any radix-16 stage is rewritten as a DFT-16 matmul this says what it could buy.)
//
// Each 256-thread workgroup (one wave per SIMD; 1 or 2 workgroups per CU = 1 or 2 waves per SIMD, as in
// store_rates.hip) loops ROWS times over FMA_PER_ROW independent v_fma_f64 (16 accumulators: the row kernel's
// VALU count) with K v_mfma_f64_16x16x4_f64 spread evenly through them, on NACC independent accumulator tiles
// (so no MFMA waits for another's result).  Also: MFMAs alone (the cadence of the matrix pipe) and a
// dependent chain on one accumulator (its latency).
// Build: hipcc -O3 -Wno-unused-result --offload-arch=gfx950 -o mfma_coexec mfma_coexec.hip
// Output per variant: microseconds per row (HIP events) and shader cycles per row (s_memtime of wave 0 of
// workgroup 0), from which: cycles per MFMA alone, and (t_K - t_0) / K = what one interleaved MFMA costs the
// vector stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int ROWS = 100;
constexpr int FMA_PER_ROW = 3200;
constexpr int NACC = 4;

template <int K, int FMAS>  // K MFMAs and FMAS vector FMAs per row
__global__ __launch_bounds__(256, 2) void k_mix(double *out, unsigned long long *cyc, double seed)
{
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x * 1e-9 + i;
    const double c = seed * 0.999, d = seed * 1e-3;
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{seed, seed * 2, seed * 3, seed * 4};
    const double a = seed * 1e-3 + threadIdx.x * 1e-6, b = seed * 2e-3;
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int row = 0; row < ROWS; ++row) {
        constexpr int SEG = K > 0 ? K : 1;       // K segments, one MFMA at the end of each
        constexpr int PER = FMAS / SEG;          // vector FMAs per segment
#pragma unroll
        for (int s = 0; s < SEG; ++s) {
#pragma unroll
            for (int i = 0; i < PER; ++i)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i & 15]) : "v"(c), "v"(d));
            if (K > 0)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[s % NACC]) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// MFMAs alone.  DEP = 1: one accumulator, every MFMA waits for the one before (latency); else NACC tiles round-robin.
template <int DEP>
__global__ __launch_bounds__(256, 2) void k_mfma(double *out, unsigned long long *cyc, double seed)
{
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{seed, seed * 2, seed * 3, seed * 4};
    const double a = seed * 1e-3 + threadIdx.x * 1e-6, b = seed * 2e-3;
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int row = 0; row < ROWS; ++row) {
#pragma unroll
        for (int s = 0; s < 64; ++s)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[DEP ? 0 : s % NACC]) : "v"(a), "v"(b));
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (s == 12345.678) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename F>
static void timeit(const char *name, int wg_per_cu, int cus, double *out, unsigned long long *cyc, F launch, int mfma_per_row,
                   int fma_per_row)
{
    const int grid = cus * wg_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch(grid);
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch(grid);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD=%d  %8.3f us/row  %9.1f cyc/row  (mfma/row %3d, fma/row %4d)\n", name, wg_per_cu, ms * 1e3 / ROWS,
           (double)c / ROWS, mfma_per_row, fma_per_row);
    hipEventDestroy(a);
    hipEventDestroy(b);
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no HIP device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz; ROWS %d\n", prop.gcnArchName, cus, prop.clockRate, ROWS);
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, 64);
    hipMalloc(&cyc, 64);
#define MIX(K, F) timeit("fma + " #K " mfma_f64_16x16x4", w, cus, out, cyc, [&](int g) { k_mix<K, F><<<g, 256>>>(out, cyc, 1.0); }, K, F)
    for (int w : {1, 2}) {
        MIX(0, FMA_PER_ROW);
        MIX(8, FMA_PER_ROW);
        MIX(16, FMA_PER_ROW);
        MIX(32, FMA_PER_ROW);
        MIX(64, FMA_PER_ROW);
        MIX(128, FMA_PER_ROW);
        // what a DFT-16 stage moved to the matrix pipe would look like: 64 MFMAs in, ~200 vector instructions out
        MIX(64, 3008);
        timeit("mfma only, 4 independent tiles", w, cus, out, cyc, [&](int g) { k_mfma<0><<<g, 256>>>(out, cyc, 1.0); }, 64, 0);
        timeit("mfma only, dependent chain", w, cus, out, cyc, [&](int g) { k_mfma<1><<<g, 256>>>(out, cyc, 1.0); }, 64, 0);
    }
#undef MIX
    hipFree(out);
    hipFree(cyc);
    return 0;
}
