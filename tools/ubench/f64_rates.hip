// Micro-benchmark: wave-level issue rate of FP64 VALU instructions on gfx950.
// Each kernel runs N_ITER iterations of 16 independent chains of one instruction kind.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITER 4096

template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, double seed)
{
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x * 1e-9 + i;
    const double c = seed * 0.999, d = seed * 1e-3;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(d));
            if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
            if (KIND == 3) asm volatile("v_fma_f64 %0, %0, 1.0, %1" : "+v"(x[i]) : "v"(d));   // add via fma
            if (KIND == 4) { float f = (float)x[i]; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"((float)c), "v"((float)d)); x[i] = f; }
            if (KIND == 5) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
            if (KIND == 6) { int lo = (int)__double_as_longlong(x[i]); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(7)); x[i] = lo; }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int wg_per_cu)
{
    double *out;
    const int cus = 256, grid = cus * wg_per_cu;
    hipMalloc(&out, grid * 256 * sizeof(double));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<KIND><<<grid, 256>>>(out, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<grid, 256>>>(out, 1.0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per SIMD: wg_per_cu waves (256 threads = 4 waves = 1 per SIMD), each N_ITER*16 instrs
    const double instr_per_simd = (double)wg_per_cu * N_ITER * 16;
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name,
           wg_per_cu, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f64", w);
        run<1>("v_mul_f64", w);
        run<2>("v_fma_f64", w);
        run<3>("v_fma_f64(x,1.0,d)", w);
        run<5>("v_fmac_f64", w);
        run<4>("v_fma_f32(+cvt)", w);
        run<6>("v_cndmask_b32(+cvt)", w);
    }
    return 0;
}
