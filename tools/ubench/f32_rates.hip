// Micro-benchmark: wave-level issue rate of FP32 VALU instructions on gfx950, scalar vs packed.
// Each kernel runs N_ITER iterations of 16 independent chains of one instruction kind.
// Build: hipcc -O3 --offload-arch=gfx950 -o f32_rates f32_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define N_ITER 4096
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed)
{
    float x[16];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = seed + threadIdx.x * 1e-6f + i; p[i] = v2f{x[i], x[i] * 0.5f}; }
    const float c = seed * 0.999f, d = seed * 1e-3f;
    const v2f pc = {c, c * 0.99f}, pd = {d, d * 1.01f};
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
            if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(d));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pc), "v"(pd));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pd));
            if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
            if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(p[i]) : "v"(pc), "v"(pd));
            if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "s"(pc), "v"(pd));
            if (KIND == 7) asm volatile("v_mov_b32 %0, %0" : "+v"(x[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int wg_per_cu)
{
    float *out;
    const int cus = 256, grid = cus * wg_per_cu;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<KIND><<<grid, 256>>>(out, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<grid, 256>>>(out, 1.0f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    // per SIMD: wg_per_cu waves (256 threads = 4 waves = 1 per SIMD), each N_ITER*16 instrs
    const double instr_per_simd = (double)wg_per_cu * N_ITER * 16;
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wg_per_cu, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 3, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_add_f32", w);
        run<2>("v_pk_fma_f32", w);
        run<3>("v_pk_add_f32", w);
        run<4>("v_pk_mul_f32", w);
        run<5>("v_pk_fma_f32 op_sel/neg", w);
        run<6>("v_pk_fma_f32 sgpr src", w);
        run<7>("v_mov_b32", w);
    }
    return 0;
}
