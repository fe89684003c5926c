// Micro-benchmark: what a burst of 16-B-per-lane global stores costs a compute-bound wave on gfx950.
// Each 256-thread workgroup (one wave per SIMD) loops ROWS times over:
//   FMA_PER_ROW dependent-free v_fma_f64  +  NST stores of 16 B (or 8 B) per lane,
// in three placements: no stores / stores in one burst at the end of the row / stores spread
// evenly through the FMAs.  Reported: time per row, i.e. the cost of the stores on top of the math.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int ROWS = 200;
constexpr int FMA_PER_ROW = 3200;  // ~ the row kernel's VALU count

template <int MODE, int WIDTH, int ALT = 0>  // MODE 0 none, 1 burst, 2 spread; WIDTH 16 or 8 bytes per lane; ALT: even/odd lanes -> rows 2 KiB apart
__global__ __launch_bounds__(256, 2) void k(unsigned char *out, size_t region, double seed)
{
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x * 1e-9 + i;
    const double c = seed * 0.999, d = seed * 1e-3;
    constexpr int NST = WIDTH == 16 ? 16 : 32;               // 64 KiB per row either way
    unsigned char *base = out + (size_t)blockIdx.x * region;  // this workgroup's output rows
    for (int row = 0; row < ROWS; ++row) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)(row % 4) * 65536, 0, 65536, 0x00020000);
        auto store = [&](int j) {
            const long long a = __double_as_longlong(x[j & 15]);
            if constexpr (WIDTH == 16) {
                v4u v = {(unsigned)a, (unsigned)(a >> 32), (unsigned)a, (unsigned)(a >> 32)};
                const unsigned off = ALT ? (threadIdx.x >> 1) * 16 + (threadIdx.x & 1) * 2048 + (threadIdx.x >> 6) * 0 : threadIdx.x * 16;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, j * 4096, 16);
            } else {
                v2u v = {(unsigned)a, (unsigned)(a >> 32)};
                __builtin_amdgcn_raw_buffer_store_b64(v, rs, threadIdx.x * 8, j * 2048, 16);
            }
        };
#pragma unroll
        for (int blk = 0; blk < NST; ++blk) {
#pragma unroll
            for (int i = 0; i < FMA_PER_ROW / NST; ++i)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i & 15]) : "v"(c), "v"(d));
            if (MODE == 2) store(blk);
        }
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < NST; ++j) store(j);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678) out[0] = 1;
}

template <int MODE, int WIDTH, int ALT = 0>
void run(const char *name, int wg_per_cu, unsigned char *out, size_t region)
{
    const int grid = 256 * wg_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<MODE, WIDTH, ALT><<<grid, 256>>>(out, region, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE, WIDTH, ALT><<<grid, 256>>>(out, region, 1.0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-40s wg/CU=%d  %.3f ms  -> %.2f us per row\n", name, wg_per_cu, ms, ms * 1e3 / ROWS);
}

// Same math with NL LDS operations of 16 B per lane per row spread through it: KIND 0 = ds_write_b128,
// 1 = ds_read_b128 (results consumed once per row), conflict-free unit-stride addresses.
template <int KIND, int NL>
__global__ __launch_bounds__(256, 2) void kl(unsigned char *out, double seed)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x * 1e-9 + i;
    const double c = seed * 0.999, d = seed * 1e-3;
    double2 *L = reinterpret_cast<double2 *>(smem) + threadIdx.x;
    double acc = 0;
    for (int row = 0; row < ROWS; ++row) {
#pragma unroll
        for (int blk = 0; blk < NL; ++blk) {
#pragma unroll
            for (int i = 0; i < FMA_PER_ROW / NL; ++i)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i & 15]) : "v"(c), "v"(d));
            if (KIND == 0) {
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"((unsigned)(size_t)L), "v"(*reinterpret_cast<v4u *>(&x[(2 * blk) & 14])), "n"((blk & 15) * 4096) : "memory");
            } else {
                v4u r;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"((unsigned)(size_t)L), "n"((blk & 15) * 4096) : "memory");
                if (blk == NL - 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); acc += (double)r.x; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    double s = acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678) out[0] = 1;
}

template <int KIND, int NL>
void runl(const char *name, int wg_per_cu, unsigned char *out)
{
    const int grid = 256 * wg_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    kl<KIND, NL><<<grid, 256>>>(out, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    kl<KIND, NL><<<grid, 256>>>(out, 1.0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-40s wg/CU=%d  %.3f ms  -> %.2f us per row\n", name, wg_per_cu, ms, ms * 1e3 / ROWS);
}

int main()
{
    const size_t region = 4 * 65536;
    unsigned char *out;
    hipMalloc(&out, region * 512);
    for (int w : {1, 2}) {
        run<0, 16>("math only", w, out, region);
        run<1, 16>("math + 16 x 16-B stores, burst", w, out, region);
        run<2, 16>("math + 16 x 16-B stores, spread", w, out, region);
        run<1, 16, 1>("math + 16 x 16-B alt-row stores, burst", w, out, region);
        run<2, 16, 1>("math + 16 x 16-B alt-row stores, spread", w, out, region);
        run<1, 8>("math + 32 x 8-B stores, burst", w, out, region);
        run<2, 8>("math + 32 x 8-B stores, spread", w, out, region);
        runl<0, 128>("math + 128 ds_write_b128, spread", w, out);
        runl<1, 128>("math + 128 ds_read_b128, spread", w, out);
        runl<0, 64>("math + 64 ds_write_b128, spread", w, out);
    }
    hipFree(out);
    return 0;
}
