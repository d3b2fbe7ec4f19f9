// ubench_valu.hip -- how fast do v_fma_f32 and v_pk_fma_f32 issue on gfx950?  (tuning aid, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    if (MODE == 0) {
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = threadIdx.x + i;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = __builtin_fmaf(acc[i], a, b);
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        v2f acc[8];
        v2f va = {a, a * 1.5f}, vb = {b, b + 1.f};
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = v2f{(float)threadIdx.x + i, (float)i};
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) acc[i] = __builtin_elementwise_fma(acc[i], va, vb);
        }
        v2f s = {0, 0};
#pragma unroll
        for (int i = 0; i < 8; i++) s += acc[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    }
}

template <int MODE>
void run(const char *name, int wgs_per_cu)
{
    float *d;
    const int grid = 256 * wgs_per_cu;
    hipMalloc(&d, grid * 256 * sizeof(float));
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, 100, 0.999f, 0.001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 256 * iters * 16 * 2;   // both modes: 16 fma lanes-worth per iter per thread
    const double winstr = (double)grid * 4 * iters * (MODE == 0 ? 16 : 8);
    printf("%-10s wgs/cu=%d  %.3f ms  %.1f TFLOP/s  %.2f cycles/wave-instr/SIMD @2.4GHz\n", name, wgs_per_cu, ms, flop / ms / 1e9,
           ms * 1e-3 * 2.4e9 * 1024 / winstr);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma", w);
    }
    return 0;
}
