// ubench_hbm_read.hip -- what read bandwidth does a 1 GB streaming read reach on one MI355X, as a function of how
// the reads are issued?  (tuning aid for k_tail, not product)
//   mode 0: grid-stride, each thread U x 16-byte loads in flight, persistent grid (G workgroups per CU)
//   mode 1: one workgroup per 15 KB span (k_tail_s's shape): 256 threads x 4 loads, then exit
//   mode 2: one workgroup per 100 KB row, chunks of 6 KB, D chunks in flight (k_tail_stream's shape)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void k_stride(const v4f *__restrict__ x, size_t n, float *out)
{
    v4f acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc.x == 123.456f) out[0] = acc.y + acc.z + acc.w;
}

__global__ __launch_bounds__(256) void k_span(const v4f *__restrict__ x, size_t n, float *out, int span16)
{
    __shared__ v4f buf[1024];
    const size_t base = (size_t)blockIdx.x * span16;
    v4f acc = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < span16; i += 256)
        if (base + i < n) buf[i] = x[base + i];
    __syncthreads();
    acc = buf[(threadIdx.x * 7) & 1023 % span16];
    if (acc.x == 123.456f) out[0] = acc.y;
}

template <int D>
__global__ __launch_bounds__(192) void k_rows(const v4f *__restrict__ x, size_t row16, int n_rows, float *out)
{
    __shared__ v4f buf[2][384];
    const v4f *row = x + (size_t)blockIdx.x * row16;
    const int n_chunks = (int)(row16 / 384);
    v4f r[D][2];
#pragma unroll
    for (int d = 0; d < D; d++)
        for (int k = 0; k < 2; k++) r[d][k] = d < n_chunks ? row[d * 384 + threadIdx.x * 2 + k] : v4f{0, 0, 0, 0};
    v4f acc = {0, 0, 0, 0};
    for (int c = 0; c < n_chunks; c += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            buf[d & 1][threadIdx.x * 2] = r[d][0];
            buf[d & 1][threadIdx.x * 2 + 1] = r[d][1];
            const int cn = c + d + D;
            for (int k = 0; k < 2; k++) r[d][k] = cn < n_chunks ? row[cn * 384 + threadIdx.x * 2 + k] : v4f{0, 0, 0, 0};
            __syncthreads();
            acc += buf[d & 1][(threadIdx.x * 5) % 384];
        }
    }
    if (acc.x == 123.456f) out[0] = acc.y;
}

int main()
{
    const size_t bytes = (size_t)10000 * 12864 * 8;      // the stage-2 scratch of 10 000 channels
    const size_t n = bytes / 16;
    v4f *d;
    float *o;
    hipMalloc(&d, bytes);
    hipMalloc(&o, 64);
    hipMemset(d, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        launch();
        hipDeviceSynchronize();
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-44s %.3f ms  %.2f TB/s\n", name, best, bytes / best * 1e-9);
    };
    for (int g : {4, 8, 16}) {
        char nm[64];
        snprintf(nm, 64, "grid-stride U=4, %d WG/CU", g);
        time(nm, [&] { hipLaunchKernelGGL(k_stride<4>, dim3(256 * g), dim3(256), 0, 0, d, n, o); });
        snprintf(nm, 64, "grid-stride U=8, %d WG/CU", g);
        time(nm, [&] { hipLaunchKernelGGL(k_stride<8>, dim3(256 * g), dim3(256), 0, 0, d, n, o); });
    }
    for (int span : {512, 963, 1024}) {
        char nm[64];
        snprintf(nm, 64, "one WG per %d x16 B span", span);
        time(nm, [&] { hipLaunchKernelGGL(k_span, dim3((unsigned)((n + span - 1) / span)), dim3(256), 0, 0, d, n, o, span); });
    }
    const size_t row16 = 12864 * 8 / 16;
    time("row stream, 1 chunk ahead", [&] { hipLaunchKernelGGL(k_rows<1>, dim3(10000), dim3(192), 0, 0, d, row16, 10000, o); });
    time("row stream, 2 chunks ahead", [&] { hipLaunchKernelGGL(k_rows<2>, dim3(10000), dim3(192), 0, 0, d, row16, 10000, o); });
    time("row stream, 4 chunks ahead", [&] { hipLaunchKernelGGL(k_rows<4>, dim3(10000), dim3(192), 0, 0, d, row16, 10000, o); });
    return 0;
}
