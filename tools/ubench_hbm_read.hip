// ubench_hbm_read.hip -- what read bandwidth does a 1 GB streaming read reach on one MI355X, as a function of how
// the reads are issued?  (tuning aid for k_tail, not product)
//   mode 0: grid-stride, each thread U x 16-byte loads in flight, persistent grid (G workgroups per CU)
//   mode 1: one workgroup per 15 KB span (k_tail_s's shape): 256 threads x 4 loads, then exit
//   mode 2: one workgroup per 100 KB row, chunks of 6 KB, D chunks in flight (k_tail_stream's shape)
//   mode 3: k_tail_stream's loads exactly: 256 threads, chunks of NB x 6 KB as 8-row x 8-segment pieces, D chunks in flight in registers
//           (inline-assembly loads, counted waits), 16-byte LDS stores, two barriers per chunk, rows last to first
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

template <int NB, int D>
__global__ __launch_bounds__(256) void k_rows2(const v4f *__restrict__ x, size_t row16, int n_rows, float *out, int lds_pad16)
{
    extern __shared__ v4f buf2[];
    constexpr int NWP = NB * 6, NPC = (NWP + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const v4f *row = x + (size_t)(n_rows - 1 - (int)blockIdx.x) * row16;
    const int n_chunks = (int)((row16 < 6400 ? row16 : 6400) / (384 * NB));
    int goff[NPC], loff[NPC];
#pragma unroll
    for (int u = 0; u < NPC; u++) {
        const int w = wave + 4 * u, kb = w / 6, ww = w - 6 * kb;
        const int r = 8 * (ww >> 1) + (lane & 7), sg = 8 * (ww & 1) + (lane >> 3);
        goff[u] = w < NWP ? kb * 384 + r * 16 + sg : 0;
        loff[u] = w < NWP ? kb * 384 + sg * 24 + r : -1;
    }
    v4f r[D][NPC];
    auto fetch = [&](int c, v4f (&q)[NPC]) __attribute__((always_inline)) {
        c = c < n_chunks ? c : n_chunks - 1;
#pragma unroll
        for (int u = 0; u < NPC; u++) {
            const v4f *p = row + (size_t)c * 384 * NB + goff[u];
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[u]) : "v"(p) : "memory");
        }
    };
#pragma unroll
    for (int d = 0; d < D; d++) fetch(d, r[d]);
    v4f acc = {0, 0, 0, 0};
    for (int c = 0; c < n_chunks; c += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"((D - 1) * NPC) : "memory");
#pragma unroll
            for (int u = 0; u < NPC; u++) asm volatile("" : "+v"(r[d][u]));
#pragma unroll
            for (int u = 0; u < NPC; u++)
                if (loff[u] >= 0) buf2[loff[u]] = r[d][u];
            fetch(c + d + D, r[d]);
            __syncthreads();
            acc += buf2[(tid * 5) % (384 * NB)];
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc.x == 123.456f) out[0] = acc.y;
    (void)lds_pad16;
}

__global__ void k_fill(float *x, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
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
    auto rows2 = [&](const char *tag) {
        char nm[96];
        for (int lds : {16, 36}) {       // KB of LDS per workgroup: what decides the workgroups per CU (9 / 4)
            snprintf(nm, 96, "k_tail_stream's loads NB=2 D=3, %d KB LDS%s", lds, tag);
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rows2<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds * 1024);
            time(nm, [&] { hipLaunchKernelGGL((k_rows2<2, 3>), dim3(10000), dim3(256), lds * 1024, 0, d, row16, 10000, o, 0); });
            snprintf(nm, 96, "k_tail_stream's loads NB=1 D=4, %d KB LDS%s", lds, tag);
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rows2<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds * 1024);
            time(nm, [&] { hipLaunchKernelGGL((k_rows2<1, 4>), dim3(10000), dim3(256), lds * 1024, 0, d, row16, 10000, o, 0); });
        }
    };
    rows2(" (constant data)");
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (float *)d, bytes / 4);
    hipDeviceSynchronize();
    rows2(" (random data)");
    // the row pitch: 10 000 rows of 6 400 pieces read, rows `pitch` pieces apart (the buffer holds 10 000 x 6 432)
    for (int pitch : {6432, 6400, 6408, 6416, 6424}) {
        char nm[96];
        snprintf(nm, 96, "k_tail_stream's loads NB=1 D=4, row pitch %d B (%s)", pitch * 16, pitch * 16 % 4096 == 0 ? "multiple of 4 KB" : "not a multiple of 4 KB");
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rows2<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024);
        time(nm, [&] { hipLaunchKernelGGL((k_rows2<1, 4>), dim3(10000), dim3(256), 36 * 1024, 0, d, (size_t)pitch, 10000, o, 0); });
    }
    // sustained: the same span reader 2 000 times back to back (0.3 s of nothing but reads), mean of the last 500
    {
        auto launch = [&] { hipLaunchKernelGGL(k_span, dim3((unsigned)((n + 963 - 1) / 963)), dim3(256), 0, 0, d, n, o, 963); };
        for (int i = 0; i < 1500; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 500; i++) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %.3f ms  %.2f TB/s\n", "one WG per 963 x16 B span, sustained (mean of launches 1 501-2 000)", ms / 500, bytes / (ms / 500) * 1e-9);
    }
    time("row stream, 1 chunk ahead (random data)", [&] { hipLaunchKernelGGL(k_rows<1>, dim3(10000), dim3(192), 0, 0, d, row16, 10000, o); });
    time("one WG per 963 x16 B span (random data)", [&] { hipLaunchKernelGGL(k_span, dim3((unsigned)((n + 963 - 1) / 963)), dim3(256), 0, 0, d, n, o, 963); });
    return 0;
}
