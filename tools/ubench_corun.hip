// ubench_corun.hip -- can a small persistent memory-bound kernel (one workgroup per CU, 18 KB of LDS, few registers) stream beside
// the compute-bound matrix-pipe kernel (two 4-wave workgroups per CU, 67 KB of LDS, ~210 VGPRs) without either starving?
// M: the phase loop of tools/ubench_mfma_fir.hip (MODE 2); R: a reader that keeps 16 KB per workgroup in flight over a 1 GB buffer
// until a stop flag is raised.  Reports M alone, R alone, and both together (R launched first).  Tuning aid, not product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int PITCH = 816, ROWS = 32;

__global__ __launch_bounds__(256, 2) void kM(const float2 *__restrict__ src, float *__restrict__ out, int bodies, float h16, const float *__restrict__ h2g)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < ROWS * PITCH / 8; i += 256) reinterpret_cast<float2 *>(lds)[i] = src[(blockIdx.x * 131 + i) & 0xffff];
    __syncthreads();
    const int lo = lane & 15, hi = lane >> 4;
    const char *tA = lds + lo * PITCH + hi * 8, *xB = lds + (16 + lo) * PITCH + hi * 8;
    float hA[4], h2[19];
    for (int j = 0; j < 4; j++) hA[j] = 0.01f * (4 * j + hi + 1);
    for (int j = 0; j < 19; j++) h2[j] = h2g[j];
    v2f acc2[4][8], ysum[8];
    for (int a = 0; a < 4; a++) for (int i = 0; i < 8; i++) acc2[a][i] = v2f{0.f, 0.f};
    for (int i = 0; i < 8; i++) ysum[i] = v2f{0.f, 0.f};
    for (int b = 0; b < bodies; b++) {
        int boff = (b & 1) << 3;
        asm volatile("" : "+v"(boff));
#pragma unroll
        for (int p = 0; p < 20; p++) {
            v4f Pr = {0, 0, 0, 0}, Pi = Pr, Qr = Pr, Qi = Pr;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                v2f t = *reinterpret_cast<const v2f *>(tA + boff + (5 * p + 4 * j) * 8) * v2f{hA[j], hA[j]};
                const v2f x = *reinterpret_cast<const v2f *>(xB + boff + (5 * p + 4 * j) * 8);
                Pr = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, x.x, Pr, 0, 0, 0);
                Pi = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, x.y, Pi, 0, 0, 0);
                Qr = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, x.x, Qr, 0, 0, 0);
                Qi = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, x.y, Qi, 0, 0, 0);
            }
            v2f Y[8] = {v2f{Pr[0], Pr[1]} - v2f{Qi[0], Qi[1]}, v2f{Pr[2], Pr[3]} - v2f{Qi[2], Qi[3]}, v2f{Pi[0], Pi[1]} + v2f{Qr[0], Qr[1]},
                        v2f{Pi[2], Pi[3]} + v2f{Qr[2], Qr[3]}, v2f{Pr[0], Pr[1]} + v2f{Qi[0], Qi[1]}, v2f{Pr[2], Pr[3]} + v2f{Qi[2], Qi[3]},
                        v2f{Pi[0], Pi[1]} - v2f{Qr[0], Qr[1]}, v2f{Pi[2], Pi[3]} - v2f{Qr[2], Qr[3]}};
            const int u = p % 5, s = p / 5;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int j = u + 5 * d;
                if (j > 18) continue;
                const int a = (s - d) & 3;
#pragma unroll
                for (int i = 0; i < 8; i++) acc2[a][i] = __builtin_elementwise_fma(Y[i], v2f{h2[j], h2[j]}, acc2[a][i]);
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("" : "+v"(acc2[a][i]));
            }
            if (u == 3) for (int i = 0; i < 8; i++) ysum[i] += acc2[(s - 3) & 3][i];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sres = 0.f;
    for (int i = 0; i < 8; i++) sres += ysum[i].x + ysum[i].y;
    out[blockIdx.x * 256 + tid] = sres;
}

// persistent reader: workgroup w reads 16 KB chunks w, w + gridDim, ... of `buf` (n_chunks in all, wrapping) until *stop != 0
// or it has done `max_iter` chunks; every chunk goes through LDS (as a tail kernel would stage it); bytes read are counted
__global__ __launch_bounds__(256) void kR(const v4f *__restrict__ buf, size_t n_chunks, const int *stop, unsigned long long *bytes, float *sink, int max_iter)
{
    __shared__ v4f tile[1024 + 128];   // 16 KB + 2 KB = 18 KB
    const int tid = threadIdx.x;
    float acc = 0.f;
    unsigned long long mine = 0;
    size_t c = blockIdx.x;
    for (int it = 0; it < max_iter; it++, c += gridDim.x) {
        if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const v4f *src = buf + (c % n_chunks) * 1024;
        v4f v0 = src[tid], v1 = src[tid + 256], v2 = src[tid + 512], v3 = src[tid + 768];
        tile[tid] = v0; tile[tid + 256] = v1; tile[tid + 512] = v2; tile[tid + 768] = v3;
        __syncthreads();
        for (int k = 0; k < 8; k++) acc += tile[(tid * 7 + k * 131) & 1023].x;
        __syncthreads();
        mine += 16384;
    }
    if (tid == 0) atomicAdd(bytes, mine);
    sink[blockIdx.x * 256 + tid] = acc;
}

__global__ void kStop(int *stop, int v) { *stop = v; }

int main()
{
    std::vector<float2> h(65536);
    srand(1);
    for (auto &v : h) v = make_float2((rand() % 2001 - 1000) * 1e-3f, (rand() % 2001 - 1000) * 1e-3f);
    float2 *src; hipMalloc(&src, h.size() * sizeof(float2));
    hipMemcpy(src, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
    float h2h[19]; for (int j = 0; j < 19; j++) h2h[j] = 0.05f + 0.001f * j;
    float *h2; hipMalloc(&h2, sizeof(h2h)); hipMemcpy(h2, h2h, sizeof(h2h), hipMemcpyHostToDevice);
    const int gridM = 1440, bodies = 13;
    const size_t ldsM = 67584;
    hipFuncSetAttribute((const void *)kM, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsM);
    float *outM; hipMalloc(&outM, gridM * 256 * sizeof(float));
    const size_t n_chunks = 65536;     // 1 GiB
    v4f *buf; hipMalloc(&buf, n_chunks * 16384); hipMemset(buf, 0, n_chunks * 16384);
    int *stop; hipMalloc(&stop, 4);
    unsigned long long *bytes; hipMalloc(&bytes, 8);
    float *sink; hipMalloc(&sink, 2048 * 256 * sizeof(float));
    hipStream_t sA, sB; hipStreamCreateWithFlags(&sA, hipStreamNonBlocking); hipStreamCreateWithFlags(&sB, hipStreamNonBlocking);
    hipEvent_t e0, e1, r0, r1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&r0); hipEventCreate(&r1);
    auto zero = [&]() { hipMemset(stop, 0, 4); hipMemset(bytes, 0, 8); hipDeviceSynchronize(); };
    for (int i = 0; i < 50; i++) kM<<<gridM, 256, ldsM, sA>>>(src, outM, bodies, 0.01f, h2);
    hipDeviceSynchronize();
    float msM = 0, msR = 0; unsigned long long nb = 0;
    // M alone
    hipEventRecord(e0, sA);
    for (int i = 0; i < 10; i++) kM<<<gridM, 256, ldsM, sA>>>(src, outM, bodies, 0.01f, h2);
    hipEventRecord(e1, sA); hipEventSynchronize(e1); hipEventElapsedTime(&msM, e0, e1);
    printf("M alone: %.3f ms per launch (%d workgroups x %d bodies)\n", msM / 10, gridM, bodies);
    const float msM1 = msM / 10;
    // R alone, as many workgroups as fit (2048) and one per CU (256), 1 GiB each
    for (int g : {2048, 256, 512}) {
        zero();
        hipEventRecord(r0, sB);
        kR<<<g, 256, 0, sB>>>(buf, n_chunks, stop, bytes, sink, (int)(n_chunks / g));
        hipEventRecord(r1, sB); hipEventSynchronize(r1); hipEventElapsedTime(&msR, r0, r1);
        hipMemcpy(&nb, bytes, 8, hipMemcpyDeviceToHost);
        printf("R alone, %4d workgroups: %.3f ms for %.2f GB = %.2f TB/s\n", g, msR, nb / 1e9, nb / 1e9 / msR);
    }
    // together: R (persistent, g workgroups) first, then 3 launches of M, then the stop flag
    for (int g : {256, 512}) {
        zero();
        hipEventRecord(r0, sB);
        kR<<<g, 256, 0, sB>>>(buf, n_chunks, stop, bytes, sink, 1 << 30);
        hipEventRecord(r1, sB);
        hipEventRecord(e0, sA);
        for (int i = 0; i < 3; i++) kM<<<gridM, 256, ldsM, sA>>>(src, outM, bodies, 0.01f, h2);
        hipEventRecord(e1, sA);
        kStop<<<1, 1, 0, sA>>>(stop, 1);
        hipEventSynchronize(r1); hipEventSynchronize(e1);
        hipEventElapsedTime(&msM, e0, e1); hipEventElapsedTime(&msR, r0, r1);
        hipMemcpy(&nb, bytes, 8, hipMemcpyDeviceToHost);
        printf("together, R with %3d workgroups: M %.3f ms per launch (alone %.3f: +%.1f %%), R read %.2f GB in %.3f ms = %.2f TB/s beside it\n",
               g, msM / 3, msM1, 100.0 * (msM / 3 / msM1 - 1.0), nb / 1e9, msR, nb / 1e9 / msR);
    }
    return 0;
}
