// Lab: what the PCIe link of the box gives the drop-in entry (sdrxgpu_bank_process: 2.56 MB in, 61.4 MB out per 10 000-channel
// block).  D2H of the block's output volume as one copy, as pieces on one stream, as two streams side by side, with an H2D beside
// it, and as a KERNEL that stores into the mapped host buffer (no SDMA engine, no copy command).  Page-locked by hipHostMalloc
// and by hipHostRegister of malloc'ed memory (what a caller's ring chunk is).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/lab/pcie_d2h.bin tools/lab/pcie_d2h.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);              \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// every lane moves 16 bytes per step; consecutive lanes consecutive addresses (whole 1 KB bursts per wave)
__global__ void k_blit(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

// a stand-in for "kernels are running": spins the vector ALU of every CU for roughly `iters` fma's per lane
__global__ void k_busy(float *out, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; i++) a = fmaf(a, b, 1e-6f);
    if (a == 123.456f) out[0] = a;
}

int main(int argc, char **argv)
{
    const size_t total = (size_t)10000 * 512 * 12;                  // 61.44 MB: IQ + audio rows of the 10 000-channel block
    const size_t audio = (size_t)10000 * 512 * 4;
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    void *d_src = nullptr, *d_in = nullptr;
    CK(hipMalloc(&d_src, total));
    CK(hipMalloc(&d_in, 2560000));
    CK(hipMemset(d_src, 1, total));
    void *h_malloc = nullptr, *h_reg = nullptr, *h_in = nullptr;
    CK(hipHostMalloc(&h_malloc, total, hipHostMallocDefault));
    if (posix_memalign(&h_reg, 4096, total)) return 1;
    memset(h_reg, 0, total);
    CK(hipHostRegister(h_reg, total, hipHostRegisterPortable | hipHostRegisterMapped));
    CK(hipHostMalloc(&h_in, 2560000, hipHostMallocDefault));
    hipStream_t s1, s2, s3;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    float *d_dummy = nullptr;
    CK(hipMalloc((void **)&d_dummy, 64));

    auto timeit = [&](const char *what, size_t bytes, auto &&fn) -> int {
        for (int w = 0; w < 2; w++) {
            fn();
            CK(hipDeviceSynchronize());
        }
        double best = 1e30, sum = 0;
        for (int r = 0; r < reps; r++) {
            const double t0 = now_ms();
            fn();
            CK(hipDeviceSynchronize());
            const double dt = now_ms() - t0;
            best = dt < best ? dt : best;
            sum += dt;
        }
        printf("%-78s %7.3f ms best %7.3f ms mean  %6.1f GB/s (best)\n", what, best, sum / reps, bytes / best / 1e6);
        fflush(stdout);
        return 0;
    };

    for (int kind = 0; kind < 2; kind++) {
        char *h = (char *)(kind ? h_reg : h_malloc);
        const char *nm = kind ? "hipHostRegister" : "hipHostMalloc  ";
        char buf[160];
        snprintf(buf, sizeof buf, "%s D2H 61.44 MB, one copy", nm);
        timeit(buf, total, [&] { (void)hipMemcpyAsync(h, d_src, total, hipMemcpyDeviceToHost, s1); });
        snprintf(buf, sizeof buf, "%s D2H 20.48 MB (audio rows), one copy", nm);
        timeit(buf, audio, [&] { (void)hipMemcpyAsync(h, d_src, audio, hipMemcpyDeviceToHost, s1); });
        snprintf(buf, sizeof buf, "%s D2H 2.05 MB (audio rows of 1 000 channels), one copy", nm);
        timeit(buf, audio / 10, [&] { (void)hipMemcpyAsync(h, d_src, audio / 10, hipMemcpyDeviceToHost, s1); });
        for (int pieces : {4, 8, 16}) {
            snprintf(buf, sizeof buf, "%s D2H 61.44 MB as %d pieces on ONE stream", nm, pieces);
            timeit(buf, total, [&] {
                const size_t per = total / pieces;
                for (int p = 0; p < pieces; p++) (void)hipMemcpyAsync(h + p * per, (char *)d_src + p * per, per, hipMemcpyDeviceToHost, s1);
            });
        }
        for (int streams : {2, 3}) {
            snprintf(buf, sizeof buf, "%s D2H 61.44 MB as %d parts on %d streams side by side", nm, streams, streams);
            timeit(buf, total, [&] {
                hipStream_t ss[3] = {s1, s2, s3};
                const size_t per = total / streams;
                for (int p = 0; p < streams; p++) (void)hipMemcpyAsync(h + p * per, (char *)d_src + p * per, per, hipMemcpyDeviceToHost, ss[p]);
            });
        }
        snprintf(buf, sizeof buf, "%s D2H 61.44 MB with the 2.56 MB H2D of the next block beside it", nm);
        timeit(buf, total, [&] {
            (void)hipMemcpyAsync(h, d_src, total, hipMemcpyDeviceToHost, s1);
            (void)hipMemcpyAsync(d_in, h_in, 2560000, hipMemcpyHostToDevice, s2);
        });
        float4 *h_dev = nullptr;
        CK(hipHostGetDevicePointer((void **)&h_dev, h, 0));
        for (int blocks : {8, 16, 32, 64, 256}) {
            snprintf(buf, sizeof buf, "%s kernel stores into the mapped host buffer, 61.44 MB, %d workgroups of 256", nm, blocks);
            timeit(buf, total, [&] { hipLaunchKernelGGL(k_blit, dim3(blocks), dim3(256), 0, s1, (const float4 *)d_src, h_dev, total / 16); });
        }
        snprintf(buf, sizeof buf, "%s kernel stores, 20.48 MB (audio rows), 32 workgroups", nm);
        timeit(buf, audio, [&] { hipLaunchKernelGGL(k_blit, dim3(32), dim3(256), 0, s1, (const float4 *)d_src, h_dev, audio / 16); });
        // the copy (engine / kernel) while every CU is busy with something else: does the copy slow down, does the other work?
        snprintf(buf, sizeof buf, "%s busy kernel alone (1024 x 256 lanes, 400 000 fma each)", nm);
        timeit(buf, 0, [&] { hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, s2, d_dummy, 400000); });
        snprintf(buf, sizeof buf, "%s D2H copy of 61.44 MB beside the busy kernel", nm);
        timeit(buf, total, [&] {
            hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, s2, d_dummy, 400000);
            (void)hipMemcpyAsync(h, d_src, total, hipMemcpyDeviceToHost, s1);
        });
        snprintf(buf, sizeof buf, "%s 32-workgroup store kernel of 61.44 MB beside the busy kernel", nm);
        timeit(buf, total, [&] {
            hipLaunchKernelGGL(k_busy, dim3(1024), dim3(256), 0, s2, d_dummy, 400000);
            hipLaunchKernelGGL(k_blit, dim3(32), dim3(256), 0, s1, (const float4 *)d_src, h_dev, total / 16);
        });
    }
    snprintf((char *)h_in, 16, "x");
    timeit("H2D 2.56 MB (one block of 10 MS/s)", 2560000, [&] { (void)hipMemcpyAsync(d_in, h_in, 2560000, hipMemcpyHostToDevice, s1); });
    // hand-over latency: event on s1 -> wait on s2 -> tiny copy
    timeit("empty kernel + event + wait on a second stream + 4 KB D2H (hand-over latency)", 4096, [&] {
        hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s1, d_dummy, 1);
        hipEvent_t e;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        (void)hipEventRecord(e, s1);
        (void)hipStreamWaitEvent(s2, e, 0);
        (void)hipMemcpyAsync(h_malloc, d_src, 4096, hipMemcpyDeviceToHost, s2);
        (void)hipEventDestroy(e);
    });
    timeit("empty kernel + 4 KB D2H on the SAME stream", 4096, [&] {
        hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s1, d_dummy, 1);
        (void)hipMemcpyAsync(h_malloc, d_src, 4096, hipMemcpyDeviceToHost, s1);
    });
    timeit("empty kernel alone (launch + sync)", 0, [&] { hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s1, d_dummy, 1); });
    return 0;
}
