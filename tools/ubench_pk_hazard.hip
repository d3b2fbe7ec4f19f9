// ubench_pk_hazard.hip -- does gfx950 need a software wait state between a packed-fp32 VALU op and a VALU op that
// reads its result?  Dependent chains written in inline asm WITHOUT any s_nop, checked against the same arithmetic in
// plain C++ (tuning aid, not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void k(const float2 *in, float2 *out_asm, float2 *out_ref, float hs0, float hs1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    v2f a = {in[i].x, in[i].y}, b = {in[i + 4096].x, in[i + 4096].y}, c = {in[i + 8192].x, in[i + 8192].y};
    const v2f hp = {hs0, hs1};
    v2f r0, r1, r2, r3, s0, s1;
    asm volatile(
        "v_pk_mul_f32 %0, %6, %7\n\t"                                  // r0 = a*b
        "v_pk_fma_f32 %1, %0, %8, %0\n\t"                              // r1 = r0*c + r0          (packed reads packed, next instruction)
        "v_pk_add_f32 %2, %1, %0\n\t"                                  // r2 = r1 + r0
        "v_pk_fma_f32 %3, %2, %9, %1 op_sel_hi:[1,0,1]\n\t"            // r3 = r2*hp.lo + r1      (SGPR pair, low half broadcast)
        "v_pk_fma_f32 %3, %2, %9, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"   // r3 += r2*hp.hi
        "v_pk_mul_f32 %4, %3, %3\n\t"                                  // s0 = r3^2
        "v_pk_fma_f32 %5, %3, %4, %4 op_sel:[1,0,0] op_sel_hi:[0,1,1]\n\t"   // s1 = swap(r3)*s0 + s0
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(s0), "=&v"(s1)
        : "v"(a), "v"(b), "v"(c), "s"(hp));
    out_asm[i] = make_float2(s1.x, s1.y);
    v2f q0 = a * b;
    v2f q1 = __builtin_elementwise_fma(q0, c, q0);
    v2f q2 = q1 + q0;
    v2f q3 = __builtin_elementwise_fma(q2, v2f{hs0, hs0}, q1);
    q3 = __builtin_elementwise_fma(q2, v2f{hs1, hs1}, q3);
    const v2f t0 = q3 * q3;
    const v2f t1 = __builtin_elementwise_fma(v2f{q3.y, q3.x}, t0, t0);
    out_ref[i] = make_float2(t1.x, t1.y);
}

int main()
{
    const int n = 4096;
    std::vector<float2> h(3 * n);
    for (int i = 0; i < 3 * n; i++) h[i] = make_float2((float)((i * 7919) % 1000) * 1e-3f - 0.5f, (float)((i * 104729) % 1000) * 1e-3f - 0.5f);
    float2 *d, *oa, *orf;
    hipMalloc(&d, h.size() * sizeof(float2));
    hipMalloc(&oa, n * sizeof(float2));
    hipMalloc(&orf, n * sizeof(float2));
    hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
    int bad = 0;
    for (int rep = 0; rep < 50; rep++) {
        k<<<n / 256, 256>>>(d, oa, orf, 0.37f, -1.25f);
        std::vector<float2> a(n), r(n);
        hipMemcpy(a.data(), oa, n * sizeof(float2), hipMemcpyDeviceToHost);
        hipMemcpy(r.data(), orf, n * sizeof(float2), hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++)
            if (a[i].x != r[i].x || a[i].y != r[i].y) bad++;
    }
    printf("packed-fp32 dependent chains without wait states: %d mismatches in %d values\n", bad, 50 * n);
    return bad != 0;
}
