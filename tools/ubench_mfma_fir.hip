// ubench_mfma_fir.hip -- can the f32 matrix pipe carry the mixer + stage-1 FIR while the vector pipe does the rest?
// (tuning aid, not product).  One wave = one tile of 16 conjugate channel pairs x 16 time segments; per stage-1
// output ("phase") 16 x v_mfma_f32_16x16x4_f32 (taps 0..15 of the fused mixer*FIR coefficients, the four real products
// Cr*xr, Cr*xi, Ci*xr, Ci*xi), the 17th tap + the +/- combinations + a transposed-form 19-tap stage 2 on the VALU.
// Operands come from a static LDS tile (no refill: this measures the pipes, not the memory system).
//   MODE 0: MFMAs + their LDS operand reads only      MODE 1: + coefficient build (t*h), tap 16, combinations
//   MODE 2: + stage 2 (the whole phase)                MODE 3: VALU part only (no MFMA)
//   MODE 4: stages 2 and 3 as ONE 119-tap filter in transposed form instead of stage 2: six rotating accumulator sets per value,
//           six per-lane coefficients per phase read from an LDS table (round 4: what a fused stage 3 would cost the phase)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int PITCH = 816;          // bytes per LDS row (100 + 2 samples)
constexpr int ROWS = 32;            // 16 translator rows + 16 input rows

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const float2 *__restrict__ src, float *__restrict__ out, int bodies, float h16,
                                               const float *__restrict__ h2g, unsigned long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < (ROWS * PITCH + 170 * 32) / 8; i += 256) reinterpret_cast<float2 *>(lds)[i] = src[(blockIdx.x * 131 + i) & 0xffff];
    __syncthreads();
    const int lo = lane & 15, hi = lane >> 4;
    const char *tA = lds + lo * PITCH + hi * 8;                 // A operand: t[slot lo][5p + 4j + hi]
    const char *xB = lds + (16 + lo) * PITCH + hi * 8;          // B operand: x[seg lo][5p + 4j + hi]
    const char *tD = lds + (4 * hi) * PITCH;                    // tap 16, D layout: t[slot 4 hi + r][5p + 16]
    const char *xD = lds + (16 + lo) * PITCH;                   //                   x[seg lo][5p + 16]
    float hA[4];
#pragma unroll
    for (int j = 0; j < 4; j++) hA[j] = 0.01f * (4 * j + hi + 1);
    float h2[19];
#pragma unroll
    for (int j = 0; j < 19; j++) h2[j] = h2g[j];
    v2f acc2[4][8];                                            // stage-2 accumulators (transposed form), 8 packed values each
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc2[a][i] = v2f{0.f, 0.f};
    v2f acc6[6][8];                                            // MODE 4: combined stage-2/3 accumulators
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc6[a][i] = v2f{0.f, 0.f};
    // MODE 4: per-lane row of the coefficient table (five phase classes among the 16 segments), 32-byte rows behind the tiles
    const char *wrow = lds + ROWS * PITCH + ((lo % 5) * 7 % 150) * 32;
    v2f ysum[8];
#pragma unroll
    for (int i = 0; i < 8; i++) ysum[i] = v2f{0.f, 0.f};

    unsigned long long t0 = 0, r0 = 0;
    if (stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    struct Ops { v2f t[4], x[4], t16[4], x16; };
    struct Acc { v4f Pr, Pi, Qr, Qi; };
    Ops ops[2];
    Acc acc[2];
    int boff = 0;
    // LDS reads of phase p (all offsets immediates; boff keeps the compiler from hoisting them out of the body loop)
    auto load_ops = [&](int p, Ops &o) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            o.t[j] = *reinterpret_cast<const v2f *>(tA + boff + (5 * p + 4 * j) * 8);
            o.x[j] = *reinterpret_cast<const v2f *>(xB + boff + (5 * p + 4 * j) * 8);
        }
        if (MODE >= 1) {
            o.x16 = *reinterpret_cast<const v2f *>(xD + boff + (5 * p + 16) * 8);
#pragma unroll
            for (int r = 0; r < 4; r++) o.t16[r] = *reinterpret_cast<const v2f *>(tD + boff + r * PITCH + (5 * p + 16) * 8);
        }
    };
    auto mfma_phase = [&](Ops &o, Acc &c) {
        if (MODE >= 1) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const v2f t16 = o.t16[r] * v2f{h16, h16};
                c.Pr[r] = t16.x * o.x16.x;
                c.Pi[r] = t16.x * o.x16.y;
                c.Qr[r] = t16.y * o.x16.x;
                c.Qi[r] = t16.y * o.x16.y;
            }
        } else {
            c.Pr = c.Pi = c.Qr = c.Qi = v4f{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v2f t = o.t[j];
            const v2f x = o.x[j];
            if (MODE >= 1) t *= v2f{hA[j], hA[j]};
            if (MODE != 3) {
                c.Pr = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, x.x, c.Pr, 0, 0, 0);
                c.Pi = __builtin_amdgcn_mfma_f32_16x16x4f32(t.x, x.y, c.Pi, 0, 0, 0);
                c.Qr = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, x.x, c.Qr, 0, 0, 0);
                c.Qi = __builtin_amdgcn_mfma_f32_16x16x4f32(t.y, x.y, c.Qi, 0, 0, 0);
            } else {
                c.Pr[j] += t.x * x.x; c.Pi[j] += t.x * x.y; c.Qr[j] += t.y * x.x; c.Qi[j] += t.y * x.y;
            }
        }
    };
    auto post = [&](int p, const Acc &c) {
        v2f Y[8];
        if (MODE >= 1) {
            Y[0] = v2f{c.Pr[0], c.Pr[1]} - v2f{c.Qi[0], c.Qi[1]};
            Y[1] = v2f{c.Pr[2], c.Pr[3]} - v2f{c.Qi[2], c.Qi[3]};
            Y[2] = v2f{c.Pi[0], c.Pi[1]} + v2f{c.Qr[0], c.Qr[1]};
            Y[3] = v2f{c.Pi[2], c.Pi[3]} + v2f{c.Qr[2], c.Qr[3]};
            Y[4] = v2f{c.Pr[0], c.Pr[1]} + v2f{c.Qi[0], c.Qi[1]};
            Y[5] = v2f{c.Pr[2], c.Pr[3]} + v2f{c.Qi[2], c.Qi[3]};
            Y[6] = v2f{c.Pi[0], c.Pi[1]} - v2f{c.Qr[0], c.Qr[1]};
            Y[7] = v2f{c.Pi[2], c.Pi[3]} - v2f{c.Qr[2], c.Qr[3]};
        } else {
            Y[0] = v2f{c.Pr[0], c.Pr[1]}; Y[1] = v2f{c.Pr[2], c.Pr[3]}; Y[2] = v2f{c.Pi[0], c.Pi[1]}; Y[3] = v2f{c.Pi[2], c.Pi[3]};
            Y[4] = v2f{c.Qr[0], c.Qr[1]}; Y[5] = v2f{c.Qr[2], c.Qr[3]}; Y[6] = v2f{c.Qi[0], c.Qi[1]}; Y[7] = v2f{c.Qi[2], c.Qi[3]};
        }
        if (MODE == 4) {
            // six per-lane coefficients of this phase: one 16-byte and one 8-byte LDS read at a per-lane address
            const v4f c03 = *reinterpret_cast<const v4f *>(wrow + boff + p * 32);
            const v2f c45 = *reinterpret_cast<const v2f *>(wrow + boff + p * 32 + 16);
            const v2f cp[3] = {v2f{c03.x, c03.y}, v2f{c03.z, c03.w}, c45};
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if ((a & 1) == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc6[a][i]) : "v"(Y[i]), "v"(cp[a >> 1]));
                    else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc6[a][i]) : "v"(Y[i]), "v"(cp[a >> 1]));
                }
            if (p == 7) {       // (stands in for the store + reset of a finished set, once per 20 phases here, per 25 in the kernel)
#pragma unroll
                for (int i = 0; i < 8; i++) { ysum[i] += acc6[2][i]; acc6[2][i] = v2f{0.f, 0.f}; }
            }
        } else if (MODE >= 2) {
            // transposed-form stage 2: input u = p % 5 of super-step s = p / 5 feeds the outputs s, s-1, s-2, (s-3)
            const int u = p % 5, s = p / 5;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int j = u + 5 * d;
                if (j > 18) continue;
                const int a = (s - d) & 3;                 // (static: p is a compile-time constant after unrolling)
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    if (d == 0 && u == 0) acc2[a][i] = Y[i] * v2f{h2[j], h2[j]};
                    else acc2[a][i] = __builtin_elementwise_fma(Y[i], v2f{h2[j], h2[j]}, acc2[a][i]);
                }
            }
            if (u == 3) {
                const int a = (s - 3) & 3;
#pragma unroll
                for (int i = 0; i < 8; i++) ysum[i] += acc2[a][i];   // (stands in for the store of a finished output)
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) ysum[i] += Y[i];
        }
    };
    load_ops(0, ops[0]);
    mfma_phase(ops[0], acc[1]);      // (so that the first post() has something defined to read)
    for (int b = 0; b < bodies; b++) {
        boff = (b & 1) << 3;
        asm volatile("" : "+v"(boff));
#pragma unroll
        for (int p = 0; p < 20; p++) {
            // window p: operands of phase p+1 are fetched, phase p runs on the matrix pipe, phase p-1 is finished on the VALU
            load_ops((p + 1) % 20, ops[(p + 1) & 1]);
            mfma_phase(ops[p & 1], acc[p & 1]);
            post((p + 19) % 20, acc[(p + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (stamps && lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[2 * (blockIdx.x * 4 + (tid >> 6))] = t1 - t0;
        stamps[2 * (blockIdx.x * 4 + (tid >> 6)) + 1] = r1 - r0;
    }
    float sres = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) sres += ysum[i].x + ysum[i].y;
    out[blockIdx.x * 256 + tid] = sres;
}

template <int MODE, int WPS>
void run(const char *name, int wgs_per_cu, const float2 *src, const float *h2)
{
    const int grid = 256 * wgs_per_cu * 2;                       // wgs_per_cu resident (the LDS request says so), two rounds
    const size_t ldsb = wgs_per_cu == 1 ? 160 * 1024 - 2048 : 160 * 1024 / wgs_per_cu - 1024;
    hipFuncSetAttribute((const void *)k<MODE, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    float *d;
    unsigned long long *st;
    hipMalloc(&d, grid * 256 * sizeof(float));
    hipMalloc(&st, grid * 8 * sizeof(unsigned long long));
    const int bodies = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 30; i++) k<MODE, WPS><<<grid, 256, ldsb>>>(src, d, bodies, 0.01f, h2, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) k<MODE, WPS><<<grid, 256, ldsb>>>(src, d, bodies, 0.01f, h2, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    k<MODE, WPS><<<grid, 256, ldsb>>>(src, d, bodies, 0.01f, h2, st);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hs(grid * 8);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < grid * 4; i++) { cyc += hs[2 * i]; rt += hs[2 * i + 1]; }
    const double clk_ghz = cyc / rt * 0.1;                       // s_memrealtime ticks at 100 MHz
    const double phases = (double)grid * 4 * bodies * 20;        // wave-phases
    const double ns_per_phase_simd = ms * 1e6 / (phases / 1024); // per SIMD
    // one phase = 512 stage-1 outputs (32 channels x 16 segments) = 2560 channel-samples
    const double chs = phases * 2560 / (ms * 1e-3);
    printf("%-34s wgs/cu=%d waves/simd<=%d  %.3f ms  %.1f ns/phase/SIMD = %.0f cycles @%.2f GHz (stamped)  -> %.2fe12 ch-samples/s  (10k x 320k block: %.3f ms)\n",
           name, wgs_per_cu, WPS, ms, ns_per_phase_simd, ns_per_phase_simd * clk_ghz, clk_ghz, chs / 1e12, 3.2e9 / chs * 1e3);
    hipFree(d);
    hipFree(st);
}

// ---- round 6: the same phase on v_mfma_f32_32x32x2_f32 --------------------------------------------------
// One wave = 16 slots x 32 time segments: A = 32 rows {Cr of slot 0..15, Ci of slot 0..15} x 2 taps, B = 2 taps x 32 segments of
// xr (first MFMA) or xi (second): D_r = [Cr; Ci] xr, D_i = [Cr; Ci] xi, 16 accumulator registers each; 8 K steps per phase for
// taps 0..15.  Per MAC it reads half the B operands and a quarter fewer operand dwords in all than the 16 x 16 x 4 tiling (per
// K step one dword of A and one b64 of B for 2 048 MACs, against 4 dwords for 1 024), for the same matrix-pipe time (16 passes
// for twice the MACs).  The price is state: the tile is twice as large, so a wave's stage-2 accumulators are 128 registers instead
// of 64 and its MFMA accumulators 64 instead of 32 -- one wave per SIMD (512 registers) where the 16 x 16 x 4 kernel runs two.
// Lane l holds D rows 8 (i / 4) + 4 (l / 32) + i % 4, i = 0..15, of column l % 32: row s (Cr, slot s) in register i, row 16 + s
// (Ci) in register i + 8 -- both halves of every combination in one lane.
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int ROWS32 = 48;          // 16 translator rows + 32 input rows

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k32(const float2 *__restrict__ src, float *__restrict__ out, int bodies, float h16,
                                                 const float *__restrict__ h2g, unsigned long long *stamps)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < (ROWS32 * PITCH) / 8; i += 256) reinterpret_cast<float2 *>(lds)[i] = src[(blockIdx.x * 131 + i) & 0xffff];
    __syncthreads();
    const int lo = lane & 31, kh = lane >> 5;
    const char *tA = lds + (lo & 15) * PITCH + kh * 8 + (lo >> 4) * 4;     // A: (re | im) of t[slot lo & 15][5p + 2j + kh]
    const char *xB = lds + (16 + lo) * PITCH + kh * 8;                     // B: x[seg lo][5p + 2j + kh]
    const char *tD = lds + (4 * kh) * PITCH;                               // tap 16, D layout: slots 4 kh + i % 4 + 8 (i / 4), i = 0..7
    const char *xD = lds + (16 + lo) * PITCH;
    float hA[8];
#pragma unroll
    for (int j = 0; j < 8; j++) hA[j] = 0.01f * (2 * j + kh + 1);
    float h2[19];
#pragma unroll
    for (int j = 0; j < 19; j++) h2[j] = h2g[j];
    v2f acc2[4][16];                                           // stage-2 accumulators (transposed form): 8 slots x 2 channels, (re, im)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc2[a][i] = v2f{0.f, 0.f};
    v2f ysum[16];
#pragma unroll
    for (int i = 0; i < 16; i++) ysum[i] = v2f{0.f, 0.f};
    unsigned long long t0 = 0, r0 = 0;
    if (stamps) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    struct Ops { float t[8]; v2f x[8]; v2f t16[8]; v2f x16; };
    struct Acc { v16f Dr, Di; };
    Ops ops[2];
    Acc acc[2];
    int boff = 0;
    auto load_ops = [&](int p, Ops &o) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            o.t[j] = *reinterpret_cast<const float *>(tA + boff + (5 * p + 2 * j) * 8);
            o.x[j] = *reinterpret_cast<const v2f *>(xB + boff + (5 * p + 2 * j) * 8);
        }
        if (MODE >= 1) {
            o.x16 = *reinterpret_cast<const v2f *>(xD + boff + (5 * p + 16) * 8);
#pragma unroll
            for (int i = 0; i < 8; i++) o.t16[i] = *reinterpret_cast<const v2f *>(tD + boff + ((i & 3) + 8 * (i >> 2)) * PITCH + (5 * p + 16) * 8);
        }
    };
    auto mfma_phase = [&](Ops &o, Acc &c) {
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const v2f t16 = o.t16[i] * v2f{h16, h16};
                c.Dr[i] = t16.x * o.x16.x;
                c.Dr[i + 8] = t16.y * o.x16.x;
                c.Di[i] = t16.x * o.x16.y;
                c.Di[i + 8] = t16.y * o.x16.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) c.Dr[i] = c.Di[i] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float t = o.t[j];
            const v2f x = o.x[j];
            if (MODE >= 1) t *= hA[j];
            if (MODE != 3) {
                c.Dr = __builtin_amdgcn_mfma_f32_32x32x2f32(t, x.x, c.Dr, 0, 0, 0);
                c.Di = __builtin_amdgcn_mfma_f32_32x32x2f32(t, x.y, c.Di, 0, 0, 0);
            } else {
                c.Dr[j] += t * x.x; c.Di[j] += t * x.y; c.Dr[j + 8] += t * x.y; c.Di[j + 8] += t * x.x;
            }
        }
    };
    auto post = [&](int p, const Acc &c) {
        v2f Y[16];
        if (MODE >= 1) {
            // slot i (Cr in register i, Ci in i + 8): channel A = (Pr - Qi, Pi + Qr), channel B = (Pr + Qi, Pi - Qr)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                Y[2 * i] = v2f{c.Dr[i] - c.Di[i + 8], c.Di[i] + c.Dr[i + 8]};
                Y[2 * i + 1] = v2f{c.Dr[i] + c.Di[i + 8], c.Di[i] - c.Dr[i + 8]};
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) { Y[2 * i] = v2f{c.Dr[i], c.Dr[i + 8]}; Y[2 * i + 1] = v2f{c.Di[i], c.Di[i + 8]}; }
        }
        if (MODE >= 2) {
            const int u = p % 5, s = p / 5;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int j = u + 5 * d;
                if (j > 18) continue;
                const int a = (s - d) & 3;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (d == 0 && u == 0) acc2[a][i] = Y[i] * v2f{h2[j], h2[j]};
                    else acc2[a][i] = __builtin_elementwise_fma(Y[i], v2f{h2[j], h2[j]}, acc2[a][i]);
                }
            }
            if (u == 3) {
                const int a = (s - 3) & 3;
#pragma unroll
                for (int i = 0; i < 16; i++) ysum[i] += acc2[a][i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) ysum[i] += Y[i];
        }
    };
    load_ops(0, ops[0]);
    mfma_phase(ops[0], acc[1]);
    for (int b = 0; b < bodies; b++) {
        boff = (b & 1) << 3;
        asm volatile("" : "+v"(boff));
#pragma unroll
        for (int p = 0; p < 20; p++) {
            load_ops((p + 1) % 20, ops[(p + 1) & 1]);
            mfma_phase(ops[p & 1], acc[p & 1]);
            post((p + 19) % 20, acc[(p + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (stamps && lane == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        stamps[2 * (blockIdx.x * 4 + (tid >> 6))] = t1 - t0;
        stamps[2 * (blockIdx.x * 4 + (tid >> 6)) + 1] = r1 - r0;
    }
    float sres = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) sres += ysum[i].x + ysum[i].y;
    out[blockIdx.x * 256 + tid] = sres;
}

template <int MODE, int WPS>
void run32(const char *name, int wgs_per_cu, const float2 *src, const float *h2)
{
    const int grid = 256 * wgs_per_cu * 2;
    const size_t ldsb = wgs_per_cu == 1 ? 160 * 1024 - 2048 : 160 * 1024 / wgs_per_cu - 1024;
    hipFuncSetAttribute((const void *)k32<MODE, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    float *d;
    unsigned long long *st;
    hipMalloc(&d, grid * 256 * sizeof(float));
    hipMalloc(&st, grid * 8 * sizeof(unsigned long long));
    const int bodies = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 30; i++) k32<MODE, WPS><<<grid, 256, ldsb>>>(src, d, bodies, 0.01f, h2, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) k32<MODE, WPS><<<grid, 256, ldsb>>>(src, d, bodies, 0.01f, h2, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    k32<MODE, WPS><<<grid, 256, ldsb>>>(src, d, bodies, 0.01f, h2, st);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hs(grid * 8);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < grid * 4; i++) { cyc += hs[2 * i]; rt += hs[2 * i + 1]; }
    const double clk_ghz = cyc / rt * 0.1;
    const double phases = (double)grid * 4 * bodies * 20 * 2;    // in units of the 16 x 16 x 4 kernel's wave-phase (16 slots x 16 segments)
    const double ns_per_phase_simd = ms * 1e6 / (phases / 1024);
    const double chs = phases * 2560 / (ms * 1e-3);
    printf("%-34s wgs/cu=%d waves/simd<=%d  %.3f ms  %.1f ns per 16x16-tile phase/SIMD = %.0f cycles @%.2f GHz (stamped)  -> %.2fe12 ch-samples/s  (10k x 320k block: %.3f ms)\n",
           name, wgs_per_cu, WPS, ms, ns_per_phase_simd, ns_per_phase_simd * clk_ghz, clk_ghz, chs / 1e12, 3.2e9 / chs * 1e3);
    hipFree(d);
    hipFree(st);
}

int main()
{
    std::vector<float2> h(65536);
    srand(1);
    for (auto &v : h) v = make_float2((rand() % 2001 - 1000) * 1e-3f, (rand() % 2001 - 1000) * 1e-3f);
    float2 *src;
    hipMalloc(&src, h.size() * sizeof(float2));
    hipMemcpy(src, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
    float h2h[19];
    for (int j = 0; j < 19; j++) h2h[j] = 0.05f + 0.001f * j;
    float *h2;
    hipMalloc(&h2, sizeof(h2h));
    hipMemcpy(h2, h2h, sizeof(h2h), hipMemcpyHostToDevice);
    if (getenv("UBENCH_ALL")) {
    run<0, 1>("mfma + operand reads", 1, src, h2);
    run<1, 1>("+ coefficients, tap 16, combos", 1, src, h2);
    run<2, 1>("+ stage 2 (whole phase)", 1, src, h2);
    run<3, 1>("VALU part only", 1, src, h2);
    }
    run<0, 2>("mfma + operand reads", 2, src, h2);
    run<1, 2>("+ coefficients, tap 16, combos", 2, src, h2);
    run<2, 2>("+ stage 2 (whole phase)", 2, src, h2);
    run<3, 2>("VALU part only", 2, src, h2);
    printf("-- v_mfma_f32_32x32x2_f32: 16 slots x 32 segments per wave (per 16 x 16-tile phase, to compare with the rows above)\n");
    run32<0, 1>("32x32x2: mfma + operand reads", 1, src, h2);
    run32<1, 1>("32x32x2: + coefficients, tap 16, combos", 1, src, h2);
    run32<2, 1>("32x32x2: + stage 2 (whole phase)", 1, src, h2);
    run32<0, 2>("32x32x2: mfma + operand reads", 2, src, h2);
    run32<1, 2>("32x32x2: + coefficients, tap 16, combos", 2, src, h2);
    run32<2, 2>("32x32x2: + stage 2 (whole phase)", 2, src, h2);
    run<0, 1>("16x16x4: mfma + operand reads", 1, src, h2);
    run<2, 1>("16x16x4: + stage 2 (whole phase)", 1, src, h2);
    run<4, 1>("stages 2+3 as one transposed filter", 1, src, h2);
    run<4, 2>("stages 2+3 as one transposed filter", 2, src, h2);
    return 0;
}
