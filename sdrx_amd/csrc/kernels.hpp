// kernels.hpp -- gfx950 (MI355X, CDNA4) kernels of the sdrx channel bank.
//
// Three kernels per IQ block, launched back to back on the bank's stream:
//
//   k_stage12   NCO mixer + decimator stages 1 and 2 fused (reference MSD::decimate mixer path
//               src/msd.hpp:132-155 + S::addSample/calculateOutput msd.hpp:228-335 for the first
//               two stages).  > 90 % of all work: stage 1 runs at the full input rate.
//   k_tail      remaining one or two stages (msd.hpp:251-335 again), output [ch][out_stride].
//   k_agc       IF AGC -> |.| -> LF AGC (src/agc.hpp:48-69, sdrx.cpp:178-180, agc.hpp:106-131).
//
// Design of k_stage12 (see DESIGN.md for the derivation):
//   * one wavefront = ONE channel; its 64 lanes are 64 time segments spaced exactly N input
//     samples apart (N = translator length).  Lane l therefore meets translator entry
//     t[(phase + i) mod N] at the same loop step i as every other lane: the NCO value is
//     WAVE-UNIFORM and comes through the scalar unit (s_load -> SGPR operand of the VALU
//     multiply), costing no vector memory and no LDS traffic.  The table the caller built with
//     the reference formula is used as is (no on-device sincos: the reference table carries
//     float-rounded phases).
//   * the W wavefronts of a workgroup are W different channels over the SAME 64 time segments,
//     so they share one input tile: 64 rows x SB samples of the IQ stream staged once in LDS
//     (coalesced global reads along rows) and read row-per-lane with a padded row stride
//     (conflict-free ds_read_b128).  Input bytes from L2 per channel-sample = 8/W.
//   * per lane the 17-/19-tap delay lines of stages 1 and 2 live in VGPRs (fully unrolled
//     bodies of SB = M1*M2*Y2B samples, static indexing -> pure register renaming);
//     filter taps are kernel arguments = SGPRs; folded symmetric FIR exactly like the reference.
//   * no per-channel delay-line state is kept between blocks: the FIR cascade is non-recursive,
//     so a lane re-creates its history by running WU warm-up bodies over the samples in front
//     of its segment (the bank keeps the last few input samples of the previous block).
//   * stage-2 outputs go to a per-channel scratch row in time order; k_tail picks them up.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sdrxgpu {

constexpr int kWave = 64;
constexpr int kMaxHalfTaps = 64;   // k_tail: stages with up to 127 taps

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    // (a.x + j a.y)(b.x + j b.y); same two products per part as std::complex<float> operator*
    // (msd.hpp:134); the compiler contracts to mul+fma.
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Folded symmetric FIR over a window w[0..T-1] (w[0] oldest), taps h[0..(T-1)/2]:
//   sum_{i<half} (w[i] + w[T-1-i]) * h[i] + w[half] * h[half]          (msd.hpp:309-331)
// accumulated in the reference's order i = 0 .. half.
template <int T>
__device__ __forceinline__ float2 folded_fir(const float2 (&w)[T], const float *__restrict__ h)
{
    constexpr int half = (T - 1) / 2;
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < half; i++) {
        acc.x = fmaf(w[i].x + w[T - 1 - i].x, h[i], acc.x);
        acc.y = fmaf(w[i].y + w[T - 1 - i].y, h[i], acc.y);
    }
    acc.x = fmaf(w[half].x, h[half], acc.x);
    acc.y = fmaf(w[half].y, h[half], acc.y);
    return acc;
}

// ------------------------------------------------------------------------------------------
// k_stage12
// ------------------------------------------------------------------------------------------
template <int T1, int T2>
struct Stage12Taps {
    float h1[(T1 + 1) / 2];
    float h2[(T2 + 1) / 2];
};

// The three pointers are separate __restrict__ kernel parameters (not struct members) so that
// the compiler may prove the translator table is never written by this kernel and fetch it
// through the scalar cache (s_load):
//   x      input ring: [history | new block | slack]
//   tables [n_ch][tbl_stride] translator tables, wrapped by SB entries
//   y2     [n_ch][y2_stride]; new stage-2 outputs go to y2_hist + j
struct Stage12Args {
    int n_ch;
    int tbl_stride;
    int N;                  // translator length (table wraps every N entries)
    int spacing;            // lane spacing in input samples: a multiple of N and of SB
    int y2_stride;
    int y2_hist;
    int n_y2_new;           // stage-2 outputs this launch must produce (per channel)
    int x_base;             // x index of the first warm-up sample of lane 0 of time-wave 0
    int tphase0;            // translator index of that sample
    int nb;                 // bodies per lane (divides spacing/SB)
    int P;                  // (spacing/SB)/nb sub-phases per segment
    int n_wt;               // time-waves = G*P
    int n_chgroups;
};

// Row stride of the LDS input tile in float2 units: SB samples + pad.  With SB*8 a multiple of
// 16 bytes, pad = 2 keeps rows 16-byte aligned for ds_read_b128; the resulting dword stride
// (2*SB+4) makes the 16-lane groups of a b128 read hit 16 distinct 4-bank slots (checked on the
// host for each instantiation, see lds_row_conflict_free()).
template <int SB>
struct TileGeom {
    static constexpr int row_f2 = SB + 2;
    static constexpr int bytes = kWave * row_f2 * 8;
};

template <int M1, int T1, int M2, int T2, int Y2B, int W>
__global__ __launch_bounds__(kWave * W, (W >= 8 ? 4 : (W >= 4 ? 3 : 1))) void k_stage12(const float2 *__restrict__ x_in, const float2 *__restrict__ tables,
                                                             float2 *__restrict__ y2_out, const Stage12Args a,
                                                             const Stage12Taps<T1, T2> taps)
{
    constexpr int SB = M1 * M2 * Y2B;                      // input samples per body
    constexpr int NY1 = M2 * Y2B;                          // stage-1 outputs per body
    constexpr int HALO = M1 * (T2 - M2) + (T1 - M1);       // input history a lane needs
    constexpr int WU = (HALO + SB - 1) / SB;               // warm-up bodies
    constexpr int ROW = TileGeom<SB>::row_f2;
    constexpr int C1 = T1 - M1;                            // carried mixed samples
    constexpr int C2 = T2 - M2;                            // carried stage-1 outputs
    static_assert(SB % 2 == 0, "body must hold an even number of samples (16-byte LDS reads)");
    static_assert(SB >= C1 && NY1 >= C2, "a body must refresh both delay lines completely");

    __shared__ __attribute__((aligned(16))) float2 tile[kWave * ROW];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // blockIdx -> (channel group, time-wave).  Consecutive blocks walk the time-waves of one
    // channel group first, so the blocks that share a translator table are dispatched together.
    const int wt = blockIdx.x % a.n_wt;
    const int cg = blockIdx.x / a.n_wt;
    const int g = wt / a.P;
    const int p = wt - g * a.P;
    const int bps = a.nb * a.P;                            // bodies per segment = spacing / SB

    int ch = cg * W + wave;
    const bool ch_ok = ch < a.n_ch;
    if (!ch_ok) ch = a.n_ch - 1;                           // keeps every wave in the barriers

    // first real body slot of lane 0 in this time-wave; whole workgroup leaves if it is past the end
    const int slot0 = g * kWave * bps + a.nb * p;
    if (slot0 * Y2B >= a.n_y2_new) return;

    int tph = a.tphase0 + p * a.nb * SB;                   // translator index at the first warm-up sample
    tph %= a.N;
    const float2 *__restrict__ tbl = tables + (size_t)ch * a.tbl_stride;
    const float2 *__restrict__ xsrc = x_in + a.x_base + (size_t)slot0 * SB;   // lane 0, first warm-up sample

    float2 w1c[C1];   // last C1 mixed samples        (stage-1 delay line, newest last)
    float2 w2c[C2];   // last C2 stage-1 outputs      (stage-2 delay line, newest last)
#pragma unroll
    for (int i = 0; i < C1; i++) w1c[i] = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < C2; i++) w2c[i] = make_float2(0.f, 0.f);

    const int n_bodies = WU + a.nb;
    const float4 *__restrict__ xrow = reinterpret_cast<const float4 *>(tile + lane * ROW);

    // y2 destination of this lane's first real output
    const int seg = g * kWave + lane;
    int jrel = (seg * bps + a.nb * p) * Y2B;
    float2 *__restrict__ ydst = y2_out + (size_t)ch * a.y2_stride + a.y2_hist;

    for (int b = 0; b < n_bodies; b++) {
        // ---- stage the 64 x SB input tile: row r = lane r's next SB samples -----------------
        __syncthreads();
        {
            const float2 *__restrict__ src = xsrc + (size_t)b * SB;
#pragma unroll 4
            for (int e = tid; e < kWave * SB; e += kWave * W) {
                const int r = e / SB;
                const int c = e - r * SB;
                tile[r * ROW + c] = src[(size_t)r * a.spacing + c];
            }
        }
        __syncthreads();

        const float2 *__restrict__ t = tbl + tph;          // wave-uniform -> scalar loads
        tph += SB;
        if (tph >= a.N) tph -= a.N;

        float2 y2v[Y2B];
#pragma unroll
        for (int q = 0; q < Y2B; q++) {
            // keep the scheduler from hoisting every LDS/scalar load of the body to its top
            // (that costs > 256 VGPRs and SGPR spills); one stage-2 output is one scheduling region
            __builtin_amdgcn_sched_barrier(0);
            float2 win2[T2];
#pragma unroll
            for (int i = 0; i < C2; i++) win2[i] = w2c[i];
#pragma unroll
            for (int u = 0; u < M2; u++) {
                // M1 new mixed samples -> one stage-1 output
                __builtin_amdgcn_sched_barrier(0);
                float2 win[T1];
#pragma unroll
                for (int i = 0; i < C1; i++) win[i] = w1c[i];
#pragma unroll
                for (int e = 0; e < M1; e++) {
                    const int idx = (q * M2 + u) * M1 + e;
                    const float4 xx = xrow[idx >> 1];
                    const float2 xv = (idx & 1) ? make_float2(xx.z, xx.w) : make_float2(xx.x, xx.y);
                    win[C1 + e] = cmul(xv, t[idx]);
                }
                win2[C2 + u] = folded_fir<T1>(win, taps.h1);
                // pin the stage-1 output here: without it LLVM sinks the arithmetic down to the
                // stage-2 FIR and every LDS read of the body stays live (spills)
                asm volatile("" : "+v"(win2[C2 + u].x), "+v"(win2[C2 + u].y));
#pragma unroll
                for (int i = 0; i < C1; i++) w1c[i] = win[i + M1];
            }
            y2v[q] = folded_fir<T2>(win2, taps.h2);
            asm volatile("" : "+v"(y2v[q].x), "+v"(y2v[q].y));
#pragma unroll
            for (int i = 0; i < C2; i++) w2c[i] = win2[i + M2];
        }

        if (b >= WU) {
            if (ch_ok) {
#pragma unroll
                for (int q = 0; q < Y2B; q++)
                    if (jrel + q < a.n_y2_new) ydst[jrel + q] = y2v[q];
            }
            jrel += Y2B;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_tail: the remaining one or two decimator stages; one workgroup per channel.
// ------------------------------------------------------------------------------------------
struct TailArgs {
    float2 *y2;             // [n_ch][y2_stride]: [0,hist) history, [hist, hist+n_new) new
    float2 *iq_out;         // [n_ch][out_stride] or nullptr
    float2 *iq_keep;        // [n_ch][keep_stride] compact copy for k_agc (never null)
    int n_ch;
    int y2_stride, y2_hist, n_y2_new;
    int out_stride, keep_stride;
    int has_a;              // 1: stage A then stage B; 0: stage B only
    int mA, tA, mB, tB;
    int n_a;                // stage-A outputs to compute
    int pA0;                // y2 index (in the [hist|new] row) of the first A window's oldest sample
    int n_b;                // stage-B outputs = outputs of the call
    int pB0;                // index of the first B window's oldest sample (in A outputs, or in y2 if !has_a)
    float hA[kMaxHalfTaps];
    float hB[kMaxHalfTaps];
};

__device__ __forceinline__ float2 folded_fir_lds(const float2 *__restrict__ w, int T, const float *__restrict__ h)
{
    const int half = (T - 1) >> 1;
    float2 acc = make_float2(0.f, 0.f);
    for (int i = 0; i < half; i++) {
        const float2 lo = w[i], hi = w[T - 1 - i];
        acc.x = fmaf(lo.x + hi.x, h[i], acc.x);
        acc.y = fmaf(lo.y + hi.y, h[i], acc.y);
    }
    const float2 c = w[half];
    acc.x = fmaf(c.x, h[half], acc.x);
    acc.y = fmaf(c.y, h[half], acc.y);
    return acc;
}

__global__ __launch_bounds__(256) void k_tail(const TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    __shared__ float hA[kMaxHalfTaps], hB[kMaxHalfTaps];

    const int ch = blockIdx.x;
    const int tid = threadIdx.x;
    const int n2 = a.y2_hist + a.n_y2_new;
    float2 *__restrict__ row = a.y2 + (size_t)ch * a.y2_stride;
    float2 *s2 = smem;                 // [n2]
    float2 *s3 = smem + n2;            // [n_a]

    if (tid < kMaxHalfTaps) {
        hA[tid] = a.hA[tid];
        hB[tid] = a.hB[tid];
    }
    for (int i = tid; i < n2; i += blockDim.x) s2[i] = row[i];
    __syncthreads();

    // history for the next call: the last y2_hist stage-2 outputs, time order kept
    for (int i = tid; i < a.y2_hist; i += blockDim.x) row[i] = s2[a.n_y2_new + i];

    const float2 *bsrc = s2;
    if (a.has_a) {
        for (int r = tid; r < a.n_a; r += blockDim.x)
            s3[r] = folded_fir_lds(s2 + a.pA0 + a.mA * r, a.tA, hA);
        __syncthreads();
        bsrc = s3;
    }
    for (int s = tid; s < a.n_b; s += blockDim.x) {
        const float2 y = folded_fir_lds(bsrc + a.pB0 + a.mB * s, a.tB, hB);
        if (a.iq_out) a.iq_out[(size_t)ch * a.out_stride + s] = y;
        a.iq_keep[(size_t)ch * a.keep_stride + s] = y;
    }
}

// ------------------------------------------------------------------------------------------
// k_agc: lane = channel, strictly serial over the block's outputs (the only true recurrence
// of the path).  64 channels per single-wave workgroup; the [ch][n] rows are transposed
// through LDS in tiles of 64 outputs so that global traffic stays row-contiguous.
// ------------------------------------------------------------------------------------------
struct AgcParams {
    float attack, decay, reference, min_gain, max_gain, alpha;
};

struct AgcArgs {
    const float2 *iq;        // [n_ch][iq_stride] decimated IQ
    float        *audio;     // [n_ch][out_stride]
    float4       *state;     // [n_ch] {agc.gain, agc.power, lf.gain, lf.power}
    int n_ch, n_out, iq_stride, out_stride;
    int lf_active;
    AgcParams agc, lf;
};

__global__ __launch_bounds__(kWave) void k_agc(const AgcArgs a)
{
    constexpr int TS = 64;                    // outputs per tile
    __shared__ float2 tin[kWave * (TS + 1)];
    __shared__ float tout[kWave * (TS + 1)];

    const int lane = threadIdx.x;
    const int ch0 = blockIdx.x * kWave;
    const int ch = ch0 + lane;
    const bool ok = ch < a.n_ch;

    float4 st = ok ? a.state[ch] : make_float4(1.f, 0.f, 1.f, 0.f);
    float g = st.x, pw = st.y, gl = st.z, pl = st.w;

    for (int base = 0; base < a.n_out; base += TS) {
        const int cnt = min(TS, a.n_out - base);
        // rows in: lane = sample within the tile
        for (int r = 0; r < kWave; r++) {
            const int c = ch0 + r;
            if (c < a.n_ch && lane < cnt) tin[r * (TS + 1) + lane] = a.iq[(size_t)c * a.iq_stride + base + lane];
        }
        __syncthreads();
        if (ok) {
            for (int i = 0; i < cnt; i++) {
                const float2 x = tin[lane * (TS + 1) + i];
                // AGC::adjust, agc.hpp:48-69 (same operation order, unfused)
                const float yr = x.x * g, yi = x.y * g;
                const float nrm = __fadd_rn(__fmul_rn(yr, yr), __fmul_rn(yi, yi));
                pw = __fadd_rn(pw, __fmul_rn(a.agc.alpha, __fsub_rn(nrm, pw)));
                const float err = __fsub_rn(a.agc.reference, pw);
                g = __fadd_rn(g, __fmul_rn(err > 0.0f ? a.agc.decay : a.agc.attack, err));
                g = g < a.agc.min_gain ? a.agc.min_gain : (g > a.agc.max_gain ? a.agc.max_gain : g);
                // Demod::demod AM: std::abs, sdrx.cpp:178-180
                const float s = __fsqrt_rn(__fadd_rn(__fmul_rn(yr, yr), __fmul_rn(yi, yi)));
                // LfAGC::adjust, agc.hpp:106-131
                const float sl = __fmul_rn(s, gl);
                pl = __fadd_rn(pl, __fmul_rn(a.lf.alpha, __fsub_rn(__fmul_rn(sl, sl), pl)));
                const float errl = __fsub_rn(a.lf.reference, __fsqrt_rn(pl));
                gl = __fadd_rn(gl, __fmul_rn(errl > 0.0f ? a.lf.decay : a.lf.attack, errl));
                gl = gl < a.lf.min_gain ? a.lf.min_gain : (gl > a.lf.max_gain ? a.lf.max_gain : gl);
                tout[lane * (TS + 1) + i] = a.lf_active ? sl : s;
            }
        }
        __syncthreads();
        for (int r = 0; r < kWave; r++) {
            const int c = ch0 + r;
            if (c < a.n_ch && lane < cnt) a.audio[(size_t)c * a.out_stride + base + lane] = tout[r * (TS + 1) + lane];
        }
        __syncthreads();
    }
    if (ok) a.state[ch] = make_float4(g, pw, gl, pl);
}

// copy the last `hist` samples of [0, hist + n_new) to the front (ranges may overlap)
__global__ void k_roll_history(float2 *x, int hist, int n_new)
{
    const int i = threadIdx.x;
    float2 v = make_float2(0.f, 0.f);
    if (i < hist) v = x[n_new + i];
    __syncthreads();
    if (i < hist) x[i] = v;
}

}  // namespace sdrxgpu
