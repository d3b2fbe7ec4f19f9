// kernels.hpp -- gfx950 (MI355X, CDNA4) kernels of the sdrx channel bank.
//
// Per IQ block, back to back on the bank's stream:
//
//   k_stage12      NCO mixer + decimator stages 1 and 2 fused (reference MSD::decimate mixer path
//                  src/msd.hpp:132-155 + S::addSample/calculateOutput msd.hpp:228-335 for the first
//                  two stages).  > 90 % of all arithmetic: stage 1 runs at the full input rate.
//   k_tail_s/k_tail remaining one or two stages (msd.hpp:251-335 again), output [ch][out_stride];
//                  k_tail_s has the stages' (m, taps) as template parameters, k_tail takes them at run time.
//   k_agc<SQ, FM>  IF AGC -> |.| (or the FM phase step) -> LF AGC (src/agc.hpp:48-69, sdrx.cpp:178-194,
//                  agc.hpp:106-131), gated by the squelch state (sdrx.cpp:458-503) when SQ; one extra
//                  block rolls the input history to the front of the ring.
// Optional, after those:
//   k_squelch      window x 512-point FFT -> SNR -> open/close (sdrx.cpp:533-573)
//   k_audio_fir    FIR / FIR2 (src/fir.hpp:79-183) on audio rows
//   k_mix_partial / k_mix_reduce / k_mix_out   pan-mix of all channels, S16 (sdrx.cpp:469-497, 649-656)
//
// Design of k_stage12 (see DESIGN.md 4.1 for the derivation and the measurements behind each choice):
//   * the 64 lanes of a wavefront = CPW channels x 64/CPW time segments spaced a multiple of N input
//     samples apart (N = translator length).  All lanes of one channel therefore meet the same
//     translator entry at the same loop step: one LDS row per channel, read as a broadcast.  The
//     table the caller built with the reference formula is used as is (no on-device sincos: the
//     reference table carries float-rounded phases).
//   * the W wavefronts of a workgroup are W*CPW channels over the SAME time segments, so they share
//     one input tile.  Tiles (input rows + translator rows, SB samples each) go global -> LDS by
//     LDS-DMA (global_load_lds_dwordx4), double-buffered, one barrier per body; rows are padded to a
//     pitch that keeps ds_read_b128 conflict-free.
//   * per lane the delay lines of stages 1 and 2 live in VGPRs (fully unrolled bodies of
//     SB = M1*M2*Y2B samples, static indexing -> pure register renaming); arithmetic is packed fp32
//     (one complex sample = one 64-bit register pair); filter taps are kernel arguments = SGPRs;
//     folded symmetric FIR exactly like the reference.
//   * no per-channel delay-line state is kept between blocks: the FIR cascade is non-recursive,
//     so a lane re-creates its history by running WU warm-up bodies over the samples in front
//     of its segment (the bank keeps the last few input samples of the previous block).
//   * stage-2 outputs go to a per-channel scratch row in BODY-MAJOR order (full-line writes); the
//     tail kernels undo the permutation while they stage their tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
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

// ---- packed-fp32 forms (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) ----------------------------
// Measured on MI355X (tools/ubench_valu.hip): a wave64 v_fma_f32 and a wave64 v_pk_fma_f32 both
// issue in ~4 cycles per SIMD, so the packed forms do twice the work per issue slot.  A complex
// sample {re, im} is one 64-bit register pair and is processed by ONE packed instruction.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f cmul_pk(v2f x, v2f t)
{
    // {xr*tr - xi*ti, xi*tr + xr*ti} in two packed instructions:
    //   p = {xr, xi} * {tr, tr}                    (pk_mul, t.lo broadcast)
    //   r = {xi, xr} * {-ti, ti} + p               (pk_fma, x halves swapped, t.hi broadcast, neg_lo)
    // Every shuffle/negation is an op_sel / neg source modifier.  Written as inline asm because
    // hipcc (ROCm 7.2) materialises {-ti, ti} with a v_xor + v_mov per sample instead of neg_lo when
    // t lives in VGPRs (+30 % VALU instructions in this kernel).  The s_nop 0 after each packed op is
    // the wait state gfx950 needs before a VALU instruction may read a packed op's result (hipcc
    // inserts the same nop in its own code; it does not look inside asm).
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "s_nop 0\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "s_nop 0"
        : "=&v"(r)
        : "v"(x), "v"(t));
    return r;
}

// Folded symmetric FIR, packed.  Two interleaved accumulators (even / odd tap pairs) so that no
// packed instruction reads the result of the instruction right before it (that adjacency costs a
// wait state on gfx950); the two partial sums are added at the end.  Same products as the
// reference's folded form (msd.hpp:309-331), summed in a different order (fp32 rounding only).
// Five complex multiplies as one instruction group: mul0..mul4 then fma0..fma4, so no packed
// instruction reads the result of the one right before it and only the group's end needs the wait
// state (10 packed instructions + 1 s_nop instead of 10 + 10).
__device__ __forceinline__ void cmul5_pk(const v2f (&x)[5], const v2f (&t)[5], v2f *out)
{
    v2f r0, r1, r2, r3, r4;
    asm("v_pk_mul_f32 %0, %5, %10 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %1, %6, %11 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %2, %7, %12 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %3, %8, %13 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %4, %9, %14 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %5, %10, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %6, %11, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %2, %7, %12, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %3, %8, %13, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %4, %9, %14, %4 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "s_nop 0"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]));
    out[0] = r0;
    out[1] = r1;
    out[2] = r2;
    out[3] = r3;
    out[4] = r4;
}

template <int T>
__device__ __forceinline__ v2f folded_fir_pk(const v2f (&w)[T], const float *__restrict__ h)
{
    constexpr int half = (T - 1) / 2;
    v2f s[half];
#pragma unroll
    for (int i = 0; i < half; i++) s[i] = w[i] + w[T - 1 - i];
    // one accumulator, the reference's order (msd.hpp:309-332: pairs first, centre tap last).  The pair sums are
    // independent of the chain, so the scheduler puts one between any two dependent FMAs (a packed op needs one
    // wait state after the instruction that wrote its source); two interleaved accumulators cost an extra add per
    // output (3.7 % of the kernel's instructions) for nothing.
    v2f acc = s[0] * v2f{h[0], h[0]};
#pragma unroll
    for (int i = 1; i < half; i++) acc = __builtin_elementwise_fma(s[i], v2f{h[i], h[i]}, acc);
    return __builtin_elementwise_fma(w[half], v2f{h[half], h[half]}, acc);
}

// ---- folded FIR over a window in LDS (or in registers), fully unrolled, taps as SGPR operands -----------------
// acc += x * h[K] on both halves, the tap read from one half of an SGPR pair (op_sel picks the half): two taps per
// pair instead of the (h, h) pair per tap the compiler would build -- 45 taps of the 10 MS/s tail would otherwise
// need 90 SGPRs and spill into v_readlane/v_writelane traffic.  The s_nop is the wait state a packed op needs
// when its source was written by the instruction right before it (hipcc cannot see into the asm).
template <int K>
__device__ __forceinline__ void fma_tap(v2f &acc, const v2f x, const v2f hpair)
{
    if constexpr ((K & 1) == 0)
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "s"(hpair));
    else
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "s"(hpair));
}

template <int T, int NH, int I = 0>
__device__ __forceinline__ void folded_fir_steps(v2f &acc, const v2f *__restrict__ w, const float (&h)[NH])
{
    static_assert(NH % 2 == 0 && NH >= (T + 1) / 2, "tap array: an even number of entries covering h[0 .. (T-1)/2]");
    constexpr int half = (T - 1) >> 1;
    if constexpr (I <= half) {
        const v2f hp = {h[I & ~1], h[(I & ~1) + 1]};
        if constexpr (I < half)
            fma_tap<I>(acc, w[I] + w[T - 1 - I], hp);
        else
            fma_tap<I>(acc, w[half], hp);
        folded_fir_steps<T, NH, I + 1>(acc, w, h);
    }
}

template <int T, int NH>
__device__ __forceinline__ v2f folded_fir_unrolled(const v2f *__restrict__ w, const float (&h)[NH])
{
    v2f acc = {0.f, 0.f};
    folded_fir_steps<T, NH>(acc, w, h);
    return acc;
}

// The same with the window's LDS reads issued D tap pairs AHEAD of their use: left to the compiler, the reads of a folded FIR over
// a window in LDS come one pair at a time in front of the add that needs them (it schedules for few registers), and a 67-tap
// output is 33 LDS round trips in a row -- 2 000-2 400 cycles per tile of k_tail_s (in-kernel stamps), with one wave of the
// workgroup active.  Volatile reads in their own address space keep their place (and stay single ds_read_b64).  Same operands,
// same order, same bits as folded_fir_unrolled.
#ifndef SDRX_FIR_AHEAD_A
#define SDRX_FIR_AHEAD_A 5
#endif
#ifndef SDRX_FIR_AHEAD_B
#define SDRX_FIR_AHEAD_B 8
#endif
typedef const volatile __attribute__((address_space(3))) v2f lds_cv2f_t;
__device__ __forceinline__ v2f lds_read_here(const v2f *p) { return *(lds_cv2f_t *)(uint32_t)(uintptr_t)p; }

template <int T, int NH, int D, int I = 0>
__device__ __forceinline__ void folded_fir_ahead_steps(v2f &acc, v2f (&lo)[D], v2f (&hi)[D], const v2f *__restrict__ w, const float (&h)[NH])
{
    constexpr int half = (T - 1) >> 1;
    if constexpr (I <= half) {
        const v2f hp = {h[I & ~1], h[(I & ~1) + 1]};
        if constexpr (I < half) {
            const v2f s = lo[I % D] + hi[I % D];
            if constexpr (I + D < half) { lo[I % D] = lds_read_here(w + I + D); hi[I % D] = lds_read_here(w + T - 1 - I - D); }
            else if constexpr (I + D == half) { lo[I % D] = lds_read_here(w + half); }
            fma_tap<I>(acc, s, hp);
        } else {
            fma_tap<I>(acc, lo[I % D], hp);
        }
        folded_fir_ahead_steps<T, NH, D, I + 1>(acc, lo, hi, w, h);
    }
}

template <int T, int NH, int D>
__device__ __forceinline__ v2f folded_fir_ahead(const v2f *__restrict__ w, const float (&h)[NH])
{
    constexpr int half = (T - 1) >> 1;
    static_assert(D >= 1 && D <= half, "prefetch distance in tap pairs");
    v2f lo[D], hi[D];
#pragma unroll
    for (int i = 0; i < D; i++) { lo[i] = lds_read_here(w + i); hi[i] = lds_read_here(w + T - 1 - i); }
    v2f acc = {0.f, 0.f};
    folded_fir_ahead_steps<T, NH, D>(acc, lo, hi, w, h);
    return acc;
}

// folded_fir_ahead for TWO windows at once (k_tail_agc's producer waves, alone on their SIMD: two independent accumulation chains
// instead of one halve the time a wave spends waiting for its own previous instruction).  Per output the same operands in the same
// order as folded_fir_ahead / folded_fir_unrolled: same bits.  (Defined below fma_tap2.)
template <int T, int NH, int D>
__device__ __forceinline__ void folded_fir2_ahead(v2f &r0, v2f &r1, const v2f *__restrict__ w0, const v2f *__restrict__ w1, const float (&h)[NH]);

// Two outputs at once (windows w0 and w1, same taps): the two accumulation chains alternate, so neither packed FMA
// reads the result of the instruction before it and one wait state serves both (fma_tap spends one per tap).  Per
// output the same instruction sequence on the same operands as folded_fir_unrolled: bit-identical results.
template <int K>
__device__ __forceinline__ void fma_tap2(v2f &acc0, v2f &acc1, const v2f x0, const v2f x1, const v2f hpair)
{
    if constexpr ((K & 1) == 0)
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]"
            : "+v"(acc0), "+v"(acc1) : "v"(x0), "v"(x1), "s"(hpair));
    else
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
            : "+v"(acc0), "+v"(acc1) : "v"(x0), "v"(x1), "s"(hpair));
}

template <int T, int NH, int D, int I = 0>
__device__ __forceinline__ void folded_fir2_ahead_steps(v2f &acc0, v2f &acc1, v2f (&lo0)[D], v2f (&hi0)[D], v2f (&lo1)[D], v2f (&hi1)[D],
                                                        const v2f *__restrict__ w0, const v2f *__restrict__ w1, const float (&h)[NH])
{
    constexpr int half = (T - 1) >> 1;
    if constexpr (I <= half) {
        const v2f hp = {h[I & ~1], h[(I & ~1) + 1]};
        if constexpr (I < half) {
            const v2f s0 = lo0[I % D] + hi0[I % D], s1 = lo1[I % D] + hi1[I % D];
            if constexpr (I + D < half) {
                lo0[I % D] = lds_read_here(w0 + I + D); hi0[I % D] = lds_read_here(w0 + T - 1 - I - D);
                lo1[I % D] = lds_read_here(w1 + I + D); hi1[I % D] = lds_read_here(w1 + T - 1 - I - D);
            } else if constexpr (I + D == half) {
                lo0[I % D] = lds_read_here(w0 + half);
                lo1[I % D] = lds_read_here(w1 + half);
            }
            fma_tap2<I>(acc0, acc1, s0, s1, hp);
        } else {
            fma_tap2<I>(acc0, acc1, lo0[I % D], lo1[I % D], hp);
        }
        folded_fir2_ahead_steps<T, NH, D, I + 1>(acc0, acc1, lo0, hi0, lo1, hi1, w0, w1, h);
    }
}

template <int T, int NH, int D>
__device__ __forceinline__ void folded_fir2_ahead(v2f &r0, v2f &r1, const v2f *__restrict__ w0, const v2f *__restrict__ w1, const float (&h)[NH])
{
    constexpr int half = (T - 1) >> 1;
    static_assert(D >= 1 && D <= half, "prefetch distance in tap pairs");
    v2f lo0[D], hi0[D], lo1[D], hi1[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        lo0[i] = lds_read_here(w0 + i); hi0[i] = lds_read_here(w0 + T - 1 - i);
        lo1[i] = lds_read_here(w1 + i); hi1[i] = lds_read_here(w1 + T - 1 - i);
    }
    r0 = v2f{0.f, 0.f};
    r1 = v2f{0.f, 0.f};
    folded_fir2_ahead_steps<T, NH, D>(r0, r1, lo0, hi0, lo1, hi1, w0, w1, h);
}

template <int T, int NH, int I = 0>
__device__ __forceinline__ void folded_fir2_steps(v2f &acc0, v2f &acc1, const v2f *__restrict__ w0, const v2f *__restrict__ w1,
                                                  const float (&h)[NH])
{
    static_assert(NH % 2 == 0 && NH >= (T + 1) / 2, "tap array: an even number of entries covering h[0 .. (T-1)/2]");
    constexpr int half = (T - 1) >> 1;
    if constexpr (I <= half) {
        const v2f hp = {h[I & ~1], h[(I & ~1) + 1]};
        if constexpr (I < half)
            fma_tap2<I>(acc0, acc1, w0[I] + w0[T - 1 - I], w1[I] + w1[T - 1 - I], hp);
        else
            fma_tap2<I>(acc0, acc1, w0[half], w1[half], hp);
        folded_fir2_steps<T, NH, I + 1>(acc0, acc1, w0, w1, h);
    }
}

// Three outputs at once: with three alternating chains no packed FMA reads the result of the one before it, so no wait
// state at all.  Per output the same instruction sequence on the same operands as folded_fir_unrolled.
template <int K>
__device__ __forceinline__ void fma_tap3(v2f &a0, v2f &a1, v2f &a2, const v2f x0, const v2f x1, const v2f x2, const v2f hpair)
{
    if constexpr ((K & 1) == 0)
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %3, %6, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %4, %6, %1 op_sel_hi:[1,0,1]\n\t"
            "v_pk_fma_f32 %2, %5, %6, %2 op_sel_hi:[1,0,1]"
            : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x0), "v"(x1), "v"(x2), "s"(hpair));
    else
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %3, %6, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %1, %4, %6, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
            "v_pk_fma_f32 %2, %5, %6, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]"
            : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(x0), "v"(x1), "v"(x2), "s"(hpair));
}

template <int T, int NH, int I = 0>
__device__ __forceinline__ void folded_fir3_steps(v2f &a0, v2f &a1, v2f &a2, const v2f *__restrict__ w0, const v2f *__restrict__ w1,
                                                  const v2f *__restrict__ w2, const float (&h)[NH])
{
    static_assert(NH % 2 == 0 && NH >= (T + 1) / 2, "tap array: an even number of entries covering h[0 .. (T-1)/2]");
    constexpr int half = (T - 1) >> 1;
    if constexpr (I <= half) {
        const v2f hp = {h[I & ~1], h[(I & ~1) + 1]};
        if constexpr (I < half)
            fma_tap3<I>(a0, a1, a2, w0[I] + w0[T - 1 - I], w1[I] + w1[T - 1 - I], w2[I] + w2[T - 1 - I], hp);
        else
            fma_tap3<I>(a0, a1, a2, w0[half], w1[half], w2[half], hp);
        folded_fir3_steps<T, NH, I + 1>(a0, a1, a2, w0, w1, w2, h);
    }
}

// folded_fir_ahead for THREE windows at once (k_tail_agc's A-waves: with three alternating chains no packed FMA reads the result of
// the one before it).  Per output the same operands in the same order as folded_fir_ahead / folded_fir_unrolled: same bits.
template <int T, int NH, int D, int I = 0>
__device__ __forceinline__ void folded_fir3_ahead_steps(v2f &a0, v2f &a1, v2f &a2, v2f (&lo0)[D], v2f (&hi0)[D], v2f (&lo1)[D], v2f (&hi1)[D],
                                                        v2f (&lo2)[D], v2f (&hi2)[D], const v2f *__restrict__ w0, const v2f *__restrict__ w1,
                                                        const v2f *__restrict__ w2, const float (&h)[NH])
{
    constexpr int half = (T - 1) >> 1;
    if constexpr (I <= half) {
        const v2f hp = {h[I & ~1], h[(I & ~1) + 1]};
        if constexpr (I < half) {
            const v2f s0 = lo0[I % D] + hi0[I % D], s1 = lo1[I % D] + hi1[I % D], s2 = lo2[I % D] + hi2[I % D];
            if constexpr (I + D < half) {
                lo0[I % D] = lds_read_here(w0 + I + D); hi0[I % D] = lds_read_here(w0 + T - 1 - I - D);
                lo1[I % D] = lds_read_here(w1 + I + D); hi1[I % D] = lds_read_here(w1 + T - 1 - I - D);
                lo2[I % D] = lds_read_here(w2 + I + D); hi2[I % D] = lds_read_here(w2 + T - 1 - I - D);
            } else if constexpr (I + D == half) {
                lo0[I % D] = lds_read_here(w0 + half);
                lo1[I % D] = lds_read_here(w1 + half);
                lo2[I % D] = lds_read_here(w2 + half);
            }
            fma_tap3<I>(a0, a1, a2, s0, s1, s2, hp);
        } else {
            fma_tap3<I>(a0, a1, a2, lo0[I % D], lo1[I % D], lo2[I % D], hp);
        }
        folded_fir3_ahead_steps<T, NH, D, I + 1>(a0, a1, a2, lo0, hi0, lo1, hi1, lo2, hi2, w0, w1, w2, h);
    }
}

template <int T, int NH, int D>
__device__ __forceinline__ void folded_fir3_ahead(v2f &r0, v2f &r1, v2f &r2, const v2f *__restrict__ w0, const v2f *__restrict__ w1,
                                                  const v2f *__restrict__ w2, const float (&h)[NH])
{
    constexpr int half = (T - 1) >> 1;
    static_assert(D >= 1 && D <= half, "prefetch distance in tap pairs");
    v2f lo0[D], hi0[D], lo1[D], hi1[D], lo2[D], hi2[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        lo0[i] = lds_read_here(w0 + i); hi0[i] = lds_read_here(w0 + T - 1 - i);
        lo1[i] = lds_read_here(w1 + i); hi1[i] = lds_read_here(w1 + T - 1 - i);
        lo2[i] = lds_read_here(w2 + i); hi2[i] = lds_read_here(w2 + T - 1 - i);
    }
    r0 = v2f{0.f, 0.f};
    r1 = v2f{0.f, 0.f};
    r2 = v2f{0.f, 0.f};
    folded_fir3_ahead_steps<T, NH, D>(r0, r1, r2, lo0, hi0, lo1, hi1, lo2, hi2, w0, w1, w2, h);
}

// Two outputs whose windows w0 and w1 both lie in LDS (any distance apart): nothing is shared between them but the
// taps and the instruction slots -- the point is the alternating accumulation chains of fma_tap2.
template <int T, int NH>
__device__ __forceinline__ void folded_fir2_lds(v2f &r0, v2f &r1, const v2f *__restrict__ w0, const v2f *__restrict__ w1, const float (&h)[NH])
{
    r0 = v2f{0.f, 0.f};
    r1 = v2f{0.f, 0.f};
    folded_fir2_steps<T, NH>(r0, r1, w0, w1, h);
}

// ------------------------------------------------------------------------------------------
// k_stage12
// ------------------------------------------------------------------------------------------
template <int T1, int T2, int TA = 1>
struct Stage12Taps {
    float h1[(T1 + 1) / 2];
    float h2[(T2 + 1) / 2];
    float h3[(((TA + 1) / 2) + 1) & ~1];   // fused stage 3 only (an even number of entries: read as SGPR pairs)
};

// Pointer parameters of k_stage12 (separate __restrict__ kernel parameters):
//   x      input ring: [history | new block | slack]; the first sample of every tile row is at an
//          even index (16-byte aligned source for the 16-byte LDS-DMA)
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
    int x_base;             // x index of the first warm-up sample of lane 0 of time-wave 0 (even)
    int tphase0;            // translator-table index of that sample (even; see the host's odd-phase copy)
    int nb;                 // bodies per lane (divides spacing/SB)
    int P;                  // (spacing/SB)/nb sub-phases per segment
    int n_wt;               // time-waves = G*P
    int cg0;                // first channel group of this launch (a launch may cover part of the bank)
    int n_chgroups;         // one past its last channel group
    int n_cgx;              // channel groups of the launch rounded up to a multiple of 8 (XCD-aware block map)
    // fused stage 3 (k_stage12<.., MA, TA> with MA > 0; P == 1): stage-3 output rr of this call (0-based) has the call's
    // stage-2 outputs MA*rr + c0 - (TA-1) .. MA*rr + c0 as its window
    int y3_stride;
    int c0;
    int n_y3;
    int seam_stride;        // float2 per channel of the seam buffer: [range][2][TA-1] (first / last TA-1 outputs of each range)
    // k_stage12m: slot-group sets below this one store their stage-2 outputs with the streaming hint (`nt`); 0: none
    int nt_sets = 0;
};

// LDS tile geometry.  A wave's 64 lanes are CPW channels x SPW = 64/CPW time segments (lane =
// csub * SPW + segment); the W waves of a workgroup cover the same SPW segments for W*CPW channels.
// One buffer = SPW input rows (one per time segment) + W*CPW translator rows (one per channel),
// every row SB samples (8 B each) + 16 B pad.  The pad keeps rows 16-byte
// aligned and makes the row pitch (2*SB+4 dwords) fall on distinct 4-bank slots within each 16-lane
// group of a ds_read_b128 (conflict-free row-per-lane reads; translator rows are read as broadcasts).
template <int SB, int W, int CPW>
struct TileGeom {
    static constexpr int row_bytes = SB * 8 + 16;
    static constexpr int x_rows = kWave / CPW;              // time segments per wave
    static constexpr int t_rows = W * CPW;                  // channels per workgroup
    static constexpr int rows = x_rows + t_rows;
    static constexpr int buf_bytes = rows * row_bytes;
    static constexpr int lanes_per_row = SB / 2;            // 16-byte DMA pieces per row
};

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): waits for n or fewer.
__device__ __forceinline__ void wait_vmcnt_le(int n)
{
#define SDRX_VMCNT_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        SDRX_VMCNT_CASE(1) SDRX_VMCNT_CASE(2) SDRX_VMCNT_CASE(3) SDRX_VMCNT_CASE(4) SDRX_VMCNT_CASE(5) SDRX_VMCNT_CASE(6)
        SDRX_VMCNT_CASE(7) SDRX_VMCNT_CASE(8) SDRX_VMCNT_CASE(9) SDRX_VMCNT_CASE(10) SDRX_VMCNT_CASE(11) SDRX_VMCNT_CASE(12)
        SDRX_VMCNT_CASE(13) SDRX_VMCNT_CASE(14) SDRX_VMCNT_CASE(15) SDRX_VMCNT_CASE(16) SDRX_VMCNT_CASE(17) SDRX_VMCNT_CASE(18)
        SDRX_VMCNT_CASE(19) SDRX_VMCNT_CASE(20) SDRX_VMCNT_CASE(21) SDRX_VMCNT_CASE(22) SDRX_VMCNT_CASE(23) SDRX_VMCNT_CASE(24)
    default:
        if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (n > 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef SDRX_VMCNT_CASE
}

// One row of SB samples, global -> LDS, without touching VGPRs (global_load_lds_dwordx4):
// lane i (< SB/2) moves 16 bytes from src + 2*i samples to dst + 16*i.
template <int SB>
__device__ __forceinline__ void dma_row(const float2 *src_row /* wave-uniform */, char *lds_row /* wave-uniform */, int lane)
{
    // a row of SB samples = SB/2 16-byte pieces, 64 per instruction
    constexpr int PIECES = (SB / 2 + kWave - 1) / kWave;
#pragma unroll
    for (int pc = 0; pc < PIECES; pc++) {
        if (pc * kWave + lane < SB / 2) {
            // uniform 64-bit base + 32-bit per-lane byte offset -> the saddr + voffset form of the
            // instruction (no 64-bit per-lane address registers).  generic -> global / LDS address spaces:
            // a global generic pointer is its own global address; the low 32 bits of a generic LDS pointer
            // are the LDS offset.
            const char *g = reinterpret_cast<const char *>(src_row) + (uint32_t)((pc * kWave + lane) * 16);
            __builtin_amdgcn_global_load_lds((gbl_void *)(uintptr_t)g, (lds_void *)(uint32_t)(uintptr_t)(lds_row + pc * kWave * 16), 16, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Stage 3 of a workgroup's OWN ranges, behind the stage-1/2 main loop (k_stage12<.., MA, TA> / k_stage12p<.., MA, TA>).
// With one body range per lane (P == 1) the workgroup has just written, for each of its channels, one contiguous range
// of RLEN = SPW * nb * Y2B stage-2 outputs.  Every stage-3 output whose window lies inside that range ("interior":
// all but the ~(TA-1)/MA outputs at each range boundary) is computed here and goes to y3_out[ch][rr]; the first and the
// last TA-1 outputs of the range go compactly to seam_out for k_tail4, which computes the boundary outputs with the
// same instruction sequence and runs the last stage.
//
// Every wave (of the first `wa`: as many as the LDS has rows for, stage3_split) works through its own n_my channels with
// LDS rows of its own (the tiles' LDS, free now): a row = one
// channel's range, fetched back by LDS-DMA with permuted source addresses so that it lands in TIME order; the
// read-back of the next rows is in flight while one row's outputs are computed, and there is no barrier after the one
// at the top (the range was written by this workgroup's waves: s_waitcnt vmcnt(0) + barrier is all the ordering
// needed).  DMA loads return in order, so "at most k * ndma loads outstanding" means every older row has landed,
// whatever the stores in between are doing.  A lane computes three neighbouring outputs from the TA + 2 MA samples
// their windows span (fma_tap3: three alternating accumulation chains; per output the same instruction sequence on
// the same operands as anywhere else).
//   mych: lane i (< n_my) holds the wave's i-th channel (place wave + i * wa of the workgroup's), or -1 (no channel: a
//         row is moved all the same, so that the counts stay uniform, from chn0 -- any valid channel of the workgroup --
//         and nothing is computed)
//   NMY:  n_my when all W waves take part (the compile-time variant of the loop)
// ------------------------------------------------------------------------------------------
// how the workgroup's n_places channels are dealt to its waves: the first wa waves, n_my channels each
template <int SPW, int Y2B, int W>
__device__ __forceinline__ void stage3_split(const Stage12Args &a, const int n_places, const int lds_bytes, int &wa, int &n_my)
{
    const int ndma = (SPW * a.nb * Y2B / 2 + kWave - 1) / kWave;
    wa = max(1, min(W, lds_bytes / (ndma * kWave * 16)));    // (host: the LDS holds one row at least)
    n_my = (n_places + wa - 1) / wa;
}

template <int MA, int TA, int SPW, int Y2B, int W, int NMY, int NH3>
__device__ __forceinline__ void stage3_epilogue(const Stage12Args &a, const float (&h3)[NH3], const float2 *__restrict__ y2_out,
                                                float2 *__restrict__ y3_out, float2 *__restrict__ seam_out, char *lds, const int lds_bytes,
                                                const int g, const int wave, const int lane, const int wa, const int n_my,
                                                const int mych, const int chn0)
{
    static_assert(Y2B == 4, "fused stage 3: groups of 4 stage-2 outputs (two 16-byte pieces)");
    const int seg_y2 = a.nb * Y2B;                           // stage-2 outputs per segment (P == 1: nb = bodies per segment)
    const int rlen = SPW * seg_y2;                           // ... per workgroup range
    const int base = g * rlen;                               // first stage-2 output (index within the call) of this range
    const int end = min(base + rlen, a.n_y2_new);
    // interior outputs rr_lo .. rr_hi: window [MA*rr + c0 - (TA-1), MA*rr + c0] inside [base, end)
    const int rr_lo = (base + (TA - 1) - a.c0 + MA - 1) / MA;
    const int rr_hi = (end - 1 - a.c0 >= 0) ? min((end - 1 - a.c0) / MA, a.n_y3 - 1) : -1;
    const int n_int = rr_hi - rr_lo + 1;
    const int npc = rlen / 2;                                // 16-byte pieces of one channel's range
    // every store of this workgroup's range has left its wave, and every wave is done with the tiles
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float rcp_seg = __builtin_amdgcn_rcpf((float)seg_y2);
    auto source_offset = [&](int d0) -> uint32_t {           // LDS piece d0 + lane (samples 2d, 2d+1 in time order) <- where the main loop stored it
        const int tt = 2 * min(d0 + lane, npc - 1);
        const int seg = (int)(((float)tt + 0.5f) * rcp_seg);
        const int rem = tt - seg * seg_y2;
        return (uint32_t)((((rem / Y2B) * SPW + seg) * Y2B + (rem % Y2B)) * 8);
    };
    constexpr int NS = 8;
    uint32_t soff[NS];                                       // (what does not depend on the channel is computed once)
#pragma unroll
    for (int k = 0; k < NS; k++) soff[k] = source_offset(k * kWave);
    const int n_third = (n_int + 2) / 3;
    const float2 *__restrict__ ysrc = y2_out + a.y2_hist + base;
    const int ndma = (npc + kWave - 1) / kWave;              // DMA instructions per row
    const int rslot = ndma * kWave * 16;                     // bytes per row, whole instructions (the last one's spare lanes land in the pad)
    const int rows_w = min(lds_bytes / rslot / wa, n_my);    // (>= 1: stage3_split)
    char *const myrows = lds + (size_t)wave * rows_w * rslot;
    if (chn0 < 0 || rows_w < 1 || wave >= wa) return;
    // (ndma_tag, rows_tag) = compile-time (ndma, rows_w), or (0, 0): the run-time values.  With constants the loop over
    // the wave's channels unrolls and every wait is an immediate.
    auto rows_pipeline = [&](auto ndma_tag, auto rows_tag) {
        constexpr bool CT = decltype(ndma_tag)::value > 0;
        const int nd = CT ? decltype(ndma_tag)::value : ndma;
        const int rw = CT ? decltype(rows_tag)::value : rows_w;
        auto issue_row = [&](int i) {
            int chn = __builtin_amdgcn_readlane(mych, i);
            if (chn < 0) chn = chn0;
            const char *grow = reinterpret_cast<const char *>(ysrc + (size_t)chn * a.y2_stride);
            char *lrow = myrows + (size_t)(i % rw) * rslot;
#pragma unroll
            for (int k = 0; k < NS; k++) {
                if (k < nd)
                    __builtin_amdgcn_global_load_lds((gbl_void *)(uintptr_t)(grow + soff[k]), (lds_void *)(uint32_t)(uintptr_t)(lrow + k * kWave * 16), 16, 0, 0);
            }
            if constexpr (!CT || decltype(ndma_tag)::value > NS) {
                for (int k = NS; k < nd; k++)
                    __builtin_amdgcn_global_load_lds((gbl_void *)(uintptr_t)(grow + source_offset(k * kWave)), (lds_void *)(uint32_t)(uintptr_t)(lrow + (size_t)k * kWave * 16), 16, 0, 0);
            }
        };
        auto row_outputs = [&](int i) {
            const int chn = __builtin_amdgcn_readlane(mych, i);
            if (chn < 0) return;
            const v2f *__restrict__ lrow = reinterpret_cast<const v2f *>(myrows + (size_t)(i % rw) * rslot);
            constexpr int SE = 2 * (TA - 1);
            static_assert(SE <= kWave, "seam entries of one range: one per lane");
            if (lane < SE) seam_out[(size_t)chn * a.seam_stride + g * SE + lane] = *reinterpret_cast<const float2 *>(&lrow[lane < TA - 1 ? lane : rlen - SE + lane]);
            float2 *__restrict__ y3row = y3_out + (size_t)chn * a.y3_stride + rr_lo;
            const v2f *__restrict__ wrow = lrow + (MA * rr_lo + a.c0 - (TA - 1) - base);
            for (int kp = lane; kp < n_third; kp += kWave) {
                const int r = 3 * kp;
                v2f win[TA + 2 * MA];
#pragma unroll
                for (int j = 0; j < TA + 2 * MA; j++) win[j] = wrow[MA * r + j];
                v2f r0 = {0.f, 0.f}, r1 = {0.f, 0.f}, r2 = {0.f, 0.f};
                folded_fir3_steps<TA>(r0, r1, r2, win, win + MA, win + 2 * MA, h3);
                y3row[r] = make_float2(r0.x, r0.y);
                if (r + 1 < n_int) y3row[r + 1] = make_float2(r1.x, r1.y);
                if (r + 2 < n_int) y3row[r + 2] = make_float2(r2.x, r2.y);
            }
        };
        const int nm = CT ? NMY : n_my;
        auto step = [&](int i) {
            wait_vmcnt_le(min(rw - 1, nm - 1 - i) * nd);
            row_outputs(i);
            if (i + rw < nm) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the row's reads are done before it is filled again
                __builtin_amdgcn_sched_barrier(0);
                issue_row(i + rw);
            }
        };
        if constexpr (CT) {
#pragma unroll
            for (int i = 0; i < decltype(rows_tag)::value; i++) issue_row(i);
#pragma unroll
            for (int i = 0; i < NMY; i++) step(i);
        } else {
            for (int i = 0; i < rw; i++) issue_row(i);
            for (int i = 0; i < nm; i++) step(i);
        }
    };
    if constexpr (NMY >= 8) {
        if (ndma == 6 && rows_w == 3 && wa == W && n_my == NMY) {   // k_stage12p at N = 1200: ranges of 768 outputs, 6 KB per row, 3 rows per wave
            rows_pipeline(std::integral_constant<int, 6>{}, std::integral_constant<int, 3>{});
            return;
        }
    }
    rows_pipeline(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
}

// First stages with a large decimation factor (6 MS/s: m=15/53 taps, 2.56 MS/s: m=20/51 taps) keep more
// than 128 registers live; they run at a lower occupancy and without the double-buffered LDS reads.
//
// MA > 0: stage 3 (decimation MA, TA taps) of the workgroup's OWN range runs as an epilogue (stage3_epilogue above).
template <int M1, int T1, int M2, int T2, int Y2B, int W, int CPW, int MA = 0, int TA = 1>
__global__ __launch_bounds__(kWave * W, (M1 > 6 ? 1 : 4))
void k_stage12(const float2 *__restrict__ x_in, const float2 *__restrict__ tables, float2 *__restrict__ y2_out,
               float2 *__restrict__ y3_out, float2 *__restrict__ seam_out, const Stage12Args a, const Stage12Taps<T1, T2, TA> taps)
{
    constexpr int SB = M1 * M2 * Y2B;                      // input samples per body
    constexpr int NY1 = M2 * Y2B;                          // stage-1 outputs per body
    constexpr int HALO = M1 * (T2 - M2) + (T1 - M1);       // input history a lane needs
    constexpr int WU = (HALO + SB - 1) / SB;               // warm-up bodies
    constexpr int C1 = T1 - M1;                            // carried mixed samples
    constexpr int C2 = T2 - M2;                            // carried stage-1 outputs
    using G = TileGeom<SB, W, CPW>;
    constexpr int SPW = kWave / CPW;                       // time segments per wave
    static_assert(SB % 2 == 0, "body must hold an even number of samples (16-byte LDS accesses)");
    static_assert(kWave % CPW == 0, "CPW must divide 64");

    __shared__ __attribute__((aligned(16))) char lds[2 * G::buf_bytes];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int csub = lane / SPW;                           // which of the wave's CPW channels
    const int sl = lane - csub * SPW;                      // which of the wave's SPW time segments

    // blockIdx -> (channel group, time-wave).  Blocks are dealt round-robin over the 8 XCDs
    // (b % 8 labels the XCD group), so the time-waves of one channel group -- which read the same
    // translator tables -- are given the same b % 8 and consecutive b / 8: they meet in one XCD's L2.
    // Whole channel groups go to the XCDs eight at a time; the last (groups % 8) of them are dealt out by units (time-waves), in
    // eight equal runs -- a bank of fewer than eight groups reaches every XCD all the same (as k_stage12m, kernels_mfma.hpp).
    const int xcd = blockIdx.x & 7;
    const int kq = blockIdx.x >> 3;
    const int n_groups = a.n_chgroups - a.cg0;
    const int n_full = (n_groups >> 3) * a.n_wt, n_rem = (n_groups & 7) * a.n_wt;
    int wt, cg;
    if (kq < n_full) {
        wt = kq % a.n_wt;
        cg = a.cg0 + (kq / a.n_wt) * 8 + xcd;
    } else {
        const int w = ((xcd * n_rem) >> 3) + (kq - n_full);
        if (w >= (((xcd + 1) * n_rem) >> 3)) return;
        wt = w % a.n_wt;
        cg = a.cg0 + (n_groups & ~7) + w / a.n_wt;
    }
    const int g = wt / a.P;
    const int p = wt - g * a.P;
    const int bps = a.nb * a.P;                            // bodies per segment = spacing / SB

    const int ch0 = (cg * W + wave) * CPW;                 // first channel of this wave (wave-uniform)
    const int ch = ch0 + csub;                             // this lane's channel
    const bool ch_ok = ch < a.n_ch;

    // first real body slot of segment 0 of this time-wave; whole workgroup leaves if it is past the end
    const int slot0 = g * SPW * bps + a.nb * p;
    if (slot0 * Y2B >= a.n_y2_new) return;

    int tph = a.tphase0 + p * a.nb * SB;                   // translator index at the first warm-up sample
    tph %= a.N;
    const float2 *xsrc = x_in + a.x_base + (size_t)slot0 * SB;   // segment 0, first warm-up sample

    const int n_bodies = WU + a.nb;

    // issue the DMA of body `b` into LDS buffer `bi`: this wave's share of the SPW input rows + the
    // translator rows of its own CPW channels (a channel past the end re-reads the last one's table)
    auto issue_body = [&](int b, int bi, int tphb) {
        char *buf = lds + bi * G::buf_bytes;
        for (int r = wave; r < SPW; r += W)
            dma_row<SB>(xsrc + (size_t)b * SB + (size_t)r * a.spacing, buf + r * G::row_bytes, lane);
#pragma unroll
        for (int c = 0; c < CPW; c++) {
            const int tc = min(ch0 + c, a.n_ch - 1);
            dma_row<SB>(tables + (size_t)tc * a.tbl_stride + tphb, buf + (SPW + wave * CPW + c) * G::row_bytes, lane);
        }
    };

    v2f w1c[C1];   // last C1 mixed samples        (stage-1 delay line, newest last)
    v2f w2c[C2];   // last C2 stage-1 outputs      (stage-2 delay line, newest last)
#pragma unroll
    for (int i = 0; i < C1; i++) w1c[i] = v2f{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < C2; i++) w2c[i] = v2f{0.f, 0.f};

    // Stage-2 outputs leave in BODY-MAJOR order: row[((wt*nb + body)*SPW + segment)*Y2B + q].  The SPW
    // lanes of one channel then write SPW*Y2B*8 contiguous bytes per body (512 B = four whole 128-byte
    // lines for the 10 MS/s chain) in one go.  In time order each lane would write 32 B into its own
    // 384-byte span per body: partial lines that L2 evicts before they fill, which measured as a
    // read-for-ownership plus two writes per byte (3 GB of HBM traffic per launch instead of 1 GB).
    // k_tail undoes the permutation while it stages its tiles (y2_storage_index below).
    float2 *__restrict__ ydst = y2_out + (size_t)(ch_ok ? ch : 0) * a.y2_stride + a.y2_hist
                                + ((size_t)wt * a.nb * SPW + sl) * Y2B;
    v2f y2v[Y2B];
    auto store_y2 = [&](int body) {          // body = 0 .. nb-1 (real bodies only)
        if (!ch_ok) return;
        float2 *dst = ydst + (size_t)body * SPW * Y2B;
        if (Y2B % 2 == 0) {
#pragma unroll
            for (int q = 0; q < Y2B; q += 2)
                *reinterpret_cast<v4f *>(dst + q) = v4f{y2v[q].x, y2v[q].y, y2v[q + 1].x, y2v[q + 1].y};
        } else {
#pragma unroll
            for (int q = 0; q < Y2B; q++) dst[q] = make_float2(y2v[q].x, y2v[q].y);
        }
    };

    issue_body(0, 0, tph);

    for (int b = 0; b < n_bodies; b++) {
        // body b's tile has landed (this wave's DMA: vmcnt; everybody's: barrier).  The barrier also
        // says every wave is done reading the other buffer, so the next body may be fetched into it.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the previous body's outputs are stored HERE, not at the end of that body: the vmcnt(0)
        // above would otherwise wait for the store round trip every body; now the stores are a whole
        // body old by the time the next wait comes
        if (b > WU) store_y2(b - 1 - WU);
        tph += SB;
        if (tph >= a.N) tph -= a.N;
        if (b + 1 < n_bodies) issue_body(b + 1, (b + 1) & 1, tph);

        const char *buf = lds + (b & 1) * G::buf_bytes;
        const v4f *__restrict__ xrow = reinterpret_cast<const v4f *>(buf + sl * G::row_bytes);
        const v4f *__restrict__ trow = reinterpret_cast<const v4f *>(buf + (SPW + wave * CPW + csub) * G::row_bytes);

        // LDS -> registers for stage-1 output k (samples k*M1 .. k*M1+M1-1), 16-byte reads only; when a
        // region starts on an odd sample its first sample is the overhang (.zw) of the previous region's
        // last read, kept in `carry`.
        auto load_region = [&](const v4f *__restrict__ row, int k, v2f (&f)[M1], v2f &carry) {
            const int idx0 = k * M1;
            int first = idx0;
            if (idx0 & 1) {
                f[0] = carry;
                first = idx0 + 1;
            }
#pragma unroll
            for (int j = first; j < idx0 + M1; j += 2) {
                const v4f v = row[j >> 1];
                f[j - idx0] = v.xy;
                if (j + 1 < idx0 + M1) f[j + 1 - idx0] = v.zw;
                else carry = v.zw;
            }
        };

        // The arithmetic of one body.  FIRST = the lane's very first (warm-up) body: its stage-2 outputs are never
        // stored and only the last C2 stage-1 outputs of the warm-up survive (as the stage-2 delay line), so the
        // stage-2 FIRs, the first KS stage-1 FIRs and the mixer products only those would have read are left out
        // (a third of the body for the 10 MS/s chain; 1/13 of all bodies are warm-up).
        auto run_body = [&](auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr int KS = FIRST ? ((WU * NY1 - C2 > 0) ? ((WU * NY1 - C2 < NY1) ? WU * NY1 - C2 : NY1) : 0) : 0;
            constexpr int KM = (KS * M1 - C1 > 0) ? (KS * M1 - C1) / M1 : 0;   // regions whose mixer products nobody reads
            v2f cx = {0.f, 0.f}, ct = {0.f, 0.f};
            constexpr bool PREFETCH = (M1 <= 6);
            v2f fx[PREFETCH ? 2 : 1][M1];  // input samples: region k+1 is fetched while region k is computed
            v2f ft[M1];                    // translator entries (broadcast reads): fetched one region ahead too,
                                           // but consumed first, so one buffer is enough
            load_region(xrow, 0, fx[0], cx);
            load_region(trow, 0, ft, ct);

#pragma unroll
            for (int q = 0; q < Y2B; q++) {
                v2f win2[T2];
#pragma unroll
                for (int i = 0; i < C2; i++) win2[i] = w2c[i];
#pragma unroll
                for (int u = 0; u < M2; u++) {
                    const int k = q * M2 + u;
                    // the scheduling barriers keep the compiler from hoisting all reads of the body
                    // (VGPR blow-up) and fix the software pipeline: mix region k, then fetch region k+1
                    // (LDS latency hides behind the FIR arithmetic), then the FIR of region k
                    __builtin_amdgcn_sched_barrier(0);
                    v2f win[T1];
#pragma unroll
                    for (int i = 0; i < C1; i++) win[i] = w1c[i];
                    if (k >= KM) {
                        if constexpr (M1 == 5) {
                            cmul5_pk(fx[k & 1], ft, &win[C1]);
                        } else {
#pragma unroll
                            for (int e = 0; e < M1; e++) win[C1 + e] = cmul_pk(fx[PREFETCH ? (k & 1) : 0][e], ft[e]);
                        }
#pragma unroll
                        for (int e = 0; e < M1; e++) asm volatile("" : "+v"(win[C1 + e]));
                    } else {
#pragma unroll
                        for (int e = 0; e < M1; e++) win[C1 + e] = v2f{0.f, 0.f};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 1 < NY1) {
                        load_region(xrow, k + 1, fx[PREFETCH ? ((k + 1) & 1) : 0], cx);
                        load_region(trow, k + 1, ft, ct);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (k >= KS) {
                        win2[C2 + u] = folded_fir_pk<T1>(win, taps.h1);
                        // pin the stage-1 output here: without it LLVM sinks the arithmetic down to the
                        // stage-2 FIR and every LDS read of the body stays live (spills)
                        asm volatile("" : "+v"(win2[C2 + u]));
                    } else {
                        win2[C2 + u] = v2f{0.f, 0.f};
                    }
#pragma unroll
                    for (int i = 0; i < C1; i++) w1c[i] = win[i + M1];
                }
                if constexpr (!FIRST) {
                    y2v[q] = folded_fir_pk<T2>(win2, taps.h2);
                    asm volatile("" : "+v"(y2v[q]));
                }
#pragma unroll
                for (int i = 0; i < C2; i++) w2c[i] = win2[i + M2];
            }
        };
        if (b == 0) run_body(std::true_type{});
        else run_body(std::false_type{});
    }
    store_y2(a.nb - 1);   // the last body's outputs

    if constexpr (MA > 0) {
        // the wave's channels: places wave, wave + wa, ... of the workgroup's W * CPW
        int wa, n_my;
        stage3_split<SPW, Y2B, W>(a, W * CPW, 2 * G::buf_bytes, wa, n_my);
        const int c_first = cg * W * CPW;
        const int place = wave + min(lane, n_my - 1) * wa;
        stage3_epilogue<MA, TA, SPW, Y2B, W, CPW>(a, taps.h3, y2_out, y3_out, seam_out, lds, 2 * G::buf_bytes, g, wave, lane, wa, n_my,
                                                  (place < W * CPW && c_first + place < a.n_ch) ? c_first + place : -1, c_first);
    }
}

// ------------------------------------------------------------------------------------------
// k_stage12p: k_stage12 with TWO channels per lane ("slot" = a pair of channels).
//
// The two channels of a lane share the input samples (one LDS read serves both) and, when their translator tables
// are complex conjugates of each other -- channels at +k and -k steps from the tuner centre: sdrx.cpp:1687 gives
// exp(j*float(phase)) and exp(j*float(-phase)), and libm's sincosf is exactly odd/even -- also the translator reads and
// half of the mixer: x*t and x*conj(t) are P + {-xi*ti, xr*ti} and P + {xi*ti, -xr*ti} with the same P = {xr*tr, xi*tr},
// three packed instructions for the two products instead of four.  Each channel's values come out of the same
// instruction sequence on the same operands as in k_stage12 (bit-identical, tested).  The price is registers: two
// delay-line sets per lane, two waves per SIMD instead of three -- with two independent channels per wave the
// instruction streams interleave as well as three single-channel waves did.
//   CONJ = true : slot = (channel, its conjugate partner or none); one translator row per slot
//   CONJ = false: slot = any two channels (or one); two translator rows per slot
// slots[s] = {channel A, channel B or -1}; lanes of a wave = CPW slots x SPW segments as in k_stage12.
// ------------------------------------------------------------------------------------------
// x[e]*t[e] and x[e]*conj(t[e]) for e = 0..4: 15 packed instructions, every dependent pair five instructions apart
__device__ __forceinline__ void cmul5_conj_pk(const v2f (&x)[5], const v2f (&t)[5], v2f *outA, v2f *outB)
{
    v2f a0, a1, a2, a3, a4, b0, b1, b2, b3, b4;
    asm("v_pk_mul_f32 %5, %10, %15 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %6, %11, %16 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %7, %12, %17 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %8, %13, %18 op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %9, %14, %19 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %10, %15, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %11, %16, %6 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %2, %12, %17, %7 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %3, %13, %18, %8 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %4, %14, %19, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %5, %10, %15, %5 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %6, %11, %16, %6 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %7, %12, %17, %7 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %8, %13, %18, %8 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %9, %14, %19, %9 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
        "s_nop 0"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3), "=&v"(b4)
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(t[4]));
    outA[0] = a0; outA[1] = a1; outA[2] = a2; outA[3] = a3; outA[4] = a4;
    outB[0] = b0; outB[1] = b1; outB[2] = b2; outB[3] = b3; outB[4] = b4;
}

template <int SB, int W, int CPW, int TR>
struct TileGeomP {
    static constexpr int row_bytes = SB * 8 + 16;
    static constexpr int x_rows = kWave / CPW;
    static constexpr int t_rows = TR * W * CPW;              // TR translator rows per slot
    static constexpr int buf_bytes = (x_rows + t_rows) * row_bytes;
};

struct PairArgs {
    const int2 *slots;       // [n_slots] {channel A, channel B or -1}; {-1, -1}: empty
    const unsigned char *group_conj;   // [n_slots / (W*CPW)]: 1 = the group's pairs are conjugate partners
    int n_slots;
    int lds_bytes;           // dynamic LDS of the launch (>= 2 tile buffers; the rest is room for the stage-3 epilogue)
};

template <int T1, int T2, int Y2B, int W, int CPW, bool CONJ, int MA, int TA>
__device__ __forceinline__ void stage12p_body(const float2 *__restrict__ x_in, const float2 *__restrict__ tables, float2 *__restrict__ y2_out,
                                              float2 *__restrict__ y3_out, float2 *__restrict__ seam_out, const Stage12Args &a, const PairArgs &pa,
                                              const Stage12Taps<T1, T2, TA> &taps, char *lds, const int cg, const int wt)
{
    constexpr int M1 = 5, M2 = 5;                          // (the 10 MS/s chain: cmul5_conj_pk)
    constexpr int SB = M1 * M2 * Y2B;
    constexpr int NY1 = M2 * Y2B;
    constexpr int HALO = M1 * (T2 - M2) + (T1 - M1);
    constexpr int WU = (HALO + SB - 1) / SB;
    constexpr int C1 = T1 - M1;
    constexpr int C2 = T2 - M2;
    constexpr int TR = CONJ ? 1 : 2;
    using G = TileGeomP<SB, W, CPW, TR>;
    constexpr int SPW = kWave / CPW;
    static_assert(WU == 1, "one warm-up body");

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int csub = lane / SPW;
    const int sl = lane - csub * SPW;
    const int g = wt / a.P;
    const int p = wt - g * a.P;
    const int bps = a.nb * a.P;

    const int s0 = (cg * W + wave) * CPW;                  // first slot of this wave (wave-uniform)
    const int2 slot = pa.slots[min(s0 + csub, pa.n_slots - 1)];
    const bool okA = s0 + csub < pa.n_slots && slot.x >= 0;     // (slot lists are padded with empty slots {-1, -1})
    const bool okB = okA && slot.y >= 0;
    const int chA = okA ? slot.x : 0, chB = okB ? slot.y : chA; // an empty place works on channel 0's table and stores nothing

    const int slot0 = g * SPW * bps + a.nb * p;
    if (slot0 * Y2B >= a.n_y2_new) return;

    int tph = a.tphase0 + p * a.nb * SB;
    tph %= a.N;
    const float2 *xsrc = x_in + a.x_base + (size_t)slot0 * SB;
    const int n_bodies = WU + a.nb;

    // translator rows of this wave's slots: channel A's table (CONJ: the partner's is its conjugate), else A's and B's
    int trowA[CPW], trowB[CPW];
#pragma unroll
    for (int c = 0; c < CPW; c++) {
        trowA[c] = __builtin_amdgcn_readlane(chA, c * SPW);
        trowB[c] = __builtin_amdgcn_readlane(chB, c * SPW);
    }
    auto issue_body = [&](int b, int bi, int tphb) {
        char *buf = lds + bi * G::buf_bytes;
        for (int r = wave; r < SPW; r += W)
            dma_row<SB>(xsrc + (size_t)b * SB + (size_t)r * a.spacing, buf + r * G::row_bytes, lane);
#pragma unroll
        for (int c = 0; c < CPW; c++) {
            dma_row<SB>(tables + (size_t)trowA[c] * a.tbl_stride + tphb, buf + (SPW + (wave * CPW + c) * TR) * G::row_bytes, lane);
            if constexpr (!CONJ)
                dma_row<SB>(tables + (size_t)trowB[c] * a.tbl_stride + tphb, buf + (SPW + (wave * CPW + c) * TR + 1) * G::row_bytes, lane);
        }
    };

    v2f w1a[C1], w1b[C1], w2a[C2], w2b[C2];
#pragma unroll
    for (int i = 0; i < C1; i++) w1a[i] = w1b[i] = v2f{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < C2; i++) w2a[i] = w2b[i] = v2f{0.f, 0.f};

    const size_t yoff = a.y2_hist + ((size_t)wt * a.nb * SPW + sl) * Y2B;
    float2 *__restrict__ ydA = y2_out + (size_t)chA * a.y2_stride + yoff;
    float2 *__restrict__ ydB = y2_out + (size_t)chB * a.y2_stride + yoff;
    v2f yA[Y2B], yB[Y2B];
    auto store_y2 = [&](int body) {
        static_assert(Y2B % 2 == 0, "16-byte stores");
        if (okA) {
            float2 *dst = ydA + (size_t)body * SPW * Y2B;
#pragma unroll
            for (int q = 0; q < Y2B; q += 2) *reinterpret_cast<v4f *>(dst + q) = v4f{yA[q].x, yA[q].y, yA[q + 1].x, yA[q + 1].y};
        }
        if (okB) {
            float2 *dst = ydB + (size_t)body * SPW * Y2B;
#pragma unroll
            for (int q = 0; q < Y2B; q += 2) *reinterpret_cast<v4f *>(dst + q) = v4f{yB[q].x, yB[q].y, yB[q + 1].x, yB[q + 1].y};
        }
    };

    issue_body(0, 0, tph);

    for (int b = 0; b < n_bodies; b++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (b > WU) store_y2(b - 1 - WU);
        tph += SB;
        if (tph >= a.N) tph -= a.N;
        if (b + 1 < n_bodies) issue_body(b + 1, (b + 1) & 1, tph);

        const char *buf = lds + (b & 1) * G::buf_bytes;
        const v4f *__restrict__ xrow = reinterpret_cast<const v4f *>(buf + sl * G::row_bytes);
        const v4f *__restrict__ trA = reinterpret_cast<const v4f *>(buf + (SPW + (wave * CPW + csub) * TR) * G::row_bytes);
        const v4f *__restrict__ trB = reinterpret_cast<const v4f *>(buf + (SPW + (wave * CPW + csub) * TR + (CONJ ? 0 : 1)) * G::row_bytes);

        auto load_region = [&](const v4f *__restrict__ row, int k, v2f (&f)[M1], v2f &carry) {
            const int idx0 = k * M1;
            int first = idx0;
            if (idx0 & 1) {
                f[0] = carry;
                first = idx0 + 1;
            }
#pragma unroll
            for (int j = first; j < idx0 + M1; j += 2) {
                const v4f v = row[j >> 1];
                f[j - idx0] = v.xy;
                if (j + 1 < idx0 + M1) f[j + 1 - idx0] = v.zw;
                else carry = v.zw;
            }
        };

        auto run_body = [&](auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            constexpr int KS = FIRST ? ((WU * NY1 - C2 > 0) ? ((WU * NY1 - C2 < NY1) ? WU * NY1 - C2 : NY1) : 0) : 0;
            constexpr int KM = (KS * M1 - C1 > 0) ? (KS * M1 - C1) / M1 : 0;
            v2f cx = {0.f, 0.f}, cta = {0.f, 0.f}, ctb = {0.f, 0.f};
            v2f fx[2][M1];
            v2f fta[M1], ftb[CONJ ? 1 : M1];
            load_region(xrow, 0, fx[0], cx);
            load_region(trA, 0, fta, cta);
            if constexpr (!CONJ) load_region(trB, 0, ftb, ctb);
#pragma unroll
            for (int q = 0; q < Y2B; q++) {
                v2f win2a[T2], win2b[T2];
#pragma unroll
                for (int i = 0; i < C2; i++) {
                    win2a[i] = w2a[i];
                    win2b[i] = w2b[i];
                }
#pragma unroll
                for (int u = 0; u < M2; u++) {
                    const int k = q * M2 + u;
                    __builtin_amdgcn_sched_barrier(0);
                    v2f wina[T1], winb[T1];
#pragma unroll
                    for (int i = 0; i < C1; i++) {
                        wina[i] = w1a[i];
                        winb[i] = w1b[i];
                    }
                    if (k >= KM) {
                        if constexpr (CONJ) {
                            cmul5_conj_pk(fx[k & 1], fta, &wina[C1], &winb[C1]);
                        } else {
                            cmul5_pk(fx[k & 1], fta, &wina[C1]);
                            cmul5_pk(fx[k & 1], ftb, &winb[C1]);
                        }
#pragma unroll
                        for (int e = 0; e < M1; e++) {
                            asm volatile("" : "+v"(wina[C1 + e]));
                            asm volatile("" : "+v"(winb[C1 + e]));
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < M1; e++) wina[C1 + e] = winb[C1 + e] = v2f{0.f, 0.f};
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (k + 1 < NY1) {
                        load_region(xrow, k + 1, fx[(k + 1) & 1], cx);
                        load_region(trA, k + 1, fta, cta);
                        if constexpr (!CONJ) load_region(trB, k + 1, ftb, ctb);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (k >= KS) {
                        win2a[C2 + u] = folded_fir_pk<T1>(wina, taps.h1);
                        win2b[C2 + u] = folded_fir_pk<T1>(winb, taps.h1);
                        asm volatile("" : "+v"(win2a[C2 + u]));
                        asm volatile("" : "+v"(win2b[C2 + u]));
                    } else {
                        win2a[C2 + u] = win2b[C2 + u] = v2f{0.f, 0.f};
                    }
#pragma unroll
                    for (int i = 0; i < C1; i++) {
                        w1a[i] = wina[i + M1];
                        w1b[i] = winb[i + M1];
                    }
                }
                if constexpr (!FIRST) {
                    yA[q] = folded_fir_pk<T2>(win2a, taps.h2);
                    yB[q] = folded_fir_pk<T2>(win2b, taps.h2);
                    asm volatile("" : "+v"(yA[q]));
                    asm volatile("" : "+v"(yB[q]));
                }
#pragma unroll
                for (int i = 0; i < C2; i++) {
                    w2a[i] = win2a[i + M2];
                    w2b[i] = win2b[i + M2];
                }
            }
        };
        if (b == 0) run_body(std::true_type{});
        else run_body(std::false_type{});
    }
    store_y2(a.nb - 1);

    if constexpr (MA > 0) {
        // stage 3 of the workgroup's own ranges; its channels are its slots' A and B: place i = slot i/2, A or B
        const int sbase = cg * W * CPW;                      // first slot of the workgroup
        auto place_channel = [&](int i) -> int {             // -1: empty
            const int s = sbase + (i >> 1);
            if (s >= pa.n_slots) return -1;
            const int2 sp = pa.slots[s];
            return (i & 1) ? sp.y : sp.x;
        };
        int wa, n_my;                                        // places per wave: wave, wave + wa, ...
        stage3_split<SPW, Y2B, W>(a, 2 * W * CPW, pa.lds_bytes, wa, n_my);
        // (read once, here: a global load inside the epilogue's loop would make hipcc wait for vmcnt(0) and drain the DMAs)
        const int place = wave + min(lane, n_my - 1) * wa;
        const int mych = place < 2 * W * CPW ? place_channel(place) : -1;
        const int chn0 = __builtin_amdgcn_readfirstlane(place_channel(0));
        stage3_epilogue<MA, TA, SPW, Y2B, W, 2 * CPW>(a, taps.h3, y2_out, y3_out, seam_out, lds, pa.lds_bytes, g, wave, lane, wa, n_my, mych, chn0);
    }
}

// One launch covers both kinds of slot groups (a second, small launch for the few general groups would run mostly
// alone on the chip: measured 0.1 ms for 6 % of the channels): group_conj[cg] says which body a workgroup takes.
template <int T1, int T2, int Y2B, int W, int CPW, int MA = 0, int TA = 1>
__global__ __launch_bounds__(kWave * W, 2)
void k_stage12p(const float2 *__restrict__ x_in, const float2 *__restrict__ tables, float2 *__restrict__ y2_out,
                float2 *__restrict__ y3_out, float2 *__restrict__ seam_out, const Stage12Args a, const PairArgs pa,
                const Stage12Taps<T1, T2, TA> taps)
{
    extern __shared__ __attribute__((aligned(16))) char lds_p[];
    // (whole slot groups to the XCDs eight at a time, the last (groups % 8) by units: as k_stage12)
    const int xcd = blockIdx.x & 7;
    const int kq = blockIdx.x >> 3;
    const int n_groups = a.n_chgroups - a.cg0;
    const int n_full = (n_groups >> 3) * a.n_wt, n_rem = (n_groups & 7) * a.n_wt;
    int wt, cg;                                            // time-wave, slot group
    if (kq < n_full) {
        wt = kq % a.n_wt;
        cg = a.cg0 + (kq / a.n_wt) * 8 + xcd;
    } else {
        const int w = ((xcd * n_rem) >> 3) + (kq - n_full);
        if (w >= (((xcd + 1) * n_rem) >> 3)) return;
        wt = w % a.n_wt;
        cg = a.cg0 + (n_groups & ~7) + w / a.n_wt;
    }
    if (pa.group_conj[cg])
        stage12p_body<T1, T2, Y2B, W, CPW, true, MA, TA>(x_in, tables, y2_out, y3_out, seam_out, a, pa, taps, lds_p, cg, wt);
    else
        stage12p_body<T1, T2, Y2B, W, CPW, false, MA, TA>(x_in, tables, y2_out, y3_out, seam_out, a, pa, taps, lds_p, cg, wt);
}

// ------------------------------------------------------------------------------------------
// k_tail: the remaining one or two decimator stages.  One workgroup per (channel, tile of
// tile_b final outputs); the stage-2 outputs a tile needs (its own span plus the FIR history in
// front of it) are staged in LDS, stage A is computed into LDS, stage B from there.  Tiles are
// small (tens of KB of LDS) so that several workgroups per CU overlap their HBM reads with each
// other's arithmetic: this kernel is bound by streaming the stage-2 scratch (8 B per 25 input
// samples per channel).
//
// The stage-2 history a channel carries from call to call lives in a separate ping-pong buffer
// (hist_in -> hist_out) so that no tile reads what another tile of the same launch writes.
// Row coordinates below: index p of the virtual row [hist (H entries) | new (n_y2_new entries)].
// ------------------------------------------------------------------------------------------
struct TailArgs {
    const float2 *y2_new;   // [n_ch][y2_stride]
    const float2 *hist_in;  // [n_ch][hist]
    float2       *hist_out; // [n_ch][hist]
    float2       *iq_out;   // [n_ch][out_stride] or nullptr
    float2       *iq_keep;  // [n_ch][keep_stride] compact copy for k_agc / k_squelch, or nullptr: they read iq_out
    int n_ch;               // one past the last channel of this launch
    int ch_lo;              // its first channel (a launch may cover part of the bank)
    int y2_stride, hist, n_y2_new;
    int out_stride, keep_stride;
    int has_a;              // 1: stage A then stage B; 0: stage B only
    int mA, tA, mB, tB;
    int pA0;                // row index of the oldest sample of stage-A output 0's window
    int pB0;                // index (in stage-A outputs, or in the row if !has_a) of the oldest sample of output 0's window
    int n_b;                // stage-B outputs = outputs of the call
    int tile_b;             // outputs per tile
    int n_tiles;            // >= 1
    // body-major storage of the new stage-2 outputs (see k_stage12): geometry of the launch that wrote them
    int y2b, seg_len, ph_len, spw_shift, spw, nb, P;       // seg_len = bps*Y2B, ph_len = nb*Y2B
    int y2b_shift;          // log2(y2b)
    int half_split;         // 1: written by k_stage12m -- a body's four outputs of a segment lie as two pairs, one in each of the
                            // body's two half-body rows: [time-wave][2 body + h][segment][2] (kernels_mfma.hpp); y2b is 4
    float rcp_seg_len, rcp_ph_len, rcp_y2b;
    float hA[kMaxHalfTaps];
    float hB[kMaxHalfTaps];
};

// time index j (0 = first new stage-2 output of this call) -> index in the channel's body-major row
__device__ __forceinline__ int y2_storage_index(const TailArgs &a, int j)
{
    // small non-negative j: floor((j + 0.5) / d) through a float reciprocal is exact
    const int seg = (int)(((float)j + 0.5f) * a.rcp_seg_len);
    const int r = j - seg * a.seg_len;
    const int pp = (int)(((float)r + 0.5f) * a.rcp_ph_len);
    const int r2 = r - pp * a.ph_len;
    const int body = (int)(((float)r2 + 0.5f) * a.rcp_y2b);
    const int q = r2 - body * a.y2b;
    const int g = seg >> a.spw_shift;
    const int sl = seg & (a.spw - 1);
    const int wt = g * a.P + pp;
    if (a.half_split) return ((wt * 2 * a.nb + 2 * body + (q >> 1)) * a.spw + sl) * 2 + (q & 1);
    return ((wt * a.nb + body) * a.spw + sl) * a.y2b + q;
}

__device__ __forceinline__ v2f folded_fir_lds(const v2f *__restrict__ w, int T, const float *__restrict__ h)
{
    const int half = (T - 1) >> 1;
    v2f acc = {0.f, 0.f};
    for (int i = 0; i < half; i++) acc = __builtin_elementwise_fma(w[i] + w[T - 1 - i], v2f{h[i], h[i]}, acc);
    return __builtin_elementwise_fma(w[half], v2f{h[half], h[half]}, acc);
}

__global__ __launch_bounds__(256) void k_tail(const TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    __shared__ float hA[kMaxHalfTaps], hB[kMaxHalfTaps];

    // blockIdx -> (channel, tile): the tiles of one channel share b % 8 (= one XCD's L2 under the
    // round-robin block placement) and follow each other, so a tile's FIR history -- the tail of the
    // previous tile's span -- is an L2 hit, not a second HBM read.
    const int xcd = blockIdx.x & 7;
    const int qq = blockIdx.x >> 3;
    const int q = qq % a.n_tiles;
    const int chf = (qq / a.n_tiles) * 8 + xcd;
    if (chf >= a.n_ch) return;
    // channels are walked from the last to the first: the stage-1/2 kernel wrote the high channel
    // groups last, so their rows are the ones still held by the Infinity Cache / L2
    const int ch = a.n_ch - 1 - chf;
    if (ch < a.ch_lo) return;
    const int tid = threadIdx.x;
    const float2 *__restrict__ ynew = a.y2_new + (size_t)ch * a.y2_stride;
    const float2 *__restrict__ hin = a.hist_in + (size_t)ch * a.hist;

    if (tid < kMaxHalfTaps) {
        hA[tid] = a.hA[tid];
        hB[tid] = a.hB[tid];
    }

    // history for the next call = the last `hist` entries of the row [hist | new]
    if (q == a.n_tiles - 1) {
        float2 *__restrict__ hout = a.hist_out + (size_t)ch * a.hist;
        for (int i = tid; i < a.hist; i += blockDim.x) {
            const int idx = a.n_y2_new + i;
            hout[i] = idx < a.hist ? hin[idx] : ynew[y2_storage_index(a, idx - a.hist)];
        }
    }

    const int s0 = q * a.tile_b;
    const int nbt = min(a.n_b - s0, a.tile_b);             // outputs of this tile
    if (nbt <= 0) return;

    // spans (all in units of samples of the respective rate)
    int r0 = 0, nr = 0, p0, np;
    if (a.has_a) {
        r0 = a.pB0 + a.mB * s0;                            // first stage-A output needed
        nr = a.mB * (nbt - 1) + a.tB;
        p0 = a.pA0 + a.mA * r0;
        np = a.mA * (nr - 1) + a.tA;
    } else {
        p0 = a.pB0 + a.mB * s0;
        np = a.mB * (nbt - 1) + a.tB;
    }
    // stage the span with 16-byte loads: start on an even row index (hist is even and both rows are
    // 16-byte aligned, so a pair never straddles the hist|new seam)
    const int p0a = p0 & ~1;
    const int odd = p0 - p0a;
    const int npair = (np + odd + 1) >> 1;
    v4f *s2v = reinterpret_cast<v4f *>(smem);              // [npair] pairs
    v2f *s2 = reinterpret_cast<v2f *>(smem) + odd;         // sample i of the span
    v2f *s3 = reinterpret_cast<v2f *>(smem) + 2 * npair;   // stage-A outputs, after the span

    const int n_row = a.hist + a.n_y2_new;                 // valid row entries
    // all loads of a thread first, then its LDS writes (guarded loads would serialise into load -> wait -> write
    // per piece, see k_tail_s); a pair past the end of the row re-reads the last pair (never inside a FIR window),
    // and the second entry of the last pair, when n_row is odd, lies inside the same allocated group of Y2B
    const int p_lastpair = (n_row - 1) & ~1;
    auto pair_src = [&](int i) -> const v4f * {
        const int p = min(p0a + 2 * i, p_lastpair);
        return reinterpret_cast<const v4f *>(p < a.hist ? hin + p : ynew + y2_storage_index(a, p - a.hist));
    };
    for (int i0 = 0; i0 < npair; i0 += 4 * 256) {
        v4f v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *pair_src(i0 + tid + 256 * u);
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i0 + tid + 256 * u < npair) s2v[i0 + tid + 256 * u] = v[u];
    }
    __syncthreads();

    const v2f *bsrc = s2;
    if (a.has_a) {
        for (int r = tid; r < nr; r += blockDim.x) s3[r] = folded_fir_lds(s2 + a.mA * r, a.tA, hA);
        __syncthreads();
        bsrc = s3;
    }
    // stage B: two lanes per output, each takes half of the folded tap pairs; the halves meet through
    // one cross-lane add (DPP), so all 256 threads work on 128 outputs
    {
        const int s = tid >> 1;
        const int hsel = tid & 1;
        const int half = (a.tB - 1) >> 1;
        const int split = (half + 1) >> 1;                 // lane 0: pairs [0, split); lane 1: [split, half) + centre
        v2f acc = {0.f, 0.f};
        if (s < nbt) {
            const v2f *w = bsrc + a.mB * s;
            const int i0 = hsel ? split : 0, i1 = hsel ? half : split;
            for (int i = i0; i < i1; i++) acc = __builtin_elementwise_fma(w[i] + w[a.tB - 1 - i], v2f{hB[i], hB[i]}, acc);
            if (hsel) acc = __builtin_elementwise_fma(w[half], v2f{hB[half], hB[half]}, acc);
        }
        acc.x += __shfl_xor(acc.x, 1);
        acc.y += __shfl_xor(acc.y, 1);
        if (s < nbt && hsel == 0) {
            const float2 o = make_float2(acc.x, acc.y);
            if (a.iq_out) a.iq_out[(size_t)ch * a.out_stride + s0 + s] = o;
            if (a.iq_keep) a.iq_keep[(size_t)ch * a.keep_stride + s0 + s] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_tail_s: k_tail with the tail stages' (m, taps) as template parameters -- same tiles and the
// same staging, the FIRs in the reference's summation order, less than half the instructions:
// measured on the 10 000-channel workload k_tail spends 0.22 ms in VALU issue alone (run-time
// tap loops, LDS-resident taps, three float-reciprocal divisions per 16-byte load) next to
// 0.25 ms of loads.  Here the FIR loops are fully unrolled (LDS reads with immediate offsets, taps
// as SGPR operands), and the span is staged by groups of Y2B consecutive outputs (contiguous both
// in time and in the body-major storage), one index computation per group.
// MA == 0: chains with three stages (no stage A).
// ------------------------------------------------------------------------------------------
// group index (time j, a multiple of Y2B) -> index in the channel's body-major row
__device__ __forceinline__ int y2_group_index(const TailArgs &a, int j)
{
    // P == 1 (one workgroup row per segment) is the only layout whose offset splits by shifts alone,
    // anything else goes through the general mapping
    if (a.P == 1) {
        const int seg = (int)(((float)j + 0.5f) * a.rcp_seg_len);
        const int r = j - __mul24(seg, a.seg_len);
        const int body = r >> a.y2b_shift;
        const int g = seg >> a.spw_shift, sl = seg & (a.spw - 1);
        // (half_split, y2b == 4: the group's first pair; its second pair lies one half-body row = 2 * spw entries further)
        if (a.half_split) return ((__mul24(g, a.nb) + body) << (a.spw_shift + 2)) | (sl << 1);
        return ((__mul24(g, a.nb) + body) << (a.spw_shift + a.y2b_shift)) | (sl << a.y2b_shift);
    }
    return y2_storage_index(a, j);
}

// one (channel, tile) of k_tail_s: everything a workgroup of that kernel does
template <int MA, int TA, int MB, int TB>
__device__ __forceinline__ void tail_s_unit(const TailArgs &a, const int ch, const int q, float2 *smem)
{
    constexpr bool HAS_A = MA > 0;
    const int tid = threadIdx.x;
    const float2 *__restrict__ ynew = a.y2_new + (size_t)ch * a.y2_stride;
    const float2 *__restrict__ hin = a.hist_in + (size_t)ch * a.hist;

    if (q == a.n_tiles - 1) {
        float2 *__restrict__ hout = a.hist_out + (size_t)ch * a.hist;
        for (int i = tid; i < a.hist; i += 256) {
            const int idx = a.n_y2_new + i;
            hout[i] = idx < a.hist ? hin[idx] : ynew[y2_storage_index(a, idx - a.hist)];
        }
    }

    const int s0 = q * a.tile_b;
    const int nbt = min(a.n_b - s0, a.tile_b);
    if (nbt <= 0) return;

    int r0 = 0, nr = 0, p0, np;
    if constexpr (HAS_A) {
        r0 = a.pB0 + MB * s0;
        nr = MB * (nbt - 1) + TB;
        p0 = a.pA0 + MA * r0;
        np = MA * (nr - 1) + TA;
    } else {
        p0 = a.pB0 + MB * s0;
        np = MB * (nbt - 1) + TB;
    }
    // stage by groups of G = Y2B outputs (G = 2 or 4; hist is a multiple of 64, so a group never straddles
    // the hist|new seam and every group starts 16-byte aligned in both rows)
    const int G = a.y2b, gsh = a.y2b_shift;
    const int p0g = p0 & ~(G - 1);
    const int lead = p0 - p0g;
    const int ngrp = (np + lead + G - 1) >> gsh;
    v4f *s2v = reinterpret_cast<v4f *>(smem);
    v2f *s2 = reinterpret_cast<v2f *>(smem) + lead;             // sample i of the span
    v2f *s3 = reinterpret_cast<v2f *>(smem) + (ngrp << gsh);    // stage-A outputs, after the span
    const int n_row = a.hist + a.n_y2_new;
    // Every thread issues ALL its loads, then writes them to LDS: one round trip to memory per tile.  (Guarded
    // loads compile to load -> wait -> LDS write, one after the other: four dependent round trips per thread,
    // measured as 0.27 ms for the loads of the 10 000-channel launch.)  So the loads are unconditional: a group
    // past the end of the row re-reads the last group -- whole groups are always allocated and written
    // (k_stage12 stores Y2B outputs at a time), and entries >= n_row are never inside a FIR window.
    const int p_last = (n_row - 1) & ~(G - 1);
    // off2: where the group's second pair lies -- the next two entries, or (half_split, among the new outputs) one half-body row further
    auto group_src = [&](int i, int &off2) -> const float2 * {
        const int p = min(p0g + (i << gsh), p_last);
        off2 = (a.half_split && p >= a.hist) ? 2 * a.spw : 2;
        return p < a.hist ? hin + p : ynew + y2_group_index(a, p - a.hist);
    };
    for (int i0 = 0; i0 < ngrp; i0 += 512) {
        const int ia = i0 + tid, ib = i0 + tid + 256;
        int sa2, sb2;
        const float2 *sa = group_src(ia, sa2), *sb = group_src(ib, sb2);
        if (G == 4) {
#ifdef SDRX_LAB_TAIL_LD_NT                                  // lab: streaming loads of the stage-2 scratch
            const v4f a0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(sa)), a1 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(sa + sa2));
            const v4f b0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(sb)), b1 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(sb + sb2));
#else
            const v4f a0 = *reinterpret_cast<const v4f *>(sa), a1 = *reinterpret_cast<const v4f *>(sa + sa2);
            const v4f b0 = *reinterpret_cast<const v4f *>(sb), b1 = *reinterpret_cast<const v4f *>(sb + sb2);
#endif
            if (ia < ngrp) { s2v[2 * ia] = a0; s2v[2 * ia + 1] = a1; }
            if (ib < ngrp) { s2v[2 * ib] = b0; s2v[2 * ib + 1] = b1; }
        } else {
            const v4f a0 = *reinterpret_cast<const v4f *>(sa);
            const v4f b0 = *reinterpret_cast<const v4f *>(sb);
            if (ia < ngrp) s2v[ia] = a0;
            if (ib < ngrp) s2v[ib] = b0;
        }
    }
    __syncthreads();

    const v2f *bsrc = s2;
    if constexpr (HAS_A) {
        // (two neighbouring outputs per lane from one register window of TA + MA entries -- 26 LDS reads for two outputs instead of 84, one
        // pass of 191 lanes per tile -- measured SLOWER here, 0.2335 against 0.2285 ms at 10 000 channels, 0.0474 / 0.0453 at 2 048: 58
        // registers instead of 28, and this kernel lives on the number of workgroups whose loads are in flight; in k_tail_agc, whose
        // waves are few and long-lived, the same change pays: kernels_tailagc.hpp)
        for (int r = tid; r < nr; r += 256) s3[r] = folded_fir_ahead<TA, kMaxHalfTaps, SDRX_FIR_AHEAD_A>(s2 + MA * r, a.hA);
        __syncthreads();
        bsrc = s3;
    }
    // stage B: one lane per output, the reference's summation order (msd.hpp:300-335); the tile's 64 outputs are
    // one wave's work, the other waves are done (k_tail splits an output over two lanes: more parallel, but three
    // times the issue slots, and issue slots are what this kernel runs out of)
    if (tid < nbt) {
        const v2f r = folded_fir_ahead<TB, kMaxHalfTaps, SDRX_FIR_AHEAD_B>(bsrc + MB * tid, a.hB);
        const float2 o = make_float2(r.x, r.y);
        if (a.iq_out) a.iq_out[(size_t)ch * a.out_stride + s0 + tid] = o;
        if (a.iq_keep) a.iq_keep[(size_t)ch * a.keep_stride + s0 + tid] = o;
    }
}

template <int MA, int TA, int MB, int TB>
__global__ __launch_bounds__(256) void k_tail_s(const TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int xcd = blockIdx.x & 7;
    const int qq = blockIdx.x >> 3;
    const int q = qq % a.n_tiles;
    const int chf = (qq / a.n_tiles) * 8 + xcd;
    if (chf >= a.n_ch) return;
    const int ch = a.n_ch - 1 - chf;
    if (ch < a.ch_lo) return;
    tail_s_unit<MA, TA, MB, TB>(a, ch, q, smem);
}

// ------------------------------------------------------------------------------------------
// k_tail4: what follows a k_stage12 with fused stage 3.  One workgroup per channel:
//   * the call's stage-3 outputs [hist3 | interior outputs from y3] are staged in LDS;
//   * the few stage-3 outputs per range boundary that k_stage12 left out (window across two workgroup ranges, or
//     reaching into the previous call) are computed here from the stage-2 scratch / hist2 -- the same instruction
//     sequence on the same operands as the interior ones, so the split into ranges never shows in a result;
//   * last stage (MB, TB) -> iq_out / iq_keep, one lane per output, the reference's summation order;
//   * hist2 / hist3 (the last H2 stage-2 / H3 stage-3 outputs of the stream) are rolled for the next call.
// ------------------------------------------------------------------------------------------
struct Tail4Args {
    TailArgs t;              // y2_new, n_ch, y2_stride, n_y2_new, out/keep strides and pointers, n_b, body-major geometry (P == 1), hA, hB
    const float2 *y3;        // [n_ch][y3_stride] interior stage-3 outputs of this call
    const float2 *seam;      // [n_ch][seam_stride]: per range of k_stage12 its first and last TA-1 stage-2 outputs
    int seam_stride;
    float2 *hist2;           // [n_ch][H2]
    float2 *hist3;           // [n_ch][H3]
    int y3_stride, n_y3, c0; // stage-3 output rr: stage-2 window MA*rr + c0 - (TA-1) .. MA*rr + c0   (indices within the call)
    int d0;                  // final output k: stage-3 window MB*k + d0 - (TB-1) .. MB*k + d0
    int H2, H3;
    int rlen, n_bnd;         // stage-2 outputs per workgroup range of k_stage12; range boundaries inside the call (incl. 0)
};

template <int MA, int TA, int MB, int TB>
__global__ __launch_bounds__(256) void k_tail4(const Tail4Args a)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const TailArgs &t = a.t;
    const int ch = t.n_ch - 1 - blockIdx.x;                  // newest-written rows first (Infinity Cache)
    const int tid = threadIdx.x;
    const float2 *__restrict__ y2 = t.y2_new + (size_t)ch * t.y2_stride;
    const float2 *__restrict__ y3 = a.y3 + (size_t)ch * a.y3_stride;
    float2 *__restrict__ h2 = a.hist2 + (size_t)ch * a.H2;
    float2 *__restrict__ h3 = a.hist3 + (size_t)ch * a.H3;
    v4f *__restrict__ row4 = reinterpret_cast<v4f *>(smem);
    v2f *__restrict__ row = reinterpret_cast<v2f *>(smem);   // row[H3 + rr] = stage-3 output rr of this call

    // (1) every global load of the workgroup is issued before anything waits.
    // boundary outputs: thread (bnd, u) takes the u-th stage-3 output whose window crosses stage-2 index bnd*rlen
    const int bnd = tid >> 2;
    const int B = bnd * a.rlen;
    const int rr_s = (B - a.c0 + MA - 1) / MA + (tid & 3);
    const int jn_s = MA * rr_s + a.c0;
    const bool seam = bnd < a.n_bnd && rr_s < a.n_y3 && jn_s <= B + TA - 2;
    static_assert((TA - 1 + MA - 1) / MA <= 4, "at most four stage-3 outputs per range boundary");
    v2f sw[TA];
    if (seam) {
        // window j = jn_s - (TA-1) .. jn_s around B: below B the last TA-1 outputs of range bnd-1 (bnd 0: of the previous
        // calls, hist2), from B on the first TA-1 outputs of range bnd
        constexpr int SE = 2 * (TA - 1);
        const float2 *__restrict__ sm = a.seam + (size_t)ch * a.seam_stride + bnd * SE;
        const float2 *__restrict__ lo = (bnd == 0) ? h2 + a.H2 : sm;             // lo[j - B], j < B (range bnd-1's last TA-1 end right before sm)
#pragma unroll
        for (int i = 0; i < TA; i++) {
            const int jr = jn_s - (TA - 1) + i - B;          // -(TA-1) .. TA-2
            const float2 v = (jr < 0) ? lo[jr] : sm[jr];
            sw[i] = v2f{v.x, v.y};
        }
    }
    // the last H2 stage-2 outputs of the stream, for the next call's first boundary
    float2 nh2 = make_float2(0.f, 0.f);
    const int hi = tid - 128;
    if (hi >= 0 && hi < a.H2) {
        const int j = t.n_y2_new + hi - a.H2;
        nh2 = (j < 0) ? h2[a.H2 + j] : y2[y2_storage_index(t, j)];
    }
    // (2) [hist3 | y3] -> LDS, 16-byte pieces (H3 is even: a piece never straddles the seam); one batch for a 32 ms block
    const int npieces = (a.H3 + a.n_y3 + 1) >> 1;
    constexpr int PU = 6;                                    // 256 x 6 pieces = 3 072 samples >= 80 + 2 561
    for (int i0 = 0; i0 < npieces; i0 += 256 * PU) {
        v4f pv[PU];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const int p = min(i0 + tid + 256 * u, npieces - 1);
            const float2 *src = (2 * p < a.H3) ? h3 + 2 * p : y3 + (2 * p - a.H3);
            pv[u] = *reinterpret_cast<const v4f *>(src);
        }
#pragma unroll
        for (int u = 0; u < PU; u++)
            if (i0 + tid + 256 * u < npieces) row4[i0 + tid + 256 * u] = pv[u];
    }
    __syncthreads();
    // (3) boundary outputs over the placeholders
    if (seam) {
        v2f acc = {0.f, 0.f};
        folded_fir_steps<TA, kMaxHalfTaps>(acc, sw, t.hA);
        row[a.H3 + rr_s] = acc;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (nh2: every read of the old hist2 is complete before anyone rewrites it)
    __syncthreads();
    // (4) last stage, two outputs per lane (k and k + 256: alternating accumulation chains, see fma_tap2)
    const v2f *__restrict__ brow = row + a.H3 + a.d0 - (TB - 1);
    for (int k = tid; k < t.n_b; k += 512) {
        const bool second = k + 256 < t.n_b;
        const int k2 = second ? k + 256 : k;
        v2f r0, r1;
        folded_fir2_lds<TB>(r0, r1, brow + MB * k, brow + MB * k2, t.hB);
        const float2 o0 = make_float2(r0.x, r0.y), o1 = make_float2(r1.x, r1.y);
        if (t.iq_out) {
            t.iq_out[(size_t)ch * t.out_stride + k] = o0;
            if (second) t.iq_out[(size_t)ch * t.out_stride + k2] = o1;
        }
        if (t.iq_keep) {
            t.iq_keep[(size_t)ch * t.keep_stride + k] = o0;
            if (second) t.iq_keep[(size_t)ch * t.keep_stride + k2] = o1;
        }
    }
    // (5) histories for the next call (every read of the old ones is behind the barriers above)
    for (int i = tid; i < a.H3; i += 256) {
        const v2f v = row[a.n_y3 + i];
        h3[i] = make_float2(v.x, v.y);
    }
    if (hi >= 0 && hi < a.H2) h2[hi] = nh2;
}

// ------------------------------------------------------------------------------------------
// k_agc: lane = channel, strictly serial over the block's outputs (the only true recurrence of the path: 512
// dependent steps per channel per block).  What bounds it is not memory and not the dependency chain but ISSUE: a
// wave that is alone on its SIMD issues one VALU instruction per 4 cycles, and one step of both loops is ~24
// instructions.  So the two recurrences run on TWO waves (two SIMDs) of a 128-thread workgroup per 64 channels:
//   wave 0: IF AGC (AGC::adjust, agc.hpp:48-69) over the decimated IQ; publishes |y|^2 (FM banks: y too) per
//           sample through LDS, 8 samples at a time;
//   wave 1: one chunk behind: envelope (or FM phase step), LF AGC (agc.hpp:106-131), squelch gating, audio out.
// One barrier per 8 samples.  Each value is computed by the same expressions as before the split.
// Each lane streams its own row (16-byte loads, 8 samples ahead), so no transposition is needed.
// ------------------------------------------------------------------------------------------
struct AgcParams {
    float attack, decay, reference, min_gain, max_gain, alpha;
    float sel;          // -inf if decay <= attack, +inf otherwise: picks the gain update without a compare (if_agc_step)
};

struct AgcArgs {
    const float2 *iq;        // [n_ch][iq_stride] decimated IQ; rows 16-byte aligned
    float        *audio;     // [n_ch][out_stride]
    float4       *state;     // [n_ch] {agc.gain, agc.power, lf.gain, lf.power}
    int n_ch, n_out, iq_stride, out_stride;   // n_ch: one past the last channel of this launch
    int ch_lo;               // its first channel
    int lf_active;
    int audio_vec;           // 1: audio rows are 16-byte aligned (vector stores)
    AgcParams agc, lf;
    // squelch gating (sdrx.cpp:458-503); sql == nullptr: every channel is treated as open
    const uchar2 *sql;       // [n_ch] {state, state_prev} left by the previous chunk
    const float *ramp_up, *ramp_down;   // [512]
    // per-channel modulation (k_agc<.., true> only): fm[c] != 0 -> FM demodulator, fm_prev = Demod::prev_sample
    const unsigned char *fm;
    float2 *fm_prev;
    // input-history roll riding along as one extra block: roll_from[0 .. roll_cnt) (the tail of the input ring this
    // chunk was read from) -> roll_to[0 ..) (the head of the other ring); roll_cnt <= 1024, 0: none
    const float2 *roll_from;
    float2 *roll_to;
    int roll_cnt;
};

// AGC::adjust, agc.hpp:48-69, same operation order; returns y = x * gain-before-update through yr, yi and |y|^2.
// a + b*c contracts to FMA here, as it does in the reference's own -march=native build; measured (round 1): a
// contract(off) build is no closer to the portable golden vectors (the 1-ulp v_sqrt and the incoming IQ error
// dominate) and 5 % slower.
__device__ __forceinline__ void if_agc_step(const float2 x, float &g, float &pw, const AgcParams &A, float &yr, float &yi, float &nrm)
{
    // the FMAs are written out (and contraction is off) so that the instruction sequence does not depend on what the
    // surrounding code looks like to the compiler: these are the contractions hipcc chose for the round-1 kernel
#pragma clang fp contract(off)
    // Two of the step's pairs of equal operations go out as ONE packed instruction each (round 6): x g for both components
    // (v_pk_mul_f32, the gain taken for both halves by op_sel) and the two candidate gain updates (v_pk_fma_f32 on (decay, attack)).
    // The same IEEE operations on the same operands -- the same bits (tools/lab/hash_outputs.py) -- in 9 instructions per step instead
    // of 11: the wave that runs this chain has its SIMD to itself, and what it pays for is instructions (a shorter chain of MORE
    // instructions was slower, profiles/r06_agc_lab_ab.txt): k_agc 24.9 -> 23.3 us, the 64-channel block 0.0443 -> 0.0427 ms.
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    const v2f_ y = v2f_{x.x, x.y} * v2f_{g, g};
    yr = y.x;
    yi = y.y;
    nrm = __builtin_fmaf(yr, yr, yi * yi);
    pw = __builtin_fmaf(A.alpha, nrm - pw, pw);
    const float err = A.reference - pw;
    // gain += (err > 0 ? decay : attack) * err (agc.hpp:56-61) without the compare: an fma is monotonic in its product, so with decay <=
    // attack the update the reference selects is the SMALLER of fma(err, decay, g) and fma(err, attack, g) whatever the sign of err
    // (err > 0: err * decay <= err * attack; err < 0: the other way round, and the reference takes attack; err = 0: both are g) -- the
    // larger one with decay > attack.  v_med3_f32 against -inf / +inf (A.sel, set on the host) is that min / max: the same bits, and
    // compare -> select -> fma on the recurrence's critical path becomes two independent fmas -> med3 (k_agc 25.5 -> 24.0 us).
    const v2f_ gg = __builtin_elementwise_fma(v2f_{err, err}, v2f_{A.decay, A.attack}, v2f_{g, g});
    g = __builtin_amdgcn_fmed3f(gg.x, gg.y, A.sel);
    g = __builtin_amdgcn_fmed3f(g, A.min_gain, A.max_gain);      // the clamp of agc.hpp:63-67 in one instruction
}

// mode: 0 = squelch closed (IF AGC only, silence), 1 = open, 2 = just opened (ramp_up), 3 = just closed (ramp_down).
// Branch-free: a branch per step would sit on the loop's critical path.
template <bool SQ, bool FM>
__device__ __forceinline__ float demod_lf_step(float yr, float yi, float nrm, float &gl_, float &pl_, const AgcParams &Lf, int lf_active,
                                               int mode, float ramp, bool is_fm, float2 &fm_prev)
{
    // Demod::demod AM: std::abs, sdrx.cpp:178-180.  v_sqrt_f32 (1 ulp) instead of the correctly rounded software sequence
    float s = __builtin_amdgcn_sqrtf(nrm);
    if constexpr (FM) {
        // Demod::demod FM (EXPERIMENTAL in the reference, sdrx.cpp:181-194): phase step between consecutive
        // amplitude-normalised samples.  Called only while the squelch is open (sdrx.cpp:458-461); the ramp-down
        // chunk takes std::abs whatever the modulation (sdrx.cpp:499).  AM lanes of the same wave compute it too and
        // drop it (no divergence in the loop).
        const float i = yr / s, q = yi / s;                                 // complex<float> / float, sdrx.cpp:185
        const float ph = atan2f(q * fm_prev.x - i * fm_prev.y, i * fm_prev.x + q * fm_prev.y);
        const bool use = is_fm && (!SQ || mode == 1 || mode == 2);
        fm_prev.x = use ? i : fm_prev.x;
        fm_prev.y = use ? q : fm_prev.y;
        s = use ? ph : s;
    }
    // LfAGC::adjust, agc.hpp:106-131 -- only while the squelch is open or ramping down (sdrx.cpp:458-463, 498-501)
    float pl, gl;
    const float sl = s * gl_;
    {
#pragma clang fp contract(off)
        pl = __builtin_fmaf(Lf.alpha, __builtin_fmaf(sl, sl, -pl_), pl_);   // (sl*sl - power is one rounding, as hipcc contracted it in round 1)
        const float errl = Lf.reference - __builtin_amdgcn_sqrtf(pl);
        typedef float v2f_ __attribute__((ext_vector_type(2)));
        const v2f_ gg = __builtin_elementwise_fma(v2f_{errl, errl}, v2f_{Lf.decay, Lf.attack}, v2f_{gl_, gl_});   // (one v_pk_fma_f32, as if_agc_step)
        gl = __builtin_amdgcn_fmed3f(gg.x, gg.y, Lf.sel);
    }
    gl = __builtin_amdgcn_fmed3f(gl, Lf.min_gain, Lf.max_gain);
    float out = lf_active ? sl : s;
    if constexpr (SQ) {
        const bool live = mode != 0;
        pl_ = live ? pl : pl_;
        gl_ = live ? gl : gl_;
        out = ramp * out;                     // ramp = ramp_up[i] / ramp_down[i] / 1 (open) / 0 (closed)
        return live ? out : 0.0f;
    } else {
        pl_ = pl;
        gl_ = gl;
        return out;
    }
}

constexpr int kAgcTile = 32;        // samples per tile (one barrier per tile)
constexpr int kAgcIqPitch = kAgcTile + 2;    // float2 per channel row of an IQ tile: 272 B, conflict-free ds_read_b128 by lane = channel
constexpr int kAgcAuPitch = kAgcTile + 4;    // floats per channel row of an audio tile: 144 B, conflict-free ds_write_b128

template <bool FM>
struct AgcLds {
    float2 iq[3][kWave][kAgcIqPitch];                        // wave 2 fills tile t+2 while wave 0 reads tile t
    float ex[2][kAgcTile][FM ? 3 : 1][kWave];                // wave 0 -> wave 1: |y|^2 (FM: and y) per sample, lane = channel
    float au[2][kWave][kAgcAuPitch];                         // wave 1 -> wave 3
};

// Four waves (one per SIMD) per 64 channels, a software pipeline over tiles of 32 samples:
//   wave 2  IQ rows, tile t+2: coalesced 16-byte loads (16 lanes per channel row) -> LDS.  (Each lane reading its own
//           row straight from memory was the round-1 kernel: every load touched 64 different lines, and the address
//           path, at 64 lines per instruction, made a step cost ~190 cycles.)
//   wave 0  IF AGC on tile t (lane = channel), publishes |y|^2 through LDS
//   wave 1  tile t-1: envelope (or FM phase step), LF AGC, squelch gating -> audio tile in LDS
//   wave 3  tile t-2: audio tile -> coalesced stores
template <bool SQ, bool FM>
__global__ __launch_bounds__(4 * kWave) void k_agc(const AgcArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char agc_smem[];
    AgcLds<FM> &L = *reinterpret_cast<AgcLds<FM> *>(agc_smem);
    // one extra block copies the input history to the front of the other input ring for the next call
    // (k_roll_history's job; riding along here saves a launch on the critical path)
    if (a.roll_cnt > 0 && blockIdx.x == gridDim.x - 1) {
        for (int i = threadIdx.x; i < a.roll_cnt; i += 4 * kWave) a.roll_to[i] = a.roll_from[i];
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & (kWave - 1);
    const int ch0 = a.ch_lo + blockIdx.x * kWave;            // first channel of this workgroup
    const int chr = ch0 + lane;
    const bool ok = chr < a.n_ch;                            // (lanes past the last channel work on the last row and store nothing)
    const int ch = ok ? chr : a.n_ch - 1;
    const int n_tiles = (a.n_out + kAgcTile - 1) / kAgcTile;

    if (wave == 2) {
        // ---- loader: tile t -> L.iq[t % 3]; instruction j covers channels 4j .. 4j+3, 16 lanes x 16 B per row ----
        const int sub = lane >> 4, piece = lane & 15;
        auto load_tile = [&](int t, v4f (&v)[16]) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int c = min(ch0 + 4 * j + sub, a.n_ch - 1);
                const int smp = min(t * kAgcTile + 2 * piece, max(a.n_out - 1, 0) & ~1);   // (a piece past the row's end re-reads the last one)
                v[j] = *reinterpret_cast<const v4f *>(a.iq + (size_t)c * a.iq_stride + smp);
            }
        };
        auto put_tile = [&](int t, const v4f (&v)[16]) {
#pragma unroll
            for (int j = 0; j < 16; j++) *reinterpret_cast<v4f *>(&L.iq[t % 3][4 * j + sub][2 * piece]) = v[j];
        };
        // tiles 0 and 1 before the pipeline starts; from then on the loads of tile i+4 are issued in iteration i and
        // land in LDS in iteration i+2 (two iterations = ~1.5 us: a round trip to memory), two register sets in turn
        v4f va[16], vb[16];
        for (int t = 0; t < 2 && t < n_tiles; t++) {
            load_tile(t, va);
            put_tile(t, va);
        }
        if (2 < n_tiles) load_tile(2, va);
        if (3 < n_tiles) load_tile(3, vb);
        __syncthreads();
        for (int i = 0; i < n_tiles + 2; i += 2) {
            if (i + 2 < n_tiles) put_tile(i + 2, va);         // (buffer (i+2) % 3 was last read in iteration i-1)
            if (i + 4 < n_tiles) load_tile(i + 4, va);
            __syncthreads();
            if (i + 1 < n_tiles + 2) {
                if (i + 3 < n_tiles) put_tile(i + 3, vb);
                if (i + 5 < n_tiles) load_tile(i + 5, vb);
                __syncthreads();
            }
        }
    } else if (wave == 0) {
        // ---- IF AGC ----
        const float4 s4 = a.state[ch];
        float g = s4.x, pw = s4.y;
        __syncthreads();
        // (a tile that lies wholly inside the call takes the branch-free body; only the last one can be short)
        auto if_tile = [&](int i, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            const v4f *__restrict__ row = reinterpret_cast<const v4f *>(&L.iq[i % 3][lane][0]);
            // the whole tile into registers first: the reads do not depend on the recurrence, and left inside the loop every one of them
            // stands in the chain's way (k_tail_agc's IF wave, which reads its tile up front, runs the same 32 steps in 41 ns each
            // where this loop took 54)
            v4f xt[kAgcTile / 2];
#pragma unroll
            for (int k2 = 0; k2 < kAgcTile / 2; k2++) xt[k2] = row[k2];
#pragma unroll
            for (int k2 = 0; k2 < kAgcTile / 2; k2++) {
                const v4f xx = xt[k2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int k = 2 * k2 + h;
                    if (FULL || i * kAgcTile + k < a.n_out) {
                        float yr, yi, nrm;
                        if_agc_step(h ? make_float2(xx.z, xx.w) : make_float2(xx.x, xx.y), g, pw, a.agc, yr, yi, nrm);
                        L.ex[i & 1][k][0][lane] = nrm;
                        if constexpr (FM) {
                            L.ex[i & 1][k][1][lane] = yr;
                            L.ex[i & 1][k][2][lane] = yi;
                        }
                    }
                }
            }
        };
        for (int i = 0; i < n_tiles + 2; i++) {
            if (i < n_tiles) {
                if ((i + 1) * kAgcTile <= a.n_out) if_tile(i, std::true_type{});
                else if_tile(i, std::false_type{});
            }
            __syncthreads();
        }
        if (ok) reinterpret_cast<float2 *>(a.state + ch)[0] = make_float2(g, pw);
    } else if (wave == 1) {
        // ---- envelope / FM phase step -> LF AGC -> squelch gating, one tile behind wave 0 ----
        const float4 s4 = a.state[ch];
        float gl = s4.z, pl = s4.w;
        bool is_fm = false;
        float2 fmp = make_float2(0.f, 0.f);
        if constexpr (FM) {
            is_fm = a.fm[ch] != 0;
            fmp = a.fm_prev[ch];
        }
        int mode = 1;
        const float *__restrict__ ramp = a.ramp_down;
        if constexpr (SQ) {
            const uchar2 q = a.sql[ch];
            mode = q.x ? (q.y ? 1 : 2) : (q.y ? 3 : 0);
            ramp = (mode == 2) ? a.ramp_up : a.ramp_down;
        }
        const bool ramped = SQ && mode >= 2;
        __syncthreads();
        auto lf_tile = [&](int t, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            float nt[kAgcTile];          // (the tile's |y|^2 up front, as above)
#pragma unroll
            for (int k = 0; k < kAgcTile; k++) nt[k] = L.ex[t & 1][k][0][lane];
#pragma unroll
            for (int k4 = 0; k4 < kAgcTile / 4; k4++) {
                float y[4];
                float4 r4 = make_float4(1.f, 1.f, 1.f, 1.f);
                if constexpr (SQ) {      // the ramp tables are 512 long = one chunk (n_out == 512 with squelch)
                    if (ramped) r4 = reinterpret_cast<const float4 *>(ramp)[((t * kAgcTile + 4 * k4) & 511) / 4];
                }
                const float r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const int k = 4 * k4 + h;
                    y[h] = 0.f;
                    if (FULL || t * kAgcTile + k < a.n_out) {
                        const float nrm = nt[k];
                        float yr = 0.f, yi = 0.f;
                        if constexpr (FM) {
                            yr = L.ex[t & 1][k][1][lane];
                            yi = L.ex[t & 1][k][2][lane];
                        }
                        y[h] = demod_lf_step<SQ, FM>(yr, yi, nrm, gl, pl, a.lf, a.lf_active, mode, r[h], is_fm, fmp);
                    }
                }
                *reinterpret_cast<float4 *>(&L.au[t & 1][lane][4 * k4]) = make_float4(y[0], y[1], y[2], y[3]);
            }
        };
        for (int i = 0; i < n_tiles + 2; i++) {
            const int t = i - 1;
            if (t >= 0 && t < n_tiles) {
                if ((t + 1) * kAgcTile <= a.n_out) lf_tile(t, std::true_type{});
                else lf_tile(t, std::false_type{});
            }
            __syncthreads();
        }
        if (ok) {
            reinterpret_cast<float2 *>(a.state + ch)[1] = make_float2(gl, pl);
            if constexpr (FM) a.fm_prev[ch] = fmp;
        }
    } else {
        // ---- storer: audio tile t-2 -> [ch][out_stride], 8 lanes x 16 B per row and instruction (or 4-byte stores) ----
        __syncthreads();
        for (int i = 0; i < n_tiles + 2; i++) {
            const int t = i - 2;
            if (t >= 0) {
                const int s0 = t * kAgcTile;
                if (a.audio_vec) {
                    const int sub = lane >> 3, piece = lane & 7;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int cl = 8 * j + sub;
                        const float4 v = *reinterpret_cast<const float4 *>(&L.au[t & 1][cl][4 * piece]);
                        float *dst = a.audio + (size_t)(ch0 + cl) * a.out_stride + s0 + 4 * piece;
                        if (ch0 + cl < a.n_ch) {
                            if (s0 + 4 * piece + 4 <= a.n_out) {
                                *reinterpret_cast<float4 *>(dst) = v;
                            } else {
                                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                                for (int h = 0; h < 4; h++)
                                    if (s0 + 4 * piece + h < a.n_out) dst[h] = e[h];
                            }
                        }
                    }
                } else {
                    const int sub = lane >> 5, k = lane & 31;
                    for (int j = 0; j < 32; j++) {
                        const int cl = 2 * j + sub;
                        if (ch0 + cl < a.n_ch && s0 + k < a.n_out) a.audio[(size_t)(ch0 + cl) * a.out_stride + s0 + k] = L.au[t & 1][cl][k];
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_squelch: per channel, per 512-output chunk: Hamming window x 512-point forward FFT of the decimated
// (pre-AGC) IQ, in-band power vs. shaped out-of-band noise power, SNR in dB, open/close with 3 dB of
// hysteresis (reference src/sdrx.cpp:533-573; FFTW there, an in-LDS radix-2 FFT here).  One wave per channel.
// ------------------------------------------------------------------------------------------
struct SquelchArgs {
    const float2 *iq;          // [n_ch][iq_stride], 512 valid outputs
    const float *window;       // [512]
    const float *pb;           // [512] passband_shape
    const float2 *tw;          // [256] exp(-j 2 pi k / 512)
    const float *sql_level;    // [n_ch]
    float *snr;                // [n_ch]
    uchar2 *sql;               // [n_ch] {state, state_prev}
    int n_ch, iq_stride;
};

__global__ __launch_bounds__(256) void k_squelch(const SquelchArgs a)
{
    constexpr int N = 512, LOGN = 9;
    __shared__ float2 buf[4][N];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ch = blockIdx.x * 4 + wave;
    const bool ok = ch < a.n_ch;
    float2 *b = buf[wave];
    // windowed input, stored bit-reversed for the decimation-in-time butterflies
    for (int i = lane; i < N; i += kWave) {
        float2 v = make_float2(0.f, 0.f);
        if (ok) {
            const float2 x = a.iq[(size_t)ch * a.iq_stride + i];
            const float w = a.window[i];
            v = make_float2(x.x * w, x.y * w);               // fft_in[i] = iq * window[i], sdrx.cpp:533
        }
        b[__brev((unsigned)i) >> (32 - LOGN)] = v;
    }
    __syncthreads();
    for (int s = 1; s <= LOGN; s++) {
        const int half = 1 << (s - 1);
        for (int bf = lane; bf < N / 2; bf += kWave) {
            const int j = bf & (half - 1);
            const int i0 = ((bf >> (s - 1)) << s) + j, i1 = i0 + half;
            const float2 w = a.tw[j << (LOGN - s)];
            const float2 u = b[i0], v = b[i1];
            const float2 t = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
            b[i0] = make_float2(u.x + t.x, u.y + t.y);
            b[i1] = make_float2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
    // power sums, sdrx.cpp:541-565 (per-bin terms as there; the order of the additions differs)
    float sig = 0.f, hi = 0.f, lo = 0.f;
    for (int i = 3 + lane; i < 91; i += kWave) {
        const float2 p = b[i], q = b[N - i];
        sig += p.x * p.x + p.y * p.y;
        sig += q.x * q.x + q.y * q.y;
    }
    for (int i = 112 + lane; i < 157; i += kWave) {
        const float2 p = b[i], q = b[N - i];
        const float wp = a.pb[i], wq = a.pb[N - i];
        const float pr = p.x * wp, pi = p.y * wp, qr = q.x * wq, qi = q.y * wq;
        hi += pr * pr + pi * pi;
        lo += qr * qr + qi * qi;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        sig += __shfl_xor(sig, o);
        hi += __shfl_xor(hi, o);
        lo += __shfl_xor(lo, o);
    }
    if (ok && lane == 0) {
        const float sig_level = sig / 176;
        const float noise_level = (hi / 45 + lo / 45) / 2;
        const float snr = 10 * log10f(sig_level / noise_level);
        const float lvl = a.sql_level[ch];
        uchar2 st = a.sql[ch];
        st.y = st.x;                                          // sql_state_prev = sql_state, sdrx.cpp:536
        if (snr > lvl + 3 || lvl == 0.0f) st.x = 1;           // sdrx.cpp:569-573
        else if (snr < lvl) st.x = 0;
        a.sql[ch] = st;
        a.snr[ch] = snr;
    }
}

// ------------------------------------------------------------------------------------------
// k_audio_fir: optional per-channel audio FIR (reference FIR::filter, src/fir.hpp:85-102): direct
// form, c[0] meets the OLDEST sample of the window, products accumulated in tap order.  One
// workgroup per channel; the channel's T-1 history samples + the block live in LDS; in place.
// ------------------------------------------------------------------------------------------
struct AudioFirArgs {
    float *audio;            // [n_ch][stride], filtered in place
    float *hist;             // [n_ch][T-1] previous inputs
    const float *taps;       // [T] gain-adjusted (c_adj_, fir.hpp:105-115)
    int n_ch, n_out, stride, T;
};

__global__ __launch_bounds__(256) void k_audio_fir(const AudioFirArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float *c = fsm;                       // [T]
    float *buf = fsm + a.T;               // [T-1 + n_out]
    const int ch = blockIdx.x;
    const int tid = threadIdx.x;
    float *row = a.audio + (size_t)ch * a.stride;
    float *h = a.hist + (size_t)ch * (a.T - 1);
    for (int i = tid; i < a.T; i += blockDim.x) c[i] = a.taps[i];
    for (int i = tid; i < a.T - 1; i += blockDim.x) buf[i] = h[i];
    for (int i = tid; i < a.n_out; i += blockDim.x) buf[a.T - 1 + i] = row[i];
    __syncthreads();
    for (int i = tid; i < a.n_out; i += blockDim.x) {
        float acc = 0.0f;
        for (int k = 0; k < a.T; k++) acc = fmaf(c[k], buf[i + k], acc);
        row[i] = acc;
    }
    for (int i = tid; i < a.T - 1; i += blockDim.x) h[i] = buf[a.n_out + i];
}

// ------------------------------------------------------------------------------------------
// Stereo mix-down of the gated channel audio (optional): the back half of alsa_write_cb.
//   k_mix_partial : pan-weighted sums over groups of kMixGroup consecutive channels, channel order inside a
//                   group as the reference's `for (auto &ch : channels)` (sdrx.cpp:452, 469-497);
//   k_mix_reduce  : group sums added in group order -> planar mix rows [2][stride];
//   (k_audio_fir on the two rows = FIR2, fir.hpp:137-157, sdrx.cpp:646)
//   k_mix_out     : interleave + float -> S16 with the clamp of sdrx.cpp:649-656.
// A bank of <= kMixGroup channels reproduces the reference's summation order exactly; larger banks differ from it
// by re-association only (products and adds are unfused and individually rounded as in the portable build).
// ------------------------------------------------------------------------------------------
constexpr int kMixGroup = 64;

struct MixArgs {
    const float *audio;      // [n_ch][stride] gated mono audio of this chunk
    const float2 *w;         // [n_ch] pan weights (even slot, odd slot)
    float *partial;          // [n_groups][2][pstride]
    float *mix;              // [2][pstride] planar
    float *out_f32;          // [n_out][2] interleaved
    short *out_s16;          // [n_out][2] interleaved
    int n_ch, n_out, stride, pstride, n_groups;
};

// grid (ceil(n_out/256), n_groups), 64 threads; a lane owns 4 consecutive output samples
__global__ __launch_bounds__(64) void k_mix_partial(const MixArgs a)
{
#pragma clang fp contract(off)   // w*s is rounded before the add, as in the reference's portable build
    const int i0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int g = blockIdx.y;
    const int c0 = g * kMixGroup;
    const int c1 = min(c0 + kMixGroup, a.n_ch);
    if (i0 >= a.n_out) return;
    const bool vec = (i0 + 4 <= a.n_out) && ((a.stride & 3) == 0) && (((uintptr_t)a.audio & 15) == 0);
    float e[4] = {0.f, 0.f, 0.f, 0.f}, o[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;
    for (int c = c0; c < c1; c += U) {
        float s[U][4];
        float2 w[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int cc = min(c + u, c1 - 1);
            const float *row = a.audio + (size_t)cc * a.stride + i0;
            w[u] = a.w[cc];
            if (vec) {
                const float4 v = *reinterpret_cast<const float4 *>(row);
                s[u][0] = v.x; s[u][1] = v.y; s[u][2] = v.z; s[u][3] = v.w;
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++) s[u][q] = (i0 + q < a.n_out) ? row[q] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (c + u < c1) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    e[q] = e[q] + w[u].x * s[u][q];
                    o[q] = o[q] + w[u].y * s[u][q];
                }
            }
        }
    }
    float *pe = a.partial + ((size_t)g * 2 + 0) * a.pstride + i0;
    float *po = a.partial + ((size_t)g * 2 + 1) * a.pstride + i0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (i0 + q < a.n_out) { pe[q] = e[q]; po[q] = o[q]; }
}

// grid (ceil(n_out/64), 2), 256 threads: the four waves each add a contiguous quarter of the groups in order for
// the same 64 samples, then the quarters are added in order
__global__ __launch_bounds__(256) void k_mix_reduce(const MixArgs a)
{
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int lr = blockIdx.y;
    const int per = (a.n_groups + 3) / 4;
    const int g0 = wv * per, g1 = min(g0 + per, a.n_groups);
    float acc = 0.f;
    if (i < a.n_out) {
        constexpr int U = 8;
        for (int g = g0; g < g1; g += U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = a.partial[((size_t)min(g + u, g1 - 1) * 2 + lr) * a.pstride + i];
#pragma unroll
            for (int u = 0; u < U; u++)
                if (g + u < g1) acc = __fadd_rn(acc, v[u]);
        }
    }
    part[wv][lane] = acc;
    __syncthreads();
    if (wv == 0 && i < a.n_out) {
        float t = part[0][lane];
        if (per < a.n_groups) t = __fadd_rn(t, part[1][lane]);
        if (2 * per < a.n_groups) t = __fadd_rn(t, part[2][lane]);
        if (3 * per < a.n_groups) t = __fadd_rn(t, part[3][lane]);
        a.mix[(size_t)lr * a.pstride + i] = t;
    }
}

__global__ __launch_bounds__(256) void k_mix_out(const MixArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_out) return;
    const float x0 = a.mix[i], x1 = a.mix[a.pstride + i];
    reinterpret_cast<float2 *>(a.out_f32)[i] = make_float2(x0, x1);
    auto cvt = [](float x) -> short {                      // sdrx.cpp:649-656
        if (x > 1.0f) return (short)32767;
        if (x < -1.0f) return (short)-32767;
        return (short)(int)__fmul_rn(x, 32767.0f);         // truncation toward zero, as the C cast
    };
    reinterpret_cast<short2 *>(a.out_s16)[i] = make_short2(cvt(x0), cvt(x1));
}

// input history -> head of the other input ring (cnt <= blockDim.x = 1024; the two rings never overlap)
__global__ void k_roll_history(const float2 *__restrict__ from, float2 *__restrict__ to, int cnt)
{
    const int i = threadIdx.x;
    if (i < cnt) to[i] = from[i];
}

}  // namespace sdrxgpu
