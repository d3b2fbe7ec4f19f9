// kernels_tailagc.hpp -- k_tail_agc: the remaining decimator stage(s) AND the IF-AGC -> |.| -> LF-AGC recurrence in ONE kernel, for
// banks whose tail is a latency chain (round 5).
//
// Reference lines computed here: the last one or two stages of MSD::decimate (src/msd.hpp:132-155, S::calculateOutput
// msd.hpp:251-335) and the per-sample lines of alsa_write_cb (src/sdrx.cpp:455-463: AGC::adjust src/agc.hpp:48-69, Demod::demod
// sdrx.cpp:178-194, LfAGC::adjust agc.hpp:106-131) -- in the reference these run back to back per sample with nothing in
// between; k_tail_s + k_agc put a launch boundary and a round trip through d_keep between them.
//
// Why: the AGC is the path's one true recurrence -- 512 dependent steps per channel and block at the instruction latency floor
// (DESIGN 4.6): 27.6 us as k_agc, whatever the bank's size -- and for banks of up to a thousand channels the tail in front of it is
// a latency chain too (one tile round trip per workgroup: 24 us at 1 024 channels, 7 us at 64).  As two kernels the two chains ADD.
// Here a workgroup owns NC channels, and per channel two producer waves hand tiles of 32 outputs, in TIME order, through LDS to
// the AGC waves, which are by then running the previous tiles' serial steps (the iteration scheme is at the kernel).  The chain's
// length stays; what goes is the tail's own chain, the launch boundary and the d_keep round trip.
// EVERY value is computed by the same functions on the same operands as in k_tail_s (tail_s_unit: folded_fir_ahead in the
// reference's order) and k_agc (if_agc_step / demod_lf_step): outputs are bit-identical to the two-kernel sequence (tested,
// tests/test_gpu_tailagc.py), so which of the two a bank runs is a geometry choice like any other of sdrxgpu_tuning.
#pragma once
#include "kernels.hpp"

namespace sdrxgpu {

constexpr int kTaMaxCpg = 16;          // channels per workgroup (lanes of the AGC waves in use), at most

struct TailAgcArgs {
    TailArgs t;                        // as for k_tail_s; tile_b / n_tiles are not used (tiles of kAgcTile outputs)
    AgcArgs g;                         // as for k_agc; iq / iq_stride are not used (the tiles come through LDS)
    int cpg;                           // channels per workgroup, 1 .. kTaMaxCpg
    int n_groups;                      // groups of y2b stage-2 outputs this call has: entries of the index table in LDS
    const uint32_t *gtab;              // [>= n_groups] where group g of a call's stage-2 outputs lies in a channel's scratch row (the
                                       //   bank's geometry: built once on the host at bank_create, copied to LDS per workgroup)
};

template <int MA, int TA, int MB, int TB>
struct TailAgcGeom {
    static constexpr bool HAS_A = MA > 0;
    static constexpr int NR = HAS_A ? MB * (kAgcTile - 1) + TB : 0;                          // stage-A outputs under a full tile
    static constexpr int NP = HAS_A ? MA * (NR - 1) + TA : MB * (kAgcTile - 1) + TB;         // stage-2 row entries under it
    static constexpr int NPAIR = (NP + 3 + 1) / 2;                                           // 16-byte pairs incl. the lead (<= 3 entries)
    static constexpr int RD = (NPAIR + kWave - 1) / kWave;                                   // LDS-DMA instructions per tile
    static constexpr int S2N = RD * kWave * 2;                                               // span entries of one buffer
    static constexpr int S3N = HAS_A ? ((NR + 1) & ~1) : 0;
    static constexpr int NBUF = HAS_A ? 3 : 4;      // span ring: tiles k (read by the A-wave), k+1, k+2 -- and k-1 where the B-wave reads spans itself
    static constexpr size_t chan_lds = (size_t)(NBUF * S2N + 2 * S3N) * sizeof(float2);     // per channel: span ring + two stage-A buffers
    // The A-wave computes three neighbouring stage-A outputs per lane from one window of TA + 2 MA entries; for the last lane of a
    // tile the second and third are past the tile's end (computed, not stored) and their entries can lie up to 2 MA + 3 entries past
    // the span proper (lead <= 3 + NP) -- for the (5,19,5,45) tail 3 entries past the BUFFER (S2N = 1 024, NP = 1 014).  What they
    // then read is the head of the same channel's next ring buffer or stage-A buffer: inside the channel's own block, whatever is
    // there.  A change of RD / NBUF that would let that read leave the block is stopped here.
    static_assert(!HAS_A || NP + 2 * MA + 3 <= S2N + 2 * S3N, "k_tail_agc: the three-output window's over-read must stay inside the channel's LDS block");
};

template <bool FM, int NC>
struct TailAgcLds {
    float2 xa[2][kAgcTile][NC];                        // B-wave -> IF wave: stage-4 outputs, [sample][channel]
    float ex[2][kAgcTile][FM ? 3 : 1][NC];             // IF wave -> LF wave
};

// (+ the index table: one 32-bit entry per group of y2b stage-2 outputs of the call, behind the channels' buffers)
template <int MA, int TA, int MB, int TB, bool FM, int NC>
constexpr size_t tail_agc_lds_bytes()
{
    return sizeof(TailAgcLds<FM, NC>) + NC * TailAgcGeom<MA, TA, MB, TB>::chan_lds;
}

// within ONE wave: what its lanes wrote to LDS is visible to its other lanes (LDS serves a wave's instructions in order); this
// only keeps the compiler from moving accesses across it
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The workgroup's barrier per tile, ordering LDS only: __syncthreads() is a release + acquire of GLOBAL memory as well, i.e.
// s_waitcnt vmcnt(0) -- a wave would wait there for the spans it has just asked for (two tiles ahead), and the prefetch would be
// worth nothing.  Not even a fence of the LDS address space alone will do: LDS-DMAs ARE pending LDS writes to the compiler, and
// a release waits for them.  So: the wave's own ds_ instructions done -- lgkmcnt(0) --, then the raw barrier; the "memory"
// clobber keeps the compiler from moving any access across.  The spans in flight are waited for where they are needed, by count.
// (What the waves hand each other goes through LDS; what they write to global memory nobody in this kernel reads.)
__device__ __forceinline__ void wg_barrier_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NC channels per workgroup; waves: [0, NC) A-waves (stage A of their channel), [NC, 2 NC) B-waves (LDS-DMA of the spans, stage B,
// outputs), 2 NC the IF-AGC wave, 2 NC + 1 the LF wave.  Iteration k (one workgroup barrier each):
//   B-wave   asks for the span of tile k+2 (LDS-DMA into ring buffer (k+2) % 3 -- free since the A-wave read tile k-1 there an
//            iteration ago), runs stage B of tile k-1 from the stage-A buffer (k-1) & 1 -> iq_out / iq_keep / xa[(k-1) & 1], then
//            waits until the span of tile k+1 (asked for an iteration ago) has landed: the A-wave reads it after the barrier
//   A-wave   stage A of tile k: span ring k % 3 -> stage-A buffer k & 1
//   IF wave  tile k-2;  LF wave  tile k-3
// A-waves of a workgroup of NC channels: one per channel, or -- four channels -- one per two (stage A is a pass of 54 lanes per tile
// and channel since its second version; eight waves on the CU's four SIMDs instead of ten)
constexpr int tail_agc_a_waves(int nc) { return nc >= 4 ? nc / 2 : nc; }
constexpr int tail_agc_waves(int nc) { return tail_agc_a_waves(nc) + nc + 2; }

template <int MA, int TA, int MB, int TB, bool SQ, bool FM, int NC>
__global__ __launch_bounds__(tail_agc_waves(NC) * kWave) void k_tail_agc(const TailAgcArgs a)
{
    using G = TailAgcGeom<MA, TA, MB, TB>;
    constexpr bool HAS_A = G::HAS_A;
    constexpr int NA = tail_agc_a_waves(NC);
    constexpr int NTHREADS = tail_agc_waves(NC) * kWave;
    extern __shared__ __attribute__((aligned(16))) char ta_smem[];
    TailAgcLds<FM, NC> &L = *reinterpret_cast<TailAgcLds<FM, NC> *>(ta_smem);
    const TailArgs &t = a.t;
    const AgcArgs &g = a.g;

    // one extra block copies the input history to the front of the other input ring for the next call (as in k_agc)
    if (g.roll_cnt > 0 && blockIdx.x == gridDim.x - 1) {
        for (int i = threadIdx.x; i < g.roll_cnt; i += NTHREADS) g.roll_to[i] = g.roll_from[i];
        return;
    }
    // (Waves w and w + 4 of a workgroup share a SIMD -- probed: s_getreg HW_ID.  Giving the IF and the LF wave a SIMD of their own, the
    // producers the other three (4 channels per workgroup), measured SLOWER, 43.4 against 37.8 us at 1 024 channels; IF and LF each
    // alone on a SIMD (2 channels per workgroup) 32.9 against 32.5: the recurrence is not held up by its neighbours' instructions, it
    // is the pipeline's pace itself -- 19 iterations of a 32-step chain.)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & (kWave - 1);
    const int ch0 = t.ch_lo + blockIdx.x * NC;
    const int n_here = min(NC, t.n_ch - ch0);                       // channels of this workgroup (>= 1)
    const int n_b = t.n_b;
    const int n_tiles = (n_b + kAgcTile - 1) / kAgcTile;
    const int n_iter = n_tiles + 3;

    // Where the stage-1/2 kernel left group g of a call's stage-2 outputs in a channel's scratch row (body-major, any layout): the
    // bank's geometry only, so the host builds the table once at bank_create and a workgroup copies it into LDS.  (Worked out per
    // DMA instruction and lane -- y2_group_index: three divisions by float reciprocal and the layout's branches, ~100 instructions --
    // it was two thirds of a producer wave's work; worked out per workgroup it cost 4.5 us of start-up.)
    uint32_t *gtab = reinterpret_cast<uint32_t *>(ta_smem + tail_agc_lds_bytes<MA, TA, MB, TB, FM, NC>());
    for (int gi = threadIdx.x * 4; gi < a.n_groups; gi += NTHREADS * 4)                  // (the table is padded to whole 16-byte pieces)
        *reinterpret_cast<uint4 *>(gtab + gi) = *reinterpret_cast<const uint4 *>(a.gtab + gi);
    __syncthreads();

    const int Gsz = t.y2b, gsh = t.y2b_shift;
    // span of tile i: outputs, stage-A outputs, first row entry (group-aligned) and the lead in front of the window
    auto span = [&](int i, int &nbt, int &nr, int &p0g, int &lead) {
        const int s0 = i * kAgcTile;
        nbt = min(n_b - s0, kAgcTile);
        int p0;
        if constexpr (HAS_A) {
            const int r0 = t.pB0 + MB * s0;
            nr = MB * (nbt - 1) + TB;
            p0 = t.pA0 + MA * r0;
        } else {
            nr = 0;
            p0 = t.pB0 + MB * s0;
        }
        p0g = p0 & ~(Gsz - 1);
        lead = p0 - p0g;
    };

    if (wave < NA + NC) {
        const bool is_b = wave >= NA;
        const int cl = is_b ? wave - NA : wave;                     // channel within the workgroup (an A-wave's first one)
        const bool live = cl < n_here;
        const int ch = ch0 + min(cl, n_here - 1);
        float2 *cbase = reinterpret_cast<float2 *>(ta_smem + sizeof(TailAgcLds<FM, NC>) + (size_t)cl * G::chan_lds);
        float2 *s3base = cbase + G::NBUF * G::S2N;
        if (!is_b) {
            // --------------------------------------------------------------------------------------------------------
            // A-wave: stage A of tile k in iteration k, for the channels wave, wave + NA, ... of the workgroup
            // --------------------------------------------------------------------------------------------------------
            wg_barrier_lds();
            for (int k = 0; k < n_iter; k++) {
                if constexpr (HAS_A) {
#pragma unroll
                    for (int ca = wave; ca < NC; ca += NA)
                    if (ca < n_here && k < n_tiles) {
                        float2 *cbase = reinterpret_cast<float2 *>(ta_smem + sizeof(TailAgcLds<FM, NC>) + (size_t)ca * G::chan_lds);
                        float2 *s3base = cbase + G::NBUF * G::S2N;
                        int nbt, nr, p0g, lead;
                        span(k, nbt, nr, p0g, lead);
                        const v2f *s2 = reinterpret_cast<const v2f *>(cbase + (k % G::NBUF) * G::S2N) + lead;
                        v2f *s3 = reinterpret_cast<v2f *>(s3base + (k & 1) * G::S3N);
#if !defined(SDRX_LAB_TA_NOFIR) && !defined(SDRX_LAB_TA_NOFIR_A)
                        // The first TB - MB stage-A outputs of tile k are the last ones of tile k-1 (stage B's window reaches back over
                        // them): copied from the other buffer, not computed again (28 % of a tile's stage-A outputs).  The new ones:
                        // three per lane, one pass per full tile.  Per output folded_fir_unrolled's arithmetic on the same operands:
                        // the same bits.
                        constexpr int CARRY = TB - MB;
                        int r_first = 0;
                        if (k > 0) {
                            const v2f *s3p = reinterpret_cast<const v2f *>(s3base + ((k - 1) & 1) * G::S3N) + MB * kAgcTile;
                            for (int i = lane; i < CARRY; i += kWave) s3[i] = s3p[i];
                            r_first = CARRY;
                        }
                        for (int rb = r_first + 3 * lane; rb < nr; rb += 3 * kWave) {
                            // three NEIGHBOURING outputs per lane: their windows overlap, TA + 2 MA entries are read where three separate
                            // windows take 3 TA (this kernel's waves run out of LDS cycles before anything else: four channels' stage A
                            // and B per CU), and three independent accumulation chains (fma_tap3).  An output past the end works on
                            // entries inside the channel's LDS block (TailAgcGeom's static_assert) and is not stored.
                            v2f win[TA + 2 * MA];
                            const v2f *w = s2 + MA * rb;
#pragma unroll
                            for (int j = 0; j < TA + 2 * MA; j++) win[j] = w[j];
                            v2f o0 = {0.f, 0.f}, o1 = {0.f, 0.f}, o2 = {0.f, 0.f};
                            folded_fir3_steps<TA>(o0, o1, o2, win, win + MA, win + 2 * MA, t.hA);
                            s3[rb] = o0;
                            if (rb + 1 < nr) s3[rb + 1] = o1;
                            if (rb + 2 < nr) s3[rb + 2] = o2;
                        }
#endif
                    }
                }
                wg_barrier_lds();
            }
        } else {
            // --------------------------------------------------------------------------------------------------------
            // B-wave: spans in (LDS-DMA), stage B of tile k-1 in iteration k, outputs, the tile for the AGC waves
            // --------------------------------------------------------------------------------------------------------
            const float2 *__restrict__ ynew = t.y2_new + (size_t)ch * t.y2_stride;
            const float2 *__restrict__ hin = t.hist_in + (size_t)ch * t.hist;
            const int n_row = t.hist + t.n_y2_new;
            const int p_lastpair = (n_row - 1) & ~1;
            const int pair2 = t.half_split ? 2 * t.spw : 2;
            // The span of a tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, nothing waits): lane l of
            // instruction r moves the 16-byte pair q = 64 r + l of the span -- row entries p0g + 2 q, + 1 -- from wherever the
            // body-major scratch (or the history) keeps it to s2[2 q], so the span lands in TIME order.  A pair past the end of
            // the row re-reads the last pair (never inside a FIR window).  All source addresses first, then the DMA
            // instructions back to back (address arithmetic between two of them made the compiler wait for the first before it
            // let the second go).
            auto issue_tile = [&](int i) {
                if (!live || i >= n_tiles) return;
                int nbt, nr, p0g, lead;
                span(i, nbt, nr, p0g, lead);
                char *dst = reinterpret_cast<char *>(cbase + (i % G::NBUF) * G::S2N);
                // A span that lies wholly among the call's new outputs and inside the row (every tile but the first one or two and the
                // last) needs none of the clamps and selects below: pair 64 r + l is group (first + 2 l + 128 r) >> gsh of the table,
                // its half of the group the same for every r -- one table read and one add per DMA instruction (the general form below
                // was 1 060 of a B-wave's 2 550 cycles per tile).
                if (p0g >= t.hist && p0g + 2 * (G::RD * kWave - 1) <= p_lastpair) {
                    const int j0 = p0g - t.hist + 2 * lane;
                    const uint32_t off2 = (Gsz == 4 && (j0 & 2)) ? (uint32_t)pair2 : 0u;
                    const uint32_t *gp = gtab + (j0 >> gsh);
                    const int gstep = (2 * kWave) >> gsh;
                    uint32_t e[G::RD];
#pragma unroll
                    for (int r = 0; r < G::RD; r++) e[r] = (gp[r * gstep] + off2) * (uint32_t)sizeof(float2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < G::RD; r++)
                        __builtin_amdgcn_global_load_lds((gbl_void *)(uintptr_t)(reinterpret_cast<const char *>(ynew) + e[r]),
                                                         (lds_void *)(uint32_t)(uintptr_t)(dst + r * kWave * 16), 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    return;
                }
                uintptr_t src[G::RD];
#pragma unroll
                for (int r = 0; r < G::RD; r++) {
                    const int p = min(p0g + 2 * (lane + kWave * r), p_lastpair);
                    const int j = max(p - t.hist, 0);
                    // (the group's second pair: the next two entries, or -- half_split -- one half-body row further)
                    const int off2 = (Gsz == 4 && (j & 2)) ? pair2 : 0;
                    const int e_new = (int)gtab[j >> gsh] + off2;
                    const bool in_hist = p < t.hist;
                    src[r] = (uintptr_t)((in_hist ? hin : ynew) + (in_hist ? p : e_new));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < G::RD; r++)
                    __builtin_amdgcn_global_load_lds((gbl_void *)src[r], (lds_void *)(uint32_t)(uintptr_t)(dst + r * kWave * 16), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            const int n_st = (t.iq_out ? 1 : 0) + (t.iq_keep ? 1 : 0);      // stores a tile's stage B issues
            issue_tile(0);
            issue_tile(1);
            // (tile 0's span: everything but tile 1's DMAs has landed)
            if (live && n_tiles > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::RD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wg_barrier_lds();
            for (int k = 0; k < n_iter; k++) {
                issue_tile(k + 2);
                const int tb = k - 1;
                const bool has_b = live && tb >= 0 && tb < n_tiles;
                if (has_b) {
                    int nbt, nr, p0g, lead;
                    span(tb, nbt, nr, p0g, lead);
                    const v2f *bsrc = HAS_A ? reinterpret_cast<const v2f *>(s3base + (tb & 1) * G::S3N)
                                            : reinterpret_cast<const v2f *>(cbase + (tb % G::NBUF) * G::S2N) + lead;
                    if (lane < nbt) {
#if !defined(SDRX_LAB_TA_NOFIR) && !defined(SDRX_LAB_TA_NOFIR_B)
                        const v2f r = folded_fir_ahead<TB, kMaxHalfTaps, SDRX_FIR_AHEAD_B>(bsrc + MB * lane, t.hB);
#else
                        const v2f r = bsrc[lane];
#endif
                        const float2 o = make_float2(r.x, r.y);
                        const int s0 = tb * kAgcTile;
                        if (t.iq_out) t.iq_out[(size_t)ch * t.out_stride + s0 + lane] = o;
                        if (t.iq_keep) t.iq_keep[(size_t)ch * t.keep_stride + s0 + lane] = o;
                        // (as an instruction of my own: a ds_write the compiler knows of makes it wait for every LDS-DMA in flight --
                        // LDS writes that might alias.  The barrier's lgkmcnt(0) covers it.)
                        asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)&L.xa[tb & 1][lane][cl]), "v"(r) : "memory");
                    }
                }
                // The A-wave reads the span of tile k+1 after this barrier: asked for an iteration ago.  DMAs return in order; younger
                // than that span's are only this iteration's: the DMAs of tile k+2 (if it exists) and stage B's stores.
                {
                    const int young = ((live && k + 2 < n_tiles) ? G::RD : 0) + (has_b ? n_st : 0);
                    if (young >= G::RD + 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::RD + 2) : "memory");
                    else if (young == G::RD + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::RD + 1) : "memory");
                    else if (young == G::RD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::RD) : "memory");
                    else if (young == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else if (young == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                wg_barrier_lds();
            }
            // (last, off the tiles' critical path:) history for the next call = the last `hist` entries of the row [hist | new]
            if (live) {
                float2 *__restrict__ hout = t.hist_out + (size_t)ch * t.hist;
                for (int i = lane; i < t.hist; i += kWave) {
                    const int idx = t.n_y2_new + i;
                    const int j = max(idx - t.hist, 0), q = j & (Gsz - 1);
                    const int e = (int)gtab[j >> gsh] + ((Gsz == 4 && (q & 2)) ? pair2 : 0) + (q & 1);
                    hout[i] = idx < t.hist ? hin[idx] : ynew[e];
                }
            }
        }
    } else if (wave == NA + NC) {
        // ------------------------------------------------------------------------------------------------------------
        // IF AGC (AGC::adjust), one tile behind the producers; lane = channel
        // ------------------------------------------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(3);
        const int cl = min(lane, n_here - 1);
        const bool ok = lane < n_here;
        const int ch = ch0 + cl;
        const float4 s4 = g.state[ch];
        float gain = s4.x, pw = s4.y;
        wg_barrier_lds();
        auto if_tile = [&](int ti, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            float2 x[kAgcTile];
#pragma unroll
            for (int k = 0; k < kAgcTile; k++) x[k] = L.xa[ti & 1][k][cl];
#pragma unroll
            for (int k = 0; k < kAgcTile; k++) {
                if (FULL || ti * kAgcTile + k < n_b) {
                    float yr, yi, nrm;
                    if_agc_step(x[k], gain, pw, g.agc, yr, yi, nrm);
                    L.ex[ti & 1][k][0][cl] = nrm;
                    if constexpr (FM) {
                        L.ex[ti & 1][k][1][cl] = yr;
                        L.ex[ti & 1][k][2][cl] = yi;
                    }
                }
            }
        };
        for (int i = 0; i < n_iter; i++) {
            const int ti = i - 2;
#ifndef SDRX_LAB_TA_NOAGC
            if (ti >= 0 && ti < n_tiles) {
                if ((ti + 1) * kAgcTile <= n_b) if_tile(ti, std::true_type{});
                else if_tile(ti, std::false_type{});
            }
#endif
            wg_barrier_lds();
        }
        if (ok) reinterpret_cast<float2 *>(g.state + ch)[0] = make_float2(gain, pw);
    } else {
        // ------------------------------------------------------------------------------------------------------------
        // envelope / FM phase step -> LF AGC -> squelch gating -> audio rows, two tiles behind the producers
        // ------------------------------------------------------------------------------------------------------------
        __builtin_amdgcn_s_setprio(3);
        const int cl = min(lane, n_here - 1);
        const bool ok = lane < n_here;
        const int ch = ch0 + cl;
        const float4 s4 = g.state[ch];
        float gl = s4.z, pl = s4.w;
        bool is_fm = false;
        float2 fmp = make_float2(0.f, 0.f);
        if constexpr (FM) {
            is_fm = g.fm[ch] != 0;
            fmp = g.fm_prev[ch];
        }
        int mode = 1;
        const float *__restrict__ ramp = g.ramp_down;
        if constexpr (SQ) {
            const uchar2 q = g.sql[ch];
            mode = q.x ? (q.y ? 1 : 2) : (q.y ? 3 : 0);
            ramp = (mode == 2) ? g.ramp_up : g.ramp_down;
        }
        const bool ramped = SQ && mode >= 2;
        float *__restrict__ arow = g.audio + (size_t)ch * g.out_stride;
        wg_barrier_lds();
        auto lf_tile = [&](int ti, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            float nrm[kAgcTile];
#pragma unroll
            for (int k = 0; k < kAgcTile; k++) nrm[k] = L.ex[ti & 1][k][0][cl];
#pragma unroll
            for (int k4 = 0; k4 < kAgcTile / 4; k4++) {
                float y[4];
                float4 r4 = make_float4(1.f, 1.f, 1.f, 1.f);
                if constexpr (SQ) {
                    if (ramped) r4 = reinterpret_cast<const float4 *>(ramp)[((ti * kAgcTile + 4 * k4) & 511) / 4];
                }
                const float r[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
                for (int h = 0; h < 4; h++) {
                    const int k = 4 * k4 + h;
                    y[h] = 0.f;
                    if (FULL || ti * kAgcTile + k < n_b) {
                        float yr = 0.f, yi = 0.f;
                        if constexpr (FM) {
                            yr = L.ex[ti & 1][k][1][cl];
                            yi = L.ex[ti & 1][k][2][cl];
                        }
                        y[h] = demod_lf_step<SQ, FM>(yr, yi, nrm[k], gl, pl, g.lf, g.lf_active, mode, r[h], is_fm, fmp);
                    }
                }
                if (ok) {
                    const int s0 = ti * kAgcTile + 4 * k4;
                    if (g.audio_vec && (FULL || s0 + 4 <= n_b)) {
                        *reinterpret_cast<float4 *>(arow + s0) = make_float4(y[0], y[1], y[2], y[3]);
                    } else {
#pragma unroll
                        for (int h = 0; h < 4; h++)
                            if (FULL || s0 + h < n_b) arow[s0 + h] = y[h];
                    }
                }
            }
        };
        for (int i = 0; i < n_iter; i++) {
            const int ti = i - 3;
#ifndef SDRX_LAB_TA_NOAGC
            if (ti >= 0 && ti < n_tiles) {
                if ((ti + 1) * kAgcTile <= n_b) lf_tile(ti, std::true_type{});
                else lf_tile(ti, std::false_type{});
            }
#endif
            wg_barrier_lds();
        }
        if (ok) {
            reinterpret_cast<float2 *>(g.state + ch)[1] = make_float2(gl, pl);
            if constexpr (FM) g.fm_prev[ch] = fmp;
        }
    }
}

}  // namespace sdrxgpu
