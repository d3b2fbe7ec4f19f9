// kernels_tailagc.hpp -- k_tail_agc: the remaining decimator stage(s) AND the IF-AGC -> |.| -> LF-AGC recurrence in ONE kernel, for
// banks whose tail is a latency chain (round 5).
//
// Reference lines computed here: the last one or two stages of MSD::decimate (src/msd.hpp:132-155, S::calculateOutput
// msd.hpp:251-335) and the per-sample lines of alsa_write_cb (src/sdrx.cpp:455-463: AGC::adjust src/agc.hpp:48-69, Demod::demod
// sdrx.cpp:178-194, LfAGC::adjust agc.hpp:106-131) -- in the reference these run back to back per sample with nothing in
// between; k_tail_s + k_agc put a launch boundary and a round trip through d_keep between them.
//
// Why: the AGC is the path's one true recurrence -- 512 dependent steps per channel and block, ~54 ns each at the instruction
// latency floor = 27.6 us whatever the bank's size -- and for banks of up to a few thousand channels the tail in
// front of it is a latency chain too (one tile round trip per workgroup: 24 us at 1 024 channels, 7 us at 64).  As two kernels
// the two chains ADD (0.052 of the 1 024-channel block's 0.130 ms).  Here a workgroup owns `cpg` channels; NT tail waves
// produce their stage-4 outputs tile by tile in TIME order (32 outputs per tile, the tile's stage-2 span prefetched in
// registers while the previous unit is computed) and hand each tile to the AGC waves through LDS, which are by then running
// the previous tile's 32 serial steps:
//   waves 0..NT-1  tail producers: unit = (channel, tile); [stage-2 span -> LDS] -> stage A -> stage B -> iq_out / iq_keep and the
//                  tile buffer xa (time-major, lane = channel for the consumer).  Wave-private LDS, no barrier inside a unit.
//   wave NT        IF AGC on tile i-1, lane = channel, publishes |y|^2 (FM: and y) through LDS
//   wave NT+1      tile i-2: envelope / FM phase step, LF AGC, squelch gating -> audio rows (16-byte stores, 4 samples per lane)
// One workgroup barrier per tile.  The chain's length stays (16 + 2 tiles x 1.73 us); what goes is the tail's own chain, the
// launch boundary and the d_keep round trip.
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
};

template <int MA, int TA, int MB, int TB>
struct TailAgcGeom {
    static constexpr bool HAS_A = MA > 0;
    static constexpr int NR = HAS_A ? MB * (kAgcTile - 1) + TB : 0;                          // stage-A outputs under a full tile
    static constexpr int NP = HAS_A ? MA * (NR - 1) + TA : MB * (kAgcTile - 1) + TB;         // stage-2 row entries under it
    static constexpr int NPAIR = (NP + 3 + 1) / 2;                                           // 16-byte pairs incl. the lead (<= 3 entries)
    static constexpr int RD = (NPAIR + kWave - 1) / kWave;                                   // LDS-DMA instructions per unit
    static constexpr int S2N = RD * kWave * 2;                                               // span entries of one buffer
    static constexpr int S3N = (NR + 1) & ~1;
    static constexpr int NBUF = 3;                                                           // the spans of units u, u+1, u+2
    static constexpr size_t wave_lds = (size_t)(NBUF * S2N + S3N) * sizeof(float2);
};

template <bool FM>
struct TailAgcLds {
    float2 xa[2][kAgcTile][kTaMaxCpg];                 // tail -> IF wave: stage-4 outputs, [sample][channel]
    float ex[2][kAgcTile][FM ? 3 : 1][kTaMaxCpg];      // IF wave -> LF wave
};

// (+ the index table: one 32-bit entry per group of y2b stage-2 outputs of the call, behind the producers' buffers)
template <int MA, int TA, int MB, int TB, bool FM, int NT>
constexpr size_t tail_agc_lds_bytes()
{
    return sizeof(TailAgcLds<FM>) + NT * TailAgcGeom<MA, TA, MB, TB>::wave_lds;
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
// s_waitcnt vmcnt(0) -- every producer wave would wait there for the spans it has just asked for (two units ahead), and the
// prefetch would be worth nothing.  What the waves hand each other goes through LDS; what they write to global memory nobody
// in this kernel reads.
// (Not even a fence of the LDS address space alone will do for the producers: their LDS-DMAs ARE pending LDS writes to the
// compiler, and a release waits for them.  So: the wave's own ds_ instructions done -- lgkmcnt(0) --, then the raw barrier; the
// "memory" clobber keeps the compiler from moving any access across.  The spans in flight are waited for where they are
// needed, by count.)
__device__ __forceinline__ void wg_barrier_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int MA, int TA, int MB, int TB, bool SQ, bool FM, int NT>
__global__ __launch_bounds__((NT + 2) * kWave) void k_tail_agc(const TailAgcArgs a)
{
    using G = TailAgcGeom<MA, TA, MB, TB>;
    constexpr bool HAS_A = G::HAS_A;
    extern __shared__ __attribute__((aligned(16))) char ta_smem[];
    TailAgcLds<FM> &L = *reinterpret_cast<TailAgcLds<FM> *>(ta_smem);
    const TailArgs &t = a.t;
    const AgcArgs &g = a.g;

    // one extra block copies the input history to the front of the other input ring for the next call (as in k_agc)
    if (g.roll_cnt > 0 && blockIdx.x == gridDim.x - 1) {
        for (int i = threadIdx.x; i < g.roll_cnt; i += (NT + 2) * kWave) g.roll_to[i] = g.roll_from[i];
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & (kWave - 1);
    const int ch0 = t.ch_lo + blockIdx.x * a.cpg;
    const int n_here = min(a.cpg, t.n_ch - ch0);                    // channels of this workgroup (>= 1)
    const int n_b = t.n_b;
    const int n_tiles = (n_b + kAgcTile - 1) / kAgcTile;

    // Where the stage-1/2 kernel left group g of this call's stage-2 outputs (body-major scratch, any layout): worked out once
    // per workgroup, by everybody, into a table in LDS.  Per DMA instruction and lane the arithmetic of y2_group_index -- three
    // divisions by float reciprocal and the layout's branches, ~100 instructions -- was two thirds of a producer wave's work.
    uint32_t *gtab = reinterpret_cast<uint32_t *>(ta_smem + tail_agc_lds_bytes<MA, TA, MB, TB, FM, NT>());
    for (int gi = threadIdx.x; gi < a.n_groups; gi += (NT + 2) * kWave) gtab[gi] = (uint32_t)y2_group_index(t, gi << t.y2b_shift);
    __syncthreads();

    if (wave < NT) {
        // ------------------------------------------------------------------------------------------------------------
        // tail producer: units (tile i, channel ch0 + wave + NT * j), tile-major
        // ------------------------------------------------------------------------------------------------------------
        float2 *smem = reinterpret_cast<float2 *>(ta_smem + sizeof(TailAgcLds<FM>) + (size_t)wave * G::wave_lds);
        const int n_mine = wave < n_here ? (n_here - wave + NT - 1) / NT : 0;
        const int Gsz = t.y2b, gsh = t.y2b_shift;
        const int n_row = t.hist + t.n_y2_new;
        const int p_last = (n_row - 1) & ~(Gsz - 1);

        // history for the next call = the last `hist` entries of the row [hist | new] (tail_s_unit's last tile does this)
        for (int j = 0; j < n_mine; j++) {
            const int ch = ch0 + wave + NT * j;
            const float2 *__restrict__ ynew = t.y2_new + (size_t)ch * t.y2_stride;
            const float2 *__restrict__ hin = t.hist_in + (size_t)ch * t.hist;
            float2 *__restrict__ hout = t.hist_out + (size_t)ch * t.hist;
            for (int i = lane; i < t.hist; i += kWave) {
                const int idx = t.n_y2_new + i;
                hout[i] = idx < t.hist ? hin[idx] : ynew[y2_storage_index(t, idx - t.hist)];
            }
        }

        // span of unit (tile i): outputs, stage-A outputs, first row entry (group-aligned) and the lead in front of the window
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
        // The span of a unit goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, nothing waits): lane l of
        // instruction r moves the 16-byte pair q = 64 r + l of the span -- row entries p0g + 2 q, + 1 -- from wherever the
        // body-major scratch (or the history) keeps it to s2[2 q], so the span lands in TIME order.  A pair past the end of
        // the row re-reads the last pair (never inside a FIR window).  (First version: 16-byte loads into registers, written
        // to LDS a unit later -- 40 registers, two of them spilled, and a spilled load is a load that is waited for at once:
        // 59 us for the kernel where the chain is 30.)
        const int p_lastpair = (n_row - 1) & ~1;
        const int pair2 = t.half_split ? 2 * t.spw : 2;
        auto issue = [&](int ch, int p0g, int buf) {
            const float2 *__restrict__ ynew = t.y2_new + (size_t)ch * t.y2_stride;
            const float2 *__restrict__ hin = t.hist_in + (size_t)ch * t.hist;
            char *dst = reinterpret_cast<char *>(smem + buf * G::S2N);
            // all source addresses first, then the DMA instructions back to back (address arithmetic between two of them made the
            // compiler wait for the first before it let the second go)
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
        // my units in order: u = i * n_mine + j  <->  (tile i, channel ch0 + wave + NT * j)
        const int n_units = n_mine * n_tiles;
        auto issue_unit = [&](int u) {
            if (u >= n_units) return;
            const int i = u / n_mine, j = u - i * n_mine;
            int nbt, nr, p0g, lead;
            span(i, nbt, nr, p0g, lead);
            issue(ch0 + wave + NT * j, p0g, u % G::NBUF);
        };
        issue_unit(0);
        issue_unit(1);
        wg_barrier_lds();
        int u = 0;
        for (int i = 0; i < n_tiles + 2; i++) {
            if (i < n_tiles) {
                for (int j = 0; j < n_mine; j++, u++) {
                    const int cl = wave + NT * j;                  // channel within the workgroup
                    const int ch = ch0 + cl;
                    int nbt, nr, p0g, lead;
                    span(i, nbt, nr, p0g, lead);
                    // unit u's span has landed: DMAs return in order, and the only VM instructions younger than them that may be
                    // outstanding are unit u+1's (stores in between are older than those and only make the wait stricter)
                    if (u + 1 < n_units) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::RD) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    wave_lds_sync();
                    const v2f *s2 = reinterpret_cast<const v2f *>(smem + (u % G::NBUF) * G::S2N) + lead;
                    v2f *s3 = reinterpret_cast<v2f *>(smem + G::NBUF * G::S2N);
                    const v2f *bsrc = s2;
                    if constexpr (HAS_A) {
#ifndef SDRX_LAB_TA_NOFIR
                        for (int r = lane; r < nr; r += kWave) s3[r] = folded_fir_ahead<TA, kMaxHalfTaps, SDRX_FIR_AHEAD_A>(s2 + MA * r, t.hA);
#endif
                        wave_lds_sync();
                        bsrc = s3;
                    }
                    if (lane < nbt) {
#ifndef SDRX_LAB_TA_NOFIR
                        const v2f r = folded_fir_ahead<TB, kMaxHalfTaps, SDRX_FIR_AHEAD_B>(bsrc + MB * lane, t.hB);
#else
                        const v2f r = bsrc[lane];
#endif
                        const float2 o = make_float2(r.x, r.y);
                        const int s0 = i * kAgcTile;
                        if (t.iq_out) t.iq_out[(size_t)ch * t.out_stride + s0 + lane] = o;
                        if (t.iq_keep) t.iq_keep[(size_t)ch * t.keep_stride + s0 + lane] = o;
                        // (as an instruction of my own: a ds_write the compiler knows of makes it wait for every LDS-DMA in flight --
                        // LDS writes that might alias --, i.e. for the span of the NEXT unit.  The barrier's lgkmcnt(0) covers it.)
                        asm volatile("ds_write_b64 %0, %1" ::"v"((uint32_t)(uintptr_t)&L.xa[i & 1][lane][cl]), "v"(r) : "memory");
                    }
                    // the buffer two units back is free (its reads are this wave's own, done): the span of unit u + 2 goes there
                    wave_lds_sync();
#ifndef SDRX_LAB_TA_NOLOAD
                    issue_unit(u + 2);
#endif
                }
            }
            wg_barrier_lds();
        }
    } else if (wave == NT) {
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
        for (int i = 0; i < n_tiles + 2; i++) {
            const int ti = i - 1;
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
        for (int i = 0; i < n_tiles + 2; i++) {
            const int ti = i - 2;
#ifndef SDRX_LAB_TA_NOAGC
            if (ti >= 0) {
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
