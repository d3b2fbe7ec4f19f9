// kernels_mfma6.hpp -- k_stage12m6: the NCO mixer and decimator stage 1 of the 6 MS/s chain ((15, 53)(5, 21), sdrx.cpp:1651-1658) on the
// f32 matrix pipe, stage 2 in transposed form on the vector ALU (round 5).  Same job, inputs, outputs and scratch
// layout as k_stage12<15, 53, 5, 21, 2, ...> (kernels.hpp); reference: MSD::decimate mixer path src/msd.hpp:132-155, S::addSample /
// calculateOutput msd.hpp:228-335, translator table sdrx.cpp:1682-1690.
//
// The formulation is k_stage12m's (kernels_mfma.hpp): for 8 conjugate channel pairs ("slots") x 16 time segments that start a multiple
// of N samples apart, stage-1 output k of all of them is a dense product of A[slot, part][tap] = (re, im) of t_c[..] (the caller's
// table as it is) with B[tap][segment] = h1[tap] x[..]; the four real sums P = Cr x, Q = Ci x give both channels of a pair,
// y+ = (Pr - Qi) + j (Pi + Qr), y- = (Pr + Qi) + j (Pi - Qr).  v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain.  What differs is the
// shape: 53 taps are 13 K = 4 steps + tap 52 (which starts the accumulators), a phase advances 15 samples, and a half-body row would
// be 150 + 38 samples -- k_stage12m's (2, 2) workgroup would need 195 KB of LDS.  So here:
//   * a wave is ONE MFMA tile pair (8 slots x 16 segments; 4 channel-segments per lane), a workgroup 4 such slot half-groups x 2
//     segment groups = 8 waves that share 32 input rows and 32 translator rows;
//   * rows cover a CHUNK of 75 samples = 5 phases = one stage-2 output: 75 + 38 samples of window + 1 so that every row starts on an
//     even sample (16-byte LDS-DMA sources: odd chunks start one sample early and read one entry further in) = 114 samples = 57
//     pieces of 16 bytes, no padding: the 64 rows of a tile are 57 whole DMA instructions, two tile buffers 117 KB, one workgroup per CU;
//   * the K steps run outside-in (0, 12, 1, 11, ...: the small outer taps first, as in the reference's folded FIR, msd.hpp:309-331);
//   * stage 2 (21 taps, / 5) transposed: a stage-1 output is multiplied into the at most five stage-2 outputs whose windows hold it;
//     five accumulators per channel-segment, shifted down by one when the oldest completes (one completion per chunk, so the chunk
//     loop needs no unrolling over bodies).  Each output is one fma chain over its 21 inputs in time order, whatever body, lane or
//     call it falls in: any cut of the stream gives the same bits;
//   * outputs leave per body (2 outputs = 16 bytes per channel-segment) in k_stage12's body-major layout [time-wave][body][segment][2]:
//     256 contiguous bytes per channel and store, the tail kernels read them as they read the vector kernel's (TailArgs.half_split = 0).
// A channel without partner runs as a pair whose second half is dropped; an empty slot works on channel 0's table and stores into the
// spare row: a channel's bits never depend on who else is in the bank.  Not bit-identical to the vector kernels (another summation
// order); held to the golden vectors of the compiled reference and to the oracle (tests/test_gpu_matrix6.py).
#pragma once
#include "kernels.hpp"

namespace sdrxgpu {

template <int GS, int GT>
struct TileGeomM6 {
    static constexpr int M1 = 15, T1 = 53, M2 = 5, T2 = 21, Y2B = 2;
    static constexpr int CH = 75;                       // samples per chunk = 5 phases = one stage-2 output
    static constexpr int PH = 5;
    static constexpr int HALO = T1 - M1;                // 38
    static constexpr int ROW = CH + HALO + 1;           // 114 samples: window + the even-start sample
    static constexpr int row_bytes = ROW * 8;           // 912 = 57 pieces of 16 bytes; == 16 (mod 32)
    static constexpr int pieces = row_bytes / 16;
    static constexpr int x_rows = GT * 16, t_rows = GS * 16;
    static constexpr int rows = x_rows + t_rows;
    static constexpr int buf_bytes = rows * row_bytes;
    static constexpr int waves = 2 * GS * GT;           // half-tile waves
    static constexpr int n_dma = rows * pieces / kWave; // DMA instructions per chunk and workgroup
    static constexpr int dma_per_wave = (n_dma + waves - 1) / waves;
    static_assert(rows * pieces % kWave == 0, "a tile is whole DMA instructions");
};

struct Stage12m6Taps {
    float h1[53];
    float h2[11];       // h2[0 .. 10]; h2[20 - e] = h2[e]
};

template <int GS, int GT>
__global__ __launch_bounds__(64 * 2 * GS * GT, 1)
void k_stage12m6(const float2 *__restrict__ x_in, const float2 *__restrict__ tables, float2 *__restrict__ y2_out,
                 const Stage12Args a, const PairArgs pa, const Stage12m6Taps taps)
{
    using G = TileGeomM6<GS, GT>;
    constexpr int SB = G::M1 * G::M2 * G::Y2B;          // 150 samples per body = 2 chunks
    constexpr int SPW = 16;
    constexpr int RB = G::row_bytes;
    extern __shared__ __attribute__((aligned(16))) char lds_m6[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int hg = wave / GT, tg = wave - hg * GT;          // this wave's slot half-group (8 slots) / segment group inside the workgroup

    // item -> (slot-group set, time-wave): k_stage12m's map (whole sets to an XCD eight at a time, the last n_sets % 8 by units)
    const int n_sets = a.n_chgroups - a.cg0;
    const int n_full = (n_sets >> 3) * a.n_wt;
    const int n_rem = (n_sets & 7) * a.n_wt;
    const int n_items = a.n_cgx * 8;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int xcd = item & 7;
    const int kk = item >> 3;
    int wtm, cgm;
    if (kk < n_full) {
        wtm = kk % a.n_wt;
        cgm = a.cg0 + (kk / a.n_wt) * 8 + xcd;
    } else {
        const int w = ((xcd * n_rem) >> 3) + (kk - n_full);
        if (w >= (((xcd + 1) * n_rem) >> 3)) continue;
        wtm = w % a.n_wt;
        cgm = a.cg0 + (n_sets & ~7) + w / a.n_wt;
    }
    const int gq = wtm / a.P, p = wtm - gq * a.P;
    const int bps = a.nb * a.P;
    const int g_first = gq * GT;
    if ((g_first * SPW * bps + a.nb * p) * G::Y2B >= a.n_y2_new) continue;

    // ---- what this lane stores: slots 2 hi + e of the wave's half-group, both channels of each ----
    const int slot_base = cgm * GS * 16 + hg * 8;
    uint32_t yrow[4];                                       // [2 e + (0: channel A, 1: channel B)]
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int s = slot_base + 2 * hi + e;
        const int2 sl = s < pa.n_slots ? pa.slots[s] : make_int2(-1, -1);
        const int ca = sl.x >= 0 ? sl.x : a.n_ch, cb = (sl.x >= 0 && sl.y >= 0) ? sl.y : a.n_ch;
        yrow[2 * e] = (uint32_t)ca * (uint32_t)a.y2_stride;
        yrow[2 * e + 1] = (uint32_t)cb * (uint32_t)a.y2_stride;
    }
    const int g_mine = g_first + tg;
    const int wt = g_mine * a.P + p;                        // time-wave in k_stage12's numbering (scratch layout)
    const bool seg_live = (g_mine * SPW * bps + a.nb * p) * G::Y2B < a.n_y2_new;

    // ---- LDS-DMA duties: instruction n = wave + waves * i moves the tile's pieces 64 n .. 64 n + 63 (row-major: row = piece / 57) ----
    const int seg_max = ((a.n_y2_new + G::Y2B - 1) / G::Y2B + SPW * bps - 1) / (SPW * bps) * SPW - 1;
    uint32_t dma_off[G::dma_per_wave];
    bool dma_is_t[G::dma_per_wave];
#pragma unroll
    for (int i = 0; i < G::dma_per_wave; i++) {
        const int n = wave + i * G::waves;
        const int gpc = min(n, G::n_dma - 1) * kWave + lane;          // piece of the tile
        const int row = gpc / G::pieces, pc = gpc - row * G::pieces;
        dma_is_t[i] = row >= G::x_rows;
        if (row < G::x_rows) {
            const int seg = min((g_first + row / 16) * SPW + (row & 15), seg_max);
            dma_off[i] = (uint32_t)(((long long)seg * bps + a.nb * p) * SB * 8) + (uint32_t)pc * 16u;
        } else {
            const int s = cgm * GS * 16 + (row - G::x_rows);
            int c = s < pa.n_slots ? pa.slots[s].x : -1;
            if (c < 0) c = 0;                               // an empty slot works on channel 0's table and stores nothing
            dma_off[i] = (uint32_t)((size_t)c * a.tbl_stride * 8) + (uint32_t)pc * 16u;
        }
    }
    const char *xsrc0 = reinterpret_cast<const char *>(x_in + a.x_base);
    const char *tsrc0 = reinterpret_cast<const char *>(tables);
    // chunk c of the lane's run starts 75 c samples in; its rows start HALO (+ 1 for odd c) samples in front of it
    auto issue_chunk = [&](int c, int bi) {
        const int rel = G::CH * c - G::HALO - (c & 1);      // even; >= -38
        int tph = (a.tphase0 + p * a.nb * SB + rel) % a.N;
        if (tph < 0) tph += a.N;
        char *buf = lds_m6 + bi * G::buf_bytes;
#pragma unroll
        for (int i = 0; i < G::dma_per_wave; i++) {
            const int n = wave + i * G::waves;
            if (n < G::n_dma) {
                const char *base = dma_is_t[i] ? tsrc0 + (size_t)tph * 8 : xsrc0 + (long long)rel * 8;
                __builtin_amdgcn_global_load_lds((gbl_void *)(uintptr_t)(base + dma_off[i]), (lds_void *)(uint32_t)(uintptr_t)(buf + n * kWave * 16), 16, 0, 0);
            }
        }
    };

    // ---- operand addresses (byte offsets inside a buffer) ----
    // MFMA rows = 8 slots x {re, im of the coefficient}: row 2 s + c.  A: part lo & 1 of t[slot lo >> 1][tap 4 j + hi]; B: x[segment lo][same];
    // C/D row 4 hi + r = slot 2 hi + (r >> 1), part r & 1; column = segment lo.
    const int aoff = (G::x_rows + hg * 8 + (lo >> 1)) * RB + hi * 8 + (lo & 1) * 4;
    const int boff = (tg * 16 + lo) * RB + hi * 8;
    const int doff = (G::x_rows + hg * 8 + 2 * hi) * RB;    // tap 52: t[slot 2 hi + e][52]
    const int eoff = (tg * 16 + lo) * RB;                   //         x[segment lo][52]
    float hA[13];                                           // taps 4 j + hi
#pragma unroll
    for (int j = 0; j < 13; j++) hA[j] = taps.h1[4 * j + hi];
    const float h52 = taps.h1[52];

    struct Ops { float ta[13]; v2f x[13]; v2f x52, t52a, t52b; };        // a phase's operands: A [j], B [j], tap 52
    typedef const volatile __attribute__((address_space(3))) v2f lds_cv2f;
    typedef const volatile __attribute__((address_space(3))) float lds_cvf;
    typedef const __attribute__((address_space(3))) char lds_cc;
    lds_cc *const lbase = (lds_cc *)(uint32_t)(uintptr_t)lds_m6;

    // stage-2 accumulators: acc2[d][i] = output (current + d) of channel-segment i = 2 e + A/B
    v2f acc2[5][4];
#pragma unroll
    for (int d = 0; d < 5; d++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc2[d][i] = v2f{0.f, 0.f};
    v4f stage[4];                                           // per channel: the body's two outputs
#pragma unroll
    for (int i = 0; i < 4; i++) stage[i] = v4f{0.f, 0.f, 0.f, 0.f};
    const uint32_t ylane = (uint32_t)a.y2_hist + ((uint32_t)wt * a.nb * SPW + lo) * G::Y2B;

    // warm-up: stage 2's first real output has its window start M1 (T2 - M2) + (T1 - M1) = 278 samples in front of the lane's
    // range = two bodies of 150 (k_stage12's WU for this chain)
    constexpr int WU = 2;
    const int n_chunks = 2 * (WU + a.nb);
    issue_chunk(0, 0);
    bool stored = false;                                    // the previous chunk ended with the body's four stores
    for (int c = 0; c < n_chunks; c++) {
        // this chunk's rows have landed (own DMAs: vmcnt; everybody's: barrier); the barrier also says every wave is done with
        // the other buffer, which the next chunk's rows may now overwrite.  The DMAs are older than the previous chunk's stores
        // (in-order completion): "at most four outstanding" leaves those in flight -- waited for with vmcnt(0), as __syncthreads()
        // does, every second chunk paid a store round trip (0.456 -> ms per 10 000-channel block).  Hence the raw barrier:
        // the wave's own LDS reads are done (lgkmcnt), nothing else of this kernel is read by another wave.
        if (stored) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stored = false;
        if (c + 1 < n_chunks) issue_chunk(c + 1, (c + 1) & 1);
        if (seg_live) {
            const int bo = (c & 1) * G::buf_bytes + (c & 1) * 8;    // (odd chunks: the window starts one entry further in)
            lds_cc *const la = lbase + aoff + bo, *const lb = lbase + boff + bo, *const ld = lbase + doff + bo, *const le = lbase + eoff + bo;
            // The operands of phase ph + 1 are fetched into the registers the MFMAs of phase ph have just read, and phase ph is
            // finished on the vector pipe meanwhile (read -> wait -> multiply -> MFMA per K step, the first version, left every one
            // of the 13 LDS round trips of a phase in the wave's way: 0.50 ms for the 10 000-channel block).
            auto load_ops = [&](int ph, Ops &o) {
#pragma unroll
                for (int j = 0; j < 13; j++) {
                    o.ta[j] = *(lds_cvf *)(la + (15 * ph + 4 * j) * 8);
                    o.x[j] = *(lds_cv2f *)(lb + (15 * ph + 4 * j) * 8);
                }
                o.x52 = *(lds_cv2f *)(le + (15 * ph + 52) * 8);
                o.t52a = *(lds_cv2f *)(ld + (15 * ph + 52) * 8);
                o.t52b = *(lds_cv2f *)(ld + RB + (15 * ph + 52) * 8);
            };
            Ops ops;
            load_ops(0, ops);
#pragma unroll
            for (int ph = 0; ph < G::PH; ph++) {
                // ---- tap 52 starts the accumulators: (tr, ti) of a slot times h52 xr resp. h52 xi = the slot's (P, Q) pair ----
                const v2f xh = ops.x52 * v2f{h52, h52};
                v4f Dr = v4f{ops.t52a.x * xh.x, ops.t52a.y * xh.x, ops.t52b.x * xh.x, ops.t52b.y * xh.x};
                v4f Di = v4f{ops.t52a.x * xh.y, ops.t52a.y * xh.y, ops.t52b.x * xh.y, ops.t52b.y * xh.y};
                v2f xs[13];
#pragma unroll
                for (int j = 0; j < 13; j++) xs[j] = ops.x[j] * v2f{hA[j], hA[j]};     // the tap rides on the B operand
                // ---- 13 K steps, outside-in ----
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int jj = 0; jj < 13; jj++) {
                    const int j = (jj & 1) ? 12 - (jj >> 1) : (jj >> 1);      // 0, 12, 1, 11, 2, 10, 3, 9, 4, 8, 5, 7, 6
                    Dr = __builtin_amdgcn_mfma_f32_16x16x4f32(ops.ta[j], xs[j].x, Dr, 0, 0, 0);
                    Di = __builtin_amdgcn_mfma_f32_16x16x4f32(ops.ta[j], xs[j].y, Di, 0, 0, 0);
                }
                if (ph + 1 < G::PH) load_ops(ph + 1, ops);
                __builtin_amdgcn_s_setprio(0);
                // ---- channel A = the slot's own table: (Pr - Qi) + j (Pi + Qr); channel B = its conjugate: (Pr + Qi) + j (Pi - Qr) ----
                v2f Y[4];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float Pr = Dr[2 * e], Qr = Dr[2 * e + 1], Pi = Di[2 * e], Qi = Di[2 * e + 1];
                    Y[2 * e] = v2f{Pr - Qi, Pi + Qr};
                    Y[2 * e + 1] = v2f{Pr + Qi, Pi - Qr};
                }
                // ---- stage 2, transposed: input u = ph of the chunk's output is element 16 + u - 5 d of the window of output (current + d) ----
#pragma unroll
                for (int d = 0; d < 5; d++) {
                    const int el = 16 + ph - 5 * d;
                    if (el < 0) continue;
                    const float he = taps.h2[el < 11 ? el : 20 - el];
                    const v2f h = v2f{he, he};
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (el == 0) acc2[d][i] = Y[i] * h;
                        else acc2[d][i] = __builtin_elementwise_fma(Y[i], h, acc2[d][i]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- the chunk's output (element 20 came with its last phase) is complete: keep it for the body's store, shift the rest down ----
            const int body = (c >> 1) - WU;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (c & 1) { stage[i].z = acc2[0][i].x; stage[i].w = acc2[0][i].y; }
                else { stage[i].x = acc2[0][i].x; stage[i].y = acc2[0][i].y; }
            }
#pragma unroll
            for (int d = 0; d < 4; d++)
#pragma unroll
                for (int i = 0; i < 4; i++) acc2[d][i] = acc2[d + 1][i];
            if ((c & 1) && body >= 0) {
                const uint32_t at = ylane + (uint32_t)body * SPW * G::Y2B;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    *reinterpret_cast<v4f *>(reinterpret_cast<char *>(y2_out) + (size_t)(yrow[i] + at) * 8u) = stage[i];
                stored = true;
            }
        }
    }
    // (the next item's first rows go to buffer 0: every wave has left it -- the last chunk is odd -- before the barrier above; its own
    // reads of buffer 1 end before the next item's second barrier)
    if (item + (int)gridDim.x < n_items) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

}   // namespace sdrxgpu
