// kernels_mfma3.hpp -- k_stage123m + k_tail_c: the 10 MS/s chain with NO stage-2 scratch (round 4).
//
// k_stage123m is k_stage12m (kernels_mfma.hpp: NCO mixer + decimator stage 1 as f32 MFMAs over 16 channel pairs x 16 time
// segments; reference MSD::decimate mixer path src/msd.hpp:132-155, S::addSample / calculateOutput msd.hpp:228-335) with
// decimator stages 2 AND 3 behind it as one 119-tap filter in transposed form: every stage-1 output is multiplied into six
// rotating accumulators per channel-segment, each the running sum of one stage-3 output (fused_index.hpp says which output
// sits where, and when the wave stores it).  What leaves the kernel is the stage-3 rate -- 1/125 of the input rate, about
// 0.35 GB per 10 000-channel block with the partial sums at the segment ends, where k_stage12m wrote 1.07 GB of stage-2 outputs
// that k_tail_s read again.  A lane has no warm-up body and no history: an output whose window straddles two segments is the sum
// of what the two lanes hold, put together by k_tail_c (which then runs stage 4).  So the chain's arithmetic is the reference's
// up to the order of summation -- mixer and stage 1 as in k_stage12m, stages 2 and 3 with the taps convolved beforehand (in
// double, rounded to fp32 once) --, held to the oracle at north_star's 1e-5, and NOT bit-identical to itself under a different
// cut of the stream into calls: segments start where a call starts (tests hold cuts to 2e-6 of each other).
#pragma once
#ifndef SDRX_LAB_TAILC
#define SDRX_LAB_TAILC 0      // lab builds: 1 = no piece look-ups, 2 = no last stage (timing only)
#endif
#include "fused_index.hpp"
#include "kernels_mfma.hpp"

namespace sdrxgpu {

struct Stage123Args {
    int n_ch;
    int tbl_stride;
    int N;                  // translator length (1 200)
    int n_y1;               // stage-1 outputs of the call (per channel); lane (segment s) takes outputs 240 s .. 240 s + 239
    int x_base;             // x index of the first sample of stage-1 output 0's window (even)
    int tphase0;            // translator index that sample meets (even; the host's odd-phase table copy)
    int kph;                // K0 mod 150, K0 = stage-1 outputs of all earlier calls
    int n_wt;               // workgroup items per channel-group set = ceil(segment groups / GT)
    int cg0, n_chgroups, n_cgx;   // channel-group sets of this launch (as Stage12Args)
    int y3c_stride;         // float2 per channel row of the scratch: [segment group][entry][segment]
    const float *wtab;      // [fx::WROWS][8] coefficient table: row tau = {H[(tau - 25 p + 94) mod 150] (0 if >= 119), p = 0..5, 0, 0}
};

struct Stage123mTaps {
    float h1[17];
};

template <int GS, int GT>
struct TileGeomM3 : TileGeomM<GS, GT> {
    using B = TileGeomM<GS, GT>;
    static constexpr int wtab_bytes = fx::WROWS * 32;
    static constexpr int strip_bytes = 10 * 32;                  // ten rows: one half-body of a lane
    // [buffer 0 | buffer 1 | coefficient table | strip of the call's one partial segment | strip of zeros]
    static constexpr int off_wtab = 2 * B::buf_bytes;
    static constexpr int off_strip = off_wtab + wtab_bytes;
    static constexpr int off_zero = off_strip + strip_bytes;
    static constexpr int lds_bytes = off_zero + strip_bytes;
};

template <int GS, int GT>
__global__ __launch_bounds__(64 * GS * GT, 2)
void k_stage123m(const float2 *__restrict__ x_in, const float2 *__restrict__ tables, float2 *__restrict__ y3c,
                 const Stage123Args a, const PairArgs pa, const Stage123mTaps taps)
{
    using G = TileGeomM3<GS, GT>;
    constexpr int SPW = 16;
    constexpr int RB = G::row_bytes;
    constexpr int NHB = 2 * fx::LSEG / 20;                  // 24 half-bodies of ten phases
    extern __shared__ __attribute__((aligned(16))) char lds_m[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int sg = wave / GT, tg = wave - sg * GT;

    // the coefficient table and the strip of zeros, once per workgroup
    for (int i = tid; i < G::wtab_bytes / 16; i += 64 * GS * GT)
        reinterpret_cast<v4f *>(lds_m + G::off_wtab)[i] = reinterpret_cast<const v4f *>(a.wtab)[i];
    for (int i = tid; i < G::strip_bytes / 16; i += 64 * GS * GT)
        reinterpret_cast<v4f *>(lds_m + G::off_zero)[i] = v4f{0.f, 0.f, 0.f, 0.f};

    const int n_seg = (a.n_y1 + fx::LSEG - 1) / fx::LSEG;   // segments that hold data
    // item -> (set, time-wave) as in k_stage12m: whole sets to the XCDs eight at a time, the last n_sets % 8 sets by units
    const int n_sets = a.n_chgroups - a.cg0;
    const int n_full = (n_sets >> 3) * a.n_wt, n_rem = (n_sets & 7) * a.n_wt;
    const int n_items = a.n_cgx * 8;                        // (a.n_cgx: units of the XCD that has most)
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int xcd = item & 7;
    const int k = item >> 3;
    int wtm, cgm;
    if (k < n_full) {
        wtm = k % a.n_wt;
        cgm = a.cg0 + (k / a.n_wt) * 8 + xcd;
    } else {
        const int w = ((xcd * n_rem) >> 3) + (k - n_full);
        if (w >= (((xcd + 1) * n_rem) >> 3)) continue;
        wtm = w % a.n_wt;
        cgm = a.cg0 + (n_sets & ~7) + w / a.n_wt;
    }
    const int g_first = wtm * GT;                           // first segment group of the workgroup
    if (g_first * SPW >= n_seg) continue;

    // ---- what this lane stores: slots 4*hi + r of the wave's group, both channels of each; a place without channel stores into
    // the spare row behind the last channel's ----
    const int slot_base = (cgm * GS + sg) * 16;
    uint32_t yrow[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int s = slot_base + 8 * (i >> 1) + 2 * hi + (i & 1);
        const int2 sl = s < pa.n_slots ? pa.slots[s] : make_int2(-1, -1);
        const int ca = sl.x >= 0 ? sl.x : a.n_ch, cb = (sl.x >= 0 && sl.y >= 0) ? sl.y : a.n_ch;
        yrow[2 * i] = (uint32_t)ca * (uint32_t)a.y3c_stride;
        yrow[2 * i + 1] = (uint32_t)cb * (uint32_t)a.y3c_stride;
    }
    const int g_mine = g_first + tg;                        // this wave's segment group
    const int seg = g_mine * SPW + lo;                      // this lane's segment of the call
    const bool seg_live = g_mine * SPW < n_seg;
    // scratch: [segment group][entry][segment]: the 16 lanes of a channel write 128 contiguous bytes per entry
    const uint32_t ylane = (uint32_t)(g_mine * fx::NENT * SPW + lo);

    // ---- the lane's walk through the coefficient table ----
    // phase n of the segment meets row (kph + delta + n) mod 150; a lane whose data end inside the segment (only the call's
    // last segment can) reads its last, partial half-body from a strip whose rows past the end are zeros, and zeros after it
    const int cls = seg % 5;
    int tau = (a.kph + fx::delta_of(cls) + fx::PERIOD) % fx::PERIOD;       // row at phase 0
    const int n_end = min(max(a.n_y1 - fx::LSEG * seg, 0), fx::LSEG);
    const int hb_e = n_end / 10, m_e = n_end - 10 * hb_e;                    // (a whole segment: hb_e = 24, never reached)
    // the workgroup's strip: the one partial segment of the call, if this workgroup holds it
    {
        const int s_p = a.n_y1 / fx::LSEG;                                   // the partial segment (if n_y1 % 240 != 0)
        const int m_p = a.n_y1 - fx::LSEG * s_p;
        const int hb_p = m_p / 10, mm = m_p - 10 * hb_p;
        __syncthreads();                                                     // (the table is in; the previous item's strip is done with)
        if (mm > 0 && s_p / SPW >= g_first && s_p / SPW < g_first + GT && tid < 80) {
            const int tau_p = (a.kph + fx::delta_of(s_p % 5) + fx::PERIOD + 10 * hb_p) % fx::PERIOD;
            const int r = tid >> 3, j = tid & 7;
            reinterpret_cast<float *>(lds_m + G::off_strip)[tid] = r < mm ? reinterpret_cast<const float *>(lds_m + G::off_wtab)[(tau_p + r) * 8 + j] : 0.f;
        }
    }

    // ---- LDS-DMA duties of this wave (as k_stage12m): pairs wave, wave + waves, ... of the workgroup's row pairs ----
    const int piece = lane < G::pieces_a ? lane : lane - G::pieces_a;
    const int rsel = lane < G::pieces_a ? 0 : 1;
    const int seg_max = n_seg - 1;                          // (rows of segments past the data re-read the last one: inside the ring)
    uint32_t dma_off[G::pairs_per_wave];
    int dma_is_t[G::pairs_per_wave];
    const uint32_t piece16 = (uint32_t)piece * 16u;
#pragma unroll
    for (int i = 0; i < G::pairs_per_wave; i++) {
        const int k = wave + i * G::waves;
        dma_is_t[i] = (2 * k >= G::x_rows);
        const int row = 2 * k + rsel;
        uint32_t off;
        if (2 * k < G::x_rows) {
            const int sgm = min((g_first + row / 16) * SPW + (row & 15), seg_max);
            off = (uint32_t)((long long)sgm * a.N * 8);     // segment sgm starts N input samples (= 240 stage-1 outputs) after sgm - 1
        } else {
            const int s = cgm * GS * 16 + (row - G::x_rows);
            int c = s < pa.n_slots ? pa.slots[s].x : -1;
            if (c < 0) c = 0;
            off = (uint32_t)((size_t)c * a.tbl_stride * 8);
        }
        dma_off[i] = off + piece16;
    }
    int tph = a.tphase0 % a.N;
    const char *xsrc0 = reinterpret_cast<const char *>(x_in + a.x_base);
    const char *tsrc0 = reinterpret_cast<const char *>(tables);
    static_assert(G::pairs == G::pairs_per_wave * G::waves, "every wave moves the same number of row pairs");
    auto issue_one = [&](int i, int hb, int bi, int tphh) {
        char *buf = lds_m + bi * G::buf_bytes;
        const int k = wave + i * G::waves;
        const char *base = dma_is_t[i] ? tsrc0 + (size_t)tphh * 8 : xsrc0 + (size_t)hb * G::HB * 8;
        __builtin_amdgcn_global_load_lds((gbl_void *)(uintptr_t)(base + dma_off[i]), (lds_void *)(uint32_t)(uintptr_t)(buf + 2 * k * RB), 16, 0, 0);
    };
    auto issue_half = [&](int hb, int bi, int tphh) {
#pragma unroll
        for (int i = 0; i < G::pairs_per_wave; i++) issue_one(i, hb, bi, tphh);
    };

    // ---- operand addresses (k_stage12m's) ----
    const int aoff = (G::x_rows + sg * 16 + (lo >> 1)) * RB + hi * 8 + (lo & 1) * 4;
    const int boff = (tg * 16 + lo) * RB + hi * 8;
    const int doff = (G::x_rows + sg * 16 + 2 * hi) * RB;
    const int eoff = (tg * 16 + lo) * RB;
    v2f hA[4];
#pragma unroll
    for (int j = 0; j < 4; j++) hA[j] = v2f{taps.h1[4 * j + hi], taps.h1[4 * j + hi]};
    const float h16 = taps.h1[16];

    struct Ops { float ta[2][4]; v2f x[4], t16[4], x16; };
    struct Acc { v4f Dr[2], Di[2]; };
    struct Cf { v4f c03; v2f c45; };                         // the six coefficients of a phase: (c0, c1), (c2, c3), (c4, c5)
    Ops ops;
    Acc acc;
    Cf cf[2];                                                // by phase parity: phase ph + 1's arrive while phase ph's are in use
    // accumulators: slot p of [4 g + 2 e + A/B] = (re, im) of that channel-segment's stage-3 output in flight
    v2f acc3[fx::NSLOT][8];
#pragma unroll
    for (int q = 0; q < fx::NSLOT; q++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc3[q][i] = v2f{0.f, 0.f};

    typedef const volatile __attribute__((address_space(3))) v2f lds_cv2f;
    typedef const volatile __attribute__((address_space(3))) float lds_cvf;
    typedef const volatile __attribute__((address_space(3))) v4f lds_cv4f;
    typedef const __attribute__((address_space(3))) char lds_cc;
    lds_cc *const lbase = (lds_cc *)(uint32_t)(uintptr_t)lds_m;
    lds_cc *const la = lbase + aoff, *const lb = lbase + boff, *const ld = lbase + doff, *const le = lbase + eoff;
    uint32_t caddr = 0;                                      // LDS byte offset of the lane's coefficient row of phase 0 of the half-body
    auto load_ops = [&](int bi, int ph, Ops &o, Cf &c) {
        const int bo = bi * G::buf_bytes;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            o.ta[0][j] = *(lds_cvf *)(la + bo + (5 * ph + 4 * j) * 8);
            o.ta[1][j] = *(lds_cvf *)(la + bo + 8 * RB + (5 * ph + 4 * j) * 8);
            o.x[j] = *(lds_cv2f *)(lb + bo + (5 * ph + 4 * j) * 8);
        }
        o.x16 = *(lds_cv2f *)(le + bo + (5 * ph + 16) * 8);
#pragma unroll
        for (int i = 0; i < 4; i++) o.t16[i] = *(lds_cv2f *)(ld + bo + (8 * (i >> 1) + (i & 1)) * RB + (5 * ph + 16) * 8);
        c.c03 = *(lds_cv4f *)(lbase + caddr + 32 * ph);
        c.c45 = *(lds_cv2f *)(lbase + caddr + 32 * ph + 16);
    };
    auto mfma_phase = [&](const Ops &o, Acc &c) {
        const v2f xh = o.x16 * v2f{h16, h16};
        const v2f xr = v2f{xh.x, xh.x}, xi = v2f{xh.y, xh.y};
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const v2f r0 = o.t16[2 * g] * xr, r1 = o.t16[2 * g + 1] * xr, i0 = o.t16[2 * g] * xi, i1 = o.t16[2 * g + 1] * xi;
            c.Dr[g] = v4f{r0.x, r0.y, r1.x, r1.y};
            c.Di[g] = v4f{i0.x, i0.y, i1.x, i1.y};
        }
        constexpr int korder[4] = {0, 3, 1, 2};              // outer taps first (kernels_mfma.hpp)
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int j = korder[jj];
            const v2f x = o.x[j] * hA[j];
            c.Dr[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ta[0][j], x.x, c.Dr[0], 0, 0, 0);
            c.Di[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ta[0][j], x.y, c.Di[0], 0, 0, 0);
            c.Dr[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ta[1][j], x.x, c.Dr[1], 0, 0, 0);
            c.Di[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ta[1][j], x.y, c.Di[1], 0, 0, 0);
        }
    };
    // combinations + the transposed 119-tap filter for the stage-1 output of phase ph of the half-body
    auto post = [&](auto ph_tag, Acc &c, const Cf &k) {
        constexpr int ph = decltype(ph_tag)::value;
        // (no operand reads between the MFMAs and the combinations in the last phase: 12 wait states, tied to the accumulators by data
        // flow -- kernels_mfma.hpp.  In the other phases the compiler's order puts the next phase's reads there, which
        // tools/hazard_lint.py checks on the built code; in the 4 x 2 shape it does not (round 6: the lint found MFMAs 5 wait states
        // in front of their combinations there), so that shape waits in every phase.)
        if constexpr (ph == 9 || GS == 4) asm volatile("s_nop 7\n\ts_nop 3" : "+v"(c.Dr[0]), "+v"(c.Di[0]), "+v"(c.Dr[1]), "+v"(c.Di[1]));
        v2f Y[8];
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const v2f pq_r = {c.Dr[g][2 * e], c.Dr[g][2 * e + 1]};
                const v2f pq_i = {c.Di[g][2 * e], c.Di[g][2 * e + 1]};
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(Y[4 * g + 2 * e]) : "v"(pq_r), "v"(pq_i));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(Y[4 * g + 2 * e + 1]) : "v"(pq_r), "v"(pq_i));
            }
        asm volatile("s_nop 0");
        // six slots x eight values: one packed fma each; the coefficient is one half of a register pair, taken for both halves
        // of the product by op_sel (low half: op_sel_hi 0; high half: op_sel 1)
        const v2f cp[3] = {v2f{k.c03.x, k.c03.y}, v2f{k.c03.z, k.c03.w}, k.c45};
        auto &A3 = acc3;                                     // (an asm operand alone does not make a generic lambda capture the array)
#pragma unroll
        for (int q = 0; q < fx::NSLOT; q++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if ((q & 1) == 0)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(A3[q][i]) : "v"(Y[i]), "v"(cp[q >> 1]));
                else
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(A3[q][i]) : "v"(Y[i]), "v"(cp[q >> 1]));
            }
    };
    // store slot Q of all eight channel-segments into entry `ent` of the lane's segment, and clear it
    auto store_slot = [&](auto q_tag, int ent) {
        constexpr int Q = decltype(q_tag)::value;
        if (!seg_live) return;
        const uint32_t at = ylane + (uint32_t)ent * SPW;
#pragma unroll
        for (int i = 0; i < 8; i++)
            *reinterpret_cast<v2f *>(reinterpret_cast<char *>(y3c) + (size_t)((yrow[i] + at) * 8u)) = acc3[Q][i];
#pragma unroll
        for (int i = 0; i < 8; i++) acc3[Q][i] = v2f{0.f, 0.f};
    };
    auto store_slot_dyn = [&](int q, int ent) {
        switch (q) {
            case 0: store_slot(std::integral_constant<int, 0>{}, ent); break;
            case 1: store_slot(std::integral_constant<int, 1>{}, ent); break;
            case 2: store_slot(std::integral_constant<int, 2>{}, ent); break;
            case 3: store_slot(std::integral_constant<int, 3>{}, ent); break;
            case 4: store_slot(std::integral_constant<int, 4>{}, ent); break;
            default: store_slot(std::integral_constant<int, 5>{}, ent); break;
        }
    };

    int n_emitted = 0;
    issue_half(0, 0, tph);
    for (int hb = 0; hb < NHB; hb++) {
        const int bi = hb & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tph_prev = tph;
        tph += G::HB;
        if (tph >= a.N) tph -= a.N;
        const int nhb = min(hb + 1, NHB - 1), nbi = bi ^ 1;
        const int ntph = hb + 1 < NHB ? tph : tph_prev;
        if (!seg_live) {                                    // a wave without data: its share of the DMAs, in one go
            issue_half(nhb, nbi, ntph);
            continue;
        }
        // the slot that the schedule says is idle in every lane now goes out (fused_index.hpp: emits)
        {
            int q;
            if (fx::emits(a.kph, 10 * hb, &q)) {
                store_slot_dyn(q, n_emitted);
                n_emitted++;
            }
        }
        // the lane's coefficient rows of this half-body
        {
            const uint32_t normal = (uint32_t)(G::off_wtab + 32 * tau);
            const uint32_t special = (hb == hb_e && m_e > 0) ? (uint32_t)G::off_strip : (uint32_t)G::off_zero;
            caddr = hb < hb_e ? normal : special;
            tau += 10;
            if (tau >= fx::PERIOD) tau -= fx::PERIOD;
        }
        auto window = [&](auto ph_tag) {
            constexpr int ph = decltype(ph_tag)::value;
            __builtin_amdgcn_s_setprio(1);
            mfma_phase(ops, acc);
            if constexpr (ph + 1 < 10) load_ops(bi, ph + 1, ops, cf[(ph + 1) & 1]);
            __builtin_amdgcn_s_setprio(0);
            if constexpr (ph < G::pairs_per_wave) issue_one(ph, nhb, nbi, ntph);
            post(std::integral_constant<int, ph>{}, acc, cf[ph & 1]);
            __builtin_amdgcn_sched_barrier(0);
        };
        load_ops(bi, 0, ops, cf[0]);
        window(std::integral_constant<int, 0>{});
        window(std::integral_constant<int, 1>{});
        window(std::integral_constant<int, 2>{});
        window(std::integral_constant<int, 3>{});
        window(std::integral_constant<int, 4>{});
        window(std::integral_constant<int, 5>{});
        window(std::integral_constant<int, 6>{});
        window(std::integral_constant<int, 7>{});
        window(std::integral_constant<int, 8>{});
        window(std::integral_constant<int, 9>{});
    }
    // what the slots hold at the end of the segment (outputs still in flight, or finished too late for an emission) goes out
    // whole: entries 10 .. 15
    store_slot(std::integral_constant<int, 0>{}, fx::DUMP0 + 0);
    store_slot(std::integral_constant<int, 1>{}, fx::DUMP0 + 1);
    store_slot(std::integral_constant<int, 2>{}, fx::DUMP0 + 2);
    store_slot(std::integral_constant<int, 3>{}, fx::DUMP0 + 3);
    store_slot(std::integral_constant<int, 4>{}, fx::DUMP0 + 4);
    store_slot(std::integral_constant<int, 5>{}, fx::DUMP0 + 5);
    // (the last half-body's extra DMAs went to a buffer the next item fills first: they have to be in, everybody's)
    if (item + (int)gridDim.x < n_items) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    }
}

// ------------------------------------------------------------------------------------------
// k_tail_c: what follows k_stage123m.  One workgroup per (channel, tile of TILE final outputs):
//   * the entries of the segments the tile's stage-3 outputs live in are staged in LDS (coalesced 8-byte loads along a
//     scratch row's segments; 16 entries per segment) together with one record per segment (its slot rotation, its residual
//     phase, its first emission's output);
//   * every stage-3 output the tile needs (its own 5 TILE and the TB - 1 before them; tile 0 takes those from the stream's
//     history) is put together by ONE thread, in stream order: what earlier calls left of it ("carry", if its window began
//     before this call), the head the segment before the one it completes in holds (if its window began there), the entry of
//     the segment it completes in -- fused_index.hpp's locate(), on the staged records;
//   * last stage (MB, TB), one lane per output, the reference's summation order -> iq_out / iq_keep;
//   * the LAST tile leaves the stream's last H3 stage-3 outputs and the partial sums of the (at most five) outputs still in
//     flight for the next call, in the other of two buffers (tile 0 of this launch reads the first).
// ------------------------------------------------------------------------------------------
struct TailCArgs {
    TailArgs t;              // n_ch, ch_lo, out / keep strides and pointers, n_b, hB
    const float2 *y3c;       // [n_ch + 1][y3c_stride]
    int y3c_stride;
    const float2 *hist_in;   // [n_ch][H3 + 8]: H3 stage-3 outputs, then six carries (slot = absolute output index mod 6)
    float2 *hist_out;
    int H3;
    int n_y1, k25, r0m6;     // fused_index.hpp
    int n_out3;              // stage-3 outputs that complete in this call
    int d0;                  // final output k: stage-3 window MB*k + d0 - (TB-1) .. MB*k + d0 (indices within the call)
    int n_seg;               // segments that hold data
    int n_tiles;
};

constexpr int kTailCTile = 128;                              // final outputs per workgroup
constexpr int kTailCMaxSeg = 96;                             // segments a tile's outputs can live in (5 x 128 + 80 outputs at 9.6 per segment + ends)
constexpr int kTailCMaxGrp = 7;                              // ... and segment groups of 16

struct SegRec { int sigma, delta, base_e, pad; };            // base_e: output Rrel of the segment is emission entry Rrel + base_e

// the entry of staged segment record `r` (segment s) that holds the segment's share of output Rrel (fused_index.hpp: locate)
__device__ __forceinline__ int locate_rec(const SegRec r, const int k25, const int r0m6, const int s, const int Rrel)
{
    const int nc = fx::kk_of(k25, Rrel) - fx::LSEG * s;
    const int p = (Rrel + r0m6 + r.sigma) % fx::NSLOT;
    if (nc >= fx::LSEG) return fx::DUMP0 + p;
    const int nE = (nc + r.delta + 11 + 9) / 10 * 10;
    if (nE > 230 || nE < 10) return fx::DUMP0 + p;
    return Rrel + r.base_e;
}

template <int MB, int TB>
__global__ __launch_bounds__(256) void k_tail_c(const TailCArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    __shared__ float2 carry_in[8];
    __shared__ SegRec recs[kTailCMaxSeg];
    const TailArgs &t = a.t;
    // blockIdx -> (channel, tile): the tiles of one channel share blockIdx % 8 (one XCD's L2) and follow each other
    const int xcd = blockIdx.x & 7;
    const int qq = blockIdx.x >> 3;
    const int q = qq % a.n_tiles;
    const int chf = (qq / a.n_tiles) * 8 + xcd;
    if (chf >= t.n_ch) return;
    const int ch = t.n_ch - 1 - chf;
    if (ch < t.ch_lo) return;
    const int tid = threadIdx.x;
    const float2 *__restrict__ sc = a.y3c + (size_t)ch * a.y3c_stride;
    const float2 *__restrict__ hin = a.hist_in + (size_t)ch * (a.H3 + 8);
    const bool last = q == a.n_tiles - 1;

    // ---- the tile: final outputs k0 .. k1 - 1; stage-3 outputs (call-relative) r_lo .. r_hi - 1 into row[] ----
    const int k0 = q * kTailCTile, k1 = min(t.n_b, k0 + kTailCTile);
    int r_lo = MB * k0 + a.d0 - (TB - 1);                    // (tile 0: negative -- the stream's history)
    int r_hi = last ? a.n_out3 : MB * (k1 - 1) + a.d0 + 1;
    if (last && a.n_tiles > 1) r_lo = min(r_lo, a.n_out3 - a.H3);       // the last tile also leaves the next call's history
    const int r_base = q == 0 ? -a.H3 : r_lo;                // row[j] = stage-3 output r_base + j
    const int r_new = max(r_lo, 0);                          // first output to put together here
    v2f *__restrict__ row = reinterpret_cast<v2f *>(smem);
    const int n_row = ((r_hi - r_base + 3) & ~3) + 4;
    float2 *__restrict__ raw = smem + n_row;                 // [segment group - g_lo][16 entries][16 segments]: the scratch row's own layout

    // segments the outputs r_new .. r_hi - 1 have pieces in
    int s_lo = 0, s_hi = -1;
    if (r_hi > r_new) {
        s_lo = max(0, (fx::kk_of(a.k25, r_new) - (fx::NTAP - 1)) / fx::LSEG);
        s_hi = min(a.n_seg - 1, fx::kk_of(a.k25, r_hi - 1) / fx::LSEG);
    }
    const int g_lo = s_lo >> 4, n_g = s_hi >= s_lo ? (s_hi >> 4) - g_lo + 1 : 0;
    const int n_s = s_hi - s_lo + 1;
    // (0) stage: history (tile 0), carries, segment records, the segment groups' entries -- a group's 16 entries x 16 segments
    //     are 2 KB in a row: thread i takes float2 i of every group, no index arithmetic, all loads before any LDS write
    if (q == 0)
        for (int i = tid; i < a.H3; i += 256) {
            const float2 v = hin[i];
            row[i] = v2f{v.x, v.y};
        }
    if (tid < 8) carry_in[tid] = tid < fx::NSLOT ? hin[a.H3 + tid] : make_float2(0.f, 0.f);
    if (tid < n_s) {
        const int s = s_lo + tid, c = s % 5;
        recs[tid] = SegRec{fx::sigma_of(c), fx::delta_of(c), fx::cycle_bias(a.k25) + fx::sigma_of(c) - fx::NSLOT * fx::zshift(s), 0};
    }
    {
        // 16-byte loads: 128 threads per group, two groups per round
        constexpr int GP = fx::NENT * 16 / 2;                // 16-byte pieces per group (128)
        static_assert(GP == 128, "two groups per round of 256 threads");
        constexpr int NR = (kTailCMaxGrp + 1) / 2;
        const v4f *__restrict__ src = reinterpret_cast<const v4f *>(sc + (size_t)g_lo * (fx::NENT * 16)) + (tid & (GP - 1));
        v4f *__restrict__ dst = reinterpret_cast<v4f *>(raw) + (tid & (GP - 1));
        const int gsel = tid >> 7;
        v4f v[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) v[r] = src[(size_t)min(2 * r + gsel, max(n_g - 1, 0)) * GP];
#pragma unroll
        for (int r = 0; r < NR; r++)
            if (2 * r + gsel < n_g) dst[(2 * r + gsel) * GP] = v[r];
    }
    __syncthreads();
    // (1) the tile's stage-3 outputs
    auto piece = [&](const int s, const int i) -> v2f {
        const int ent = locate_rec(recs[s - s_lo], a.k25, a.r0m6, s, i);
        const float2 v = raw[((s >> 4) - g_lo) * (fx::NENT * 16) + ent * 16 + (s & 15)];
        return v2f{v.x, v.y};
    };
    for (int i = r_new + tid; i < r_hi; i += 256) {
        const int kk = fx::kk_of(a.k25, i);
        v2f v = {0.f, 0.f};
        if (kk - (fx::NTAP - 1) < 0) {
            const float2 c = carry_in[(a.r0m6 + i) % fx::NSLOT];
            v = v2f{c.x, c.y};
        }
        const int s_c = kk / fx::LSEG;                        // the segment it completes in
#if SDRX_LAB_TAILC == 1
        v += v2f{raw[(i + tid) & 1023].x, raw[(i * 7) & 1023].y};
#else
        if (s_c > 0 && kk - (fx::NTAP - 1) < fx::LSEG * s_c) v += piece(s_c - 1, i);
        v += piece(s_c, i);
#endif
        row[i - r_base] = v;
    }
    // (2) last tile: the outputs still in flight when the call ends -- carry for the next call, one thread per slot, pieces
    //     in stream order, straight from memory
    v2f carry_out = {0.f, 0.f};
    if (last && tid < fx::NSLOT) {
        const int r1m6 = (a.r0m6 + a.n_out3) % fx::NSLOT;
        const int Rrel = a.n_out3 + (tid - r1m6 + fx::NSLOT) % fx::NSLOT;     // the output of slot `tid` among n_out3 .. n_out3 + 5
        const int kk = fx::kk_of(a.k25, Rrel);
        if (kk - (fx::NTAP - 1) < a.n_y1) {                                     // its window has begun
            if (kk - (fx::NTAP - 1) < 0) carry_out = v2f{carry_in[tid].x, carry_in[tid].y};
            const int sl = max(0, (kk - (fx::NTAP - 1)) / fx::LSEG);
            for (int s = sl; s < a.n_seg; s++) {
                if (!fx::overlaps(a.k25, a.n_y1, s, Rrel)) continue;
                int nc;
                const int ent = fx::locate(a.k25, a.r0m6, s, Rrel, &nc);
                const float2 v = sc[((s >> 4) * fx::NENT + ent) * 16 + (s & 15)];
                carry_out += v2f{v.x, v.y};
            }
        }
    }
    __syncthreads();
    // (3) last stage: one lane per output
#if SDRX_LAB_TAILC == 2
    if (k0 + tid < k1) {
        const v2f r = row[MB * (k0 + tid) + a.d0 - r_base];
        const float2 o = make_float2(r.x, r.y);
        if (t.iq_out) t.iq_out[(size_t)ch * t.out_stride + k0 + tid] = o;
        if (t.iq_keep) t.iq_keep[(size_t)ch * t.keep_stride + k0 + tid] = o;
    }
#else
    if (k0 + tid < k1) {
        const v2f r = folded_fir_ahead<TB, kMaxHalfTaps, SDRX_FIR_AHEAD_B>(row + (MB * (k0 + tid) + a.d0 - (TB - 1) - r_base), t.hB);
        const float2 o = make_float2(r.x, r.y);
        if (t.iq_out) t.iq_out[(size_t)ch * t.out_stride + k0 + tid] = o;
        if (t.iq_keep) t.iq_keep[(size_t)ch * t.keep_stride + k0 + tid] = o;
    }
#endif
    // (4) last tile: histories for the next call
    if (last) {
        float2 *__restrict__ hout = a.hist_out + (size_t)ch * (a.H3 + 8);
        for (int i = tid; i < a.H3; i += 256) {
            const v2f v = row[a.n_out3 - a.H3 + i - r_base];   // (a short call: part of it is the history that came in, tile 0's row[0 ..])
            hout[i] = make_float2(v.x, v.y);
        }
        if (tid < fx::NSLOT) hout[a.H3 + tid] = make_float2(carry_out.x, carry_out.y);
    }
}

}   // namespace sdrxgpu
