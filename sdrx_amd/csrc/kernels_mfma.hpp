// kernels_mfma.hpp -- k_stage12m: the NCO mixer and decimator stage 1 of the 10 MS/s chain on the f32 matrix pipe
// (v_mfma_f32_16x16x4_f32), decimator stage 2 on the vector pipe.  Same job, inputs and outputs as k_stage12p
// (kernels.hpp); reference: MSD::decimate mixer path src/msd.hpp:132-155, S::addSample / calculateOutput
// msd.hpp:228-335, translator table sdrx.cpp:1682-1690.
//
// Why a matrix instruction fits a 1-D FIR here.  The mixer and the first FIR are one linear map per channel,
//     y1_c[k] = sum_i h1[i] * t_c[(5k-12+i) mod N] * x[5k-12+i],          i = 0 .. 16,
// and for time segments that start a multiple of N samples apart the coefficient h1[i]*t_c[..] is the same in every
// segment.  So for 16 channels x 16 such segments, stage-1 output k of all of them is a dense 16 x 17 by 17 x 16
// product: A[channel][tap] = h1[tap] * t_c[..] (built on the fly from the caller's table -- float-rounded phases and
// all), B[tap][segment] = x.  Channels at +k and -k steps from the tuner have exactly conjugate tables
// (k_stage12p's observation), so with A split into its real part Cr and imaginary part Ci the four REAL products
//     Pr = Cr xr,  Pi = Cr xi,  Qr = Ci xr,  Qi = Ci xi
// give both channels: y+ = (Pr - Qi) + j (Pi + Qr),  y- = (Pr + Qi) + j (Pi - Qr): 68 multiply-adds per pair of
// complex outputs against 98 lane-slots in the packed vector form, no separate mixer pass, and the matrix pipe holds
// 2.4 GHz where packed fp32 vector code runs at 1.8 (measured: tools/ubench_mfma_fir.hip).  The f32 MFMA
// is an exact fp32 fma chain (one rounding per product, k-ordered), not a reduced-precision path.
//
// A channel without partner runs as a pair whose second half is dropped: its bits do not depend on who else is in the
// bank.  Taps 0..15 go through four K = 4 steps; tap 16 starts the accumulators (vector multiply), which costs less
// issue time than a fifth, mostly empty K step (the f32 MFMA and the vector ALU do not overlap: measured additive).
//
// Geometry.  A wave = one tile of 16 slots (pairs) x 16 time segments, as two MFMA tiles of 8 slots x 16 segments: the
// 16 rows of an MFMA are 8 slots x {Cr, Ci}, so that a lane's accumulator registers hold (P, Q) of a slot side by side
// (C/D row 4*(lane>>4)+r = slot 2*(lane>>4) + (r>>1), part r&1; column = segment lane&15).  A workgroup = GS slot groups x GT segment groups = GS*GT waves that
// share GT*16 input rows and GS*16 translator rows in LDS (LDS-DMA, two buffers of half a body = 50 samples + 12 of
// history each).  Segments, bodies (100 samples = 20 stage-1 outputs = 4 stage-2 outputs), warm-up and the body-major
// stage-2 scratch are k_stage12's but for the order inside a body (half-body rows: TailArgs.half_split tells the tail kernels).
// Stage 2 (19 taps, /5) runs in TRANSPOSED form: a stage-1 output is multiplied into the (at most four) stage-2 outputs
// whose windows hold it, so a lane carries 4 accumulators per value instead of a 14-deep delay line -- what makes 8
// channel-segments per lane fit the register file.  Each output is still one fma chain over its 19 inputs in order
// j = 0..18, whatever body, lane or call it falls in: splitting the stream differently changes no bit.
#pragma once
#include "kernels.hpp"

namespace sdrxgpu {

template <int GS, int GT>
struct TileGeomM {
    static constexpr int HB = 50;                       // input samples per half-body
    static constexpr int PH = 10;                       // stage-1 outputs ("phases") per half-body
    static constexpr int HALO = 12;                     // T1 - M1: samples in front of a half-body its first window needs
    static constexpr int row_bytes = 528;               // 62 samples = 496 B, padded: pitch == 16 (mod 32) keeps the operand reads conflict-free
    static constexpr int pieces_a = 33, pieces_b = 31;  // one DMA instruction = two adjacent rows: 33 + 31 16-byte pieces
    static constexpr int x_rows = GT * 16, t_rows = GS * 16;
    static constexpr int buf_bytes = (x_rows + t_rows) * row_bytes;
    static constexpr int waves = GS * GT;
    static constexpr int pairs = (x_rows + t_rows) / 2; // DMA instructions per half-body
    static constexpr int pairs_per_wave = (pairs + waves - 1) / waves;
};

struct Stage12mTaps {
    float h1[17];
    float h2[10];       // h2[0 .. 9]; h2[18 - e] = h2[e] (the reference reads that half only, msd.hpp:309-331): 10 scalar register pairs, not 19
};

// A wave's memory instructions -- LDS-DMA for the next half-body, stores of the previous one's outputs -- go out a piece per phase
// instead of in a burst behind the barrier (-5 % on the launch).  The first six phases of a lane's warm-up body (nobody's input)
// are skipped (-1.5 %): a second path through the half-body, 256 VGPRs and a few spills (none touched inside the loop).
template <int GS, int GT>
__global__ __launch_bounds__(64 * GS * GT, 2)
void k_stage12m(const float2 *__restrict__ x_in, const float2 *__restrict__ tables, float2 *__restrict__ y2_out,
                const Stage12Args a, const PairArgs pa, const Stage12mTaps taps)
{
    using G = TileGeomM<GS, GT>;
    constexpr bool TRIM = true;
    constexpr int SB = 100, Y2B = 4, SPW = 16;
    constexpr int RB = G::row_bytes;
    extern __shared__ __attribute__((aligned(16))) char lds_m[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, hi = lane >> 4;
    const int sg = wave / GT, tg = wave - sg * GT;          // this wave's slot group / segment group inside the workgroup

    // A workgroup takes the items blockIdx.x, blockIdx.x + gridDim.x, ... (the host launches one workgroup per item; a smaller
    // grid makes the workgroups persistent).  item -> (slot-group set, time-wave): item % 8 is the XCD (round-robin placement of
    // blocks; gridDim.x is a multiple of 8).  The sets go to the XCDs eight at a time, a whole set to an XCD -- its time-waves share
    // that XCD's L2, and all XCDs work their way up the bank together (the tail kernel starts with the rows written last: Infinity
    // Cache) --, but the last n_sets % 8 sets are dealt out by units (time-waves), in eight equal runs: with whole sets to the end,
    // 17 sets were 3 on one XCD and 2 on the others (a bank of 1 024 channels: + 25 % on the launch), 163 sets 21 against 20.
    const int n_sets = a.n_chgroups - a.cg0;
    const int n_full = (n_sets >> 3) * a.n_wt;              // units of an XCD from whole rounds of eight sets
    const int n_rem = (n_sets & 7) * a.n_wt;                // units of the last, partial round
    const int n_items = a.n_cgx * 8;                        // (a.n_cgx here: units of the XCD that has most = n_full + ceil(n_rem / 8))
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int xcd = item & 7;
    const int k = item >> 3;
    int wtm, cgm;
    if (k < n_full) {
        wtm = k % a.n_wt;                                   // a.n_wt here: time-waves of GT segment groups
        cgm = a.cg0 + (k / a.n_wt) * 8 + xcd;               // set of GS slot groups
    } else {
        const int w = ((xcd * n_rem) >> 3) + (k - n_full);
        if (w >= (((xcd + 1) * n_rem) >> 3)) continue;
        wtm = w % a.n_wt;
        cgm = a.cg0 + (n_sets & ~7) + w / a.n_wt;
    }
    const bool y2_stream = cgm < a.nt_sets;
    const int gq = wtm / a.P, p = wtm - gq * a.P;
    const int bps = a.nb * a.P;
    const int g_first = gq * GT;                            // first segment group of the workgroup
    if ((g_first * SPW * bps + a.nb * p) * Y2B >= a.n_y2_new) continue;

    // ---- what this lane stores: slots 4*hi + r of the wave's group, both channels of each.  A place without channel
    // (empty slot, no partner) stores into the spare row behind the last channel's: no branch around any store ----
    const int slot_base = (cgm * GS + sg) * 16;
    uint32_t yrow[8];                                       // [4 g + 2 e + (0: channel A, 1: channel B)]: first element of the channel's scratch row
#pragma unroll
    for (int i = 0; i < 4; i++) {                           // i = 2 g + e: slot 8 g + 2 hi + e of the wave's group
        const int s = slot_base + 8 * (i >> 1) + 2 * hi + (i & 1);
        const int2 sl = s < pa.n_slots ? pa.slots[s] : make_int2(-1, -1);
        const int ca = sl.x >= 0 ? sl.x : a.n_ch, cb = (sl.x >= 0 && sl.y >= 0) ? sl.y : a.n_ch;
        yrow[2 * i] = (uint32_t)ca * (uint32_t)a.y2_stride;
        yrow[2 * i + 1] = (uint32_t)cb * (uint32_t)a.y2_stride;
    }
    const int g_mine = g_first + tg;                        // this wave's segment group
    const int wt = g_mine * a.P + p;                        // its time-wave in k_stage12's numbering (scratch layout)
    const bool seg_live = (g_mine * SPW * bps + a.nb * p) * Y2B < a.n_y2_new;

    // ---- LDS-DMA duties of this wave: pairs wave, wave + waves, ... of the workgroup's row pairs ----
    // lanes 0..32 move row 2k (33 pieces: the 31 of the row and the 2 of its pad), lanes 33..63 row 2k+1 (31 pieces)
    const int piece = lane < G::pieces_a ? lane : lane - G::pieces_a;
    const int rsel = lane < G::pieces_a ? 0 : 1;
    // (rows of segments past the last segment group that holds data re-read its last segment: inside the ring)
    const int seg_max = ((a.n_y2_new + Y2B - 1) / Y2B + SPW * bps - 1) / (SPW * bps) * SPW - 1;
    // per pair and lane: the byte offset from the pair's wave-uniform base, in a register per pair
    uint32_t dma_off[G::pairs_per_wave];
    int dma_is_t[G::pairs_per_wave];
    const uint32_t piece16 = (uint32_t)piece * 16u;
#pragma unroll
    for (int i = 0; i < G::pairs_per_wave; i++) {
        const int k = wave + i * G::waves;
        dma_is_t[i] = (2 * k >= G::x_rows);
        auto row_off = [&](int row) -> uint32_t {
            if (2 * k < G::x_rows) {
                const int seg = min((g_first + row / 16) * SPW + (row & 15), seg_max);
                return (uint32_t)(((long long)seg * bps + a.nb * p) * SB * 8);
            }
            const int s = cgm * GS * 16 + (row - G::x_rows);
            int c = s < pa.n_slots ? pa.slots[s].x : -1;
            if (c < 0) c = 0;                               // an empty slot works on channel 0's table and stores nothing
            return (uint32_t)((size_t)c * a.tbl_stride * 8);
        };
        dma_off[i] = row_off(2 * k + rsel) + piece16;
    }
    auto lane_off = [&](int i) -> uint32_t { return dma_off[i]; };
    int tph = a.tphase0 + p * a.nb * SB - G::HALO;          // translator index of the first entry of half-body 0's rows
    tph %= a.N;
    if (tph < 0) tph += a.N;
    const char *xsrc0 = reinterpret_cast<const char *>(x_in + a.x_base - G::HALO);
    const char *tsrc0 = reinterpret_cast<const char *>(tables);
    static_assert(G::pairs == G::pairs_per_wave * G::waves, "every wave moves the same number of row pairs");
    auto issue_one = [&](int i, int hb, int bi, int tphh) {
        char *buf = lds_m + bi * G::buf_bytes;
        const int k = wave + i * G::waves;
        const char *base = dma_is_t[i] ? tsrc0 + (size_t)tphh * 8 : xsrc0 + (size_t)hb * G::HB * 8;
        __builtin_amdgcn_global_load_lds((gbl_void *)(uintptr_t)(base + lane_off(i)), (lds_void *)(uint32_t)(uintptr_t)(buf + 2 * k * RB), 16, 0, 0);
    };
    auto issue_half = [&](int hb, int bi, int tphh) {
#pragma unroll
        for (int i = 0; i < G::pairs_per_wave; i++) issue_one(i, hb, bi, tphh);
    };

    // ---- operand addresses (byte offsets inside a buffer) ----
    // The MFMA's 16 rows are 8 slots x {real, imaginary part of the coefficient}: row 2 s + c.  The accumulators then hold, per
    // lane, (P, Q) of a slot in neighbouring registers -- C/D row 4 hi + r = slot 2 hi + (r >> 1), part r & 1 --, which is what
    // the packed tap-16 products write and what makes (re, im) of every output neighbours further down.  A wave's 16 slots
    // are two such half-groups (g = 0, 1) that share the B operand.
    const int aoff = (G::x_rows + sg * 16 + (lo >> 1)) * RB + hi * 8 + (lo & 1) * 4;   // A: part lo & 1 of t[slot 8 g + (lo >> 1)][5p + 4j + hi]
    const int boff = (tg * 16 + lo) * RB + hi * 8;               // B: x[segment lo][5p + 4j + hi]
    const int doff = (G::x_rows + sg * 16 + 2 * hi) * RB;        // tap 16, C/D layout: t[slot 8 g + 2 hi + e][5p + 16]
    const int eoff = (tg * 16 + lo) * RB;                        //                     x[segment lo][5p + 16]
    v2f hA[4];                                                   // taps 4j + hi, j = 0..3, as (h, h) pairs
#pragma unroll
    for (int j = 0; j < 4; j++) hA[j] = v2f{taps.h1[4 * j + hi], taps.h1[4 * j + hi]};
    const float h16 = taps.h1[16];

    // All vector arithmetic below is written in packed form (v_pk_mul/add/fma_f32 on register pairs): the f32 MFMA and
    // the vector ALU do not overlap (measured), so beside MFMAs a wave's vector instructions issue alone, at
    // 4 cycles each whether packed or not.
    struct Ops { float ta[2][4]; v2f x[4], t16[4], x16; };     // A operands [g][j]; B = x[j]; tap 16: t16[2 g + e], x16
    struct Acc { v4f Dr[2], Di[2]; };                           // [g]: A_g x (h xr), A_g x (h xi): (P, Q) of slots 8 g + 2 hi, + 1
    Ops ops;
    Acc acc;
    // stage-2 accumulators: output (s + dd) & 3 of super-step s; [4 g + 2 e + A/B] = (re, im) of that channel
    v2f acc2[4][8];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
        for (int i = 0; i < 8; i++) acc2[q][i] = v2f{0.f, 0.f};

    // LDS pointers in their own address space: every operand address is one per-lane base register + an immediate.
    // (volatile: the 8-byte operand reads stay single ds_read_b64 -- 32 lanes per LDS cycle on 64 banks, which the row pitch makes
    // conflict-free; merged into ds_read2_b64 they are served 16 lanes at a time on 32 banks: half the rate, and two-way conflicts)
    typedef const volatile __attribute__((address_space(3))) v2f lds_cv2f;
    typedef const volatile __attribute__((address_space(3))) float lds_cvf;
    typedef const __attribute__((address_space(3))) char lds_cc;
    lds_cc *const lbase = (lds_cc *)(uint32_t)(uintptr_t)lds_m;
    lds_cc *const la = lbase + aoff, *const lb = lbase + boff, *const ld = lbase + doff, *const le = lbase + eoff;
    auto load_ops = [&](int bi, int ph, Ops &o) {                // bi: buffer; ph: phase inside the half-body, 0..9
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
    };
    auto mfma_phase = [&](const Ops &o, Acc &c) {
        // tap 16 starts the accumulators: (tr, ti) of a slot times h16 xr resp. h16 xi = the slot's (P, Q) pair
        const v2f xh = o.x16 * v2f{h16, h16};
        const v2f xr = v2f{xh.x, xh.x}, xi = v2f{xh.y, xh.y};
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const v2f r0 = o.t16[2 * g] * xr, r1 = o.t16[2 * g + 1] * xr, i0 = o.t16[2 * g] * xi, i1 = o.t16[2 * g + 1] * xi;
            c.Dr[g] = v4f{r0.x, r0.y, r1.x, r1.y};
            c.Di[g] = v4f{i0.x, i0.y, i1.x, i1.y};
        }
        // K steps in the order taps 0-3, 12-15, 4-7, 8-11: the small outer taps first, the large centre taps last, so that the
        // partial sums a rounding is relative to stay small as long as possible (the reference's folded FIR adds outside-in
        // too, msd.hpp:309-331).  Under a strong carrier every weak channel's P and Q are large sums that cancel; with the steps
        // in natural order the weak partner's deviation from the reference was 1.7 x larger (strong carrier 0.9 over a floor
        // of 0.05: 1.2e-5 of the channel's rms against 7.8e-6; tests/test_gpu_pairs_stress.py, DESIGN 4.7 "Numerics")
        constexpr int korder[4] = {0, 3, 1, 2};
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int j = korder[jj];
            const v2f x = o.x[j] * hA[j];                        // the tap rides on the B operand (shared by both half-groups)
            c.Dr[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ta[0][j], x.x, c.Dr[0], 0, 0, 0);
            c.Di[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ta[0][j], x.y, c.Di[0], 0, 0, 0);
            c.Dr[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ta[1][j], x.x, c.Dr[1], 0, 0, 0);
            c.Di[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.ta[1][j], x.y, c.Di[1], 0, 0, 0);
        }
    };

    // The two stage-2 outputs a half-body completes wait in `stage` and leave right after the NEXT barrier, as one
    // 16-byte store per channel: by the
    // time the following s_waitcnt vmcnt(0) comes they are a whole half-body old.  (Stored where they complete, they
    // made every barrier wait for a store round trip: 0.85 against 0.64 ms for the launch.)
    // Scratch layout: k_stage12's body-major rows with HALF-bodies as the bodies -- [time-wave][half-body][segment][2] --, so that
    // the 16 lanes of a channel write 256 contiguous bytes per store and no 32-byte sector is written in two goes (with whole
    // bodies of four outputs per segment the two halves of every sector came ten phases apart: 2.0 GB of HBM writes per
    // launch for 1.07 GB of outputs).  The tail kernels are told so: TailArgs.half_split (y2b stays 4).
    v4f stage[8];                                           // per channel: (re, im) of the half-body's first output, of its second
    const uint32_t ylane = (uint32_t)a.y2_hist + ((uint32_t)wt * (2 * a.nb) * SPW + lo) * 2;
    auto store_staged_part = [&](int body, int h, int i0, int i1) {     // channels i0 .. i1 - 1 of the lane's eight
        if (body < 0 || !seg_live) return;
        const uint32_t at = ylane + (uint32_t)(2 * body + h) * SPW * 2;
        // (scalar base + 32-bit byte offset per lane: one address register per store -- these stores sit between the phases now,
        // where the register file is full; the host keeps the scratch below 4 GB)
        // y2_stream: a scratch larger than the Infinity Cache goes out with the streaming hint, except the sets written last, which the
        // tail kernel reads first (tail 0.243 -> 0.231 ms at 10 000 channels with every store streaming; 0.044 -> 0.049
        // at 1 024 channels, whose scratch the cache holds)
        if (y2_stream) {
#pragma unroll
            for (int i = i0; i < i1; i++)
                // Inline assembly, because LLVM folds the two arms of this branch into one ordinary store otherwise (also with
                // __builtin_nontemporal_store in this one).  The s_nop is the compiler's own rule for gfx940 and later, which it cannot
                // apply to a store it does not see: two wait states between a store of more than 8 bytes and a vector instruction
                // that overwrites its data registers (without it one set of a 2 304-channel bank came out wrong).
                // (tools/hazard_lint.py checks the built code for this wait; -DSDRX_LAB_NO_STORE_NOP=1 is the lab build it must fail on)
#if defined(SDRX_LAB_NO_STORE_NOP) && SDRX_LAB_NO_STORE_NOP
                asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"((yrow[i] + at) * 8u), "v"(stage[i]), "s"(y2_out) : "memory");
#else
                asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"((yrow[i] + at) * 8u), "v"(stage[i]), "s"(y2_out) : "memory");
#endif
        } else {
#pragma unroll
            for (int i = i0; i < i1; i++)
                *reinterpret_cast<v4f *>(reinterpret_cast<char *>(y2_out) + (size_t)((yrow[i] + at) * 8u)) = stage[i];
        }
    };
    auto store_staged = [&](int body, int h) { store_staged_part(body, h, 0, 8); };
    // combinations + stage 2 for stage-1 output P20 (0..19) of a body
    auto post = [&](auto p20_tag, Acc &c) {
        constexpr int P20 = decltype(p20_tag)::value;
        // (the combinations below are inline assembly: the compiler does not space them from the MFMAs that write their sources --
        // 12 wait states for the 8-pass f32 MFMA on gfx950.  In phases 0..8 the compiler puts the next phase's 21 operand reads in
        // between: its choice, not a rule, so tools/hazard_lint.py checks the built code object for it (build() fails otherwise).
        // The last phase of a half-body has no reads to put there: explicit wait states, tied to the accumulators by data flow --
        // the MFMAs cannot sink below them, the combinations cannot rise above.)
        if constexpr (P20 % 10 == 9) asm volatile("s_nop 7\n\ts_nop 3" : "+v"(c.Dr[0]), "+v"(c.Di[0]), "+v"(c.Dr[1]), "+v"(c.Di[1]));
        // channel A = the slot's own table: (Pr - Qi) + j (Pi + Qr); channel B = its conjugate: (Pr + Qi) + j (Pi - Qr)
        v2f Y[8];
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                // one packed add per channel: sources (Pr, Qr) and (Pi, Qi) as the accumulators hold them; the low half takes the
                // second source's HIGH half (op_sel) and the high half its LOW half (op_sel_hi), negated where the sign says so --
                // (Pr - Qi, Qr + Pi) and (Pr + Qi, Pi - Qr), the same additions the two scalar instructions per half made
                const v2f pq_r = {c.Dr[g][2 * e], c.Dr[g][2 * e + 1]};
                const v2f pq_i = {c.Di[g][2 * e], c.Di[g][2 * e + 1]};
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(Y[4 * g + 2 * e]) : "v"(pq_r), "v"(pq_i));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(Y[4 * g + 2 * e + 1]) : "v"(pq_r), "v"(pq_i));
            }
        asm volatile("s_nop 0");                                 // (a packed instruction may read what the last one above wrote)
        // stage-2 output j (absolute) has the stage-1 outputs 5j-14 .. 5j+4 as its window (S::addSample fires on every
        // 5th input, msd.hpp:228-247).  Inside a body, input u = P20 % 5 of super-step s = P20 / 5 is element 14 + u - 5 dd
        // of the window of output s + dd: output s + 3 starts at u = 1 (element 0), output s completes at u = 4 (element 18)
        constexpr int u = P20 % 5, s = P20 / 5;
#pragma unroll
        for (int dd = 0; dd < 4; dd++) {
            const int e = 14 + u - 5 * dd;
            if (e < 0) continue;
            const float he = taps.h2[e < 10 ? e : 18 - e];
            const v2f h = v2f{he, he};
            const int slot = (s + dd) & 3;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (e == 0) acc2[slot][i] = Y[i] * h;
                else acc2[slot][i] = __builtin_elementwise_fma(Y[i], h, acc2[slot][i]);
            }
            // pin the updates here: LLVM otherwise sinks every chain down to its store, five phases later, and keeps
            // all the stage-1 outputs of those phases alive meanwhile (spills)
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" : "+v"(acc2[slot][i]));
        }
        if constexpr (u == 4) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr ((s & 1) == 0) { stage[i].x = acc2[s & 3][i].x; stage[i].y = acc2[s & 3][i].y; }
                else { stage[i].z = acc2[s & 3][i].x; stage[i].w = acc2[s & 3][i].y; }
            }
        }
    };

    const int n_bodies = 1 + a.nb;                              // one warm-up body
    issue_half(0, 0, tph);
    for (int b = 0; b < n_bodies; b++) {
        const int body = b - 1;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            // this half-body's rows have landed (own DMAs: vmcnt; everybody's: barrier); the barrier also says
            // every wave is done with the other buffer, which the next half-body's rows may now overwrite
            const int hb = 2 * b + half;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int tph_prev = tph;
            tph += G::HB;
            if (tph >= a.N) tph -= a.N;
            // What a wave has to move besides goes out a piece per phase: its share of the next half-body's rows (LDS-DMA i in phase i;
            // the last one two phases before the barrier that waits for it) and the two outputs per channel the previous half-body
            // completed (8 stores: channels 2 ph, 2 ph + 1 in phases 0 .. 3; phase 4 completes the next output into the staging
            // registers).  As one burst behind the barrier these 16 memory instructions cost the wave 2 400 of its 14 200 cycles per
            // half-body (s_memtime stamps: 185 cycles per DMA, 115 per store, into a queue the other waves' bursts have filled),
            // cycles in which it feeds nothing to the matrix pipe, and the one other wave of the SIMD cannot keep that pipe busy
            // alone.  No branch may stand in a phase for this (with one the kernel needs 77 registers more than there are):
            // (behind the last half-body: the same rows once more, into the buffer nobody reads any more -- no branch in the phases)
            const int nhb = min(hb + 1, 2 * n_bodies - 1), nbi = (half + 1) & 1;
            const int ntph = hb + 1 < 2 * n_bodies ? tph : tph_prev;
            if (!seg_live) issue_half(nhb, nbi, ntph);       // (a wave without data: its share of the DMAs, in one go)
            const int sbody = half == 0 ? body - 1 : body, sh = half == 0 ? 1 : 0;
            // the ten phases: phase ph goes to the matrix pipe, the operands of phase ph+1 are fetched into the registers
            // the MFMAs have just read, then phase ph is finished on the vector pipe (the fetch covers the MFMA latency)
            auto window = [&](auto ph_tag) {
                constexpr int P20 = decltype(ph_tag)::value;   // phase inside the body
                constexpr int ph = P20 % 10;                   // ... inside the half-body
                // (the wave that feeds the matrix pipe goes first when both waves of the SIMD have something to issue:
                // measured -1.5 % on the launch)
                __builtin_amdgcn_s_setprio(1);
                mfma_phase(ops, acc);
                if constexpr (ph + 1 < 10) load_ops(P20 / 10, ph + 1, ops);
                __builtin_amdgcn_s_setprio(0);
                if constexpr (ph < G::pairs_per_wave) issue_one(ph, nhb, nbi, ntph);
                if constexpr (ph < 4) store_staged_part(sbody, sh, 2 * ph, 2 * ph + 2);
                post(std::integral_constant<int, P20>{}, acc);
                __builtin_amdgcn_sched_barrier(0);
            };
            // from6: the warm-up body's first half.  Stage 2's first real output has the stage-1 outputs from phase 6 of the
            // warm-up body on as its window: phases 0..5 of that body are nobody's input and are skipped (nothing to store yet).
            auto run_half = [&](auto h_tag, const bool from6) {
                constexpr int H = decltype(h_tag)::value;
                if (!from6) {
                    load_ops(H, 0, ops);
                    window(std::integral_constant<int, H * 10 + 0>{});
                    window(std::integral_constant<int, H * 10 + 1>{});
                    window(std::integral_constant<int, H * 10 + 2>{});
                    window(std::integral_constant<int, H * 10 + 3>{});
                    window(std::integral_constant<int, H * 10 + 4>{});
                    window(std::integral_constant<int, H * 10 + 5>{});
                } else {
#pragma unroll
                    for (int i = 0; i < 6 && i < G::pairs_per_wave; i++) issue_one(i, nhb, nbi, ntph);
                    load_ops(H, 6, ops);
                }
                window(std::integral_constant<int, H * 10 + 6>{});
                window(std::integral_constant<int, H * 10 + 7>{});
                window(std::integral_constant<int, H * 10 + 8>{});
                window(std::integral_constant<int, H * 10 + 9>{});
            };
            if (seg_live) {
                if (half == 0) run_half(std::integral_constant<int, 0>{}, TRIM && b == 0);
                else run_half(std::integral_constant<int, 1>{}, false);
            }
            // (a wave whose segment group lies past the end of the data only keeps the workgroup's barriers and its share of the DMAs)
        }
    }
    store_staged(a.nb - 1, 1);                                  // the last half-body's outputs
    // (the last half-body's extra DMAs went to buffer 0, where the next item's first rows go: they have to be in, everybody's)
    if (item + (int)gridDim.x < n_items) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    // (the next item's first rows go to buffer 0, which every wave left before the barrier of the last half-body)
    }
}

}   // namespace sdrxgpu
