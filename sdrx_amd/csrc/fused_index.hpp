// fused_index.hpp -- where the fused stage-2/3 path keeps what (k_stage123m writes, k_tail_c reads).
//
// Stages 2 (19 taps, /5) and 3 (21 taps, /5) of the 10 MS/s chain (reference src/sdrx.cpp:1660-1668; S::calculateOutput
// src/msd.hpp:251-335) are ONE 119-tap filter on the stage-1 outputs, decimating by 25:
//     y3[R] = sum_e H[e] y1[25 R - 94 + e],   H[5 e3 + e2] += h3[e3] h2[e2],   e = 0 .. 118.
// k_stage123m runs it in transposed form: a lane walks one SEGMENT of 240 stage-1 outputs of the call (the 16 lanes of an
// MFMA column group are segments N = 1 200 input samples apart) and adds every new stage-1 output into six rotating
// accumulators ("slots"), each the running sum of one stage-3 output.  240 is not a multiple of 25, so the 16 segments of a
// wave are in five different phases of the 150-output pattern; a per-segment rotation sigma of the slots leaves a residual
// phase delta of at most +-10 outputs, small enough that every slot has a window of idle phases COMMON to all lanes once per
// cycle -- there the wave stores it ("emission", at a half-body boundary) and clears it.  At the end of the segment all
// six slots are dumped.  No warm-up and no history: a stage-3 output whose 119-output window straddles segments (or calls)
// is the sum of the partial sums its segments hold, put together by k_tail_c with the functions below.
//
// Everything here is integer arithmetic on CALL-RELATIVE quantities (so 32 bits are enough on the device):
//   n_y1   stage-1 outputs of the call;  segment s covers the call's outputs 240 s .. 240 s + 239
//   k25    K0 mod 25, K0 = absolute index of the call's first stage-1 output (= stage-1 outputs of all earlier calls)
//   kph    K0 mod 150
//   Rrel   stage-3 output R - r0, r0 = K0 div 25 = stage-3 outputs completed by earlier calls
//   kk     = 25 R + 24 - K0 = 25 Rrel + 24 - k25: the call's stage-1 output that completes R
// tools/emu_fused.py is the same model in Python (checked against a direct convolution on ragged call sequences);
// tests/test_fused_index.py holds this header to it.
#pragma once

#if defined(__HIPCC__)
#define SDRX_HD __host__ __device__ __forceinline__
#else
#define SDRX_HD inline
#endif

namespace sdrxgpu {
namespace fx {

constexpr int LSEG = 240;      // stage-1 outputs per segment
constexpr int NSLOT = 6;
constexpr int PERIOD = 150;    // 25 * NSLOT
constexpr int NTAP = 119;      // 5 * (21 - 1) + 19
constexpr int NENT = 16;       // scratch entries per segment: up to DUMP0 emissions + NSLOT dump slots
constexpr int DUMP0 = 10;
constexpr int WROWS = 160;     // rows of the coefficient table in LDS: 150 + 9 (a half-body never wraps) + 1 spare; 32 bytes each

// per segment class c = s mod 5: (240 s + 25 sigma - delta) mod 150 == 0
SDRX_HD int sigma_of(int c) { return c == 0 ? 0 : c == 1 ? 2 : c == 2 ? 5 : c == 3 ? 1 : 4; }
SDRX_HD int delta_of(int c) { return c == 0 ? 0 : c == 1 ? -10 : c == 2 ? 5 : c == 3 ? -5 : 10; }
// z_s: 240 s - delta_s + 25 sigma_s = 150 z_s
SDRX_HD int zshift(int s)
{
    const int c = s % 5;
    return (LSEG * s - delta_of(c) + 25 * sigma_of(c)) / PERIOD;
}
// 1 - b, b = (the first emission event of an item lies one cycle after floor((K0 - 25) / 25)): see first_cycle() of the model
SDRX_HD int cycle_bias(int k25) { return k25 <= 9 ? 1 : 0; }

// emission schedule of an item (wave-uniform): at the half-body boundary in front of phase n (n = 10, 20 .. 230) slot
// (w / 25) is stored and cleared iff (w % 25) <= 9, w = (kph + n - 35) mod 150
SDRX_HD bool emits(int kph, int n, int *slot)
{
    const int w = (kph + n - 35 + 2 * PERIOD) % PERIOD;
    *slot = w / 25;
    return (w % 25) <= 9 && n >= 10 && n <= 230;
}

SDRX_HD int kk_of(int k25, int Rrel) { return 25 * Rrel + 24 - k25; }

// entry (0 .. 15) of segment s that holds the segment's share of output Rrel, given that the output's window overlaps the
// segment; *n_c = the phase of the segment at which the output completes (>= 240: not in this segment)
SDRX_HD int locate(int k25, int r0m6, int s, int Rrel, int *n_c)
{
    const int c = s % 5;
    const int nc = kk_of(k25, Rrel) - LSEG * s;
    *n_c = nc;
    const int p = (Rrel + r0m6 + sigma_of(c)) % NSLOT;
    if (nc >= LSEG) return DUMP0 + p;
    const int nc0 = nc + delta_of(c);
    const int nE = (nc0 + 11 + 9) / 10 * 10;                 // smallest multiple of 10 >= nc0 + 11   (nc0 + 11 >= 1)
    if (nE > 230 || nE < 10) return DUMP0 + p;
    return Rrel + cycle_bias(k25) + sigma_of(c) - NSLOT * zshift(s);
}

// does the window of output Rrel (the call's stage-1 outputs kk - 118 .. kk) overlap the live part of segment s?
SDRX_HD bool overlaps(int k25, int n_y1, int s, int Rrel)
{
    const int kk = kk_of(k25, Rrel);
    const int lo = LSEG * s, hi = (LSEG * (s + 1) < n_y1) ? LSEG * (s + 1) : n_y1;
    return kk >= lo && kk - (NTAP - 1) < hi;
}

// inverse: the output an entry of segment s holds (Rrel; may be negative: an output of an earlier call -- never valid), or
// `false` if the entry holds nothing of this call.  Valid means: locate() of that output is this entry and its window
// overlaps the segment's live part.
SDRX_HD bool entry_output(int k25, int r0m6, int n_y1, int s, int entry, int *Rrel, int *n_c)
{
    const int c = s % 5;
    int R;
    if (entry < DUMP0) {
        R = entry - cycle_bias(k25) - sigma_of(c) + NSLOT * zshift(s);
    } else {
        // what a dump slot can hold: an output still in flight at the end of the segment (completes at phase 240 .. 357: its window
        // begins at phase <= 239) or one that completed so late that its emission boundary lies past the item (phase >= 210):
        // the one output of residue class (slot - sigma) that completes in the 150 phases from 208 on
        const int p = entry - DUMP0;
        const int num = 208 + LSEG * s - 24 + k25;                 // 25 R >= num
        const int Rmin = (num + 24) / 25;
        int adj = (p - sigma_of(c) - r0m6 - Rmin) % NSLOT;
        if (adj < 0) adj += NSLOT;
        R = Rmin + adj;
    }
    if (R < 0) return false;
    int nc;
    if (locate(k25, r0m6, s, R, &nc) != entry || !overlaps(k25, n_y1, s, R)) return false;
    *Rrel = R;
    *n_c = nc;
    return true;
}

}   // namespace fx
}   // namespace sdrxgpu
