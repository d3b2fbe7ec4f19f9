#!/usr/bin/env python3
"""Index-math model of the fused stage-2/3 path (k_stage123m + k_tail_c), in plain Python / numpy -- the thing that was
debugged on the CPU before any kernel was written.  It models, per channel:

  * the kernel side: every lane walks ONE segment of LSEG = 240 stage-1 outputs of the call, feeds each into six rotating
    accumulators through a per-lane row of the 150 x 6 coefficient table, stores the accumulator that the wave-uniform
    schedule says is idle at a half-body boundary ("emission"), and dumps all six at the end of the segment;
  * the tail side: every stage-3 output of the call is put together from at most one entry of the segment it completes in
    plus the partial sums ("heads") earlier segments / earlier calls left -- by the index functions below, which are what
    sdrx_amd/csrc/fused_index.hpp restates in C++ (tests/test_fused_index.py compares the two).

Run as a script it checks the model against a direct float64 convolution on random ragged call sequences."""
import numpy as np

LSEG = 240          # stage-1 outputs per segment (N / m1 = 1200 / 5)
NSLOT = 6
PERIOD = 25 * NSLOT  # 150: the coefficient pattern repeats every 150 stage-1 outputs
NTAP = 119          # taps of stage 2 (19, /5) followed by stage 3 (21, /5) as one filter at the stage-1 rate, /25
NENT = 16           # entries per segment in the scratch: up to 10 emissions + 6 dump slots
DUMP0 = 10          # first dump entry
# per segment class (s mod 5): sigma (rotation of the slots) and delta (what is left of the segment's phase offset, in
# stage-1 outputs): (240 s + 25 sigma) mod 150 = delta mod 150, |delta| <= 10
SIGMA = (0, 2, 5, 1, 4)
DELTA = (0, -10, 5, -5, 10)
for _c in range(5):
    assert (LSEG * _c + 25 * SIGMA[_c] - DELTA[_c]) % PERIOD == 0


def zshift(s):
    """z_s with 240 s - delta_s + 25 sigma_s = 150 z_s."""
    c = s % 5
    v = LSEG * s - DELTA[c] + 25 * SIGMA[c]
    assert v % PERIOD == 0
    return v // PERIOD


def combined_taps(h2, h3):
    """H[e], e = 5 e3 + e2: stage 2 then stage 3 as one filter on stage-1 outputs (float64; the kernel rounds to fp32)."""
    H = np.zeros(NTAP)
    for e3 in range(len(h3)):
        for e2 in range(len(h2)):
            H[5 * e3 + e2] += float(h3[e3]) * float(h2[e2])
    return H


def coef_table(H):
    """W[tau][p] = H[(tau - 25 p + 94) mod 150] (0 where that index is >= 119): coefficient of physical slot p for a lane whose
    phase is tau = (K0 + delta_s + n) mod 150."""
    W = np.zeros((PERIOD, NSLOT))
    for tau in range(PERIOD):
        for p in range(NSLOT):
            e = (tau - 25 * p + 94) % PERIOD
            if e < NTAP:
                W[tau, p] = H[e]
    return W


def first_cycle(K0):
    """Cycle number of the first emission event of an item (events: one per 25 stage-1 outputs, at the half-body boundary n,
    10 <= n <= 230, with ((K0 + n - 35) mod 25) <= 9)."""
    g = K0 - 25
    return g // 25 + (0 if g % 25 <= 9 else 1)


def emits(K0, n):
    """(emit?, physical slot) at half-body boundary n (a multiple of 10)."""
    w = (K0 + n - 35) % PERIOD
    return (w % 25 <= 9 and 10 <= n <= 230), w // 25


def locate(K0, s, R):
    """Entry of segment s's 16 that holds (this segment's share of) stage-3 output R, given that R's window overlaps the segment.
    Returns (entry index, would-be completion phase n_c)."""
    c = s % 5
    n_c = 25 * R + 24 - K0 - LSEG * s
    p = (R + SIGMA[c]) % NSLOT
    if n_c >= LSEG:
        return DUMP0 + p, n_c                       # still in flight at the end of the segment
    n_c0 = n_c + DELTA[c]
    n_E = -(-(n_c0 + 11) // 10) * 10                # smallest multiple of 10 >= n_c0 + 11
    if n_E > 230 or n_E < 10:
        return DUMP0 + p, n_c                       # finished, but its emission boundary lies past the item: in the dump
    E = R + SIGMA[c] - NSLOT * zshift(s) - first_cycle(K0)
    return E, n_c


def kernel_model(y1, K0, W):
    """One channel, one call.  y1: the call's new stage-1 outputs (absolute indices K0 .. K0 + len - 1).  Returns
    scratch[n_seg][NENT] (complex)."""
    n_y1 = len(y1)
    n_seg = -(-n_y1 // LSEG)
    scratch = np.full((max(n_seg, 1), NENT), np.nan + 0j)
    for s in range(n_seg):
        c = s % 5
        acc = np.zeros(NSLOT, dtype=complex)
        E = 0
        n_end = min(LSEG, n_y1 - LSEG * s)
        for n in range(LSEG):
            if n % 10 == 0:
                em, p = emits(K0, n)
                if em:
                    scratch[s, E] = acc[p]
                    acc[p] = 0
                    E += 1
            if n < n_end:                             # (a lane past the end of the data reads rows of zeros)
                tau = (K0 + DELTA[c] + n) % PERIOD
                acc += y1[LSEG * s + n] * W[tau]
        assert E <= DUMP0
        scratch[s, DUMP0:] = acc
    return scratch


def tail_model(scratch, K0, n_y1, carry, r_first):
    """Stage-3 outputs that complete in this call (R = r_first ...) and the carry for the next one.  carry: dict R -> partial
    sum of outputs still in flight when the previous call ended."""
    K_end = K0 + n_y1
    n_seg = -(-n_y1 // LSEG)
    out = []
    R = r_first
    new_carry = {}
    while 25 * R - 94 < K_end:                       # every output whose window has begun
        v = carry.get(R, 0.0)
        s_lo = max(0, (25 * R - 94 - K0) // LSEG)
        s_hi = min(n_seg - 1, (25 * R + 24 - K0) // LSEG)
        for s in range(s_lo, s_hi + 1):
            seg_lo, seg_hi = K0 + LSEG * s, min(K0 + LSEG * (s + 1), K_end)
            if 25 * R + 24 < seg_lo or 25 * R - 94 >= seg_hi:
                continue
            e, _ = locate(K0, s, R)
            v = v + scratch[s, e]
        if 25 * R + 24 < K_end:
            out.append(v)
        else:
            new_carry[R] = v
        R += 1
    return np.array(out), new_carry


def main():
    rng = np.random.default_rng(7)
    h2 = rng.standard_normal(19)
    h3 = rng.standard_normal(21)
    H = combined_taps(h2, h3)
    W = coef_table(H)
    worst = 0.0
    for trial in range(40):
        total = int(rng.integers(3000, 90000))
        y1 = rng.standard_normal(total) + 1j * rng.standard_normal(total)
        # reference: y3[R] = sum_e H[e] y1[25 R - 94 + e], zero history
        ypad = np.concatenate([np.zeros(200, dtype=complex), y1])
        n3 = total // 25
        ref = np.array([np.dot(H, ypad[200 + 25 * R - 94: 200 + 25 * R - 94 + NTAP]) for R in range(n3)])
        # ragged calls
        cuts = [0]
        while cuts[-1] < total:
            step = int(rng.choice([1, 7, 24, 25, 26, 119, 120, 239, 240, 241, 1000, 3840, 64000, 12345]))
            cuts.append(min(total, cuts[-1] + step))
        got, carry, r_next = [], {}, 0
        for a, b in zip(cuts[:-1], cuts[1:]):
            sc = kernel_model(y1[a:b], a, W)
            o, carry = tail_model(sc, a, b - a, carry, r_next)
            got.extend(o)
            r_next += len(o)
        got = np.array(got)
        assert len(got) == n3, (len(got), n3)
        worst = max(worst, float(np.abs(got - ref).max() / np.abs(ref).max()))
    print("fused index model: 40 ragged streams, worst deviation from the direct convolution %.2e" % worst)
    assert worst < 1e-12


if __name__ == "__main__":
    main()
