#!/usr/bin/env python3
"""fp32 emulation of stage 1 of the matrix-pipe kernel (k_stage12m: P / Q sums, exact fma chains) for the WEAK partner of a conjugate
pair under a strong carrier, in several accumulation orders, against the same stage in float64 -- and of the reference's order
(mixer product rounded, folded FIR; src/msd.hpp:134-135, 309-331) for scale.  The error that matters is the part that survives
stages 2-4, i.e. the error's spectrum inside the channel: reported as the rms error after a /125 moving decimation proxy
(box-car over 125 stage-1 outputs) relative to the channel's own rms there, and raw.
  tools/lab/emu_matrix_orders.py [fixture] [n_stage1_outputs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, regen_blocks      # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "g8_one_a090_n050"
n_out = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
g = load_golden(name, "pairs")
ch, blocks = regen_blocks(g)
offs = [int(o) for o in g["offsets"]]
x = np.concatenate(blocks)[: 5 * n_out + 32].astype(np.complex64)
h = ch.stages[0].h.astype(np.float32)
T, M = len(h), ch.stages[0].m
N = ch.N
f32 = np.float32


def fma(a, b, c):          # exact product, one rounding (double holds a 24 x 24-bit product exactly)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def windows(v):            # [n_out, T]: window of output k = samples 5k .. 5k+16 (oldest first)
    idx = np.arange(n_out)[:, None] * M + np.arange(T)[None, :]
    return v[idx]


def stage1_exact(t):
    xm = x.astype(np.complex128) * np.tile(t.astype(np.complex128), len(x) // N + 1)[: len(x)]
    return (windows(xm) * h.astype(np.float64)[None, :]).sum(axis=1)


def stage1_reference(t):
    """portable build: mixer product with separate roundings, folded FIR, one accumulator per part, i = 0 .. half"""
    tt = np.tile(t, len(x) // N + 1)[: len(x)]
    re = (x.real * tt.real).astype(f32) - (x.imag * tt.imag).astype(f32)
    im = (x.real * tt.imag).astype(f32) + (x.imag * tt.real).astype(f32)
    out = []
    for part in (re.astype(f32), im.astype(f32)):
        w = windows(part)
        acc = np.zeros(n_out, f32)
        half = (T - 1) // 2
        for i in range(half):
            acc = (acc + ((w[:, i] + w[:, T - 1 - i]).astype(f32) * h[i]).astype(f32)).astype(f32)
        acc = (acc + (w[:, half] * h[half]).astype(f32)).astype(f32)
        out.append(acc)
    return out[0].astype(np.float64) + 1j * out[1].astype(np.float64)


def stage1_matrix(t_pos, order, split=None, sign=-1):
    """k_stage12m for the channel whose table is conj(t_pos) (sign = -1: the conjugate partner) or t_pos itself (+1):
    P = sum h t_r x, Q = sum h t_i x as fma chains over the taps in `order` (tap 16 first, as a product), y = P -+ Q.
    split: list of tap lists -- separate chains from zero, added pairwise at the end (tree)."""
    tt = np.tile(t_pos, len(x) // N + 1)[: len(x)]
    tr, ti = windows(tt.real.astype(f32)), windows(tt.imag.astype(f32))
    xr, xi = windows(x.real.astype(f32)), windows(x.imag.astype(f32))

    def chain(a, b, taps, first_as_product):
        acc = None
        for n_, i in enumerate(taps):
            bh = (b[:, i] * h[i]).astype(f32)                    # the tap rides on the B operand (one rounding)
            if acc is None:
                acc = (a[:, i] * bh).astype(f32) if first_as_product else fma(a[:, i], bh, np.zeros(n_out, f32))
            else:
                acc = fma(a[:, i], bh, acc)
        return acc

    def total(a, b):
        if split is None:
            return chain(a, b, order, True)
        parts = [chain(a, b, taps, k == 0) for k, taps in enumerate(split)]
        while len(parts) > 1:
            parts = [(parts[i] + parts[i + 1]).astype(f32) if i + 1 < len(parts) else parts[i] for i in range(0, len(parts), 2)]
        return parts[0]

    Pr, Pi, Qr, Qi = total(tr, xr), total(tr, xi), total(ti, xr), total(ti, xi)
    if sign < 0:     # conjugate table: (Pr + Qi) + j (Pi - Qr)
        return (Pr + Qi).astype(f32).astype(np.float64) + 1j * (Pi - Qr).astype(f32).astype(np.float64)
    return (Pr - Qi).astype(f32).astype(np.float64) + 1j * (Pi + Qr).astype(f32).astype(np.float64)


def report(label, y, exact):
    e = y - exact
    box = 125
    n = (len(e) // box) * box
    eb = e[:n].reshape(-1, box).mean(axis=1)
    sb_ = exact[:n].reshape(-1, box).mean(axis=1)
    print(f"  {label:58s} raw rms err / rms {np.sqrt(np.mean(abs(e) ** 2)) / np.sqrt(np.mean(abs(exact) ** 2)):.3e}   "
          f"in-band (box {box}) rms err / rms {np.sqrt(np.mean(abs(eb) ** 2)) / np.sqrt(np.mean(abs(sb_) ** 2)):.3e}")
    return np.sqrt(np.mean(abs(eb) ** 2)) / np.sqrt(np.mean(abs(sb_) ** 2))


strong = offs.index(37) if 37 in offs else 0
t_pos = g["tables"][strong]
print(f"{name}: carrier on +37, weak partner -37; {n_out} stage-1 outputs")
exact_w = stage1_exact(np.conj(t_pos))
natural = [16] + list(range(16))
now = [16, 0, 1, 2, 3, 12, 13, 14, 15, 4, 5, 6, 7, 8, 9, 10, 11]
updown = [16, 0, 1, 2, 3, 15, 14, 13, 12, 4, 5, 6, 7, 11, 10, 9, 8]
strict = [16, 0, 15, 1, 14, 2, 13, 3, 12, 4, 11, 5, 10, 6, 9, 7, 8]
res = {}
res["ref"] = report("reference order (portable build's arithmetic)", stage1_reference(np.conj(t_pos)), exact_w)
res["natural"] = report("matrix, taps 16, 0 .. 15", stage1_matrix(t_pos, natural), exact_w)
res["now"] = report("matrix, K steps outside-in (round 4, in the library)", stage1_matrix(t_pos, now), exact_w)
res["updown"] = report("matrix, K steps outside-in, inner steps mirrored", stage1_matrix(t_pos, updown), exact_w)
res["strict"] = report("matrix, strictly outside-in", stage1_matrix(t_pos, strict), exact_w)
res["two"] = report("matrix, two chains (outer 16,0-3,12-15 | inner 4-11) + add", stage1_matrix(t_pos, None, split=[[16, 0, 1, 2, 3, 12, 13, 14, 15], [4, 5, 6, 7, 8, 9, 10, 11]]), exact_w)
res["two_b"] = report("matrix, two chains (16,0-7 | 8-15) + add", stage1_matrix(t_pos, None, split=[[16, 0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11, 12, 13, 14, 15]]), exact_w)
res["four"] = report("matrix, four chains (one per K step) + tree", stage1_matrix(t_pos, None, split=[[16, 0, 1, 2, 3], [12, 13, 14, 15], [4, 5, 6, 7], [8, 9, 10, 11]]), exact_w)
print("  ratios to the library's order:", {k: round(v / res["now"], 3) for k, v in res.items()})
