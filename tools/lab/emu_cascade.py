#!/usr/bin/env python3
"""fp32 emulation of the whole decimator for ONE channel in the orders the kernels use, against float64:
  ref      the reference's portable order in every stage (mixer product, folded outside-in FIRs)            = vector kernels
  m12      k_stage12m + k_tail_s: stage 1 as P -+ Q fma chains (K steps outside-in), stage 2 as ONE fma chain in time order
           (transposed form), stages 3-4 folded                                                              = today's default
  m123     the cascade in registers: the same, and stage 3 as one fma chain in time order as well, stage 4 folded
  m119     k_stage123m: stages 2-3 as one 119-tap chain in time order (SDRXGPU_FLAG_FUSED_STAGE3)
Prints the comparator of the tests -- max |err| / max(|ref|, rms) over the channel's outputs -- for each, taking float64 as the
reference, and the same between ref and each (what the parity tests see up to the reference's own rounding).
  tools/lab/emu_cascade.py [fixture] [channel offset] [n_final_outputs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, regen_blocks, rel_err      # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "g9_nb_a030_n250"
g = load_golden(name, "pairs")
ch, blocks = regen_blocks(g)
offs = [int(o) for o in g["offsets"]]
want = int(sys.argv[2]) if len(sys.argv) > 2 else offs[1]
n_fin = int(sys.argv[3]) if len(sys.argv) > 3 else 300
ci = offs.index(want)
t = g["tables"][ci]
hs = [s.h.astype(np.float32) for s in ch.stages]
ms = [s.m for s in ch.stages]
N = ch.N
n_in = n_fin * 625 + 9000
x = np.concatenate(blocks)[:n_in].astype(np.complex64)
f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def win(v, m, T):
    n = (len(v) - T) // m + 1
    return v[np.arange(n)[:, None] * m + np.arange(T)[None, :]]


def fir_exact(v, h, m):
    return (win(v, m, len(h)) * h.astype(np.float64)[None, :]).sum(axis=1)


def fir_folded(v, h, m):
    """reference, portable: sum_i (w[i] + w[T-1-i]) h[i] + w[half] h[half], one accumulator, separate roundings (no fma)"""
    T = len(h)
    half = (T - 1) // 2
    out = []
    for part in (v.real.astype(f32), v.imag.astype(f32)):
        w = win(part, m, T)
        acc = np.zeros(len(w), f32)
        for i in range(half):
            acc = (acc + ((w[:, i] + w[:, T - 1 - i]).astype(f32) * h[i]).astype(f32)).astype(f32)
        acc = (acc + (w[:, half] * h[half]).astype(f32)).astype(f32)
        out.append(acc)
    return out[0].astype(np.complex64) + 1j * out[1].astype(np.complex64)


def fir_chain(v, h, m):
    """transposed form: one fma chain over the window in time order 0 .. T-1 (first term a product)"""
    T = len(h)
    out = []
    for part in (v.real.astype(f32), v.imag.astype(f32)):
        w = win(part, m, T)
        acc = (w[:, 0] * h[0]).astype(f32)
        for i in range(1, T):
            acc = fma(w[:, i], np.full(len(w), h[i], f32), acc)
        out.append(acc)
    return out[0].astype(np.complex64) + 1j * out[1].astype(np.complex64)


def stage1_matrix(tab):
    """k_stage12m for the channel with table `tab` run as the slot's own channel: y = (Pr - Qi) + j (Pi + Qr)"""
    T, m = len(hs[0]), ms[0]
    tt = np.tile(tab, len(x) // N + 1)[: len(x)]
    tr, ti = win(tt.real.astype(f32), m, T), win(tt.imag.astype(f32), m, T)
    xr, xi = win(x.real.astype(f32), m, T), win(x.imag.astype(f32), m, T)
    order = [16, 0, 1, 2, 3, 12, 13, 14, 15, 4, 5, 6, 7, 8, 9, 10, 11]

    def chain(a, b):
        acc = None
        for i in order:
            bh = (b[:, i] * hs[0][i]).astype(f32)
            acc = (a[:, i] * bh).astype(f32) if acc is None else fma(a[:, i], bh, acc)
        return acc
    Pr, Pi, Qr, Qi = chain(tr, xr), chain(tr, xi), chain(ti, xr), chain(ti, xi)
    return ((Pr - Qi).astype(f32)).astype(np.complex64) + 1j * ((Pi + Qr).astype(f32)).astype(np.complex64)


def stage1_ref(tab):
    tt = np.tile(tab, len(x) // N + 1)[: len(x)]
    re = ((x.real * tt.real).astype(f32) - (x.imag * tt.imag).astype(f32)).astype(f32)
    im = ((x.real * tt.imag).astype(f32) + (x.imag * tt.real).astype(f32)).astype(f32)
    return fir_folded(re.astype(np.complex64) + 1j * im.astype(np.complex64), hs[0], ms[0])


def exact():
    xm = x.astype(np.complex128) * np.tile(t.astype(np.complex128), len(x) // N + 1)[: len(x)]
    v = xm
    for h, m in zip(hs, ms):
        v = fir_exact(v, h, m)
    return v


h23 = np.zeros(5 * (len(hs[2]) - 1) + len(hs[1]), np.float64)
for e3, a in enumerate(hs[2].astype(np.float64)):
    for e2, b in enumerate(hs[1].astype(np.float64)):
        h23[5 * e3 + e2] += a * b
h23 = h23.astype(np.float32)

ex = exact()
paths = {
    "ref": fir_folded(fir_folded(fir_folded(stage1_ref(t), hs[1], ms[1]), hs[2], ms[2]), hs[3], ms[3]),
    "m12": fir_folded(fir_folded(fir_chain(stage1_matrix(t), hs[1], ms[1]), hs[2], ms[2]), hs[3], ms[3]),
    "m123": fir_folded(fir_chain(fir_chain(stage1_matrix(t), hs[1], ms[1]), hs[2], ms[2]), hs[3], ms[3]),
    "m119": fir_folded(fir_chain(stage1_matrix(t), h23, 25), hs[3], ms[3]),
    "v12_c3": fir_folded(fir_chain(fir_folded(stage1_ref(t), hs[1], ms[1]), hs[2], ms[2]), hs[3], ms[3]),
}
n = min(len(v) for v in paths.values())
n = min(n, len(ex))
print(f"{name}, channel {want:+d} ({n} final outputs): comparator against float64 / against the reference order")
for k, v in paths.items():
    print(f"  {k:7s} {rel_err(v[:n], ex[:n]):.3e}   {rel_err(v[:n], paths['ref'][:n]):.3e}")
