"""Fuzz of the matrix-pipe path on the GPU (not collected by pytest; run by hand: `python tests/fuzz_gpu_matrix.py <seed>
<seconds>` from the repo root).  Random banks of the 10 MS/s chain (channel count, offset pattern incl. repeated and
untranslated channels), call splits, lane lengths, workgroup shapes, --ftfir and LF AGC through k_stage12m: the same
stream in one call per block and in a random split must give the same bits; every workgroup shape the same bits as the
default one; three channels of every case against the oracle (1e-5), and all of them against the vector kernels (1e-5:
two summation orders of the same arithmetic, each within its own error of the exact result).  Round 3, final build: 9 215
cases in 420 s on one MI355X (seed 11; 8 063 more with seed 7), worst 4.5e-6 against the oracle, 5.9e-6 against the vector
kernels, every split and every workgroup shape bit-identical.  Round 5: the scratch stores' streaming cut drawn at random (7 648
cases, seed 77); a third argument "6" runs the 6 MS/s chain through k_stage12m6 instead: 8 003 cases in 300 s (seed 66), every
split and lane length bit-identical, worst 4.5e-6 against the oracle, 5.7e-6 against the vector kernels (profiles/r05_m6.txt).
Round 6: the host entry's channel parts drawn at random (automatic / equal / explicit cuts; banks up to 7 000 channels)."""
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
RATE = sys.argv[3] if len(sys.argv) > 3 else "10"            # "6": k_stage12m6 (bodies of 150 samples, 24 per segment; one workgroup shape)
ch = chains.chain(RATE)
L = ch.block_len
while time.time() < t_end:
    ftfir = RATE == "10" and rng.random() < 0.15
    n_ch = rng.choice([1, 2, 3, 17, 31, 32, 33, 64, 65, 100, 257, 1023, 1024, 1025, 1500, 2049, 3000, 2049, 3000, 4500, 7000])
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
    stream = synth.synth_iq(rng.randrange(1 << 30), 0, total, 0.25, car, [0.3] * len(car), ch.N, ch.fs // 1000)
    cuts, p = [], 0
    while p < total:
        n = min(total - p, rng.choice([L, L, L // 2, 625, 1, 0, 99999, ch.m_total * 3 + 1]))
        cuts.append(stream[p:p + n]); p += n
    whole = [stream[i:i + L] for i in range(0, total, L)]
    lf = rng.random() < 0.3
    gs, gt = rng.choice([(0, 0), (2, 2), (4, 2), (2, 4)] if RATE == "10" else [(0, 0), (2, 2)])
    hint = rng.choice([0, 0, WHOLE, 1, 2, 3, 4, 6] if RATE == "10" else [0, 0, WHOLE, 1, 2, 3, 4, 6, 8, 12])
    keep = rng.choice([0, 1, 1 + rng.randrange(0, 40), 0xFFFFFFFF])      # tuning.y2_keep_mb: which sets' scratch stores stream (same bits)

    def run(tuning, hint_, blocks):
        b = sb.ChannelBank(tabs, ch.stage_tuples(), block_hint=hint_, lf_agc=lf, ftfir=ftfir, tuning=tuning)
        info = b.info()
        outs = [b.process(x) for x in blocks]
        b.close()
        return np.concatenate([o[0] for o in outs], 1), np.concatenate([o[1] for o in outs], 1), info
    # round 6: the host entry's channel parts drawn at random too (banks of 2 048 channels and more run in parts: automatic cuts
    # by rounds, equal parts, or explicit boundaries anywhere; a pair belongs to the part of its first channel)
    parts = {}
    if n_ch >= 2048:
        how = rng.choice(["auto", "equal", "explicit", "explicit"])
        if how == "equal":
            parts = {"host_parts": rng.randrange(1, 6)}
        elif how == "explicit":
            b_ = sorted(set(rng.randrange(64, n_ch - 64) // 64 * 64 for _ in range(rng.randrange(1, 6))))
            parts = {"host_parts": len(b_) + 1, "host_part_ch": tuple(b_)}
    ref = run({"matrix_pipe": 2, "y2_keep_mb": 0xFFFFFFFF, "host_parts": 1}, 0, whole)
    got = run({"matrix_pipe": 2, "m_slot_groups": gs, "m_seg_groups": gt, "y2_keep_mb": keep, **parts}, hint, cuts)
    vec = run({"matrix_pipe": 1}, 0, whole)
    assert ref[2].matrix_pipe == 1 and got[2].matrix_pipe == 1 and vec[2].matrix_pipe == 0
    ok = np.array_equal(ref[0].view(np.uint32), got[0].view(np.uint32)) and np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))
    den = np.maximum(np.abs(vec[0]), np.sqrt(np.mean(np.abs(vec[0]) ** 2, axis=1, keepdims=True)) + 1e-30)
    e_vec = float(np.max(np.abs(got[0] - vec[0]) / den)) if got[0].size else 0.0
    pick = sorted(set([0, n_ch // 2, n_ch - 1]))
    errs = []
    for c in pick:
        y = orc.msd(tabs[c], ch.stage_tuples(), use_ftfir=ftfir and tabs[c] is not None).decimate(stream)
        d = np.maximum(np.abs(y), np.sqrt(np.mean(np.abs(y) ** 2)) + 1e-30)
        errs.append(float(np.max(np.abs(got[0][c, :len(y)] - y) / d)) if len(y) else 0.0)
    n_cases += 1
    good = ok and max(errs) < 1e-5 and e_vec < 1e-5
    print(("OK  " if good else "FAIL"), "ftfir" if ftfir else "     ", "n_ch", n_ch, kind, "hint", hint, "shape", (gs, gt), "keep", keep, "lf", lf, "calls", len(cuts),
          "slots", got[2].m_slots, "parts", parts.get("host_part_ch", parts.get("host_parts", "auto")), "oracle %.2e" % max(errs), "vector kernels %.2e" % e_vec, flush=True)
    if not good:
        sys.exit(1)
print("cases", n_cases)
