"""Lab: the drop-in entry (sdrxgpu_bank_process: host pointers, page-locked, complete on return) of the 10 000-channel bank by
how its channel parts are cut (tuning.host_parts / host_part_ch).  Prints ms per block with IQ + audio and with audio only, and checks every variant's outputs against the first's
(same bits: a channel's result does not depend on the cut).  SDRXGPU_LIB_PATH picks a lab build.

A "round" of k_stage12m is the workgroups the chip holds at once: 2 per CU x 256 CUs = 512, 9 per set of 32 slots at the bank's
lane length -> 56 sets.  cut_by_sets() finds the channel boundaries that give parts of so many sets (a pair of the whole bank's
pairing belongs to the part of its first channel)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from sdrx_amd import bank as sb, chains, synth

ch = chains.chain("10"); L = ch.block_len; n = int(os.environ.get("N_CH", "10000"))
offs = chains.offsets_cycled(n, ch.N)
UNIT, SET = 64, 32


def _firsts():
    """the first channel of every slot of the whole bank's pairing (a pair belongs to the part of its first channel)"""
    open_, first = {}, []
    for c, o in enumerate(offs):
        if open_.get(-o) and o not in (0, -ch.N // 2):
            first.append(open_[-o].pop())
        else:
            open_.setdefault(o, []).append(c)
    for v in open_.values():
        first.extend(v)
    return np.sort(np.array(first))


FIRST = _firsts()


def sets(lo, hi):
    k = int(np.searchsorted(FIRST, hi) - np.searchsorted(FIRST, lo))
    return (k + SET - 1) // SET


def cut_by_sets(targets):
    """channel boundaries: part i = the most channels (whole units) that make at most targets[i] sets; the last part takes the rest"""
    cuts, lo = [], 0
    for t in targets:
        hi = lo
        while hi + UNIT < n and sets(lo, hi + UNIT) <= t:
            hi += UNIT
        if hi <= lo or hi >= n:
            break
        cuts.append(hi)
        lo = hi
    return cuts


tabs = sb.translators_for(ch, offs)
x = synth.synth_iq(9, 0, L, 0.25); o_iq = np.zeros((n, 512), np.complex64); o_au = np.zeros((n, 512), np.float32)
for a in (x, o_iq, o_au):
    sb.host_register(a)

variants = []
def add(name, **tun):
    variants.append((name, tun))

add("automatic (IQ: a part per round of k_stage12m; audio only: whole rounds | rest)")
add("round 5: 4 equal parts", host_parts=4)
add("1 part", host_parts=1)
add("2 equal parts", host_parts=2)
add("3 equal parts", host_parts=3)
for tg in ([56, 56], [28, 56, 56], [56, 56, 34], [112], [84], [112, 28], [140], [14, 28, 56], [28, 28, 28, 28, 28], [56, 84]):
    cuts = cut_by_sets(tg)
    bounds = [0] + cuts + [n]
    real = [sets(a, b) for a, b in zip(bounds[:-1], bounds[1:])]
    add(f"sets {real} (cuts {cuts})", host_parts=len(cuts) + 1, host_part_ch=tuple(cuts))

only = [int(v) for v in sys.argv[1:]]
ref = None
for i, (name, tun) in enumerate(variants):
    if only and i not in only:
        continue
    try:
        b = sb.ChannelBank(tabs, ch.stage_tuples(), tuning=tun)
    except sb.SdrxGpuError as e:
        print(f"[{i}] {name}: refused: {e}", flush=True)
        continue
    if i == 0:
        print("automatic cuts: calls with IQ", b.host_parts(), " audio-only calls", b.host_parts(True), flush=True)
    if ref is None:
        inf = b.info()
        print(f"bank: {n} channels, {inf.m_slots} slots, {inf.k12_bodies} bodies per lane, matrix_pipe {inf.matrix_pipe}", flush=True)
    r = []
    for io in (dict(iq_out=o_iq, audio_out=o_au), dict(iq_out=None, audio_out=o_au)):
        b.reset()
        for _ in range(3):
            b.process(x, **io)
        t0 = time.perf_counter()
        for _ in range(12):
            b.process(x, **io)
        r.append((time.perf_counter() - t0) / 12 * 1e3)
    b.reset()
    o_iq[:] = 0; o_au[:] = 0
    b.process(x, iq_out=o_iq, audio_out=o_au)
    b.process(x, iq_out=o_iq, audio_out=o_au)
    h = (hash(o_iq.tobytes()), hash(o_au.tobytes()))
    if ref is None:
        ref = h
    b.close()
    print(f"[{i}] {name}: IQ + audio {r[0]:.3f} ms, audio only {r[1]:.3f} ms" + ("" if h == ref else "   *** OUTPUTS DIFFER ***"), flush=True)
