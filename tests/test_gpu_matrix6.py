"""k_stage12m6: mixer + stage 1 of the 6 MS/s chain ((15, 53)(5, 21), reference src/sdrx.cpp:1651-1658, src/msd.hpp:132-155, 228-335)
on the f32 matrix pipe (sdrx_amd/csrc/kernels_mfma6.hpp).  Like k_stage12m for the 10 MS/s chain: another summation order than the
vector kernels', so held to the reference's golden vector, to the oracle at 1e-5 -- and to ITSELF bit for bit wherever the
reference is deterministic: any cut of the stream into calls, any lane length, whoever else is in the bank, either tail path."""
import numpy as np
import pytest

import glob
import os

from conftest import GOLD, RTOL, load_golden, regen_blocks, rel_err, rel_err_rows, margin
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu

WHOLE = 0xFFFFFFFF
OFF, REQ = 1, 2
RATE = "6"


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _run(ch, tabs, blocks, tuning=None, block_hint=0, **kw):
    t = {"matrix_pipe": REQ}
    t.update(tuning or {})
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), block_hint=block_hint, tuning=t, **kw)
    info = bank.info()
    outs = [bank.process(x) for x in blocks]
    bank.close()
    return np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1), info


def _cut(stream, lens):
    out, p = [], 0
    for n in lens:
        n = min(n, len(stream) - p)
        out.append(stream[p:p + n])
        p += n
    assert p == len(stream)
    return out


def _oracle(oracle, ch, tabs, blocks, **kw):
    ob = oracle.bank(tabs, ch.stage_tuples(), **kw)
    r = [ob.process(x, threads=8) for x in blocks]
    return np.concatenate([v[0] for v in r], axis=1), np.concatenate([v[1] for v in r], axis=1)


def test_golden_vector_of_the_compiled_reference():
    g = load_golden("g7_fs6")
    ch, blocks = regen_blocks(g)
    offs = [int(o) for o in g["offsets"]]
    tabs = [g["tables"][i] if o != 0 else None for i, o in enumerate(offs)]
    iq, au, info = _run(ch, tabs, blocks)
    assert info.matrix_pipe == 1 and iq.shape == g["iq"].shape
    e_iq, e_au = rel_err_rows(iq, g["iq"]), rel_err_rows(au, g["audio_lf"])
    margin("matrix6_golden_g7_fs6", iq=e_iq, audio=e_au)
    assert e_iq <= RTOL and e_au <= RTOL


OFFSETS = [
    list(range(-20, 21)),                                  # 20 conjugate pairs + the untranslated channel
    [5, -5, 5, -5, 7, 9, 11, 0, 0, -7, -360, 13],           # repeated offsets, loners, two untranslated, -360 (N / 2: its own conjugate)
    [3],                                                   # a single channel
    [4, -4],                                               # one pair
    list(range(1, 38)),                                    # no partners at all
    chains.offsets_cycled(300, 720),                       # several slot-group sets
]


@pytest.mark.parametrize("offs", OFFSETS)
def test_matches_the_oracle_in_ragged_calls(oracle, offs):
    ch = chains.chain(RATE)
    tabs = sb.translators_for(ch, offs)
    L, M = ch.block_len, ch.m_total
    stream = synth.synth_iq(31, 0, 2 * L + 4321, 0.25, [t for t in tabs[:1] if t is not None], [0.3], ch.N, ch.fs // 1000)
    lens = [L, 0, M - 1, M, 777, L // 2, 10 ** 9]
    iq, au, info = _run(ch, tabs, _cut(stream, lens))
    assert info.matrix_pipe == 1 and info.fused_tail == 0 and info.pair_lanes == 0
    riq, rau = _oracle(oracle, ch, tabs, _cut(stream, lens))
    assert iq.shape == riq.shape
    e_iq, e_au = rel_err_rows(iq, riq), rel_err_rows(au, rau)
    margin("matrix6_vs_oracle_%dch" % len(offs), iq=e_iq, audio=e_au)
    assert e_iq <= RTOL and e_au <= RTOL


def test_close_to_the_vector_kernels_but_not_equal():
    ch = chains.chain(RATE)
    offs = list(range(-20, 21))
    tabs = sb.translators_for(ch, offs)
    blocks = [synth.synth_iq(5, 0, ch.block_len, 0.25, [tabs[3]], [0.3], ch.N, ch.fs // 1000)]
    m = _run(ch, tabs, blocks)
    v = _run(ch, tabs, blocks, tuning={"matrix_pipe": OFF})
    assert v[2].matrix_pipe == 0
    assert rel_err_rows(m[0], v[0]) <= RTOL
    assert not np.array_equal(_bits(m[0]), _bits(v[0]))


@pytest.mark.parametrize("offs", OFFSETS[:2])
def test_any_cut_of_the_stream_any_lane_length_either_tail_gives_the_same_bits(offs):
    ch = chains.chain(RATE)
    tabs = sb.translators_for(ch, offs)
    L, M = ch.block_len, ch.m_total
    stream = synth.synth_iq(77, 0, 2 * L + 1234, 0.25, [t for t in tabs[:1] if t is not None], [0.3], ch.N, ch.fs // 1000)
    whole = _run(ch, tabs, _cut(stream, [L, L, 10 ** 9]), block_hint=WHOLE)
    assert whole[2].k12_bodies == 24
    for lens in ([L, 0, M - 1, M, 777, L // 2, 10 ** 9], [1, 74, 75, 76, 149, 150, 151, 3599, 3600, 3601, 5000, 10 ** 9], [L // 3] * 7):
        for hint in (0, 1, 3, 8):
            got = _run(ch, tabs, _cut(stream, lens), block_hint=hint)
            assert np.array_equal(_bits(whole[0]), _bits(got[0])) and np.array_equal(_bits(whole[1]), _bits(got[1])), (lens[:3], hint)
    # k_tail_agc behind it instead of k_tail_s + k_agc: the same bits (the tail kernels read this kernel's scratch as the vector kernel's)
    got = _run(ch, tabs, _cut(stream, [L, 777, 10 ** 9]), tuning={"tail_agc": 2})
    assert got[2].tail_agc > 0
    assert np.array_equal(_bits(whole[0]), _bits(got[0])) and np.array_equal(_bits(whole[1]), _bits(got[1]))


def test_a_channels_bits_do_not_depend_on_who_else_is_in_the_bank():
    ch = chains.chain(RATE)
    L = ch.block_len
    stream = [synth.synth_iq(79, 0, L, 0.25)]
    t = sb.translators_for(ch, [37, -37, 0, 359])
    alone = [_run(ch, [t[0]], stream)[0][0], _run(ch, [t[1]], stream)[0][0], _run(ch, [t[3], t[2]], stream)[0][1], _run(ch, [t[3]], stream)[0][0]]
    both = _run(ch, t, stream)[0]
    many = _run(ch, t * 9 + [t[0]], stream)[0]
    swapped = _run(ch, [t[1], t[0]], stream)[0]
    for i in range(4):
        assert np.array_equal(_bits(alone[i]), _bits(both[i]))
        for k in range(9):
            assert np.array_equal(_bits(alone[i]), _bits(many[4 * k + i]))
    assert np.array_equal(_bits(alone[0]), _bits(many[36]))
    assert np.array_equal(_bits(alone[0]), _bits(swapped[1])) and np.array_equal(_bits(alone[1]), _bits(swapped[0]))


def test_large_bank_host_parts_multibank_and_refusals(oracle):
    ch = chains.chain(RATE)
    n_ch = 2600
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = [synth.synth_iq(9, b * L, L, 0.25, [tabs[650]], [0.2], ch.N, ch.fs // 1000) for b in range(2)]
    iq, au, info = _run(ch, tabs, blocks)                     # host entry: channel parts
    assert info.matrix_pipe == 1
    for c in range(0, n_ch - ch.N, 97):                       # repeated instances of an offset come out bit-identical
        assert np.array_equal(_bits(iq[c]), _bits(iq[c + ch.N]))
    pick = list(range(0, ch.N, 7)) + [n_ch - 1]
    riq, rau = _oracle(oracle, ch, [tabs[c] for c in pick], blocks)
    e_iq, e_au = rel_err_rows(iq[pick], riq), rel_err_rows(au[pick], rau)
    margin("matrix6_large_bank", iq=e_iq, audio=e_au)
    assert e_iq <= RTOL and e_au <= RTOL
    # cut over three shards: the kernel family is pinned from the whole list's choice, the bits stay
    mb = sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0, 0), tuning={"matrix_pipe": REQ})
    assert mb.info().matrix_pipe == 1
    got = [mb.process(x) for x in blocks]
    mb.close()
    assert np.array_equal(_bits(np.concatenate([g[0] for g in got], axis=1)), _bits(iq))
    # what the kernel is not built for is refused, not run on something else
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank(tabs[:8], ch.stage_tuples(), tuning={"matrix_pipe": 3})
    assert e.value.code == -5
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank(tabs[:8], ch.stage_tuples(), tuning={"matrix_pipe": REQ, "m_slot_groups": 4})
    assert e.value.code == -1
    c24 = chains.chain("2.4")
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank(sb.translators_for(c24, [1, -1]), c24.stage_tuples(), tuning={"matrix_pipe": REQ})
    assert e.value.code == -5


def test_never_automatic():
    """A caller's choice (tuning.matrix_pipe = 2), like the fused stage 3 of the 10 MS/s chain: faster from 1 024 channels with
    partners, but further from the reference under a strong carrier than the contract of the automatic paths allows (the stress
    test below)."""
    ch = chains.chain(RATE)
    for offs in (chains.offsets_cycled(2048, ch.N), [1 + (i % 359) for i in range(2048)], chains.offsets_cycled(700, ch.N)):
        bank = sb.ChannelBank(sb.translators_for(ch, offs), ch.stage_tuples())
        assert int(bank.info().matrix_pipe) == 0
        bank.close()
    with pytest.raises(sb.SdrxGpuError):
        sb.ChannelBank(sb.translators_for(ch, chains.offsets_cycled(64, ch.N)), ch.stage_tuples(), reference_order=True, tuning={"matrix_pipe": REQ})


STRESS = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "pairs6", "g[89]_*.npz")))


@pytest.mark.parametrize("name", STRESS)
def test_pairs_and_neighbours_under_a_strong_carrier(name):
    """tests/golden/pairs6: the stress of tests/test_gpu_pairs_stress.py for this chain -- outputs of the reference compiled here (its
    portable build and its Release build), a carrier of 0.3 / 0.9 of full scale on +37 over a floor of 0.25 / 0.05, channels: its
    conjugate partner, a neighbour, a second pair, the untranslated one ("one"), or 35 ... 45 ("neighbours").  The vector kernels
    (what a bank of this chain runs unless told otherwise) keep the contract of INTEGRATION.md 6: within 1e-5 of the portable build
    wherever the reference's two builds agree to 0.5e-5, never further from it than 1.7 x their spread.  k_stage12m6 does NOT: the
    carrier's conjugate partner under a carrier of 0.9 over a floor of 0.05 comes out 1.38e-5 (IQ) / 1.78e-5 (audio) from the portable
    build where the two builds are 0.42e-5 / 0.52e-5 apart -- 3.3 / 3.4 x; held here to 4 x, and that is why it is never chosen
    automatically."""
    g = load_golden(name, "pairs6")
    ch, blocks = regen_blocks(g)
    offs = [int(o) for o in g["offsets"]]
    tabs = [g["tables"][i] if o != 0 else None for i, o in enumerate(offs)]
    spread_iq, spread_au = max(float(v) for v in g["avx2_rel_dev_iq"]), max(float(v) for v in g["avx2_rel_dev_audio_lf"])
    rec = {}
    for path, mp, times in (("matrix", REQ, 4.0), ("vector", OFF, 1.7)):
        iq, au, info = _run(ch, tabs, blocks, tuning={"matrix_pipe": mp})
        assert info.matrix_pipe == (1 if mp == REQ else 0)
        e_iq = max(rel_err(iq[i], g["iq"][i]) for i in range(len(offs)))
        e_au = max(rel_err(au[i], g["audio_lf"][i]) for i in range(len(offs)))
        n_iq = max(min(rel_err(iq[i], g["iq"][i]), rel_err(iq[i], g["avx2_iq"][i])) for i in range(len(offs)))
        rec[path + "_iq"], rec[path + "_audio"], rec[path + "_nearer_build_iq"] = e_iq, e_au, n_iq
        for got, spread in ((e_iq, spread_iq), (e_au, spread_au)):
            assert got <= max(RTOL, times * spread), (path, got, spread)
            if spread <= 0.5e-5 and path == "vector":
                assert got <= RTOL, (path, got, spread)
    margin(f"pairs6[{name}]", ref_avx2_iq=spread_iq, ref_avx2_audio=spread_au, **rec)
