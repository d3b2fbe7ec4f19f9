"""Two channels per lane in the stage-1/2 kernel (k_stage12p): conjugate partners (+k / -k steps from the tuner centre)
share the input reads, the translator reads and half of the mixer; other channels are paired two by two and share the
input reads.  Every channel's values come out of the same instruction sequence on the same operands as with one
channel per lane, so the two kernels must agree bit for bit -- for every mix of partners, loners, untranslated
channels, odd counts, both tail paths, ragged calls and the host entry's channel parts."""
import numpy as np
import pytest

from conftest import RTOL, rel_err_rows
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu

WHOLE = 0xFFFFFFFF
OFF, REQ = 1, 2


def _run(ch, tabs, blocks, pair, fused, **kw):
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), block_hint=WHOLE, tuning={"pair_lanes": pair, "fused_tail": fused, "k12_waves": 4, "k12_cpw": 4}, **kw)
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


@pytest.mark.parametrize("fused", [sb.FUSED_OFF, sb.FUSED_REQUIRE])
@pytest.mark.parametrize("offs", [
    list(range(-20, 21)),                                  # 20 conjugate pairs + the untranslated channel
    [5, -5, 5, -5, 7, 9, 11, 0, 0, -7, 600 - 1200, 13],     # repeated offsets, loners, two untranslated, -600 (no partner)
    [3],                                                   # a single channel
    [4, -4],                                               # one pair
    list(range(1, 38)),                                    # no partners at all: 37 loners -> 18 general pairs + 1
])
def test_pair_lanes_equal_single_lanes_bit_for_bit(offs, fused):
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(31, 0, 2 * L + 4321, 0.25, [t for t in tabs[:1] if t is not None], [0.3], ch.N, 10000)
    lens = [L, 0, 624, 625, 777, L // 2, 10 ** 9]
    want = _run(ch, tabs, _cut(stream, lens), OFF, fused)
    got = _run(ch, tabs, _cut(stream, lens), REQ, fused)
    assert want[2].pair_lanes == 0 and got[2].pair_lanes >= 1
    assert np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32))
    assert np.array_equal(want[1].view(np.uint32), got[1].view(np.uint32))


def test_conjugate_pairs_are_found():
    ch = chains.chain("10")
    offs = list(range(-20, 21))
    bank = sb.ChannelBank(sb.translators_for(ch, offs), ch.stage_tuples(), block_hint=WHOLE, tuning={"pair_lanes": REQ, "k12_waves": 4, "k12_cpw": 4})
    # 20 conjugate pairs; 16 of them fill one slot group and share the mixer, the other four ride with the general
    # pairs (a group is all of one kind); the count is reported + 1
    assert bank.info().pair_lanes == 16 + 1
    bank.close()


def test_large_bank_pairs_by_default_and_matches_the_oracle_and_the_host_parts(oracle):
    ch = chains.chain("10")
    n_ch = 10000                                           # the bench workload: four host parts of 2 560 channels
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = [synth.synth_iq(9, b * L, L, 0.25, [tabs[650]], [0.2], ch.N, 10000) for b in range(2)]
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": 1})   # (the vector kernels' automatic path; tests/test_gpu_matrix.py covers the default)
    info = bank.info()
    assert info.pair_lanes > 4000 and info.fused_tail == 1      # (partners are looked for inside a host part: 4 712 of the 4 990 possible)
    outs = [bank.process(x) for x in blocks]                # host entry: channel parts, D2H beside the kernels
    bank.close()
    iq = np.concatenate([o[0] for o in outs], axis=1)
    au = np.concatenate([o[1] for o in outs], axis=1)
    single = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"pair_lanes": OFF, "matrix_pipe": 1})
    assert single.info().pair_lanes == 0
    o2 = [single.process(x) for x in blocks]
    single.close()
    assert np.array_equal(iq.view(np.uint32), np.concatenate([o[0] for o in o2], axis=1).view(np.uint32))
    assert np.array_equal(au.view(np.uint32), np.concatenate([o[1] for o in o2], axis=1).view(np.uint32))
    for c in (0, 1, 599, 600, 601, 1199, 1200, 2559, 2560, 9999):
        ob = oracle.bank([tabs[c]], ch.stage_tuples())
        r = [ob.process(x, threads=1) for x in blocks]
        assert rel_err_rows(iq[c:c + 1], np.concatenate([v[0] for v in r], axis=1)) <= RTOL
        assert rel_err_rows(au[c:c + 1], np.concatenate([v[1] for v in r], axis=1)) <= RTOL


def test_pair_lanes_with_ftfir_squelch_fm_and_audio_fir():
    """Everything around the stage-1/2 kernel is unchanged by the lane layout: --ftfir (tables doubled and phase-shifted: still
    conjugates of each other), squelch + FM + LF AGC + audio FIR downstream."""
    ch = chains.chain("10")
    offs = list(range(-12, 13)) + [300, -300, 7]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    pb, ru, rd = chains.squelch_tables()
    on = [False, True, True, False]
    blocks = [synth.synth_iq(3, b * L, L, 0.25, [tabs[1]] if c else [], [0.25] if c else [], ch.N, 10000) for b, c in enumerate(on)]
    kw = dict(ftfir=True, lf_agc=True, squelch=(np.full(len(offs), 6.0, np.float32), pb, ru, rd),
              modulation=["FM" if i % 5 == 0 else "AM" for i in range(len(offs))], audio_fir=chains.audio_bpf_taps())
    want = _run(ch, tabs, blocks, OFF, sb.FUSED_REQUIRE, **kw)
    got = _run(ch, tabs, blocks, REQ, sb.FUSED_REQUIRE, **kw)
    assert got[2].pair_lanes >= 1
    assert np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32))
    assert np.array_equal(want[1].view(np.uint32), got[1].view(np.uint32))


@pytest.mark.parametrize("mult,fits", [(2, True), (4, True), (16, False)])
def test_longer_translator_tables_change_the_epilogue_rows(mult, fits):
    """Translator tables of 2 N, 4 N and 16 N entries (a finer grid): a lane's segment is as long as the table, so the
    ranges the fused epilogue reads back grow to 12 KB / 24 KB / 98 KB per channel.  With 12 KB every wave of k_stage12p
    still has an LDS row of its own; with 24 KB only three of the four waves have one and share the channels out among
    themselves; 98 KB do not fit and such a bank must refuse fused stage 3 (and run the separate tail).  Same bits on
    every path."""
    ch = chains.chain("10")
    offs = [5, -5, 17, -17, 0, 33, 2 * ch.N - 1, -(2 * ch.N - 1), 9]
    N = mult * ch.N
    tabs = [sb.make_translator(N, ch.z, o) if o else None for o in offs]
    L = ch.block_len
    stream = synth.synth_iq(77, 0, L + 4 * ch.N + 123, 0.25)
    lens = [L, 0, 625, 10 ** 9]
    want = _run(ch, tabs, _cut(stream, lens), OFF, sb.FUSED_OFF)
    if fits:
        got = _run(ch, tabs, _cut(stream, lens), REQ, sb.FUSED_REQUIRE)
        assert got[2].pair_lanes >= 1 and got[2].fused_tail
    else:
        with pytest.raises(sb.SdrxGpuError):
            _run(ch, tabs, _cut(stream, lens), REQ, sb.FUSED_REQUIRE)
        got = _run(ch, tabs, _cut(stream, lens), REQ, 0)                   # automatic: two channels per lane, separate tail
        assert got[2].pair_lanes >= 1 and not got[2].fused_tail
    assert np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32))
    assert np.array_equal(want[1].view(np.uint32), got[1].view(np.uint32))
