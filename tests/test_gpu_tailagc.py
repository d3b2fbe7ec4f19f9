"""k_tail_agc (the remaining decimator stages + IF-AGC -> |.| -> LF-AGC as ONE kernel, tiles handed from producer waves to the AGC
waves through LDS; sdrx_amd/csrc/kernels_tailagc.hpp) against the two-kernel sequence k_tail_s + k_agc: the same functions on the
same operands, so the same BITS -- decimated IQ, audio, AGC state, squelch state -- for every chain, every cut of the stream
into calls, every workgroup shape, with squelch / FM / LF-AGC, through every entry.  And once against the oracle
(reference: src/msd.hpp:132-155, 251-335; src/sdrx.cpp:455-463; src/agc.hpp:48-69, 106-131)."""
import numpy as np
import pytest

from conftest import RTOL, rel_err_rows
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu

FUSED, SPLIT = 2, 1          # tuning.tail_agc: require k_tail_agc / never


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _stream(ch, n_blocks, tabs, seed=31, extra=0):
    L = ch.block_len
    car = [t for t in tabs[:40:13] if t is not None][:2]
    return synth.synth_iq(seed, 0, n_blocks * L + extra, 0.25, car, [0.3, 0.1][:len(car)], ch.N, ch.fs // 1000)


def _run(tabs, ch, calls, tail_agc, cpg=0, **kw):
    tuning = dict(kw.pop("tuning", {}))
    tuning.update({"tail_agc": tail_agc, "tail_agc_cpg": cpg})
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning=tuning, **kw)
    assert (bank.info().tail_agc > 0) == (tail_agc == FUSED)
    if cpg and tail_agc == FUSED:
        assert bank.info().tail_agc == cpg
    outs = [bank.process(x) for x in calls]
    state = bank.agc_state()
    sq = bank.squelch_state() if kw.get("squelch") is not None else None
    info = bank.info()
    bank.close()
    return (np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1), state, sq, info)


def _same(a, b):
    assert a[0].shape == b[0].shape
    assert np.array_equal(_bits(a[0]), _bits(b[0])), "decimated IQ differs"
    assert np.array_equal(_bits(a[1]), _bits(b[1])), "audio differs"
    for u, v in zip(a[2], b[2]):
        assert np.array_equal(_bits(u), _bits(v)), "AGC state differs"
    if a[3] is not None:
        assert np.array_equal(_bits(a[3][0]), _bits(b[3][0])) and np.array_equal(a[3][1], b[3][1]), "squelch state differs"
    assert a[4].samples_out == b[4].samples_out


@pytest.mark.parametrize("n_ch,cpg", [(1024, 0), (64, 0), (64, 4), (333, 2), (37, 4), (5, 2), (1, 1), (2000, 0), (300, 0)])
def test_same_bits_as_the_two_kernels_whole_blocks(oracle, n_ch, cpg):
    ch = chains.chain("10")
    offs = chains.offsets_for_config("c3") if n_ch == 1024 else chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    x = _stream(ch, 3, tabs)
    L = ch.block_len
    calls = [x[i * L:(i + 1) * L] for i in range(3)]
    a = _run(tabs, ch, calls, FUSED, cpg)
    b = _run(tabs, ch, calls, SPLIT)
    _same(a, b)
    if n_ch <= 64:
        ref = oracle.bank(tabs, ch.stage_tuples())
        r_iq, r_au = ref.process(x, threads=8)
        assert rel_err_rows(a[0], r_iq) <= RTOL and rel_err_rows(a[1], r_au) <= RTOL


@pytest.mark.parametrize("rate", ["10", "2.4", "0.96", "6", "2.56", "1.92"])
def test_every_chain_ragged_calls_lf_agc(rate):
    """Calls of 0 ... several blocks' worth of samples (tiles of 32 outputs that end anywhere, calls without any output), the LF
    AGC active, 45 channels (not a multiple of anything)."""
    ch = chains.chain(rate)
    offs = [o for o in range(-22, 23)]
    tabs = sb.translators_for(ch, offs)
    L, M = ch.block_len, ch.m_total
    x = _stream(ch, 3, tabs, seed=5, extra=777)
    lens = [0, 1, M - 1, 1, 31 * M + 5, L - 40 * M, 33 * M, L, 7]
    lens.append(len(x) - sum(lens))
    calls, p = [], 0
    for n in lens:
        calls.append(x[p:p + n])
        p += n
    a = _run(tabs, ch, calls, FUSED, lf_agc=True)
    b = _run(tabs, ch, calls, SPLIT, lf_agc=True)
    _same(a, b)
    assert a[0].shape[1] == len(x) // M


@pytest.mark.parametrize("fm", [False, True])
def test_squelch_gating_and_fm_lanes(fm):
    ch = chains.chain("10")
    n_ch = 70
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    x = _stream(ch, 5, tabs, seed=9)
    L = ch.block_len
    # carriers that go away for one block and come back: the squelch of their channels opens, stays, closes, opens again
    # (ramp_up / open / ramp_down / closed chunks all occur)
    x = x.copy()
    x[2 * L:3 * L] = synth.synth_iq(9, 2 * L, L, 0.25)
    calls = [x[i * L:(i + 1) * L] for i in range(5)]
    level = np.where(np.arange(n_ch) % 4 == 0, 0.0, 6.0).astype(np.float32)
    kw = dict(squelch=(level,) + tuple(chains.squelch_tables()))
    if fm:
        kw["modulation"] = [("FM" if c % 3 == 1 else "AM") for c in range(n_ch)]
    a = _run(tabs, ch, calls, FUSED, **kw)
    b = _run(tabs, ch, calls, SPLIT, **kw)
    _same(a, b)
    assert len(set(a[3][1].tolist())) == 2          # some channels open, some closed


def test_vector_kernels_in_front_and_the_other_entries():
    """The stage-1/2 kernel in front does not matter (vector kernels: another scratch layout, P > 1), nor does the entry: device
    pointers with unaligned output rows (the keep-buffer path), page-locked input, the audio FIR behind it."""
    import torch
    ch = chains.chain("10")
    n_ch = 200
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    x = _stream(ch, 2, tabs, seed=77)
    calls = [x[:L], x[L:]]
    fir = (np.hanning(31) / 16).astype(np.float32)
    for tun in ({"matrix_pipe": 1}, {"matrix_pipe": 2}, {"matrix_pipe": 1, "k12_cpw": 2}):
        a = _run(tabs, ch, calls, FUSED, tuning=tun, audio_fir=fir)
        b = _run(tabs, ch, calls, SPLIT, tuning=tun, audio_fir=fir)
        _same(a, b)
    # device entry, rows of 513 (8-byte aligned only): k_agc read d_keep there, k_tail_agc needs neither
    outs = {}
    for mode in (FUSED, SPLIT):
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"tail_agc": mode})
        d_iq = torch.zeros(n_ch, 513, 2, device="cuda")
        d_au = torch.zeros(n_ch, 513, device="cuda")
        got = []
        for c in calls:
            d_x = torch.from_numpy(c.view(np.float32).reshape(L, 2).copy()).cuda()
            torch.cuda.synchronize()
            assert bank.process_device(d_x.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 513) == 512
            bank.synchronize()
            got.append((d_iq.cpu().numpy().copy(), d_au.cpu().numpy().copy()))
        outs[mode] = got
        bank.close()
    for (i1, a1), (i2, a2) in zip(outs[FUSED], outs[SPLIT]):
        assert np.array_equal(_bits(i1), _bits(i2)) and np.array_equal(_bits(a1), _bits(a2))


def test_refusals_and_automatic_choice():
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, chains.offsets_cycled(64, ch.N))
    with pytest.raises(sb.SdrxGpuError) as e:      # the run-time-tap tail has no fused form
        sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"tail_agc": FUSED, "tail_generic": 1})
    assert e.value.code == -5
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"tail_agc": 3})
    assert e.value.code == -1
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"tail_agc_cpg": 3})
    assert e.value.code == -1
    # automatic where it is the faster of the two: banks of 1 .. 4 channels per CU of the device that are alone on it when they
    # are created; channels per workgroup: as few as give every CU one workgroup at most (sdrxgpu.hip, SDRX_TAILAGC_AUTO_*)
    import torch
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    for n in (n_cu // 4, n_cu, 2 * n_cu, 4 * n_cu, 8 * n_cu):
        want = 0 if (n < n_cu or n > 4 * n_cu) else 1 if n <= n_cu else 2 if n <= 2 * n_cu else 4
        bk = sb.ChannelBank(sb.translators_for(ch, chains.offsets_cycled(n, ch.N)), ch.stage_tuples())
        assert bk.info().tail_agc == want, (n, bk.info().tail_agc)
        bk.close()


def test_the_choice_is_made_once_and_kept():
    """Which of the two (bit-identical) tail sequences a bank runs is decided ONCE -- provisionally in bank_create, finally when its
    first block is processed, from the banks alive on the device at that moment -- and bank_info says which: a second bank that comes
    and goes between the calls of a stream changes neither the path nor a bit of the first bank's outputs or state; a bank created
    beside another one keeps the two short kernels; and two banks created before either has seen a block (BASELINE configs[4]'s two
    streams per GPU) both keep them, whichever was created first."""
    import torch
    ch = chains.chain("10")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    n_ch = 2 * n_cu
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    L, M = ch.block_len, ch.m_total
    x = _stream(ch, 3, tabs, seed=13, extra=99)
    lens = [L, 5 * M + 3, L - 5 * M, 17, L - 20]
    lens.append(len(x) - sum(lens))
    calls, p = [], 0
    for n in lens:
        calls.append(x[p:p + n])
        p += n
    level = np.where(np.arange(n_ch) % 4 == 0, 0.0, 6.0).astype(np.float32)
    mod = [("FM" if c % 3 == 1 else "AM") for c in range(n_ch)]
    want = _run(tabs, ch, calls, SPLIT, lf_agc=True, modulation=mod)
    auto = sb.ChannelBank(tabs, ch.stage_tuples(), lf_agc=True, modulation=mod)
    assert auto.info().tail_agc == 2                    # alone on the device at create: k_tail_agc
    outs, other = [], None
    small = sb.translators_for(ch, chains.offsets_cycled(300, ch.N))
    for k, c in enumerate(calls):
        if k in (1, 4):                                 # a second bank appears (and works) ...
            other = sb.ChannelBank(small, ch.stage_tuples())
            assert other.info().tail_agc == 0           # ... created beside another bank: the two short kernels
            other.process(x[:L])
        outs.append(auto.process(c))
        assert auto.info().tail_agc == 2                # the first bank's path is the one it was created with
        if k in (2, 4):                                 # ... and goes
            other.close()
            other = None
    got = (np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1), auto.agc_state(), None, auto.info())
    auto.close()
    _same(got, want)
    # two banks, then their first blocks: neither takes the fused kernel, and the one created first says so once it has run
    first = sb.ChannelBank(tabs, ch.stage_tuples())
    assert first.info().tail_agc == 2                   # (provisional: alone when it was created)
    second = sb.ChannelBank(tabs, ch.stage_tuples())
    assert second.info().tail_agc == 0
    a1, a2 = first.process(x[:L]), second.process(x[:L])
    assert first.info().tail_agc == 0 and second.info().tail_agc == 0
    second.close()
    b1 = first.process(x[L:2 * L])
    assert first.info().tail_agc == 0                   # ... and alone again it stays with what it ran
    first.close()
    assert np.array_equal(_bits(a1[0]), _bits(a2[0])) and np.array_equal(_bits(a1[1]), _bits(a2[1]))
    plain = _run(tabs, ch, [x[:L], x[L:2 * L]], SPLIT)
    assert np.array_equal(_bits(np.concatenate([a1[0], b1[0]], axis=1)), _bits(plain[0]))
    assert np.array_equal(_bits(np.concatenate([a1[1], b1[1]], axis=1)), _bits(plain[1]))
    # whole 512-output chunks with the squelch, the same coming and going
    kw = dict(squelch=(level,) + tuple(chains.squelch_tables()), modulation=mod)
    blocks = [x[i * L:(i + 1) * L] for i in range(3)]
    want = _run(tabs, ch, blocks, SPLIT, **kw)
    auto = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
    assert auto.info().tail_agc == 2
    outs = []
    for k, c in enumerate(blocks):
        other = sb.ChannelBank(small, ch.stage_tuples()) if k == 1 else None
        outs.append(auto.process(c))
        if other is not None:
            other.close()
    got = (np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1), auto.agc_state(), auto.squelch_state(), auto.info())
    auto.close()
    _same(got, want)
