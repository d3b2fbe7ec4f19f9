"""GPU parity of the optional stereo mix-down (pan weights, FIR2, S16): the back half of the reference's
alsa_write_cb, src/sdrx.cpp:469-497 + 646-656, with get_audio_pos (1024-1044) for the positions."""
import numpy as np
import pytest

from conftest import RTOL, margin
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu


def _pos(n):
    return [sb.audio_pos(c, n) for c in range(n)]


def test_audio_pos_matches_the_oracle(oracle):
    for n in list(range(1, 33)) + [64, 100, 999, 10000]:
        assert _pos(n) == [oracle.audio_pos(c, n) for c in range(n)]


def test_mix_fir2_s16_end_to_end_vs_oracle(oracle):
    """8 channels with squelch (some closed, opening, closing), panned, FIR2 with +6 dB volume, S16: every stage from
    the oracle's own chain (MSD -> squelch-gated audio -> pan mix -> FIR2 -> S16)."""
    from oracle import ffi as offi
    ch = chains.chain("10")
    offs = [37, -100, 5, 210, -333, 64, -7, 450]
    levels = [9.0, 0.0, 9.0, 0.0, 9.0, 0.0, 0.0, 9.0]
    n = len(offs)
    tabs = sb.translators_for(ch, offs)
    pb, ru, rd = chains.squelch_tables()
    win = chains.hamming_window()
    taps = chains.audio_bpf_taps()
    pos = _pos(n)
    on = [False, True, True, True, False, False]
    blocks = [synth.synth_iq(11, b * ch.block_len, ch.block_len, 0.25, [tabs[0], tabs[4]] if c else [], [0.2, 0.1] if c else [],
                             ch.N, 10000) for b, c in enumerate(on)]
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), squelch=(np.array(levels, np.float32), pb, ru, rd),
                          pan_pos=pos, mix_fir=taps, mix_fir_gain_db=6.0)
    ref_ch = [offi.SquelchChannel(oracle, lv, win, pb, ru, rd) for lv in levels]
    ref_msd = [oracle.msd(t, ch.stage_tuples()) for t in tabs]
    ref_fir = oracle.fir2(taps, 6.0)
    nonzero = 0
    worst = 0.0
    for b, x in enumerate(blocks):
        _, au = bank.process(x)
        mix_f, mix_q = bank.mix()
        r_au = np.stack([ref_ch[c].chunk(ref_msd[c].decimate(x))[0] for c in range(n)])
        r_mix = ref_fir.filter(oracle.pan_mix(r_au, pos))
        r_q = oracle.to_s16(r_mix)
        assert mix_f.shape == (512, 2) and mix_q.shape == (512, 2)
        scale = max(float(np.sqrt(np.mean(r_mix ** 2))), 1e-3)
        e = float(np.max(np.abs(mix_f - r_mix))) / scale
        worst = max(worst, e)
        assert e <= RTOL, (b, e)
        assert int(np.max(np.abs(mix_q.astype(np.int32) - r_q.astype(np.int32)))) <= 1, b     # truncation at an integer edge
        assert np.array_equal(mix_q, oracle.to_s16(mix_f)), b                              # conversion itself: exact
        nonzero += int(np.any(mix_q != 0))
    assert nonzero >= 3
    margin("mix_end_to_end_with_squelch", mix=worst)
    bank.close()


@pytest.mark.parametrize("n_ch", [1, 2, 7, 64])
def test_pan_mix_reproduces_the_reference_summation_order(oracle, n_ch):
    """Up to 64 channels the device adds the channels in the reference's order with unfused, individually rounded
    products: fed with the device's own channel audio, the oracle's pan mix must come out bit-identical."""
    ch = chains.chain("2.4")
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    pos = _pos(n_ch)
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), pan_pos=pos)
    for b in range(2):
        x = synth.synth_iq(21, b * ch.block_len, ch.block_len, 0.25, tabs[:1], [0.3], ch.N, 7000)
        _, au = bank.process(x)
        mix_f, mix_q = bank.mix()
        want = oracle.pan_mix(au, pos)
        assert np.array_equal(mix_f.view(np.uint32), want.view(np.uint32))
        assert np.array_equal(mix_q, oracle.to_s16(want))
    bank.close()


def test_pan_mix_of_a_large_bank_and_clipping(oracle):
    """1000 channels: groups of 64 are added in order, so the sum differs from the strictly serial one by
    re-association only; the sum of 1000 envelopes clips, which exercises both clamp branches of sdrx.cpp:651-652
    (the negative one through a negative volume-less filter tap set)."""
    ch = chains.chain("10")
    n_ch = 1000
    tabs = sb.translators_for(ch, chains.offsets_cycled(n_ch, ch.N))
    pos = _pos(n_ch)
    x = synth.synth_iq(31, 0, ch.block_len, 0.25, tabs[:1], [0.3], ch.N, 7000)
    for taps in (None, np.array([-1.0], np.float32), np.array([0.001], np.float32)):
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), pan_pos=pos, mix_fir=taps)
        _, au = bank.process(x)
        mix_f, mix_q = bank.mix()
        want = oracle.pan_mix(au, pos)
        if taps is not None:
            want = oracle.fir2(taps).filter(want)
        scale = float(np.sqrt(np.mean(want ** 2)))
        assert float(np.max(np.abs(mix_f - want))) / scale <= RTOL
        assert np.array_equal(mix_q, oracle.to_s16(mix_f))
        if taps is None:
            assert np.all(mix_q[100:] == 32767)
        elif taps[0] < 0:
            assert np.all(mix_q[100:] == -32767)
        else:
            assert np.all(np.abs(mix_q[100:].astype(np.int32)) < 32767) and np.any(mix_q != 0)
        bank.close()


def test_mix_with_ragged_calls_and_reset(oracle):
    """The FIR2 delay line carries across calls of any length; reset clears it."""
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, [37, -100, 5])
    pos = _pos(3)
    taps = chains.audio_bpf_taps()
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), pan_pos=pos, mix_fir=taps)
    x = synth.synth_iq(41, 0, 2 * ch.block_len, 0.25, tabs[:1], [0.3], ch.N, 7000)
    cuts = [0, 70000, 71250, 320000, 500000, 640000]

    def run():
        fir = oracle.fir2(taps)
        for a, b in zip(cuts[:-1], cuts[1:]):
            _, au = bank.process(x[a:b])
            mix_f, mix_q = bank.mix()
            assert mix_f.shape[0] == au.shape[1]
            want = fir.filter(oracle.pan_mix(au, pos)) if au.shape[1] else np.zeros((0, 2), np.float32)
            if au.shape[1]:
                scale = max(float(np.sqrt(np.mean(want ** 2))), 1e-3)
                e = float(np.max(np.abs(mix_f - want))) / scale
                margin("mix_ragged_calls", mix=e)
                assert e <= RTOL, e
    run()
    bank.reset()
    run()
    bank.close()


def test_mix_errors():
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, [5, 6])
    x = synth.synth_iq(3, 0, 2 * ch.block_len, 0.25)
    plain = sb.ChannelBank(tabs, ch.stage_tuples())
    plain.process(x[:ch.block_len])
    with pytest.raises(sb.SdrxGpuError) as e:
        plain.mix()
    assert e.value.code == -1
    plain.close()
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), pan_pos=[-2, 2])
    with pytest.raises(sb.SdrxGpuError) as e:
        bank.process(x)                          # two blocks in one call: one mix buffer per call
    assert e.value.code == -5
    bank.close()
