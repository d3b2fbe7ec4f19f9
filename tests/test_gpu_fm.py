"""GPU parity of the per-channel FM demodulator (Demod::demod's FM branch, reference src/sdrx.cpp:181-194,
EXPERIMENTAL there; restated in oracle/sdrx_oracle.c, parity unpinned: Demod lives inside the unbuildable sdrx.cpp)."""
import numpy as np
import pytest

from conftest import RTOL, rel_err
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu


def _fm_stream(ch, tabs, n, seed, f_mod=700.0, beta=1.5, amp=0.3):
    """noise + one FM carrier per given channel table: amp * exp(j beta sin(2 pi f_mod t)) on the channel centre"""
    x = synth.synth_iq(seed, 0, n, 0.05).astype(np.complex128)
    t = np.arange(n)
    for k, tab in enumerate(tabs):
        carrier = np.conj(np.tile(tab.astype(np.complex128), n // ch.N + 1)[:n])
        x = x + amp * carrier * np.exp(1j * beta * np.sin(2 * np.pi * (f_mod + 150 * k) * t / ch.fs))
    return x.astype(np.complex64)


def test_fm_channels_vs_oracle_and_am_channels_untouched(oracle):
    ch = chains.chain("10")
    offs = [37, -100, 5, 64]
    mods = ["FM", "AM", "FM", "AM"]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    x = _fm_stream(ch, [tabs[0], tabs[2]], 3 * L, 7)
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), modulation=mods)
    plain = sb.ChannelBank(tabs, ch.stage_tuples())
    ref_msd = [oracle.msd(t, ch.stage_tuples()) for t in tabs]
    ref_agc = [oracle.agc_pair(False) for _ in tabs]
    prev = [np.zeros(2, np.float32) for _ in tabs]
    for b in range(3):
        blk = x[b * L:(b + 1) * L]
        iq, au = bank.process(blk)
        iq_p, au_p = plain.process(blk)
        assert np.array_equal(iq.view(np.uint32), iq_p.view(np.uint32))
        for c in range(4):
            y = ref_msd[c].decimate(blk)
            ya = ref_agc[c].adjust(y)
            if mods[c] == "FM":
                want = ref_agc[c].adjust_lf(oracle.demod_fm(prev[c], ya))
                assert rel_err(au[c], want) <= RTOL, (b, c, rel_err(au[c], want))
                assert float(np.std(au[c])) > 0.05                      # it really is the demodulated tone, not an envelope
                assert not np.allclose(au[c], au_p[c], atol=1e-3)
            else:
                want = ref_agc[c].adjust_lf(oracle.demod_am(ya))
                assert rel_err(au[c], want) <= RTOL
                assert np.array_equal(au[c].view(np.uint32), au_p[c].view(np.uint32))   # AM lanes of an FM bank: bit-identical
    # the demodulated tone: 700 Hz at 16 kS/s -> dominant FFT bin of the last block's audio
    spec = np.abs(np.fft.rfft(au[0] - np.mean(au[0])))
    assert abs(int(np.argmax(spec)) - round(700.0 * 512 / 16000)) <= 1
    bank.close()
    plain.close()


def test_fm_with_squelch_gating_vs_oracle(oracle):
    """FM demodulation happens only while the squelch is open (sdrx.cpp:458-461: prev_sample freezes while closed);
    the ramp-down chunk carries std::abs whatever the modulation (sdrx.cpp:499)."""
    from oracle import ffi as offi
    ch = chains.chain("10")
    offs = [37, -100]
    tabs = sb.translators_for(ch, offs)
    pb, ru, rd = chains.squelch_tables()
    win = chains.hamming_window()
    L = ch.block_len
    on = [False, True, True, False, False, True]
    blocks = []
    for b, c in enumerate(on):
        full = _fm_stream(ch, [tabs[0]] if c else [], (b + 1) * L, 11)
        blocks.append(full[b * L:])
    levels = np.array([9.0, 0.0], np.float32)
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), squelch=(levels, pb, ru, rd), modulation=["FM", "FM"])
    ref_ch = [offi.SquelchChannel(oracle, float(lv), win, pb, ru, rd, fm=True) for lv in levels]
    ref_msd = [oracle.msd(t, ch.stage_tuples()) for t in tabs]
    states = []
    for b, blk in enumerate(blocks):
        _, au = bank.process(blk)
        _, st = bank.squelch_state()
        for c in range(2):
            r_au, _, r_st = ref_ch[c].chunk(ref_msd[c].decimate(blk))
            assert int(st[c]) == r_st, (b, c)
            scale = max(float(np.sqrt(np.mean(r_au ** 2))), 1e-3)
            assert float(np.max(np.abs(au[c] - r_au))) / scale <= 5e-5, (b, c)
        states.append(int(st[0]))
    assert 1 in states and 0 in states
    bank.close()


def test_modulation_argument_errors():
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, [5, 6])
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank(tabs, ch.stage_tuples(), modulation=[0, 7])
    assert e.value.code == -1
    # all-AM modulation list: the AM-only kernel, same results as no list at all
    x = synth.synth_iq(3, 0, ch.block_len, 0.25)
    a = sb.ChannelBank(tabs, ch.stage_tuples(), modulation=["AM", "AM"])
    b = sb.ChannelBank(tabs, ch.stage_tuples())
    assert np.array_equal(a.process(x)[1].view(np.uint32), b.process(x)[1].view(np.uint32))
    a.close()
    b.close()
