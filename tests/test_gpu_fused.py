"""Stage 3 fused behind the stage-1/2 kernel (k_stage12<.., MA, TA> + k_tail4) against the separate tail kernel, the
golden vectors and the oracle.  The fused path is what a large bank of a 4-stage chain runs (10 000 channels of the
10 MS/s chain: the bench workload); small banks are forced onto it here with tuning.fused_tail = 2.  The two paths
run the same instruction sequence on the same operands for every output, so they must agree bit for bit -- also
when the stream is cut at arbitrary points (interior outputs of one call are boundary outputs of another)."""
import numpy as np
import pytest

from conftest import RTOL, load_golden, regen_blocks, rel_err_rows
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu

WHOLE = 0xFFFFFFFF


def _bank(ch, tabs, fused, **tun):
    t = {"fused_tail": sb.FUSED_REQUIRE if fused else sb.FUSED_OFF}
    t.update(tun)
    b = sb.ChannelBank(tabs, ch.stage_tuples(), block_hint=WHOLE, tuning=t)
    assert b.info().fused_tail == int(fused)
    return b


def _run(bank, blocks):
    iq, au = [], []
    for x in blocks:
        a, b = bank.process(x)
        iq.append(a)
        au.append(b)
    bank.close()
    return np.concatenate(iq, axis=1), np.concatenate(au, axis=1)


def _cut(stream, lens):
    out, p = [], 0
    for n in lens:
        n = min(n, len(stream) - p)
        out.append(stream[p:p + n])
        p += n
    assert p == len(stream)
    return out


@pytest.mark.parametrize("rate,w,cpw", [("10", 4, 4), ("10", 16, 4), ("10", 8, 2), ("10", 4, 1), ("10", 16, 1),
                                        ("2.4", 4, 4), ("2.4", 8, 1), ("2.4", 16, 2)])
def test_fused_equals_separate_tail_bit_for_bit(rate, w, cpw):
    ch = chains.chain(rate)
    offs = list(range(-40, 37))                          # 77 channels: not a multiple of any W*CPW
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(19, 0, 3 * L, 0.25, [tabs[3]], [0.3], ch.N, ch.fs // 1000)
    lens = [L, 0, 1, ch.m_total - 1, ch.m_total, 5000, L - 6 * ch.m_total, 10, L // 3, 10 ** 9]
    want = _run(_bank(ch, tabs, False), _cut(stream, lens))
    got = _run(_bank(ch, tabs, True, k12_waves=w, k12_cpw=cpw), _cut(stream, lens))
    assert want[0].shape == got[0].shape == (len(offs), 3 * 512)
    assert np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32))
    assert np.array_equal(want[1].view(np.uint32), got[1].view(np.uint32))


@pytest.mark.parametrize("name", ["g2_fs10000", "g3_fs10000_noise", "g4_fs10000_impulse", "g4_fs10000_tone",
                                  "g6_fs10000_ragged", "g1_fs2400_c1", "g5_fs10000_ftfir"])
def test_fused_golden_vectors(name):
    g = load_golden(name)
    ch, blocks = regen_blocks(g)
    tabs = [g["tables"][i] if int(o) != 0 else None for i, o in enumerate(g["offsets"])]
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), block_hint=WHOLE, ftfir="ftfir" in name,
                          tuning={"fused_tail": sb.FUSED_REQUIRE})
    iq, au = _run(bank, blocks)
    e_iq, e_au = rel_err_rows(iq, g["iq"]), rel_err_rows(au, g["audio_lf"])
    print(f"{name} (fused): iq rel {e_iq:.3e}  audio rel {e_au:.3e}")
    assert e_iq <= RTOL and e_au <= RTOL


def test_fused_split_invariance_and_reset():
    ch = chains.chain("10")
    offs = [0, 3, -77, 415, -600, 599]
    tabs = sb.translators_for(ch, offs)
    x = synth.synth_iq(5, 0, 2 * ch.block_len, 0.25)
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), block_hint=WHOLE, tuning={"fused_tail": sb.FUSED_REQUIRE})
    one = [np.concatenate(v, axis=1) for v in zip(bank.process(x[: ch.block_len]), bank.process(x[ch.block_len:]))]
    bank.reset()
    parts = [bank.process(c) for c in _cut(x, (0, 1, 624, 625, 626, 24, 99999, 1, 0, 320000 - 1, 77777, 19200, 19199, 10 ** 9))]
    got = [np.concatenate(v, axis=1) for v in zip(*parts)]
    bank.close()
    for a, b in zip(one, got):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_large_bank_automatic_path_matches_the_oracle(oracle):
    """4 000 channels of the 10 MS/s chain, everything automatic (two channels per lane; whether stage 3 runs fused depends
    on the bodies per lane the create call picks for this size -- tests/test_gpu_pair.py pins the 10 000-channel case)."""
    ch = chains.chain("10")
    offs = chains.offsets_cycled(4000, ch.N)
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = [synth.synth_iq(77, b * L, L, 0.25, [tabs[700]], [0.2], ch.N, 10000) for b in range(2)]
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": 1})   # (the vector kernels' automatic path; tests/test_gpu_matrix.py covers the default)
    assert bank.info().pair_lanes > 1000
    iq, au = _run(bank, blocks)
    for c in (0, 1, 599, 600, 700, 1200, 3999):
        ob = oracle.bank([tabs[c]], ch.stage_tuples())
        r = [ob.process(x, threads=1) for x in blocks]
        r_iq = np.concatenate([v[0] for v in r], axis=1)
        r_au = np.concatenate([v[1] for v in r], axis=1)
        assert rel_err_rows(iq[c:c + 1], r_iq) <= RTOL and rel_err_rows(au[c:c + 1], r_au) <= RTOL
    # instances of one offset are computed independently and agree exactly
    assert np.array_equal(iq[0].view(np.uint32), iq[1200].view(np.uint32))


def test_fused_with_squelch_audio_fir_and_two_blocks_per_call(oracle):
    """Everything downstream of the decimated IQ is unchanged by the fused path; a call of two blocks (max_in_len)."""
    ch = chains.chain("10")
    offs = [37, -100, 5, 0]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(3, 0, 4 * L, 0.25, [tabs[0]], [0.2], ch.N, 10000)
    kw = dict(audio_fir=chains.audio_bpf_taps(), max_in_len=2 * L, block_hint=WHOLE)
    a = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"fused_tail": sb.FUSED_OFF}, **kw)
    b = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"fused_tail": sb.FUSED_REQUIRE}, **kw)
    for cut in (stream[: 2 * L], stream[2 * L: 3 * L], stream[3 * L:]):
        wa, wb = a.process(cut)
        ga, gb = b.process(cut)
        assert np.array_equal(wa.view(np.uint32), ga.view(np.uint32))
        assert np.array_equal(wb.view(np.uint32), gb.view(np.uint32))
    a.close()
    b.close()
