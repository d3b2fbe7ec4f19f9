"""k_stage123m + k_tail_c (SDRXGPU_FLAG_FUSED_STAGE3 for large banks of the 10 MS/s chain; tuning.matrix_pipe = 3 at any size): mixer + stage 1 on the f32 matrix
pipe and stages 2-3 behind it in the same kernel as one 119-tap filter in transposed form -- no stage-2 scratch.  Another
summation order again (the taps of stages 2 and 3 convolved beforehand; an output whose window straddles two lanes' segments is
the sum of two partial sums), so the path is held to the oracle / the reference's golden vectors at north_star's 1e-5, to itself
bit for bit wherever that is promised (workgroup shape, who else is in the bank, which entry point, repeated instances), and to
the same tolerance under different cuts of the stream into calls: segments start where a call starts, so another cut sums the
outputs in other pieces -- a second realisation of the path's rounding noise (the reference is cut-invariant bit for bit;
k_stage12m and the vector kernels are too).  What that noise is: an fp32 chain of 119 products rounds relative to its partial
sums, and those carry every signal that is still in band after stage 1 -- a carrier two channels away, say -- at full strength;
the reference's two short stages round less (this path is about 2 x the reference's own Release-vs-portable
deviation on such channels, the other paths about 1 x)."""
import os

import numpy as np
import pytest

from conftest import RTOL, load_golden, regen_blocks, rel_err, rel_err_rows, margin
from sdrx_amd import bank as sb, chains, synth
from test_gpu_matrix import OFFSETS, _cut, _oracle

pytestmark = pytest.mark.gpu

WHOLE = 0xFFFFFFFF
FUSED3 = 3


def _run(ch, tabs, blocks, tuning=None, **kw):
    t = {"matrix_pipe": FUSED3}
    t.update(tuning or {})
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning=t, **kw)
    info = bank.info()
    assert info.matrix_pipe == 2
    outs = [bank.process(x) for x in blocks]
    bank.close()
    return np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1), info


@pytest.mark.parametrize("offs", OFFSETS)
def test_fused_stage3_matches_the_oracle(oracle, offs):
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(31, 0, 2 * L + 4321, 0.25, [t for t in tabs[:1] if t is not None], [0.3], ch.N, 10000)
    for lens in ([L, 0, 624, 625, 777, L // 2, 10 ** 9], [1, 3, 1, 24, 25, 26, 99, 100, 101, 1199, 1200, 1201, 5000, 10 ** 9]):
        iq, au, info = _run(ch, tabs, _cut(stream, lens))
        riq, rau = _oracle(oracle, ch, tabs, _cut(stream, lens))
        assert iq.shape == riq.shape
        e_iq, e_au = rel_err_rows(iq, riq), rel_err_rows(au, rau)
        margin("fused3_vs_oracle_%dch" % len(offs), iq=e_iq, audio=e_au)
        assert e_iq <= RTOL and e_au <= RTOL


@pytest.mark.parametrize("name", ["g2_fs10000", "g3_fs10000_noise", "g4_fs10000_impulse", "g4_fs10000_dc", "g4_fs10000_tone",
                                  "g5_fs10000_ftfir", "g6_fs10000_ragged"])
def test_golden_vectors_through_the_fused_path(name):
    g = load_golden(name)
    ch, blocks = regen_blocks(g)
    tabs = [g["tables"][i] if int(o) != 0 else None for i, o in enumerate(g["offsets"])]
    if all(t is None for t in tabs):
        pytest.skip("nobody mixes: no translator length to space the segments by (such banks run the vector kernels)")
    iq, au, _ = _run(ch, tabs, blocks, ftfir="ftfir" in name)
    e_iq, e_au = rel_err_rows(iq, g["iq"]), rel_err_rows(au, g["audio_lf"])
    margin(f"golden_fused3[{name}]", iq=e_iq, audio=e_au)
    assert e_iq <= RTOL and e_au <= RTOL


@pytest.mark.parametrize("offs", OFFSETS[:2])
def test_other_cuts_of_the_stream_agree_to_rounding(offs):
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(77, 0, 2 * L + 1234, 0.25, [t for t in tabs[:1] if t is not None], [0.3], ch.N, 10000)
    whole = _run(ch, tabs, _cut(stream, [L, L, 10 ** 9]))
    again = _run(ch, tabs, _cut(stream, [L, L, 10 ** 9]))
    assert np.array_equal(whole[0].view(np.uint32), again[0].view(np.uint32))          # the same cut: the same bits
    worst = 0.0
    for lens in ([L, 0, 624, 625, 777, L // 2, 10 ** 9], [1, 24, 25, 26, 99, 100, 101, 1199, 1200, 1201, 5000, 10 ** 9], [L // 3] * 7):
        got = _run(ch, tabs, _cut(stream, lens))
        worst = max(worst, rel_err_rows(got[0], whole[0]), rel_err_rows(got[1], whole[1]))
    margin("fused3_cut_invariance_%dch" % len(offs), worst=worst)
    assert worst <= RTOL


@pytest.mark.parametrize("gs,gt", [(4, 2), (2, 4)])
def test_workgroup_shapes_give_the_same_bits(gs, gt):
    ch = chains.chain("10")
    offs = list(range(-40, 41)) + [0, 7, 7]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(78, 0, L + 30000, 0.25)
    lens = [L, 625, 10 ** 9]
    want = _run(ch, tabs, _cut(stream, lens))
    got = _run(ch, tabs, _cut(stream, lens), tuning={"m_slot_groups": gs, "m_seg_groups": gt})
    assert np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32)) and np.array_equal(want[1].view(np.uint32), got[1].view(np.uint32))


def test_a_channels_bits_do_not_depend_on_who_else_is_in_the_bank():
    ch = chains.chain("10")
    L = ch.block_len
    stream = [synth.synth_iq(79, 0, L, 0.25), synth.synth_iq(79, L, 12345, 0.25)]
    t = sb.translators_for(ch, [37, -37, 0, 599])
    alone = [_run(ch, [t[0]], stream)[0][0], _run(ch, [t[1]], stream)[0][0], _run(ch, [t[3], t[2]], stream)[0][1], _run(ch, [t[3]], stream)[0][0]]
    both = _run(ch, t, stream)[0]
    many = _run(ch, t * 9 + [t[0]], stream)[0]
    for i in range(4):
        assert np.array_equal(alone[i].view(np.uint32), both[i].view(np.uint32))
        for k in range(9):
            assert np.array_equal(alone[i].view(np.uint32), many[4 * k + i].view(np.uint32))


@pytest.mark.parametrize("n_ch", [3584, 10000])
def test_large_bank_with_the_flag_all_offsets_against_the_oracle(oracle, n_ch):
    """A large 10 MS/s bank created with SDRXGPU_FLAG_FUSED_STAGE3 (without it: k_stage12m + k_tail_s); 10 000 = the bench workload.
    Host entry (four channel parts, D2H beside the kernels) and device entry give the same bits; every one of the 1 200 distinct
    offsets against the oracle."""
    import torch
    ch = chains.chain("10")
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = [synth.synth_iq(9, b * L, L, 0.25, [tabs[650]], [0.2], ch.N, 10000) for b in range(2)]
    plain = sb.ChannelBank(tabs[:3600], ch.stage_tuples())
    assert plain.info().matrix_pipe == 1                     # not without being asked for
    plain.close()
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), fused_stage3=True)
    assert bank.info().matrix_pipe == 2
    outs = [bank.process(x) for x in blocks]
    bank.close()
    iq = np.concatenate([o[0] for o in outs], axis=1)
    au = np.concatenate([o[1] for o in outs], axis=1)
    for c in range(0, n_ch - ch.N, 97):                      # repeated instances: independent work, identical bits
        assert np.array_equal(iq[c].view(np.uint32), iq[c + ch.N].view(np.uint32))
    dev = sb.ChannelBank(tabs, ch.stage_tuples(), fused_stage3=True)
    d_iq = torch.zeros(n_ch, 512, 2, device="cuda")
    d_au = torch.zeros(n_ch, 512, device="cuda")
    for k, x in enumerate(blocks):
        xt = torch.from_numpy(x.view(np.float32).reshape(L, 2).copy()).cuda()
        assert dev.process_device(xt.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512) == 512
        dev.synchronize()
        assert np.array_equal(d_iq.cpu().numpy().view(np.complex64)[..., 0].view(np.uint32), iq[:, 512 * k:512 * (k + 1)].view(np.uint32))
        assert np.array_equal(d_au.cpu().numpy().view(np.uint32), au[:, 512 * k:512 * (k + 1)].view(np.uint32))
    dev.close()
    pick = list(range(ch.N)) + [c for c in (ch.N, 2559, 2560, 4999, 7777, 9999) if c < n_ch]
    ob = oracle.bank([tabs[c] for c in pick], ch.stage_tuples())
    ref = [ob.process(x, threads=os.cpu_count() or 8) for x in blocks]
    riq, rau = np.concatenate([r[0] for r in ref], axis=1), np.concatenate([r[1] for r in ref], axis=1)
    e_iq = np.array([rel_err(iq[c], riq[i]) for i, c in enumerate(pick)])
    e_au = np.array([rel_err(au[c], rau[i]) for i, c in enumerate(pick)])
    print("worst channels (iq):", [(pick[i], offs[pick[i]], float(e_iq[i])) for i in np.argsort(e_iq)[-3:]])
    margin("fused3_%dch_vs_oracle" % n_ch, iq=float(e_iq.max()), audio=float(e_au.max()))
    assert e_iq.max() <= RTOL and e_au.max() <= RTOL


def test_fused_path_with_ftfir_squelch_fm_and_audio_fir_against_the_vector_kernels():
    ch = chains.chain("10")
    offs = list(range(-12, 13)) + [300, -300, 7]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    pb, ru, rd = chains.squelch_tables()
    on = [False, True, True, False]
    blocks = [synth.synth_iq(3, b * L, L, 0.25, [tabs[1]] if c else [], [0.25] if c else [], ch.N, 10000) for b, c in enumerate(on)]
    kw = dict(ftfir=True, squelch=(np.full(len(offs), 6.0, np.float32), pb, ru, rd),
              modulation=["FM" if i % 5 == 0 else "AM" for i in range(len(offs))], audio_fir=chains.audio_bpf_taps())
    m = _run(ch, tabs, blocks, **kw)
    vb = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": 1}, **kw)
    v = [vb.process(x) for x in blocks]
    vb.close()
    v_iq, v_au = np.concatenate([o[0] for o in v], axis=1), np.concatenate([o[1] for o in v], axis=1)
    e_iq, e_au = rel_err_rows(m[0], v_iq), rel_err_rows(m[1], v_au)
    margin("fused3_ftfir_squelch_fm_vs_vector_kernels", iq=e_iq, audio=e_au)
    assert e_iq <= RTOL and e_au <= RTOL


def test_fused_path_is_refused_where_it_cannot_run():
    ch = chains.chain("10")
    with pytest.raises(sb.SdrxGpuError):
        sb.ChannelBank(sb.translators_for(ch, [1, -1]), ch.stage_tuples(), tuning={"matrix_pipe": FUSED3}, block_hint=3)   # whole segments only
    with pytest.raises(sb.SdrxGpuError):
        sb.ChannelBank(sb.translators_for(ch, [1, -1]), ch.stage_tuples(), tuning={"matrix_pipe": FUSED3}, pipelined=True)
    ch24 = chains.chain("2.4")
    with pytest.raises(sb.SdrxGpuError):
        sb.ChannelBank(sb.translators_for(ch24, [1, -1]), ch24.stage_tuples(), tuning={"matrix_pipe": FUSED3})


def test_200_blocks_in_ragged_calls_through_the_fused_path(oracle):
    from test_gpu_long_stream import N_BLOCKS, OFFS, _stream_and_cuts
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, OFFS)
    n_ch, L = len(OFFS), ch.block_len
    stream, cuts, total = _stream_and_cuts(ch, tabs)
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": FUSED3})
    ob = oracle.bank(tabs, ch.stage_tuples())
    got_iq = np.zeros((n_ch, N_BLOCKS * 512), np.complex64)
    got_au = np.zeros((n_ch, N_BLOCKS * 512), np.float32)
    p_out = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        iq, au = bank.process(stream(a, b))
        assert iq.shape[1] == b // 625 - a // 625
        got_iq[:, p_out:p_out + iq.shape[1]] = iq
        got_au[:, p_out:p_out + iq.shape[1]] = au
        p_out += iq.shape[1]
    info = bank.info()
    assert info.samples_in == total and info.samples_out == N_BLOCKS * 512 == p_out and info.matrix_pipe == 2
    bank.close()
    worst_iq = worst_au = 0.0
    for k in range(N_BLOCKS):
        r_iq, r_au = ob.process(stream(k * L, (k + 1) * L), threads=n_ch)
        sl = slice(k * 512, (k + 1) * 512)
        worst_iq = max(worst_iq, max(rel_err(got_iq[c, sl], r_iq[c]) for c in range(n_ch)))
        worst_au = max(worst_au, max(rel_err(got_au[c, sl], r_au[c]) for c in range(n_ch)))
    margin("long_stream_200_blocks[fused3]", worst_iq=worst_iq, worst_audio=worst_au)
    assert worst_iq <= RTOL and worst_au <= RTOL
