"""Mixer + stage 1 of the 10 MS/s chain on the f32 matrix pipe (k_stage12m): the same channel outputs as the vector
kernels up to fp32 rounding (another summation order: the reference's own AVX2 and portable builds differ likewise), so
this path is held to the oracle at north_star's 1e-5 -- and to ITSELF bit for bit wherever the reference is
deterministic: however the stream is cut into calls, whatever the workgroup shape or the lane length, whoever else is
in the bank (a channel's partner, repeated instances), host entry or device entry."""
import os

import numpy as np
import pytest

from conftest import RTOL, load_golden, regen_blocks, rel_err, rel_err_rows, margin
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu

WHOLE = 0xFFFFFFFF
OFF, REQ = 1, 2


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
    r = [ob.process(x) for x in blocks]
    return np.concatenate([v[0] for v in r], axis=1), np.concatenate([v[1] for v in r], axis=1)


OFFSETS = [
    list(range(-20, 21)),                                  # 20 conjugate pairs + the untranslated channel
    [5, -5, 5, -5, 7, 9, 11, 0, 0, -7, 600 - 1200, 13],     # repeated offsets, loners, two untranslated, -600 (its own conjugate)
    [3],                                                   # a single channel
    [4, -4],                                               # one pair
    list(range(1, 38)),                                    # no partners at all
]


@pytest.mark.parametrize("offs", OFFSETS)
def test_matrix_pipe_matches_the_oracle(oracle, offs):
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(31, 0, 2 * L + 4321, 0.25, [t for t in tabs[:1] if t is not None], [0.3], ch.N, 10000)
    lens = [L, 0, 624, 625, 777, L // 2, 10 ** 9]
    iq, au, info = _run(ch, tabs, _cut(stream, lens))
    assert info.matrix_pipe == 1 and info.fused_tail == 0 and info.pair_lanes == 0
    riq, rau = _oracle(oracle, ch, tabs, _cut(stream, lens))
    e_iq, e_au = rel_err_rows(iq, riq), rel_err_rows(au, rau)
    margin("matrix_pipe_vs_oracle_%dch" % len(offs), iq=e_iq, audio=e_au)
    assert e_iq <= RTOL and e_au <= RTOL


def test_matrix_pipe_is_close_to_the_vector_kernels_but_not_equal():
    ch = chains.chain("10")
    offs = list(range(-20, 21))
    tabs = sb.translators_for(ch, offs)
    blocks = [synth.synth_iq(5, 0, ch.block_len, 0.25, [tabs[3]], [0.3], ch.N, 10000)]
    m = _run(ch, tabs, blocks)
    v = _run(ch, tabs, blocks, tuning={"matrix_pipe": OFF})
    assert v[2].matrix_pipe == 0
    assert rel_err_rows(m[0], v[0]) <= RTOL              # (two summation orders, each within its own error of the exact result)
    assert not np.array_equal(m[0].view(np.uint32), v[0].view(np.uint32))      # (if this ever fails the two paths have become one)


@pytest.mark.parametrize("offs", OFFSETS[:2])
def test_any_split_of_the_stream_gives_the_same_bits(offs):
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(77, 0, 2 * L + 1234, 0.25, [t for t in tabs[:1] if t is not None], [0.3], ch.N, 10000)
    whole = _run(ch, tabs, _cut(stream, [L, L, 10 ** 9]))
    for lens in ([L, 0, 624, 625, 777, L // 2, 10 ** 9], [1, 24, 25, 26, 99, 100, 101, 1199, 1200, 1201, 5000, 10 ** 9], [L // 3] * 7):
        got = _run(ch, tabs, _cut(stream, lens))
        assert np.array_equal(whole[0].view(np.uint32), got[0].view(np.uint32))
        assert np.array_equal(whole[1].view(np.uint32), got[1].view(np.uint32))


@pytest.mark.parametrize("gs,gt", [(4, 2), (2, 4)])
def test_workgroup_shapes_and_lane_lengths_give_the_same_bits(gs, gt):
    ch = chains.chain("10")
    offs = list(range(-40, 41)) + [0, 7, 7]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(78, 0, L + 30000, 0.25)
    lens = [L, 625, 10 ** 9]
    want = _run(ch, tabs, _cut(stream, lens), block_hint=WHOLE)
    for hint in (WHOLE, 1, 3, 4):
        got = _run(ch, tabs, _cut(stream, lens), tuning={"m_slot_groups": gs, "m_seg_groups": gt}, block_hint=hint)
        assert got[2].k12_bodies == (12 if hint == WHOLE else hint)
        assert np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32))
        assert np.array_equal(want[1].view(np.uint32), got[1].view(np.uint32))


@pytest.mark.parametrize("gs,gt", [(2, 2), (4, 2), (2, 4)])
def test_streaming_scratch_stores_give_the_same_bits(gs, gt):
    """k_stage12m writes the stage-2 scratch of all but the sets written last with the streaming hint (tuning.y2_keep_mb; automatic:
    the last 192 MB of a launch stay ordinary stores).  A cache policy, not arithmetic: everything streaming, nothing streaming, a cut
    in the middle of the bank (the streaming arm is inline assembly behind a branch per workgroup) -- the same bits, over ragged calls."""
    ch = chains.chain("10")
    offs = chains.offsets_cycled(300, ch.N) + [0, 7, 7]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    stream = synth.synth_iq(91, 0, L + 40000, 0.25, [tabs[3]], [0.3], ch.N, 10000)
    lens = [L, 625, 10 ** 9]
    shape = {"m_slot_groups": gs, "m_seg_groups": gt}
    want = _run(ch, tabs, _cut(stream, lens), tuning=dict(shape, y2_keep_mb=0xFFFFFFFF))
    # (a set of 32 x gs rows x 12 800 new outputs x 8 B is 6.5 MB at gs = 2: "keep 14 MB" leaves the last two sets ordinary)
    for keep in (1, 1 + 14, 0):
        got = _run(ch, tabs, _cut(stream, lens), tuning=dict(shape, y2_keep_mb=keep))
        assert np.array_equal(want[0].view(np.uint32), got[0].view(np.uint32)), keep
        assert np.array_equal(want[1].view(np.uint32), got[1].view(np.uint32)), keep


def test_a_channels_bits_do_not_depend_on_who_else_is_in_the_bank():
    ch = chains.chain("10")
    L = ch.block_len
    stream = [synth.synth_iq(79, 0, L, 0.25)]
    t = sb.translators_for(ch, [37, -37, 0, 599])
    # each channel with as little company as a bank allows (the untranslated one needs somebody who fixes the table length)
    alone = [_run(ch, [t[0]], stream)[0][0], _run(ch, [t[1]], stream)[0][0], _run(ch, [t[3], t[2]], stream)[0][1], _run(ch, [t[3]], stream)[0][0]]
    both = _run(ch, t, stream)[0]
    many = _run(ch, t * 9 + [t[0]], stream)[0]
    swapped = _run(ch, [t[1], t[0]], stream)[0]                  # the partner comes first: roles in the slot are exchanged
    for i in range(4):
        assert np.array_equal(alone[i].view(np.uint32), both[i].view(np.uint32))
        for k in range(9):
            assert np.array_equal(alone[i].view(np.uint32), many[4 * k + i].view(np.uint32))
    assert np.array_equal(alone[0].view(np.uint32), many[36].view(np.uint32))
    assert np.array_equal(alone[0].view(np.uint32), swapped[1].view(np.uint32))
    assert np.array_equal(alone[1].view(np.uint32), swapped[0].view(np.uint32))


def test_automatic_choice_goes_with_the_partners_found_not_with_the_channel_count():
    # k_stage12m's work goes with its slots: a channel without its conjugate partner costs what a pair costs.  The automatic
    # choice (bank_create: the two kernels' launch counts) takes the matrix pipe for a bank whose channels mostly come in +k / -k
    # pairs and the vector kernels for one of the same size without any; below 1 024 channels always the vector kernels.
    ch = chains.chain("10")
    want = {}
    for name, offs in (("paired_2400", chains.offsets_cycled(2400, ch.N)),
                       ("one_sided_2400", [1 + (i % 599) for i in range(2400)]),
                       ("config_3_1024", chains.offsets_for_config("c3")),
                       ("one_sided_1024", list(range(1, 1025))),
                       ("paired_900", [o for o in range(-450, 451) if o != 0]),
                       ("one_sided_6000", [1 + (i % 599) for i in range(6000)])):
        bank = sb.ChannelBank(sb.translators_for(ch, offs), ch.stage_tuples())
        want[name] = int(bank.info().matrix_pipe)
        bank.close()
    # (what does NOT depend on the device: no partners -> never the matrix pipe; under 1 024 channels -> never.  The paired banks'
    # answer comes from the launch-count model, which counts the device's CUs: on a 256-CU MI355X it says yes for both)
    assert (want["one_sided_2400"], want["one_sided_1024"], want["paired_900"], want["one_sided_6000"]) == (0, 0, 0, 0)
    import torch
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert (want["paired_2400"], want["config_3_1024"]) == (1, 1)


@pytest.mark.parametrize("n_ch", [3584, 10000])
def test_large_bank_host_parts_and_oracle(oracle, n_ch):
    """k_stage12m + k_tail_s at the sizes it was the default for in round 3; 10 000 = the bench workload, four host parts."""
    ch = chains.chain("10")
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = [synth.synth_iq(9, b * L, L, 0.25, [tabs[650]], [0.2], ch.N, 10000) for b in range(2)]
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": REQ})     # (what such a bank takes by itself; pinned: the test is about this kernel)
    info = bank.info()
    assert info.matrix_pipe == 1
    outs = [bank.process(x) for x in blocks]                # host entry: channel parts, D2H beside the kernels
    bank.close()
    iq = np.concatenate([o[0] for o in outs], axis=1)
    au = np.concatenate([o[1] for o in outs], axis=1)
    # repeated instances of an offset are computed independently and come out bit-identical
    for c in range(0, n_ch - ch.N, 97):
        assert np.array_equal(iq[c].view(np.uint32), iq[c + ch.N].view(np.uint32))
    # every one of the 1 200 distinct offsets (channels 0 .. 1 199) and a few later instances against the oracle
    pick = list(range(ch.N)) + [c for c in (ch.N, 2559, 2560, 4999, 7777, 9999) if c < n_ch]
    ob = oracle.bank([tabs[c] for c in pick], ch.stage_tuples())
    ref = [ob.process(x, threads=os.cpu_count() or 8) for x in blocks]
    riq, rau = np.concatenate([r[0] for r in ref], axis=1), np.concatenate([r[1] for r in ref], axis=1)
    e_iq = np.array([rel_err(iq[c], riq[i]) for i, c in enumerate(pick)])
    e_au = np.array([rel_err(au[c], rau[i]) for i, c in enumerate(pick)])
    worst_iq, worst_au = float(e_iq.max()), float(e_au.max())
    print("worst channels (iq):", [(pick[i], offs[pick[i]], float(e_iq[i])) for i in np.argsort(e_iq)[-3:]])
    margin("matrix_pipe_%dch_vs_oracle" % n_ch, iq=worst_iq, audio=worst_au)
    assert worst_iq <= RTOL and worst_au <= RTOL


def test_matrix_pipe_with_ftfir_squelch_fm_lf_agc_and_audio_fir(oracle):
    """Everything around the stage-1/2 kernel is unchanged: --ftfir (tables doubled and phase-shifted: still conjugates of
    each other), squelch + FM + audio FIR downstream."""
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
    v = _run(ch, tabs, blocks, tuning={"matrix_pipe": OFF}, **kw)
    e_iq, e_au = rel_err_rows(m[0], v[0]), rel_err_rows(m[1], v[1])
    margin("matrix_pipe_ftfir_squelch_fm_vs_vector_kernels", iq=e_iq, audio=e_au)
    assert e_iq <= RTOL and e_au <= RTOL


def test_matrix_pipe_is_refused_where_it_cannot_run():
    ch = chains.chain("2.4")
    with pytest.raises(sb.SdrxGpuError):
        sb.ChannelBank(sb.translators_for(ch, [1, -1]), ch.stage_tuples(), tuning={"matrix_pipe": REQ})
    ch = chains.chain("10")
    with pytest.raises(sb.SdrxGpuError):
        sb.ChannelBank(sb.translators_for(ch, [1, -1]), ch.stage_tuples(), tuning={"matrix_pipe": REQ, "m_slot_groups": 3, "m_seg_groups": 1})
    with pytest.raises(sb.SdrxGpuError):
        sb.ChannelBank([None, None], ch.stage_tuples(), tuning={"matrix_pipe": REQ})        # nobody mixes: no table length to space the segments by


@pytest.mark.parametrize("name", ["g2_fs10000", "g3_fs10000_noise", "g4_fs10000_impulse", "g4_fs10000_dc", "g4_fs10000_tone",
                                  "g5_fs10000_ftfir", "g6_fs10000_ragged"])
def test_golden_vectors_through_the_matrix_pipe(name):
    """The reference's own outputs (compiled from its headers, tests/golden): offsets {0, +-1, 37, -415, 479, 480}, noise
    only, impulse / DC / tone known answers, --ftfir, ragged call lengths."""
    g = load_golden(name)
    ch, blocks = regen_blocks(g)
    tabs = [g["tables"][i] if int(o) != 0 else None for i, o in enumerate(g["offsets"])]
    if all(t is None for t in tabs):
        pytest.skip("nobody mixes: no translator length to space the segments by (such banks run the vector kernels)")
    iq, au, info = _run(ch, tabs, blocks, ftfir="ftfir" in name)
    assert info.matrix_pipe == 1
    e_iq, e_au = rel_err_rows(iq, g["iq"]), rel_err_rows(au, g["audio_lf"])
    margin(f"golden_matrix_pipe[{name}]", iq=e_iq, audio=e_au)
    assert e_iq <= RTOL and e_au <= RTOL


def test_200_blocks_in_ragged_calls_through_the_matrix_pipe(oracle):
    """tests/test_gpu_long_stream.py's stream (6.4 s of signal, three carriers up to 0.6 of full scale, ragged calls) through
    k_stage12m: within 1e-5 of the free-running oracle at every block, exact sample counts."""
    from test_gpu_long_stream import N_BLOCKS, OFFS, _stream_and_cuts
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, OFFS)
    n_ch, L = len(OFFS), ch.block_len
    stream, cuts, total = _stream_and_cuts(ch, tabs)
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": REQ})
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
    assert info.samples_in == total and info.samples_out == N_BLOCKS * 512 == p_out and info.matrix_pipe == 1
    bank.close()
    worst_iq = worst_au = last_iq = last_au = 0.0
    for k in range(N_BLOCKS):
        r_iq, r_au = ob.process(stream(k * L, (k + 1) * L), threads=n_ch)
        sl = slice(k * 512, (k + 1) * 512)
        last_iq = max(rel_err(got_iq[c, sl], r_iq[c]) for c in range(n_ch))
        last_au = max(rel_err(got_au[c, sl], r_au[c]) for c in range(n_ch))
        worst_iq, worst_au = max(worst_iq, last_iq), max(worst_au, last_au)
    margin("long_stream_200_blocks[matrix_pipe]", worst_iq=worst_iq, worst_audio=worst_au, last_block_iq=last_iq, last_block_audio=last_au)
    assert worst_iq <= RTOL and worst_au <= RTOL


def test_reference_order_flag_keeps_a_paired_bank_on_the_vector_kernels():
    """SDRXGPU_FLAG_REFERENCE_ORDER: what an application sets to get only the kernels that add every FIR in the reference's own
    order (msd.hpp:309-331) -- the bank that would take the matrix pipe by itself stays on the vector kernels, bit-identical to
    tuning.matrix_pipe = 1, also cut over devices; asking for the matrix pipe at the same time is refused."""
    ch = chains.chain("10")
    offs = chains.offsets_for_config("c3")
    tabs = sb.translators_for(ch, offs)
    x = synth.synth_iq(3, 0, ch.block_len, 0.25, [tabs[5]], [0.3], ch.N, 10000)
    auto = sb.ChannelBank(tabs, ch.stage_tuples())
    assert auto.info().matrix_pipe == 1
    auto.close()
    strict = sb.ChannelBank(tabs, ch.stage_tuples(), reference_order=True)
    assert strict.info().matrix_pipe == 0
    a = strict.process(x)
    strict.close()
    vec = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": 1})
    b = vec.process(x)
    vec.close()
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    mb = sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0), reference_order=True)
    assert mb.info().matrix_pipe == 0
    c = mb.process(x)
    mb.close()
    assert np.array_equal(a[0].view(np.uint32), c[0].view(np.uint32)) and np.array_equal(a[1].view(np.uint32), c[1].view(np.uint32))
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank(tabs, ch.stage_tuples(), reference_order=True, tuning={"matrix_pipe": 2})
    assert e.value.code == -1
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank(tabs, ch.stage_tuples(), reference_order=True, fused_stage3=True)
    assert e.value.code == -1
