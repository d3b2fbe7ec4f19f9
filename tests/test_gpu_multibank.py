"""The multi-device bank behind the C ABI (sdrxgpu_multibank_*): one channel list cut over several devices in ONE process --
what the reference's one-process / one-device-thread / worker-per-channel shape (src/sdrx.cpp:306-364, src/ds.hpp:43-77)
becomes across the GPUs of a node.  The box has one GPU, so the shards are rehearsed as devices = {0, 0}, {0, 0, 0}: two or
three banks, threads and stream sets on the one card -- every code path of the N-device case but the PCIe links."""
import numpy as np
import pytest
import torch

from conftest import RTOL, rel_err_rows
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu


def _blocks(ch, n, car, seed=21):
    L = ch.block_len
    return [synth.synth_iq(seed, b * L, L, 0.25, car, [0.3] * len(car), ch.N, 10000) for b in range(n)]


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("n_ch,devices", [(1024, (0, 0)), (2500, (0, 0, 0)), (64, (0, 0)), (5, (0, 0, 0, 0, 0, 0, 0, 0))])
def test_shards_give_the_single_banks_bits_and_the_oracles_values(oracle, n_ch, devices):
    ch = chains.chain("10")
    offs = chains.offsets_for_config("c3") if n_ch == 1024 else chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    blocks = _blocks(ch, 2, [tabs[3]])
    one = sb.ChannelBank(tabs, ch.stage_tuples())
    fam = one.info().matrix_pipe
    want = [one.process(x) for x in blocks]
    one.close()
    mb = sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=devices)
    info = mb.info()
    assert info.n_shards == min(len(devices), n_ch) and info.n_channels == n_ch and info.m_total == 625
    # the kernel family is the WHOLE list's choice, pinned in every shard (a 512-channel shard alone would take the vector kernels)
    assert info.matrix_pipe == fam
    sh = mb.shards()
    assert [s[3].matrix_pipe for s in sh] == [fam] * info.n_shards
    assert sh[0][1] == 0 and sh[-1][2] == n_ch and all(sh[i][2] == sh[i + 1][1] for i in range(len(sh) - 1))
    assert [(s[1], s[2]) for s in sh] == [sb.shard_range(n_ch, i, info.n_shards) for i in range(info.n_shards)]
    for k, x in enumerate(blocks):
        iq, au = mb.process(x)
        assert iq.shape == (n_ch, 512)
        assert np.array_equal(_bits(iq), _bits(want[k][0])) and np.array_equal(_bits(au), _bits(want[k][1]))
    assert mb.info().samples_in == 2 * ch.block_len and mb.info().samples_out == 1024
    if n_ch <= 1024:
        ref = oracle.bank(tabs, ch.stage_tuples())
        for k, x in enumerate(blocks):
            r_iq, r_au = ref.process(x, threads=8)
        assert rel_err_rows(iq, r_iq) <= RTOL and rel_err_rows(au, r_au) <= RTOL
    mb.close()


def test_ragged_calls_reset_state_and_refusals():
    ch = chains.chain("10")
    n_ch = 300
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    x = np.concatenate(_blocks(ch, 2, [tabs[5]]))
    one = sb.ChannelBank(tabs, ch.stage_tuples(), lf_agc=True)
    w_iq, w_au = one.process(x)
    w_state = one.agc_state()
    one.close()
    mb = sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0), lf_agc=True)
    cuts = (0, 1000, 320625, 320625, 500001, len(x))
    parts = [mb.process(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    assert [p[0].shape[1] for p in parts] == [1, 512, 0, 287, 224]
    assert np.array_equal(_bits(np.concatenate([p[0] for p in parts], axis=1)), _bits(w_iq))
    assert np.array_equal(_bits(np.concatenate([p[1] for p in parts], axis=1)), _bits(w_au))
    for a, b in zip(mb.agc_state(), w_state):
        assert np.array_equal(_bits(a), _bits(b))
    # a refused call (rows too short) is refused before any shard has been touched: the stream goes on as if it had not been made
    mb.reset()
    assert mb.info().samples_in == 0
    a1 = mb.process(x[:320000])
    small = np.zeros((n_ch, 100), np.complex64)
    with pytest.raises(sb.SdrxGpuError) as e:
        mb.process(x[320000:], iq_out=small)
    assert e.value.code == -6
    assert mb.info().samples_in == 320000
    a2 = mb.process(x[320000:])
    assert np.array_equal(_bits(np.concatenate([a1[0], a2[0]], axis=1)), _bits(w_iq))
    mb.close()
    # what a multibank does not offer fails loudly at create
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0), pan_pos=[0] * n_ch)
    assert e.value.code == -5
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 7))
    assert e.value.code == -2
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0), pipelined=True)
    assert e.value.code == -5


def test_page_locked_block_and_chunk_zero_copy_and_the_pinned_entry():
    """The caller's IQ buffer and ring chunk page-locked (sdrxgpu_host_register, portable): every shard's H2D and D2H are DMAs
    from / into the same two host ranges.  And the throughput entry: page-locked block in, per-shard device rows out."""
    ch = chains.chain("10")
    n_ch = 1200
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    blocks = _blocks(ch, 3, [tabs[9]])
    one = sb.ChannelBank(tabs, ch.stage_tuples())
    want = [one.process(x) for x in blocks]
    one.close()
    mb = sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0, 0))
    L = ch.block_len
    xin = np.zeros(L, np.complex64)
    chunk = np.zeros((n_ch, 512), np.complex64)
    audio = np.zeros((n_ch, 512), np.float32)
    for a_ in (xin, chunk, audio):
        sb.host_register(a_)
    try:
        for k, x in enumerate(blocks[:2]):
            xin[:] = x
            iq, au = mb.process(xin, iq_out=chunk, audio_out=audio)
            assert np.array_equal(_bits(iq), _bits(want[k][0])) and np.array_equal(_bits(au), _bits(want[k][1]))
        sh = mb.shards()
        d_iq = [torch.zeros(hi - lo, 512, 2, device="cuda") for _, lo, hi, _ in sh]
        d_au = [torch.zeros(hi - lo, 512, device="cuda") for _, lo, hi, _ in sh]
        xin[:] = blocks[2]
        assert mb.process_pinned(xin.ctypes.data, L, [t.data_ptr() for t in d_iq], [t.data_ptr() for t in d_au], 512) == 512
        mb.synchronize()
        got_iq = np.concatenate([t.cpu().numpy().view(np.complex64)[..., 0] for t in d_iq])
        got_au = np.concatenate([t.cpu().numpy() for t in d_au])
        assert np.array_equal(_bits(got_iq), _bits(want[2][0])) and np.array_equal(_bits(got_au), _bits(want[2][1]))
    finally:
        mb.close()
        for a_ in (xin, chunk, audio):
            sb.host_unregister(a_)


def test_squelch_and_modulation_arrays_are_cut_with_the_channels():
    ch = chains.chain("10")
    n_ch = 96
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    blocks = _blocks(ch, 3, [tabs[2], tabs[50]])
    sq_tabs = chains.squelch_tables()
    level = np.where(np.arange(n_ch) % 3 == 0, 0.0, 6.0).astype(np.float32)
    mod = [("FM" if c % 5 == 4 else "AM") for c in range(n_ch)]
    kw = dict(squelch=(level,) + tuple(sq_tabs), modulation=mod)
    one = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
    want = [one.process(x) for x in blocks]
    w_sq, w_gains = one.squelch_state(), one.gains()
    one.close()
    mb = sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0, 0), **kw)
    for k, x in enumerate(blocks):
        iq, au = mb.process(x)
        assert np.array_equal(_bits(iq), _bits(want[k][0])) and np.array_equal(_bits(au), _bits(want[k][1]))
    # what sdrx prints per channel (sdrx.cpp:593-634), over all shards in channel order
    sq, gains = mb.squelch_state(), mb.gains()
    assert np.array_equal(_bits(sq[0]), _bits(w_sq[0])) and np.array_equal(sq[1], w_sq[1]) and len(set(sq[1].tolist())) == 2
    assert np.array_equal(_bits(gains[0]), _bits(w_gains[0])) and np.array_equal(_bits(gains[1]), _bits(w_gains[1]))
    mb.close()
