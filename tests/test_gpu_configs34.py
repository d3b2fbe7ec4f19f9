"""BASELINE configs[3] and configs[4] at their FULL size on the one GPU of the box.

configs[3]: 8 192 channels from one 10 MS/s IQ stream, 1 024 per GPU on 8 GPUs -- here the whole channel list of
``dist.plan("c4", r, 8)`` over 8 shards of a sdrxgpu_multibank that all sit on device 0 (8 banks, 8 worker threads, 8 H2D
copies of every block, every shard's rows straight into its slice of the caller's chunk).  configs[4]: 16 independent 10 MS/s
streams x 1 024 channels, 2 per GPU on 8 GPUs -- here 16 banks on device 0, each driven from a thread of its own through
sdrxgpu_bank_process_pinned, as 16 device threads of 16 receivers would.  What one device cannot show is the eight PCIe
links and eight devices working at once (N > 1 unmeasured); everything else of the N-device code runs.

Reference: one independent worker per channel, src/ds.hpp:43-77; dispatch src/sdrx.cpp:327-340; the per-channel chain
src/msd.hpp:132-155, src/agc.hpp:48-69, src/sdrx.cpp:455-463."""
import threading

import numpy as np
import pytest
import torch

from conftest import RTOL, margin, rel_err_rows
from sdrx_amd import bank as sb, chains, dist as sdist, synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _cut(x, cuts):
    return [x[a:b] for a, b in zip(cuts[:-1], cuts[1:])]


def test_config3_8192_channels_over_8_shards(oracle):
    ch = chains.chain("10")
    L, M = ch.block_len, ch.m_total
    plans = [sdist.plan("c4", r, 8, ch.N) for r in range(8)]
    assert all(p.broadcast and p.channels_total == 8192 for p in plans)
    offs = [o for p in plans for o in p.offsets[0]]                     # rank r's 1 024 channels = shard r's
    n_ch = len(offs)
    assert n_ch == 8192
    tabs = sb.translators_for(ch, offs)
    car = [tabs[5], tabs[1024 + 700]]
    x = synth.synth_iq(1234, 0, 3 * L + 3 * M + 11, 0.25, car, [0.3, 0.1], ch.N, ch.fs // 1000)
    # ragged calls: a whole block, calls that end inside an output period, one without any output, one longer than a block
    # (split inside the library)
    cuts = (0, L, L + 7, L + 7 + 100 * M - 3, L + 7 + 100 * M - 3 + 2, len(x))
    calls = _cut(x, cuts)

    one = sb.ChannelBank(tabs, ch.stage_tuples())
    fam = one.info().matrix_pipe
    want = [one.process(c) for c in calls]
    w_state = one.agc_state()
    one.close()
    assert [w[0].shape[1] for w in want] == [512, 0, 100, 0, len(x) // M - 612]
    w_iq = np.concatenate([w[0] for w in want], axis=1)
    w_au = np.concatenate([w[1] for w in want], axis=1)

    # every shard's first and last channel + a spread of others (>= 64), against the oracle on the same stream
    sample = sorted({sb.shard_range(n_ch, i, 8)[0] for i in range(8)} | {sb.shard_range(n_ch, i, 8)[1] - 1 for i in range(8)} |
                    set(range(3, n_ch, 149)))
    assert len(sample) >= 64
    ref = oracle.bank([tabs[c] for c in sample], ch.stage_tuples())
    r_iq, r_au = ref.process(x, threads=8)
    e_iq, e_au = rel_err_rows(w_iq[sample], r_iq), rel_err_rows(w_au[sample], r_au)
    margin("config3_8192_single_bank_vs_oracle", iq=e_iq, audio=e_au)
    assert e_iq <= RTOL and e_au <= RTOL

    for tuning, tail_agc_in_shards in ((None, False), ({"tail_agc": 2}, True)):
        mb = sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0,) * 8, tuning=tuning)
        info = mb.info()
        assert info.n_shards == 8 and info.n_channels == n_ch and info.matrix_pipe == fam
        sh = mb.shards()
        assert [(s[1], s[2]) for s in sh] == [(1024 * i, 1024 * (i + 1)) for i in range(8)]
        assert [s[3].matrix_pipe for s in sh] == [fam] * 8
        # shards that share a device keep k_tail_s + k_agc; tuning.tail_agc = 2 makes every shard run k_tail_agc, the kernel a
        # 1 024-channel shard runs when it is alone on its GPU (8 real devices): same bits either way
        assert [s[3].tail_agc > 0 for s in sh] == [tail_agc_in_shards] * 8
        got = [mb.process(c) for c in calls]
        g_iq = np.concatenate([g[0] for g in got], axis=1)
        g_au = np.concatenate([g[1] for g in got], axis=1)
        assert np.array_equal(_bits(g_iq), _bits(w_iq)) and np.array_equal(_bits(g_au), _bits(w_au))
        for a, b in zip(mb.agc_state(), w_state):
            assert np.array_equal(_bits(a), _bits(b))
        assert mb.info().samples_in == len(x) and mb.info().samples_out == len(x) // M
        e_iq, e_au = rel_err_rows(g_iq[sample], r_iq), rel_err_rows(g_au[sample], r_au)
        assert e_iq <= RTOL and e_au <= RTOL
        mb.close()


def test_config4_16_streams_of_1024_channels_from_16_threads(oracle):
    ch = chains.chain("10")
    L = ch.block_len
    n_streams, n_blocks = 16, 3
    plans = [sdist.plan("c5", r, 8, ch.N) for r in range(8)]
    seeds = [s for p in plans for s in p.seeds]
    assert seeds == list(range(1, 17)) and all(not p.broadcast for p in plans)
    offs = plans[0].offsets[0]
    assert len(offs) == 1024 and all(p.offsets == [offs, offs] for p in plans)
    tabs = sb.translators_for(ch, offs)
    car = [tabs[100], tabs[900]]
    streams = [synth.synth_iq(seed, 0, n_blocks * L, 0.25, car, [0.3, 0.05], ch.N, ch.fs // 1000) for seed in seeds]
    # page-locked blocks (the device layer's double buffer, airspy_dev.cpp:410-414) and device rows per bank
    hosts = [np.ascontiguousarray(s_).copy() for s_ in streams]
    for h in hosts:
        sb.host_register(h)
    banks = [sb.ChannelBank(tabs, ch.stage_tuples(), device=0) for _ in range(n_streams)]
    outs = [(torch.zeros(n_blocks, 1024, 512, 2, device="cuda"), torch.zeros(n_blocks, 1024, 512, device="cuda")) for _ in range(n_streams)]
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(n_streams)

    def drive(i):
        try:
            start.wait()
            for k in range(n_blocks):
                n = banks[i].process_pinned(hosts[i][k * L:].ctypes.data, L, outs[i][0][k].data_ptr(), outs[i][1][k].data_ptr(), 512)
                assert n == 512
            banks[i].synchronize()
        except Exception as e:      # noqa: BLE001 (reported by the main thread)
            errors.append((i, repr(e)))

    try:
        threads = [threading.Thread(target=drive, args=(i,)) for i in range(n_streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        torch.cuda.synchronize()
        sample = [0, 1, 511, 512, 513, 777, 1022, 1023]               # incl. offset 0 (no translator) and both ends
        ref_bank = [tabs[c] for c in sample]
        worst = [0.0, 0.0]
        for i in range(n_streams):
            iq = np.concatenate([outs[i][0][k].cpu().numpy().view(np.complex64)[..., 0] for k in range(n_blocks)], axis=1)
            au = np.concatenate([outs[i][1][k].cpu().numpy() for k in range(n_blocks)], axis=1)
            assert iq.shape == (1024, n_blocks * 512) and np.isfinite(au).all()
            r_iq, r_au = oracle.bank(ref_bank, ch.stage_tuples()).process(streams[i], threads=8)     # ITS stream
            e_iq, e_au = rel_err_rows(iq[sample], r_iq), rel_err_rows(au[sample], r_au)
            worst = [max(worst[0], e_iq), max(worst[1], e_au)]
            assert e_iq <= RTOL and e_au <= RTOL, (i, e_iq, e_au)
            if i in (0, 9):     # and the whole bank, bit for bit, against the same stream through a bank that is alone
                solo = sb.ChannelBank(tabs, ch.stage_tuples(), device=0)
                s_iq, s_au = solo.process(streams[i])
                solo.close()
                assert np.array_equal(_bits(iq), _bits(s_iq)) and np.array_equal(_bits(au), _bits(s_au))
        margin("config4_16_streams_vs_oracle", iq=worst[0], audio=worst[1])
        # streams differ: no bank got another bank's block
        a0 = outs[0][1][0].cpu().numpy()
        assert all(not np.array_equal(a0, outs[i][1][0].cpu().numpy()) for i in range(1, n_streams))
    finally:
        for b_ in banks:
            b_.close()
        for h in hosts:
            sb.host_unregister(h)
