"""The drop-in entry's channel parts (sdrxgpu_bank_process: host pointers, complete on return; reference call site
src/sdrx.cpp:327-346, outputs into the ring chunk, src/crb.hpp:110-111).  Round 6 cut the bank where the rounds of the stage-1/2
kernel end, gave every bank two cuts (calls that take the decimated IQ / audio-only calls) and made a +k / -k pair belong to the part
of its FIRST channel, so that a part's launch computes later parts' partner channels ahead of time.  What must hold whatever the
channel list and wherever the cuts fall: the bits of the device entry (one launch over the whole bank), for both kinds of call, with
ragged calls, and the state afterwards."""
import numpy as np
import pytest
import torch

from conftest import RTOL, rel_err_rows
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _device_entry(tabs, ch, calls, **kw):
    """the same stream through sdrxgpu_bank_process_device: no parts at all"""
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
    n_ch = len(tabs)
    iq, au = [], []
    for c in calls:
        n_max = len(c) // ch.m_total + 1
        d_x = torch.from_numpy(np.ascontiguousarray(c).view(np.float32).reshape(-1, 2).copy()).cuda()
        d_iq = torch.zeros(n_ch, n_max, 2, device="cuda")
        d_au = torch.zeros(n_ch, n_max, device="cuda")
        torch.cuda.synchronize()
        n = bank.process_device(d_x.data_ptr(), len(c), d_iq.data_ptr(), d_au.data_ptr(), n_max)
        bank.synchronize()
        iq.append(d_iq.cpu().numpy().view(np.complex64)[..., 0][:, :n].copy())
        au.append(d_au.cpu().numpy()[:, :n].copy())
    state = bank.agc_state()
    fam = bank.info().matrix_pipe
    bank.close()
    return np.concatenate(iq, axis=1), np.concatenate(au, axis=1), state, fam


def _lists(ch, rng):
    """channel lists that pair differently: whole +k / -k cycles, a shuffled list (partners far apart, in other parts), a list
    with every third channel's partner missing, one with channels at the tuner centre (no table) sprinkled in"""
    n = 4224
    cyc = chains.offsets_cycled(n, ch.N)
    shuf = list(rng.permutation(cyc))
    holes = [o for i, o in enumerate(chains.offsets_cycled(n + 1500, ch.N)) if not (o > 0 and i % 3 == 0)][:n]
    centre = [0 if i % 41 == 7 else o for i, o in enumerate(cyc)]
    return {"cycled": cyc, "shuffled": [int(o) for o in shuf], "partners_missing": holes, "with_tuner_centre": centre}


@pytest.mark.parametrize("which", ["cycled", "shuffled", "partners_missing", "with_tuner_centre"])
def test_any_cut_of_any_list_gives_the_device_entrys_bits(which):
    ch = chains.chain("10")
    rng = np.random.default_rng(606)
    offs = _lists(ch, rng)[which]
    n_ch = len(offs)
    tabs = sb.translators_for(ch, offs)
    L, M = ch.block_len, ch.m_total
    x = synth.synth_iq(61, 0, 2 * L + 5 * M + 3, 0.25, [t for t in tabs[:60:17] if t is not None][:2], [0.3, 0.1], ch.N, ch.fs // 1000)
    cuts = (0, L, L + 9 * M - 1, L + 9 * M + 1, len(x))
    calls = [x[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    w_iq, w_au, w_state, fam = _device_entry(tabs, ch, calls)
    variants = [None]                                              # the automatic cuts
    for _ in range(3):                                             # ... and random ones: 2 to 6 parts, boundaries anywhere
        k = int(rng.integers(1, 6))
        b = sorted(set(int(v) // 64 * 64 for v in rng.integers(64, n_ch - 64, size=k)))
        variants.append({"host_parts": len(b) + 1, "host_part_ch": tuple(b)})
    variants.append({"host_parts": 3})                             # equal parts (round 5's rule)
    for tun in variants:
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning=tun)
        assert bank.info().matrix_pipe == fam
        for audio_only in (False, True):
            parts = bank.host_parts(audio_only)
            assert parts[0] == 0 and parts[-1] == n_ch and all(a < b for a, b in zip(parts[:-1], parts[1:]))
            assert all(p % 64 == 0 for p in parts[1:-1])
        if tun and "host_part_ch" in tun:
            assert bank.host_parts() == [0] + list(tun["host_part_ch"]) + [n_ch]
        # calls alternate between taking the IQ and taking the audio only: both cuts of the bank work on ONE stream and state
        got_iq, got_au = [], []
        for i, c in enumerate(calls):
            if i % 2 == 0:
                iq, au = bank.process(c)
                got_iq.append(iq)
            else:
                _, au = bank.process(c, want_iq=False)
                got_iq.append(None)
            got_au.append(au)
        au = np.concatenate(got_au, axis=1)
        assert np.array_equal(_bits(au), _bits(w_au)), (which, tun)
        col = 0
        for i, c in enumerate(calls):
            n = got_au[i].shape[1]
            if got_iq[i] is not None:
                assert np.array_equal(_bits(got_iq[i]), _bits(w_iq[:, col:col + n])), (which, tun, i)
            col += n
        for a, b in zip(bank.agc_state(), w_state):
            assert np.array_equal(_bits(a), _bits(b))
        bank.close()


def test_automatic_cuts_follow_the_rounds_and_small_banks_stay_whole(oracle):
    ch = chains.chain("10")
    # under 2 048 channels, or with squelch / audio FIR: one part
    small = sb.ChannelBank(sb.translators_for(ch, chains.offsets_cycled(1500, ch.N)), ch.stage_tuples())
    assert small.host_parts() == [0, 1500] and small.host_parts(True) == [0, 1500]
    small.close()
    n_ch = 10000
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    assert bank.info().matrix_pipe == 1
    with_iq, audio_only = bank.host_parts(False), bank.host_parts(True)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    if n_cu == 256:      # 512 workgroup places / 9 time-workgroups per set = 56 sets of 32 slots per round; 163 sets = 3 rounds
        assert len(with_iq) - 1 == 3 and len(audio_only) - 1 == 2
        # calls with IQ: three parts equal in sets (54 / 55 each, no part longer than a round); audio only: two whole rounds | the rest
        assert audio_only[1] >= with_iq[2] and abs((with_iq[2] - with_iq[1]) - (with_iq[1] - with_iq[0])) <= 1024
    assert len(audio_only) <= len(with_iq)
    # one block through both kinds of call against the oracle on a few channels of every part
    x = synth.synth_iq(9, 0, ch.block_len, 0.25, [tabs[650]], [0.2], ch.N, 10000)
    iq, au = bank.process(x)
    bank.reset()
    _, au2 = bank.process(x, want_iq=False)
    bank.close()
    assert np.array_equal(_bits(au), _bits(au2))
    pick = sorted({p for b in with_iq[:-1] for p in (b, b + 1)} | {b - 1 for b in with_iq[1:]} | {650, 1850, 9999})
    ref = oracle.bank([tabs[c] for c in pick], ch.stage_tuples())
    r_iq, r_au = ref.process(x, threads=8)
    assert rel_err_rows(iq[pick], r_iq) <= RTOL and rel_err_rows(au[pick], r_au) <= RTOL


def test_vector_kernel_banks_and_the_fused_stage3_family_take_the_same_cuts():
    """SDRXGPU_FLAG_REFERENCE_ORDER (vector kernels: pairs looked for inside each part) and SDRXGPU_FLAG_FUSED_STAGE3 (k_stage123m:
    the matrix family's slot list) through explicit cuts: the device entry's bits for each."""
    ch = chains.chain("10")
    n_ch = 2624
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    x = synth.synth_iq(5, 0, 2 * L, 0.25, [tabs[30]], [0.3], ch.N, ch.fs // 1000)
    calls = [x[:L], x[L:]]
    for kw in (dict(reference_order=True), dict(fused_stage3=True)):
        w_iq, w_au, w_state, fam = _device_entry(tabs, ch, calls, **kw)
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"host_parts": 3, "host_part_ch": (640, 1984)}, **kw)
        assert bank.info().matrix_pipe == fam
        outs = [bank.process(c) for c in calls]
        assert np.array_equal(_bits(np.concatenate([o[0] for o in outs], axis=1)), _bits(w_iq))
        assert np.array_equal(_bits(np.concatenate([o[1] for o in outs], axis=1)), _bits(w_au))
        bank.close()
