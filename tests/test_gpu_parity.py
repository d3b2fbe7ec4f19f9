"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the golden vectors produced by the compiled reference and against the oracle on
identical seeded input.  Tolerance: 1e-5 relative under the comparator of conftest.rel_err
(north_star: "within 1e-5 relative float tolerance")."""
import os

import numpy as np
import pytest

from conftest import RTOL, load_golden, margin, regen_blocks, rel_err, rel_err_rows
from sdrx_amd import bank as sb
from sdrx_amd import chains, synth

pytestmark = pytest.mark.gpu


def _tables(g, ch):
    return [g["tables"][i] if int(o) != 0 else None for i, o in enumerate(g["offsets"])]


def _run_golden(name, lf_agc=False, **bank_kw):
    g = load_golden(name)
    ch, blocks = regen_blocks(g)
    bank = sb.ChannelBank(_tables(g, ch), ch.stage_tuples(), lf_agc=lf_agc, **bank_kw)
    iq, au, lens = [], [], []
    for x in blocks:
        a, b = bank.process(x)
        iq.append(a)
        au.append(b)
        lens.append(a.shape[1])
    bank.close()
    iq = np.concatenate(iq, axis=1)
    au = np.concatenate(au, axis=1)
    assert lens == [int(v) for v in g["out_lens"][0]], (lens, g["out_lens"][0])
    worst_iq = rel_err_rows(iq, g["iq"])
    worst_au = rel_err_rows(au, g["audio_lf"])
    return worst_iq, worst_au, g


@pytest.mark.parametrize("name", ["g2_fs10000", "g3_fs10000_noise", "g4_fs10000_impulse", "g4_fs10000_dc",
                                  "g4_fs10000_tone", "g1_fs2400_c1"])
def test_golden_vectors(name):
    worst_iq, worst_au, _ = _run_golden(name)
    margin(f"golden[{name}]", iq=worst_iq, audio=worst_au)
    assert worst_iq <= RTOL
    assert worst_au <= RTOL


def test_golden_lf_agc_active():
    worst_iq, worst_au, _ = _run_golden("g1_fs2400_c1_lfagc", lf_agc=True)
    assert worst_iq <= RTOL and worst_au <= RTOL


def test_golden_audio_fir_config1():
    """Config 1 end to end: 2.4 MS/s -> 16 kS/s audio incl. the 197-tap audio band-pass (fir.hpp:79-127)."""
    for name, lf in (("g1_fs2400_c1", False), ("g1_fs2400_c1_lfagc", True)):
        g = load_golden(name)
        ch, blocks = regen_blocks(g)
        bank = sb.ChannelBank(_tables(g, ch), ch.stage_tuples(), lf_agc=lf, audio_fir=chains.audio_bpf_taps(), audio_fir_gain_db=0.0)
        au = np.concatenate([bank.process(x)[1] for x in blocks], axis=1)
        bank.close()
        e = rel_err_rows(au, g["audio_fir"])
        print(f"{name}: FIR-filtered audio rel {e:.3e}")
        assert e <= RTOL


def test_golden_ftfir():
    """--ftfir (msd.hpp:104-131, 192-224, 341-423) against the reference's own ftfir output."""
    worst_iq, worst_au, _ = _run_golden("g5_fs10000_ftfir", ftfir=True)
    print(f"ftfir: iq rel {worst_iq:.3e} audio rel {worst_au:.3e}")
    assert worst_iq <= RTOL and worst_au <= RTOL


@pytest.mark.parametrize("name", ["g7_fs0p96", "g7_fs1p2", "g7_fs1p44", "g7_fs1p6", "g7_fs1p92", "g7_fs6", "g7_fs2p56"])
def test_golden_other_sample_rate_chains(name):
    """The 3-stage chains of sdrx.cpp:1588-1628 (z != 1, lane spacing = k*N), incl. the tuner-centre channel."""
    worst_iq, worst_au, _ = _run_golden(name)
    print(f"{name}: iq rel {worst_iq:.3e} audio rel {worst_au:.3e}")
    assert worst_iq <= RTOL and worst_au <= RTOL


def test_ftfir_on_2p4_chain_vs_oracle(oracle):
    ch = chains.chain("2.4")
    offs = [37, 0, -5]
    tabs = sb.translators_for(ch, offs)
    blocks = [synth.synth_iq(31, b * ch.block_len, ch.block_len, 0.25) for b in range(2)]
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), ftfir=True)
    got = np.concatenate([bank.process(x)[0] for x in blocks], axis=1)
    bank.close()
    ref = []
    for t in tabs:
        m = oracle.msd(t, ch.stage_tuples(), True)
        ref.append(np.concatenate([m.decimate(x) for x in blocks]))
    assert rel_err_rows(got, np.stack(ref)) <= RTOL


def test_golden_ragged_call_lengths():
    """decimate accepts any in_len and reports out_len (msd.hpp:73-75,158): 0, 1, 624, 625 ... samples."""
    worst_iq, worst_au, g = _run_golden("g6_fs10000_ragged")
    assert worst_iq <= RTOL and worst_au <= RTOL


@pytest.mark.parametrize("hint", [1, 2, 3, 4, 6, 12])
def test_every_time_tiling_gives_the_same_answer(hint):
    """The stage-1/2 kernel's bodies-per-lane tiling must not show in the result (bit-identical)."""
    g = load_golden("g2_fs10000")
    ch, blocks = regen_blocks(g)
    ref_bank = sb.ChannelBank(_tables(g, ch), ch.stage_tuples(), block_hint=12)
    bank = sb.ChannelBank(_tables(g, ch), ch.stage_tuples(), block_hint=hint)
    for x in blocks[:2]:
        a, b = ref_bank.process(x)
        c, d = bank.process(x)
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
        assert np.array_equal(b.view(np.uint32), d.view(np.uint32))
    ref_bank.close()
    bank.close()


@pytest.mark.parametrize("w,cpw", [(16, 4), (16, 2), (16, 1), (8, 4), (8, 1), (4, 4), (4, 2), (4, 1)])
def test_every_workgroup_geometry_gives_the_same_answer(w, cpw):
    """Waves per workgroup / channels per wave of the stage-1/2 kernel must not show in the result."""
    ch = chains.chain("10")
    offs = list(range(-40, 37))            # 77 channels: not a multiple of any W*CPW
    tabs = sb.translators_for(ch, offs)
    blocks = [synth.synth_iq(8, b * ch.block_len, ch.block_len, 0.25) for b in range(2)]
    ref_bank = sb.ChannelBank(tabs, ch.stage_tuples())
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), block_hint=4, tuning={"k12_waves": w, "k12_cpw": cpw})
    for x in blocks:
        a, b = ref_bank.process(x)
        c, d = bank.process(x)
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
        assert np.array_equal(b.view(np.uint32), d.view(np.uint32))
    ref_bank.close()
    bank.close()


def _oracle_bank(oracle, ch, tabs, blocks, threads=8):
    ob = oracle.bank(tabs, ch.stage_tuples())
    outs = [ob.process(x, threads=threads) for x in blocks]
    return np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1)


def test_config2_64_channels_vs_oracle(oracle):
    """BASELINE config 2: 64 channels, 10 MS/s, 8.33 kHz grid, 3 blocks."""
    ch = chains.chain("10")
    offs = chains.offsets_for_config("c2")
    tabs = sb.translators_for(ch, offs)
    car = [tabs[offs.index(5)], tabs[offs.index(-20)]]
    blocks = [synth.synth_iq(1234, b * ch.block_len, ch.block_len, 0.25, car, [0.3, 0.1], ch.N, 10000) for b in range(3)]
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    got = [bank.process(x) for x in blocks]
    iq = np.concatenate([g[0] for g in got], axis=1)
    au = np.concatenate([g[1] for g in got], axis=1)
    assert iq.shape == (64, 1536)
    ref_iq, ref_au = _oracle_bank(oracle, ch, tabs, blocks)
    e1, e2 = rel_err_rows(iq, ref_iq), rel_err_rows(au, ref_au)
    print(f"c2: iq rel {e1:.3e} audio rel {e2:.3e}")
    assert e1 <= RTOL and e2 <= RTOL
    # AGC state readable like sdrx -v prints it
    g_agc, g_lf = bank.gains()
    assert g_agc.shape == (64,) and np.all(g_agc >= 1 / 300) and np.all(g_agc <= 300)
    bank.close()


def test_config3_1024_channels_vs_oracle(oracle):
    """BASELINE config 3: 1024 channels incl. the tuner centre (no-translator path) and 25 kHz names, 2 blocks."""
    ch = chains.chain("10")
    offs = chains.offsets_for_config("c3")
    tabs = sb.translators_for(ch, offs)
    blocks = [synth.synth_iq(42, b * ch.block_len, ch.block_len, 0.25) for b in range(2)]
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    got = [bank.process(x) for x in blocks]
    iq = np.concatenate([g[0] for g in got], axis=1)
    au = np.concatenate([g[1] for g in got], axis=1)
    ref_iq, ref_au = _oracle_bank(oracle, ch, tabs, blocks, threads=16)
    e1, e2 = rel_err_rows(iq, ref_iq), rel_err_rows(au, ref_au)
    print(f"c3: iq rel {e1:.3e} audio rel {e2:.3e}")
    assert e1 <= RTOL and e2 <= RTOL
    bank.close()


def test_full_size_10k_channels_properties_and_sampled_oracle(oracle):
    """North-star size: 10 000 channel instances from one 10 MS/s stream.  Size-independent properties +
    a sample of channels against the oracle."""
    ch = chains.chain("10")
    n_ch = 10000
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    blocks = [synth.synth_iq(7, b * ch.block_len, ch.block_len, 0.25) for b in range(2)]
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    got = [bank.process(x) for x in blocks]
    iq = np.concatenate([g[0] for g in got], axis=1)
    au = np.concatenate([g[1] for g in got], axis=1)
    assert iq.shape == (n_ch, 1024) and np.all(np.isfinite(iq.view(np.float32))) and np.all(np.isfinite(au))
    # (1) instances with the same offset are independent work but must agree bit for bit
    for c in (0, 1, 599, 1199):
        for rep in range(1, n_ch // ch.N):
            assert np.array_equal(iq[c].view(np.uint32), iq[c + rep * ch.N].view(np.uint32))
            assert np.array_equal(au[c].view(np.uint32), au[c + rep * ch.N].view(np.uint32))
    # (2) sampled channels against the oracle (incl. the ends of the offset range and the tuner centre)
    #     -- all 1 200 distinct offsets (channels 0 .. 1 199) and later instances spread over the bank
    pick = sorted(set(list(range(ch.N)) + [ch.N, 4321, 9999] + list(range(37, n_ch, 997))))
    ref_iq, ref_au = _oracle_bank(oracle, ch, [tabs[c] for c in pick], blocks, threads=os.cpu_count() or 8)
    e1, e2 = rel_err_rows(iq[pick], ref_iq), rel_err_rows(au[pick], ref_au)
    print(f"10k: iq rel {e1:.3e} audio rel {e2:.3e} over {len(pick)} sampled channels")
    assert e1 <= RTOL and e2 <= RTOL
    # (3) linearity of the FIR chain: scaling the input by 2 scales the decimated IQ by exactly 2
    bank.reset()
    iq2, _ = bank.process(blocks[0] * np.float32(2.0), want_audio=False)
    assert np.array_equal((iq[:, :512] * np.float32(2.0)).view(np.uint32), iq2.view(np.uint32))
    # (4) a real-valued input makes mirror channels (+k / -k steps: exactly conjugate tables) exact conjugates of each
    #     other -- whichever lanes, workgroups and kernels the two happen to run in -- with identical audio
    first = {}
    for c, o in enumerate(offs):
        first.setdefault(int(o), c)
    pairs = [(first[o], first[-o]) for o in range(1, ch.N // 2) if o in first and -o in first]
    assert len(pairs) == ch.N // 2 - 1
    cp, cm = np.array([p for p, _ in pairs]), np.array([m for _, m in pairs])
    conj_tables = all(np.array_equal(tabs[p].view(np.uint32), np.conj(tabs[m]).view(np.uint32)) for p, m in pairs[:50])
    print("10k: mirror-channel property", "checked" if conj_tables else "SKIPPED (libm's cexpf is not exactly odd/even on this box)")
    if conj_tables:
        bank.reset()
        xr = blocks[0].copy()
        xr.imag = 0
        iqr, aur = bank.process(xr)
        assert np.array_equal(iqr[cp].view(np.uint32), np.conj(iqr[cm]).view(np.uint32))
        assert np.array_equal(aur[cp].view(np.uint32), aur[cm].view(np.uint32))
        # ... and so are their later instances (other workgroups, other host parts)
        far = [(p + 7 * ch.N, m + 3 * ch.N) for p, m in pairs[::37]]
        assert all(np.array_equal(iqr[p].view(np.uint32), np.conj(iqr[m]).view(np.uint32)) for p, m in far)
    bank.close()


def test_block_split_invariance_and_reset():
    """Cutting the stream at arbitrary points must not change a single output (state continuity)."""
    ch = chains.chain("10")
    offs = [0, 3, -77, 415, -600, 599]
    tabs = sb.translators_for(ch, offs)
    x = synth.synth_iq(5, 0, 2 * ch.block_len, 0.25)
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    a1, b1 = bank.process(x[: ch.block_len])
    a2, b2 = bank.process(x[ch.block_len:])
    one_iq, one_au = np.concatenate([a1, a2], axis=1), np.concatenate([b1, b2], axis=1)
    bank.reset()
    parts_iq, parts_au, p = [], [], 0
    for n in (0, 1, 624, 625, 626, 24, 99999, 1, 0, 320000 - 1, 77777, 10 ** 9):
        n = min(n, len(x) - p)
        a, b = bank.process(x[p:p + n])
        parts_iq.append(a)
        parts_au.append(b)
        p += n
    assert p == len(x)
    got_iq, got_au = np.concatenate(parts_iq, axis=1), np.concatenate(parts_au, axis=1)
    assert got_iq.shape == one_iq.shape
    assert np.array_equal(got_iq.view(np.uint32), one_iq.view(np.uint32))
    assert np.array_equal(got_au.view(np.uint32), one_au.view(np.uint32))
    bank.close()


def test_long_call_is_split_internally(oracle):
    """in_len larger than one 32 ms block: still one call, same answer."""
    ch = chains.chain("10")
    offs = [11, -250]
    tabs = sb.translators_for(ch, offs)
    x = synth.synth_iq(9, 0, 3 * ch.block_len + 4321, 0.25)
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    iq, au = bank.process(x)
    ref_iq, ref_au = _oracle_bank(oracle, ch, tabs, [x], threads=2)
    assert iq.shape == ref_iq.shape
    assert rel_err_rows(iq, ref_iq) <= RTOL and rel_err_rows(au, ref_au) <= RTOL
    bank.close()


def test_conjugate_symmetry():
    """Channel +off of x and channel -off of conj(x) are complex conjugates (tables are exact conjugates)."""
    ch = chains.chain("10")
    x = synth.synth_iq(21, 0, ch.block_len, 0.25)
    tp = sb.make_translator(ch.N, ch.z, 123)
    tm = sb.make_translator(ch.N, ch.z, -123)
    if not np.array_equal(tp.view(np.uint32), np.conj(tm).view(np.uint32)):
        pytest.skip("libm's cexpf is not exactly odd/even on this box")
    b1 = sb.ChannelBank([tp], ch.stage_tuples())
    b2 = sb.ChannelBank([tm], ch.stage_tuples())
    a, _ = b1.process(x)
    b, _ = b2.process(np.conj(x))
    assert rel_err(a[0], np.conj(b[0])) <= 2e-6
    b1.close()
    b2.close()


def test_device_pointer_entry_and_ingest_buffer():
    """process_device with torch device tensors, incl. zero-copy ingest into the bank's own buffer."""
    import torch
    ch = chains.chain("10")
    offs = [7, -7, 100]
    tabs = sb.translators_for(ch, offs)
    x = synth.synth_iq(3, 0, ch.block_len, 0.25)
    host = sb.ChannelBank(tabs, ch.stage_tuples())
    ref_iq, ref_au = host.process(x)
    host.close()
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    d_x = torch.from_numpy(x.view(np.float32).copy()).cuda()
    d_iq = torch.zeros(3, 512, 2, device="cuda")
    d_au = torch.zeros(3, 512, device="cuda")
    torch.cuda.synchronize()
    n = bank.process_device(d_x.data_ptr(), ch.block_len, d_iq.data_ptr(), d_au.data_ptr(), 512)
    bank.synchronize()
    assert n == 512
    assert np.array_equal(d_iq.cpu().numpy().reshape(3, 1024).view(np.complex64).view(np.uint32), ref_iq.view(np.uint32))
    assert np.array_equal(d_au.cpu().numpy().view(np.uint32), ref_au.view(np.uint32))
    # zero-copy: write the block straight into the bank's ingest buffer (as an H2D copy or an RCCL
    # broadcast would), wrapped as a torch tensor through __cuda_array_interface__
    bank.reset()
    ptr, cap = bank.ingest_buffer()
    assert cap >= ch.block_len
    view = torch.as_tensor(sb.DeviceView(ptr, (ch.block_len, 2)), device="cuda")
    assert view.data_ptr() == ptr
    view.copy_(d_x.view(ch.block_len, 2))
    d_iq.zero_()
    torch.cuda.synchronize()
    n = bank.process_device(ptr, ch.block_len, d_iq.data_ptr(), None, 512)
    bank.synchronize()
    assert n == 512
    assert np.array_equal(d_iq.cpu().numpy().reshape(3, 1024).view(np.complex64).view(np.uint32), ref_iq.view(np.uint32))
    bank.close()


def test_pipelined_mode_is_bit_identical_one_call_late():
    """SDRXGPU_FLAG_PIPELINED: call k returns block k-1, flush returns the last one; same bits as the serial order."""
    import torch
    ch = chains.chain("10")
    offs = list(range(-20, 21))
    tabs = sb.translators_for(ch, offs)
    n_ch = len(offs)
    blocks = [synth.synth_iq(17, b * ch.block_len, ch.block_len, 0.25) for b in range(4)]
    serial = sb.ChannelBank(tabs, ch.stage_tuples())
    want = [serial.process(x) for x in blocks]
    serial.close()
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), pipelined=True)
    d_iq = torch.zeros(n_ch, 512, 2, device="cuda")
    d_au = torch.zeros(n_ch, 512, device="cuda")
    got = []
    for k, x in enumerate(blocks):
        d_x = torch.from_numpy(x.view(np.float32).copy()).cuda()
        torch.cuda.synchronize()
        n = bank.process_device(d_x.data_ptr(), ch.block_len, d_iq.data_ptr(), d_au.data_ptr(), 512)
        bank.synchronize()
        assert n == (0 if k == 0 else 512)
        if k:
            got.append((d_iq.cpu().numpy().reshape(n_ch, 1024).view(np.complex64).copy(), d_au.cpu().numpy().copy()))
    assert bank.flush(d_iq.data_ptr(), d_au.data_ptr(), 512) == 512
    bank.synchronize()
    got.append((d_iq.cpu().numpy().reshape(n_ch, 1024).view(np.complex64).copy(), d_au.cpu().numpy().copy()))
    assert bank.flush(d_iq.data_ptr(), d_au.data_ptr(), 512) == 0
    for (a, b), (c, d) in zip(want, got):
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
        assert np.array_equal(b.view(np.uint32), d.view(np.uint32))
    # host-pointer entry is refused on a pipelined bank (it promises outputs on return)
    with pytest.raises(sb.SdrxGpuError) as e:
        bank.process(blocks[0])
    assert e.value.code == -5
    bank.close()


@pytest.mark.parametrize("n_ch,zero_copy", [(1024, False), (1024, True), (40, True)])
def test_pipelined_mode_back_to_back_without_waiting(n_ch, zero_copy):
    """The two chains of a pipelined bank run free of each other (round 5): six blocks enqueued back to back, nothing waits
    until the end.  Distinct input buffers, or (zero_copy) the bank's own two input rings written on the bank's stream right before
    each call -- the ring call k reads is in use until call k+1 has been enqueued, so block k+1 goes into the other one.  Same bits
    as the serial order."""
    import torch
    ch = chains.chain("10")
    offs = chains.offsets_for_config("c3") if n_ch == 1024 else list(range(-20, 20))
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = [synth.synth_iq(23, b * L, L, 0.25, [tabs[3]], [0.3], ch.N, 10000) for b in range(6)]
    serial = sb.ChannelBank(tabs, ch.stage_tuples())
    want = [serial.process(x) for x in blocks]
    fam = serial.info().matrix_pipe
    serial.close()
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), pipelined=True, stream=ts.cuda_stream)
        assert bank.info().matrix_pipe == fam
        d_x = [torch.from_numpy(x.view(np.float32).reshape(L, 2).copy()).cuda() for x in blocks]
        d_iq = [torch.zeros(n_ch, 512, 2, device="cuda") for _ in range(7)]
        d_au = [torch.zeros(n_ch, 512, device="cuda") for _ in range(7)]
        torch.cuda.synchronize()
        for k in range(6):
            src = d_x[k].data_ptr()
            if zero_copy:
                p, cap = bank.ingest_buffer()
                torch.as_tensor(sb.DeviceView(p, (L, 2)), device="cuda").copy_(d_x[k], non_blocking=True)    # on the bank's stream
                src = p
            n = bank.process_device(src, L, d_iq[k].data_ptr(), d_au[k].data_ptr(), 512)
            assert n == (0 if k == 0 else 512)
        assert bank.flush(d_iq[6].data_ptr(), d_au[6].data_ptr(), 512) == 512
        bank.synchronize()
        torch.cuda.synchronize()
    for k in range(6):
        got_iq = d_iq[k + 1].cpu().numpy().view(np.complex64)[..., 0]
        assert np.array_equal(got_iq.view(np.uint32), want[k][0].view(np.uint32)), k
        assert np.array_equal(d_au[k + 1].cpu().numpy().view(np.uint32), want[k][1].view(np.uint32)), k
    bank.close()


def test_pipelined_mode_one_input_buffer_rewritten_in_stream_order():
    """A caller's own input buffer is the caller's again in the bank's stream order as soon as the call has been enqueued (ADVICE r5:
    round 5 had moved the input copy onto the second chain without that guarantee): ONE device buffer, refilled on the bank's
    stream right after every call -- and scribbled over in between -- gives the serial order's bits."""
    import torch
    ch = chains.chain("10")
    offs = chains.offsets_for_config("c3")
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = [synth.synth_iq(29, b * L, L, 0.25, [tabs[7]], [0.3], ch.N, 10000) for b in range(5)]
    serial = sb.ChannelBank(tabs, ch.stage_tuples())
    want = [serial.process(x) for x in blocks]
    serial.close()
    ts = torch.cuda.Stream()
    with torch.cuda.stream(ts):
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), pipelined=True, stream=ts.cuda_stream)
        staged = [torch.from_numpy(x.view(np.float32).reshape(L, 2).copy()).cuda() for x in blocks]
        junk = torch.full((L, 2), 1e30, device="cuda")
        d_x = torch.empty(L, 2, device="cuda")
        d_iq = [torch.zeros(len(offs), 512, 2, device="cuda") for _ in range(6)]
        d_au = [torch.zeros(len(offs), 512, device="cuda") for _ in range(6)]
        torch.cuda.synchronize()
        for k in range(5):
            d_x.copy_(staged[k], non_blocking=True)                       # on the bank's stream, behind call k - 1
            n = bank.process_device(d_x.data_ptr(), L, d_iq[k].data_ptr(), d_au[k].data_ptr(), 512)
            assert n == (0 if k == 0 else 512)
            d_x.copy_(junk, non_blocking=True)                            # ... and the buffer is ours again at once
        assert bank.flush(d_iq[5].data_ptr(), d_au[5].data_ptr(), 512) == 512
        bank.synchronize()
        torch.cuda.synchronize()
    for k in range(5):
        got_iq = d_iq[k + 1].cpu().numpy().view(np.complex64)[..., 0]
        assert np.array_equal(got_iq.view(np.uint32), want[k][0].view(np.uint32)), k
        assert np.array_equal(d_au[k + 1].cpu().numpy().view(np.uint32), want[k][1].view(np.uint32)), k
    bank.close()


def test_two_banks_two_threads_concurrently(oracle):
    """BASELINE config 5's per-GPU share: independent IQ streams = independent banks, each driven from its own
    host thread on its own HIP stream, at the same time; no state outside the banks."""
    import threading
    ch = chains.chain("10")
    offs = list(range(-30, 30))
    tabs = sb.translators_for(ch, offs)
    streams = {sid: [synth.synth_iq(100 + sid, b * ch.block_len, ch.block_len, 0.25) for b in range(3)] for sid in (0, 1)}
    results, errors = {}, []

    def worker(sid):
        try:
            bank = sb.ChannelBank(tabs, ch.stage_tuples())
            out = [bank.process(x) for x in streams[sid]]
            results[sid] = (np.concatenate([o[0] for o in out], axis=1), np.concatenate([o[1] for o in out], axis=1))
            bank.close()
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=worker, args=(sid,)) for sid in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for sid in (0, 1):
        ref_iq, ref_au = _oracle_bank(oracle, ch, tabs, streams[sid], threads=8)
        assert rel_err_rows(results[sid][0], ref_iq) <= RTOL
        assert rel_err_rows(results[sid][1], ref_au) <= RTOL
    assert not np.array_equal(results[0][0], results[1][0])


def test_bank_without_any_translator(oracle):
    """Every channel at the tuner centre: no translator table at all (msd.hpp:79-101 for the whole bank)."""
    for rate in ("10", "1.2"):
        ch = chains.chain(rate)
        blocks = [synth.synth_iq(55, b * ch.block_len, ch.block_len, 0.25) for b in range(2)]
        bank = sb.ChannelBank([None, None, None], ch.stage_tuples())
        iq = np.concatenate([bank.process(x)[0] for x in blocks], axis=1)
        bank.close()
        m = oracle.msd(None, ch.stage_tuples())
        ref = np.concatenate([m.decimate(x) for x in blocks])
        for c in range(3):
            assert rel_err(iq[c], ref) <= RTOL


def test_single_channel_and_tiny_calls(oracle):
    """One channel, calls shorter than the decimation factor, a 3-stage chain with z != 1."""
    ch = chains.chain("1.44")
    t = sb.make_translator(ch.N, ch.z, -11)
    x = synth.synth_iq(77, 0, 3 * ch.m_total * 7 + 5, 0.3)
    bank = sb.ChannelBank([t], ch.stage_tuples())
    parts, p = [], 0
    for n in (1, 2, 3, 50, 89, 90, 91, 1, 400, 10 ** 9):
        n = min(n, len(x) - p)
        parts.append(bank.process(x[p:p + n])[0])
        p += n
    got = np.concatenate(parts, axis=1)[0]
    bank.close()
    ref = oracle.msd(t, ch.stage_tuples()).decimate(x)
    assert got.shape == ref.shape == (21,)
    assert rel_err(got, ref) <= RTOL


def test_lf_agc_and_custom_agc_parameters(oracle):
    """LfAGC active + non-default AGC parameters (agc.hpp setters)."""
    from oracle import ffi as offi
    ch = chains.chain("2.4")
    t = sb.make_translator(ch.N, ch.z, 37)
    blocks = [synth.synth_iq(9, b * ch.block_len, ch.block_len, 0.25, [t], [0.4], ch.N, 2400) for b in range(3)]
    agc, lf = sb.sdrx_agc_params()
    agc.attack, agc.decay, agc.reference, agc.max_gain, agc.alpha = 2.0, 0.05, 0.5, 50.0, 0.02
    lf.attack, lf.decay, lf.reference, lf.max_gain, lf.alpha = 0.5, 0.02, 0.7, 20.0, 0.01
    bank = sb.ChannelBank([t], ch.stage_tuples(), lf_agc=True, agc=agc, agc_lf=lf)
    au = np.concatenate([bank.process(x)[1] for x in blocks], axis=1)[0]
    bank.close()
    pair = oracle.agc_pair(True)
    for name, src in (("agc", agc), ("lf", lf)):
        dst = getattr(pair, name)
        dst.attack, dst.decay, dst.reference, dst.alpha = src.attack, src.decay, src.reference, src.alpha
        dst.max_gain, dst.min_gain = src.max_gain, 1.0 / src.max_gain
    m = oracle.msd(t, ch.stage_tuples())
    ref = np.concatenate([pair.adjust_lf(oracle.demod_am(pair.adjust(m.decimate(x)))) for x in blocks])
    assert rel_err(au, ref) <= RTOL


@pytest.mark.parametrize("rates", [(1.0, 0.01), (0.02, 0.5), (0.3, 0.3), (0.0, 0.2)])
def test_agc_rates_in_either_order(oracle, rates):
    """The kernels take the gain update `gain += (err > 0 ? decay : attack) * err` (agc.hpp:56-61) without the compare: as the smaller
    (decay <= attack) or the larger (decay > attack) of the two fmas.  Held against the oracle's compare-and-select for both orders,
    equal rates and a zero rate, IF and LF loop with opposite orders, a carrier that comes and goes (err of both signs), through
    k_agc and through k_tail_agc."""
    att, dec = rates
    ch = chains.chain("10")
    offs = [5, -5, 6, 0]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = [synth.synth_iq(31, b * L, L, 0.02, [tabs[0]], [0.0 if b == 1 else 0.6], ch.N, 10000) for b in range(3)]
    agc, lf = sb.sdrx_agc_params()
    agc.attack, agc.decay = att, dec
    lf.attack, lf.decay = dec, att
    for ta in (1, 2):
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), lf_agc=True, agc=agc, agc_lf=lf, tuning={"tail_agc": ta})
        assert (bank.info().tail_agc > 0) == (ta == 2)
        outs = [bank.process(x) for x in blocks]
        bank.close()
        iq = np.concatenate([o[0] for o in outs], axis=1)
        au = np.concatenate([o[1] for o in outs], axis=1)
        worst = 0.0
        for c in range(len(offs)):
            pair = oracle.agc_pair(True)
            pair.agc.attack, pair.agc.decay, pair.lf.attack, pair.lf.decay = att, dec, dec, att
            ref = pair.adjust_lf(oracle.demod_am(pair.adjust(iq[c])))      # (the oracle's loops on the GPU's own decimated IQ)
            worst = max(worst, rel_err(au[c], ref))
        margin(f"agc_rates[{att},{dec},tail_agc={ta}]", audio=worst)
        assert worst <= RTOL, (rates, ta, worst)


def test_squelch_and_gated_audio_vs_oracle(oracle):
    """Squelch (sdrx.cpp:533-573) + the audio gating by it (458-503) over a carrier that comes and goes: closed ->
    opening (ramp_up) -> open -> closing (ramp_down) -> closed, plus an always-open (level 0) and a never-opening channel.
    FFT parity is UNPINNED (FFTW absent): the oracle evaluates the same DFT directly in double precision."""
    from oracle import ffi as offi
    ch = chains.chain("10")
    offs = [37, -100, 5]
    levels = [9.0, 0.0, 9.0]
    tabs = sb.translators_for(ch, offs)
    pb, ru, rd = chains.squelch_tables()
    win = offi.hamming_window(oracle)
    assert np.array_equal(win.view(np.uint32), chains.hamming_window().view(np.uint32))
    on = [False, False, True, True, True, False, False, False]
    blocks = [synth.synth_iq(5, b * ch.block_len, ch.block_len, 0.25, [tabs[0]] if c else [], [0.2] if c else [], ch.N, 10000)
              for b, c in enumerate(on)]
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), squelch=(np.array(levels, np.float32), pb, ru, rd))
    ref_ch = [offi.SquelchChannel(oracle, lv, win, pb, ru, rd) for lv in levels]
    ref_msd = [oracle.msd(t, ch.stage_tuples()) for t in tabs]
    seen_states = set()
    w_iq = w_snr = w_au = 0.0
    for b, x in enumerate(blocks):
        iq, au = bank.process(x)
        snr, st = bank.squelch_state()
        for c in range(3):
            y = ref_msd[c].decimate(x)
            r_au, r_snr, r_st = ref_ch[c].chunk(y)
            assert rel_err(iq[c], y) <= RTOL
            # SNR in dB from single-precision power sums over ~90 / ~90 FFT bins (GPU: radix-2 fp32 FFT; oracle: the DFT
            # in double): 2e-3 dB = 4.6e-4 relative on a ratio of two sums whose terms each carry ~1e-6
            assert abs(float(snr[c]) - r_snr) < 2e-3, (b, c, snr[c], r_snr)
            assert int(st[c]) == r_st, (b, c)
            scale = max(float(np.sqrt(np.mean(r_au ** 2))), 1e-3)
            e_au = float(np.max(np.abs(au[c] - r_au))) / scale
            w_iq, w_snr, w_au = max(w_iq, rel_err(iq[c], y)), max(w_snr, abs(float(snr[c]) - r_snr)), max(w_au, e_au)
            assert e_au <= RTOL, (b, c, e_au)
        seen_states.add((int(st[0]), bool(np.any(au[0] != 0))))
    margin("squelch_and_gated_audio", iq=w_iq, snr_db=w_snr, audio=w_au)
    assert (1, True) in seen_states and (0, False) in seen_states and (0, True) in seen_states   # open, closed, ramp-down
    assert np.all(bank.squelch_state()[1][[1]] == 1)       # level 0: always open
    # whole chunks only
    with pytest.raises(sb.SdrxGpuError) as e:
        bank.process(blocks[0][:1000])
    assert e.value.code == -5
    bank.close()


def test_refused_call_leaves_the_bank_untouched():
    """A call the library refuses (out_stride too small; squelch bank fed less than a 512-output chunk) must change
    nothing: the calls after it give the same bits as an uninterrupted run.  MSD::decimate cannot fail (msd.hpp:75),
    so a refusal is this library's own and may not cost the stream its state."""
    import torch
    ch = chains.chain("10")
    offs = [0, 5, -7, 599, -600]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = [synth.synth_iq(41, b * L, L, 0.25, [tabs[1]], [0.3], ch.N, 10000) for b in range(3)]
    pb, ru, rd = chains.squelch_tables()
    for kw in ({}, {"squelch": (np.array([0.0, 6.0, 6.0, 0.0, 3.0], np.float32), pb, ru, rd)}):
        ref = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
        want = [ref.process(x) for x in blocks]
        ref.close()
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
        got = [bank.process(blocks[0])]
        # host-pointer entry: room for 100 outputs per channel, the call yields 512
        small_iq = np.zeros((len(offs), 100), np.complex64)
        small_au = np.zeros((len(offs), 100), np.float32)
        with pytest.raises(sb.SdrxGpuError) as e:
            bank.process(blocks[1], iq_out=small_iq, audio_out=small_au)
        assert e.value.code == -6
        # device-pointer entry, same refusal
        d_in = torch.from_numpy(blocks[1].view(np.float32).reshape(L, 2).copy()).cuda()
        d_iq = torch.zeros(len(offs), 100, 2, device="cuda")
        d_au = torch.zeros(len(offs), 100, device="cuda")
        with pytest.raises(sb.SdrxGpuError) as e:
            bank.process_device(d_in.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 100)
        assert e.value.code == -6
        if kw:
            with pytest.raises(sb.SdrxGpuError) as e:       # squelch: whole 512-output chunks only
                bank.process(blocks[1][: L // 2])
            assert e.value.code == -5
        info = bank.info()
        assert info.samples_in == L and info.samples_out == 512
        got.append(bank.process(blocks[1]))
        got.append(bank.process(blocks[2]))
        bank.close()
        for (a, b), (c, d) in zip(want, got):
            assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
            assert np.array_equal(b.view(np.uint32), d.view(np.uint32))


def test_error_behaviour():
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, [5])
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    x = synth.synth_iq(3, 0, ch.block_len, 0.25)
    with pytest.raises(sb.SdrxGpuError) as e:
        bank.process(x, out_stride=100)          # 512 outputs do not fit
    assert e.value.code == -6
    bank.close()
    # a chain the library was not built for is refused, not approximated
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank([None], [(7, np.ones(9, dtype=np.float32) / 9)] * 3)
    assert e.value.code == -5
    # translator tables of different lengths in one bank
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank([np.ones(1200, np.complex64), np.ones(288, np.complex64)], ch.stage_tuples())
    assert e.value.code == -5


def test_profile_counters():
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, list(range(1, 33)))
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    x = synth.synth_iq(3, 0, ch.block_len, 0.25)
    bank.profile_enable(True)
    for _ in range(3):
        bank.process(x)
    ms, n = bank.profile_read(sb.K_STAGE12)
    assert n == 3 and ms > 0
    ms_t, n_t = bank.profile_read(sb.K_TOTAL)
    assert n_t == 3 and ms_t >= ms
    bank.close()


def test_two_blocks_per_call_with_a_larger_max_in_len(oracle):
    """max_in_len = 2 blocks: one call produces 1024 outputs per channel (longer scratch rows, more time-waves)."""
    ch = chains.chain("10")
    offs = chains.offsets_cycled(40, ch.N)
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    x = synth.synth_iq(77, 0, 4 * L, 0.25, [tabs[3]], [0.2], ch.N, 9000)
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), max_in_len=2 * L)
    one = sb.ChannelBank(tabs, ch.stage_tuples())
    got, ref = [], []
    for k in range(2):
        iq, au = bank.process(x[2 * k * L:(2 * k + 2) * L])
        assert iq.shape == (40, 1024)
        got.append((iq, au))
        parts = [one.process(x[(2 * k + j) * L:(2 * k + j + 1) * L]) for j in range(2)]
        ref.append((np.concatenate([p[0] for p in parts], axis=1), np.concatenate([p[1] for p in parts], axis=1)))
    for (iq, au), (riq, rau) in zip(got, ref):
        assert np.array_equal(iq.view(np.uint32), riq.view(np.uint32))      # the split of the stream never changes a result
        assert np.array_equal(au.view(np.uint32), rau.view(np.uint32))
    y = oracle.msd(tabs[3], ch.stage_tuples()).decimate(x)
    assert rel_err(np.concatenate([g[0][3] for g in got]), y) <= RTOL
    bank.close()
    one.close()


def test_many_channels_beyond_the_benchmark_size(oracle):
    """25 000 channel instances in one bank (2.7 GB of stage-2 scratch): sampled against the oracle, and every
    instance of the same offset bit-identical."""
    ch = chains.chain("10")
    n = 25000
    offs = chains.offsets_cycled(n, ch.N)
    tabs_u = {o: t for o, t in zip(offs[:ch.N], sb.translators_for(ch, offs[:ch.N]))}
    tabs = [tabs_u[o] for o in offs]
    x = synth.synth_iq(5, 0, ch.block_len, 0.25, [tabs[700]], [0.2], ch.N, 9000)
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    iq, au = bank.process(x)
    bank.close()
    for c in (0, 700, 12345, n - 1):
        y = oracle.msd(tabs[c], ch.stage_tuples()).decimate(x)
        assert rel_err(iq[c], y) <= RTOL
    assert np.array_equal(iq[700].view(np.uint32), iq[700 + ch.N].view(np.uint32))
    assert np.array_equal(au[123].view(np.uint32), au[123 + 20 * ch.N].view(np.uint32))
