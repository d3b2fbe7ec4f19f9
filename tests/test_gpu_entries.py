"""The three entry points of the C ABI give the same bits: host pointers (sdrxgpu_bank_process, a large bank run as two
channel halves with the D2H of the first beside the kernels of the second), device pointers
(sdrxgpu_bank_process_device, blocks placed in the bank's two input rings) and page-locked host input with device
outputs (sdrxgpu_bank_process_pinned, nothing waits)."""
import threading

import numpy as np
import pytest
import torch

from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu


def _blocks(ch, n, seed=11, car=None):
    L = ch.block_len
    return [synth.synth_iq(seed, b * L, L, 0.25, car or [], [0.3] if car else [], ch.N, 10000) for b in range(n)]


def _device_run(bank, blocks, n_ch, in_place):
    L = len(blocks[0])
    d_iq = torch.zeros(n_ch, 512, 2, device="cuda")
    d_au = torch.zeros(n_ch, 512, device="cuda")
    outs = []
    for k, x in enumerate(blocks):
        xt = torch.from_numpy(x.view(np.float32).reshape(L, 2).copy()).cuda()
        if in_place:       # write the block where the bank expects it (its current input ring): no copy inside the call
            p, cap = bank.ingest_buffer()
            assert cap >= L
            dst = torch.as_tensor(sb.DeviceView(p, (L, 2)), device="cuda")
            dst.copy_(xt)
            src = p
        else:
            src = xt.data_ptr()
        torch.cuda.synchronize()
        assert bank.process_device(src, L, d_iq.data_ptr(), d_au.data_ptr(), 512) == 512
        bank.synchronize()
        outs.append((d_iq.cpu().numpy().view(np.complex64)[..., 0].copy(), d_au.cpu().numpy().copy()))
    return outs


@pytest.mark.parametrize("n_ch,fused,matrix", [(2304, sb.FUSED_AUTO, 2), (2304, sb.FUSED_OFF, 1), (4100, sb.FUSED_OFF, 2), (4100, sb.FUSED_REQUIRE, 0)])
def test_host_entry_in_two_halves_equals_device_entry(n_ch, fused, matrix):
    ch = chains.chain("10")
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    blocks = _blocks(ch, 3, car=[tabs[7]])
    # (the kernel path is pinned per case: the test ends with a bit-for-bit comparison of two cuts of the stream, which holds on every
    # path but the opt-in fused one, tests/test_gpu_fused3.py)
    # matrix: 2 = k_stage12m + k_tail_s, 1 = the vector kernels with the separate tail, 0 with fused_tail required = k_stage12p + k_tail4
    kw = dict(block_hint=0xFFFFFFFF, tuning={"fused_tail": fused, "matrix_pipe": matrix})
    dev = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
    want = _device_run(dev, blocks, n_ch, in_place=True)
    dev.close()
    host = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
    iq = np.zeros((n_ch, 512), np.complex64)
    au = np.zeros((n_ch, 512), np.float32)
    for k, x in enumerate(blocks):
        if k == 1:          # audio only: no IQ rows cross PCIe
            a, b = host.process(x, audio_out=au)
            assert a is None
        else:
            a, b = host.process(x, iq_out=iq, audio_out=au)
            assert np.array_equal(a.view(np.uint32), want[k][0].view(np.uint32))
        assert np.array_equal(b.view(np.uint32), want[k][1].view(np.uint32))
    info = host.info()
    assert info.samples_in == 3 * ch.block_len and info.samples_out == 3 * 512
    # ragged calls through the same entry still agree with one long call
    host.reset()
    x = np.concatenate(blocks[:2])
    parts = [host.process(x[a:b]) for a, b in ((0, 1000), (1000, 320625), (320625, 320625), (320625, len(x)))]
    got = np.concatenate([p[0] for p in parts], axis=1)
    assert np.array_equal(got[:, :512].view(np.uint32), want[0][0].view(np.uint32))
    assert np.array_equal(got[:, 512:].view(np.uint32), want[1][0].view(np.uint32))
    host.close()


def test_process_pinned_equals_device_entry_and_two_banks_overlap():
    ch = chains.chain("10")
    offs = chains.offsets_for_config("c3")
    tabs = sb.translators_for(ch, offs)
    n_ch, L = len(offs), ch.block_len
    streams = [_blocks(ch, 4, seed=21 + i) for i in range(2)]
    want = []
    for blocks in streams:
        bank = sb.ChannelBank(tabs, ch.stage_tuples())
        want.append(_device_run(bank, blocks, n_ch, in_place=False))
        bank.close()
    banks = [sb.ChannelBank(tabs, ch.stage_tuples()) for _ in streams]
    # the device layer's double buffer (airspy_dev.cpp:410-414): two page-locked blocks per stream, used in turn
    hosts = [[torch.empty(L, 2).pin_memory() for _ in range(2)] for _ in streams]
    outs = [[(torch.zeros(n_ch, 512, 2, device="cuda"), torch.zeros(n_ch, 512, device="cuda")) for _ in range(4)] for _ in streams]

    def drive(i):
        for k, x in enumerate(streams[i]):
            hosts[i][k & 1].copy_(torch.from_numpy(x.view(np.float32).reshape(L, 2)))
            n = banks[i].process_pinned(hosts[i][k & 1].data_ptr(), L, outs[i][k][0].data_ptr(), outs[i][k][1].data_ptr(), 512)
            assert n == 512
        banks[i].synchronize()

    th = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        for k in range(4):
            got_iq = outs[i][k][0].cpu().numpy().view(np.complex64)[..., 0]
            got_au = outs[i][k][1].cpu().numpy()
            assert np.array_equal(got_iq.view(np.uint32), want[i][k][0].view(np.uint32)), (i, k)
            assert np.array_equal(got_au.view(np.uint32), want[i][k][1].view(np.uint32)), (i, k)
        # mixing the entries on one bank stays right: one more block through the host entry
        a, b = banks[i].process(streams[i][0])
        assert a.shape == (n_ch, 512)
        banks[i].close()


def test_blocks_placed_in_both_input_rings_ahead_of_time():
    """sdrxgpu_bank_ingest_buffer / _after: block k+1 is written into the bank's other ring BEFORE block k is processed."""
    ch = chains.chain("10")
    offs = [0, 9, -250, 599]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    blocks = _blocks(ch, 4, seed=5)
    ref = sb.ChannelBank(tabs, ch.stage_tuples())
    want = [ref.process(x) for x in blocks]
    ref.close()
    bank = sb.ChannelBank(tabs, ch.stage_tuples())
    d_iq = torch.zeros(len(offs), 512, 2, device="cuda")
    d_au = torch.zeros(len(offs), 512, device="cuda")
    p_cur, _ = bank.ingest_buffer()
    torch.as_tensor(sb.DeviceView(p_cur, (L, 2)), device="cuda").copy_(torch.from_numpy(blocks[0].view(np.float32).reshape(L, 2).copy()))
    for k in range(4):
        if k + 1 < 4:
            p_next = bank.ingest_buffer_after(L)
            assert p_next != p_cur
            torch.as_tensor(sb.DeviceView(p_next, (L, 2)), device="cuda").copy_(
                torch.from_numpy(blocks[k + 1].view(np.float32).reshape(L, 2).copy()))
        torch.cuda.synchronize()
        assert bank.process_device(p_cur, L, d_iq.data_ptr(), d_au.data_ptr(), 512) == 512
        bank.synchronize()
        assert bank.ingest_buffer()[0] == p_next or k == 3
        assert np.array_equal(d_iq.cpu().numpy().view(np.complex64)[..., 0].view(np.uint32), want[k][0].view(np.uint32))
        assert np.array_equal(d_au.cpu().numpy().view(np.uint32), want[k][1].view(np.uint32))
        p_cur = p_next
    bank.close()


@pytest.mark.parametrize("fm", [False, True])
@pytest.mark.parametrize("stride", [512, 513, 516])
def test_device_entry_reads_the_callers_iq_rows_in_place_squelch_and_fm(stride, fm):
    """With 16-byte aligned rows (even stride) k_agc and k_squelch read the decimated IQ from the caller's buffer and no
    compact copy is written; with an odd stride they read the copy.  Either way the outputs are those of the host entry,
    bit for bit -- also for banks with squelch (FFT of the pre-AGC rows) and FM channels."""
    ch = chains.chain("10")
    L = ch.block_len
    offs = [37, -100, 5, 0, 77, -77] * 20
    levels = np.array([9.0, 0.0, 9.0, 3.0, 9.0, 0.0] * 20, np.float32)
    tabs = sb.translators_for(ch, offs)
    pb, ru, rd = chains.squelch_tables()
    on = [False, True, True, False]
    blocks = [synth.synth_iq(5, b * L, L, 0.25, [tabs[0]] if c else [], [0.2] if c else [], ch.N, 10000) for b, c in enumerate(on)]
    kw = {"squelch": (levels, pb, ru, rd)}
    if fm:
        kw["modulation"] = ["FM" if i % 3 == 0 else "AM" for i in range(len(offs))]
    a = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
    ref = [a.process(x) for x in blocks]
    a.close()
    b = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
    d_iq = torch.zeros(len(offs), stride, 2, dtype=torch.float32, device="cuda")
    d_au = torch.zeros(len(offs), stride, dtype=torch.float32, device="cuda")
    for x, (riq, rau) in zip(blocks, ref):
        dx = torch.from_numpy(np.ascontiguousarray(x).view(np.float32).reshape(-1, 2)).cuda()
        n = b.process_device(dx.data_ptr(), len(x), d_iq.data_ptr(), d_au.data_ptr(), stride)
        b.synchronize()
        assert n == 512
        giq = d_iq[:, :n].cpu().numpy().view(np.complex64)[..., 0]
        gau = d_au[:, :n].cpu().numpy()
        assert np.array_equal(giq.view(np.uint32), riq.view(np.uint32))
        assert np.array_equal(gau.view(np.uint32), rau.view(np.uint32))
    b.close()


def test_creating_and_destroying_banks_returns_their_device_memory():
    """A loop of create / host-pointer call / destroy, nothing in between: the device's free memory must not shrink (it did, by a
    bank's translator tables per round -- 62 MB of a 3 000-channel bank -- until bank_create waited for its own uploads)."""
    ch = chains.chain("10")
    n_ch = 2000
    tabs = sb.translators_for(ch, chains.offsets_cycled(n_ch, ch.N))
    x = synth.synth_iq(3, 0, ch.block_len, 0.25)

    def one():
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": 1})
        bank.process(x)
        bank.close()
    one()                                                   # (the runtime's own first-use allocations)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(8):
        one()
    torch.cuda.synchronize()
    lost = free0 - torch.cuda.mem_get_info()[0]
    assert lost < 16 << 20, "%d MB of device memory not returned by 8 banks" % (lost >> 20)


def test_creating_a_bank_does_not_wait_for_another_banks_stream():
    """bank_create ends with a wait for the bank's OWN stream only (rounds 2-3: hipDeviceSynchronize, which stalls every other
    bank's streams -- config 5 runs 16 banks).  Bank A has some 100 ms of work queued; bank B is created meanwhile: the call must
    return while A is still running, and A's outputs are what they are without the interruption."""
    import time
    ch = chains.chain("10")
    n_ch = 3000
    tabs = sb.translators_for(ch, chains.offsets_cycled(n_ch, ch.N))
    L = ch.block_len
    xs = [torch.from_numpy(x.view(np.float32).reshape(L, 2).copy()).cuda() for x in _blocks(ch, 2)]
    d_iq = torch.zeros(n_ch, 512, 2, device="cuda")
    d_au = torch.zeros(n_ch, 512, device="cuda")

    def run(interrupt):
        stream = torch.cuda.Stream()
        a = sb.ChannelBank(tabs, ch.stage_tuples(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        n_steps = 400
        for k in range(n_steps):
            assert a.process_device(xs[k & 1].data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512) == 512
        done = torch.cuda.Event()
        done.record(stream)
        still_running = None
        if interrupt:
            t0 = time.perf_counter()
            b = sb.ChannelBank(sb.translators_for(ch, [3, -3, 5, 0]), ch.stage_tuples())
            t_create = time.perf_counter() - t0
            still_running = not done.query()
            out_b = b.process(synth.synth_iq(5, 0, L, 0.25))
            assert out_b[0].shape == (4, 512)
            b.close()
            print(f"bank B created in {t_create * 1e3:.1f} ms beside bank A; A still running afterwards: {still_running}")
        a.synchronize()
        res = (d_iq.cpu().numpy().copy(), d_au.cpu().numpy().copy())
        a.close()
        return res, still_running

    plain, _ = run(False)
    got, still_running = run(True)
    assert still_running, "bank_create returned only after the other bank's queued work had finished"
    assert np.array_equal(plain[0].view(np.uint32), got[0].view(np.uint32)) and np.array_equal(plain[1].view(np.uint32), got[1].view(np.uint32))
