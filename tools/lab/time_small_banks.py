"""Lab: ms per block of small banks (bench.py's channel lists) with k_tail_s + k_agc against k_tail_agc, no events anywhere."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from sdrx_amd import bank as sb, chains, synth
ch = chains.chain("10"); L = ch.block_len
x = torch.from_numpy(synth.synth_iq(1, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).cuda()
for n in (32, 64, 96, 128, 192, 256, 384):
    offs = chains.offsets_for_config("c2") if n == 64 else chains.offsets_cycled(n, ch.N)
    tabs = sb.translators_for(ch, offs)
    res = []
    for rep in range(2):
        for ta in (1, 2):
            b = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"tail_agc": ta})
            d_iq = torch.empty(n, 512, 2, device="cuda"); d_au = torch.empty(n, 512, device="cuda")
            for _ in range(200): b.process_device(x.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
            b.synchronize(); t0 = time.perf_counter()
            for _ in range(1000): b.process_device(x.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
            b.synchronize(); res.append((time.perf_counter() - t0))
            b.close()
    print("%4d channels: two kernels %.4f %.4f   k_tail_agc %.4f %.4f ms per block" % (n, res[0], res[2], res[1], res[3]), flush=True)
