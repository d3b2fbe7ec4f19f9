"""Lab: k_stage12m6 against the vector kernel of the 6 MS/s chain for channel lists with / without conjugate partners.
  python tools/lab/bench_m6_lists.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from sdrx_amd import bank as sb, chains, synth

ch = chains.chain("6")
L = ch.block_len
x = torch.from_numpy(synth.synth_iq(1, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).cuda()
lists = {}
for n in (512, 768, 1024, 1536, 2048, 4096, 10000):
    lists["cycled_%d" % n] = chains.offsets_cycled(n, ch.N)
lists["one_sided_2048"] = [1 + (i % 359) for i in range(2048)]
lists["one_sided_10000"] = [1 + (i % 359) for i in range(10000)]
lists["half_paired_4096"] = [((i % 359) + 1) * (1 if (i % 4) else -1) for i in range(4096)]
for name, offs in lists.items():
    tabs = sb.translators_for(ch, offs)
    res = {}
    for mp in (1, 2):
        b = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": mp})
        info = b.info()
        d_iq = torch.empty(len(offs), 512, 2, device="cuda"); d_au = torch.empty(len(offs), 512, device="cuda")
        for _ in range(60):
            b.process_device(x.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
        b.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            b.process_device(x.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
        b.synchronize(); res[mp] = (time.perf_counter() - t0) / 200 * 1e3
        slots = info.m_slots
        b.close()
    print(f"{name:20s} vector {res[1]:.4f} ms  matrix {res[2]:.4f} ms  ratio {res[2] / res[1]:.2f}  slots/channels {slots / len(offs):.2f}", flush=True)
