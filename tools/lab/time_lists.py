#!/usr/bin/env python3
"""Lab: ms per block of the matrix-pipe path against the vector kernels on channel lists with few / no conjugate partners
(what bank_create's automatic choice has to get right; bench.py's own list is 83 - 99 % paired).  GPU box only."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from sdrx_amd import bank as sb, chains, synth

ch = chains.chain("10")
L = ch.block_len
x = torch.from_numpy(synth.synth_iq(3, 0, L, 0.25).view(np.float32)).cuda()


def run(offs, matrix):
    tabs = sb.translators_for(ch, offs)
    tun = {"matrix_pipe": matrix} if matrix else None
    b = sb.ChannelBank(tabs, ch.stage_tuples(), tuning=tun)
    n = len(offs)
    iq = torch.empty((n, 512, 2), dtype=torch.float32, device="cuda")
    au = torch.empty((n, 512), dtype=torch.float32, device="cuda")
    for _ in range(30):
        b.process_device(x.data_ptr(), L, iq.data_ptr(), au.data_ptr(), 512)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(60):
        b.process_device(x.data_ptr(), L, iq.data_ptr(), au.data_ptr(), 512)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 60 * 1e3
    info = b.info()
    b.close()
    return ms, int(info.matrix_pipe)


lists_small = {
    "c3: -512..511": list(range(-512, 512)),
    "-511..511 w/o 0": [o for o in range(-511, 512) if o != 0],
    "cycled 1024": chains.offsets_cycled(1024, 1200),
    "-600..599 (1200)": list(range(-600, 600)),
    "-750..749 mod (1500)": [((i + 600) % 1200) - 600 for i in range(-750, 750)],
}
lists_big = {
    "one-sided 2400": [1 + (i % 599) for i in range(2400)],
    "one-sided 6000": [1 + (i % 599) for i in range(6000)],
    "one-sided 10000": [1 + (i % 599) for i in range(10000)],
    "half paired 4000": [(-1) ** i * (1 + (i // 2) % 599) for i in range(2000)] + [1 + (i % 599) for i in range(2000)],
    "half paired 10000": [(-1) ** i * (1 + (i // 2) % 599) for i in range(5000)] + [1 + (i % 599) for i in range(5000)],
    "3/4 paired 10000": [(-1) ** i * (1 + (i // 2) % 599) for i in range(7500)] + [1 + (i % 599) for i in range(2500)],
    "paired 6000": [(-1) ** i * (1 + (i // 2) % 599) for i in range(6000)],
}
lists = lists_small if (len(sys.argv) > 1 and sys.argv[1] == "small") else lists_big
for name, offs in lists.items():
    v, _ = run(offs, 1)
    m, _ = run(offs, 2)
    a, which = run(offs, 0)
    print("%-20s vector %.4f  matrix %.4f  automatic %.4f (%s)" % (name, v, m, a, "matrix" if which else "vector"), flush=True)
