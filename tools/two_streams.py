"""BASELINE configs[4], one GPU's share: 2 independent 10 MS/s streams x 1024 channels, every block arriving from
page-locked host memory through sdrxgpu_bank_process_pinned (H2D on each bank's copy stream into its free input ring,
kernels behind it on the bank's stream, nothing waits).  The same loop as in bench.py, alone, for
  rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats -- python3 tools/two_streams.py
(profiles/r02_two_streams_*).  Prints ms per step of both streams."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from sdrx_amd import bank as sb, chains, synth

ch = chains.chain("10")
L = ch.block_len
if "--setstream" in sys.argv:
    sys.argv.remove("--setstream")
    ts0 = torch.cuda.Stream()
    torch.cuda.set_stream(ts0)
if "--pre" in sys.argv:      # what bench.py does before this loop: a 10 000-channel bank on an explicit torch stream
    sys.argv.remove("--pre")
    ts = torch.cuda.Stream()
    big = sb.ChannelBank(sb.translators_for(ch, chains.offsets_cycled(10000, ch.N)), ch.stage_tuples(), stream=ts.cuda_stream)
    xb = torch.from_numpy(synth.synth_iq(9, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).cuda()
    oi, oa = torch.empty(10000, 512, 2, device="cuda"), torch.empty(10000, 512, device="cuda")
    for _ in range(200):
        big.process_device(xb.data_ptr(), L, oi.data_ptr(), oa.data_ptr(), 512)
    torch.cuda.synchronize()
    big.close()
n_streams, o = 2, chains.offsets_for_config("c3")
tabs = sb.translators_for(ch, o)
banks = [sb.ChannelBank(tabs, ch.stage_tuples()) for _ in range(n_streams)]
hosts = [torch.from_numpy(synth.synth_iq(1 + i, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).pin_memory() for i in range(n_streams)]
outs = [(torch.empty(len(o), 512, 2, device="cuda"), torch.empty(len(o), 512, device="cuda")) for _ in range(n_streams)]


def step():
    for i in range(n_streams):
        banks[i].process_pinned(hosts[i].data_ptr(), L, outs[i][0].data_ptr(), outs[i][1].data_ptr(), 512)


for _ in range(50):
    step()
for b in banks:
    b.synchronize()
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.perf_counter()
for _ in range(n_it):
    step()
for b in banks:
    b.synchronize()
dt = time.perf_counter() - t0
print(f"2 streams x {len(o)} channels: {dt / n_it * 1e3:.4f} ms per step of both ({n_streams * len(o) * L * n_it / dt / 1e6:.0f} Mch-samples/s)")
for b in banks:
    b.close()
