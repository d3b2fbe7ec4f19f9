"""Lab: ms per block of BASELINE's 64- and 1 024-channel banks (bench.py's lists) with lab builds of the AGC kernels (SDRXGPU_LIB_PATH):
-DSDRX_LAB_FAST_AGC=1 the IF recurrence with six dependent operations, =2 the LF recurrence skipped while --lf-agc is off, =3 both."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from sdrx_amd import bank as sb, chains, synth
ch = chains.chain("10"); L = ch.block_len
x = torch.from_numpy(synth.synth_iq(1, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).cuda()
out = []
for n, key in ((64, "c2"), (1024, "c3"), (256, None)):
    offs = chains.offsets_for_config(key) if key else chains.offsets_cycled(n, ch.N)
    tabs = sb.translators_for(ch, offs)
    for ta in (1, 2):
        b = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"tail_agc": ta})
        d_iq = torch.empty(n, 512, 2, device="cuda"); d_au = torch.empty(n, 512, device="cuda")
        for _ in range(300): b.process_device(x.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
        b.synchronize(); t0 = time.perf_counter()
        for _ in range(1000): b.process_device(x.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
        b.synchronize(); out.append("%d ch %s %.4f" % (n, "k_tail_agc" if ta == 2 else "two kernels", time.perf_counter() - t0))
        b.close()
print(os.environ.get("SDRXGPU_LIB_PATH", "product") + ": " + "; ".join(out), flush=True)
