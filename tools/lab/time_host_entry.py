"""Lab: the drop-in entry (host pointers, page-locked, complete on return) of the 10 000-channel bank: ms per block with IQ + audio
and audio only, by tuning.host_parts.  SDRXGPU_LIB_PATH picks a lab build."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from sdrx_amd import bank as sb, chains, synth
ch = chains.chain("10"); L = ch.block_len; n = 10000
tabs = sb.translators_for(ch, chains.offsets_cycled(n, ch.N))
x = synth.synth_iq(9, 0, L, 0.25); o_iq = np.zeros((n, 512), np.complex64); o_au = np.zeros((n, 512), np.float32)
for a in (x, o_iq, o_au): sb.host_register(a)
for parts in [int(v) for v in (sys.argv[1:] or ["0", "3", "4", "5", "6", "8"])]:
    b = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"host_parts": parts})
    r = []
    for io in (dict(iq_out=o_iq, audio_out=o_au), dict(iq_out=None, audio_out=o_au)):
        for _ in range(3): b.process(x, **io)
        t0 = time.perf_counter()
        for _ in range(12): b.process(x, **io)
        r.append((time.perf_counter() - t0) / 12 * 1e3)
    b.close()
    print("host_parts %d: IQ + audio %.3f ms, audio only %.3f ms" % (parts, r[0], r[1]), flush=True)
