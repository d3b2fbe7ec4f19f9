#!/usr/bin/env python3
"""Lab: small banks of the 10 MS/s chain through the vector kernels: the automatic geometry against every forced one
(W = 4 waves; CPW channels per wave; nb bodies per lane), stage-1/2 kernel ms.  SDRXGPU_LIB_PATH picks the build."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
from sdrx_amd import bank as sb, chains, synth
ch = chains.chain("10"); L = ch.block_len
x = torch.from_numpy(synth.synth_iq(9, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).cuda()
def t(tabs, n_ch, **kw):
    iq = torch.empty((n_ch, 512, 2), dtype=torch.float32, device='cuda'); au = torch.empty((n_ch, 512), dtype=torch.float32, device='cuda')
    try: bk = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
    except sb.SdrxGpuError: return None
    bk.profile_enable(1)
    for _ in range(25): bk.process_device(x.data_ptr(), L, iq.data_ptr(), au.data_ptr(), 512)
    torch.cuda.synchronize(); bk.profile_reset()
    for _ in range(40): bk.process_device(x.data_ptr(), L, iq.data_ptr(), au.data_ptr(), 512)
    torch.cuda.synchronize()
    ms, n = bk.profile_read(sb.K_STAGE12); k12 = ms / max(n, 1)
    i = bk.info(); bk.close(); return k12, (i.k12_waves, i.k12_cpw, i.k12_bodies)
for n_ch in [int(a) for a in sys.argv[1:]] or (16, 32, 64, 96, 128, 192, 256, 384, 512):
    offs = chains.offsets_cycled(n_ch, ch.N); tabs = sb.translators_for(ch, offs)
    a = t(tabs, n_ch)
    res = []
    for c in (4, 2, 1):
        for hint in (1, 2, 3, 4, 6, 12):
            r = t(tabs, n_ch, block_hint=hint if hint != 12 else 0xFFFFFFFF, tuning={"k12_waves": 4, "k12_cpw": c, "pair_lanes": 1, "fused_tail": 1})
            if r: res.append((r[0], c, hint))
    res.sort()
    print(n_ch, "auto %.4f %s" % a, "| best forced:", ["%.4f cpw%d nb%d" % r for r in res[:3]], flush=True)
