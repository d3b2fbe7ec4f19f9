#!/usr/bin/env python3
"""Lab: stage-1/2 kernel ms of the vector kernels for W = 4 x CPW in (4, 2, 1) x bodies per lane in (1, 2, 3, 4, 6, 12) over bank sizes
1 ... 1 023 (10 MS/s chain) -> gpurun_out/sweep_w4.jsonl: what the geometry model's body_time() constants are fitted to."""
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
    for _ in range(20): bk.process_device(x.data_ptr(), L, iq.data_ptr(), au.data_ptr(), 512)
    torch.cuda.synchronize(); bk.profile_reset()
    for _ in range(40): bk.process_device(x.data_ptr(), L, iq.data_ptr(), au.data_ptr(), 512)
    torch.cuda.synchronize()
    ms, n = bk.profile_read(sb.K_STAGE12); k12 = ms / max(n, 1)
    i = bk.info(); bk.close(); return k12, (i.k12_waves, i.k12_cpw, i.k12_bodies)
out = open("gpurun_out/sweep_w4.jsonl", "w")
for n_ch in (1, 8, 16, 24, 32, 48, 64, 80, 96, 128, 160, 192, 224, 256, 320, 384, 448, 512, 640, 768, 896, 1000, 1023):
    offs = chains.offsets_cycled(n_ch, ch.N); tabs = sb.translators_for(ch, offs)
    a = t(tabs, n_ch, tuning={"matrix_pipe": 1, "pair_lanes": 1})
    out.write(json.dumps({"n_ch": n_ch, "auto": a[0], "geom": a[1]}) + "\n")
    for c in (4, 2, 1):
        for hint in (1, 2, 3, 4, 6, 12):
            r = t(tabs, n_ch, block_hint=hint if hint != 12 else 0xFFFFFFFF, tuning={"k12_waves": 4, "k12_cpw": c, "pair_lanes": 1, "fused_tail": 1, "matrix_pipe": 1})
            if r: out.write(json.dumps({"n_ch": n_ch, "W": 4, "CPW": c, "nb": hint, "k12": r[0]}) + "\n")
    out.flush(); print(n_ch, flush=True)
