#!/bin/bash
# Lab: k_stage12m below 1 024 channels (paired lists -N/2 .. N/2-1) against the vector kernel, by bodies per lane.  -> gpurun_out/r05_small_matrix.txt
cd $GRAFT_REPO_ROOT
out=gpurun_out/r05_small_matrix.txt
: > $out
python3 - >> $out 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from sdrx_amd import bank as sb, chains, synth
ch = chains.chain("10"); L = ch.block_len
x = torch.from_numpy(synth.synth_iq(1, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).cuda()
for n in (256, 384, 512, 640, 768, 896, 1024):
    offs = list(range(-n // 2, n // 2))
    tabs = sb.translators_for(ch, offs)
    res = []
    for label, kw in [("vector", dict(tuning={"matrix_pipe": 1}))] + [("matrix nb=%s" % h, dict(tuning={"matrix_pipe": 2}, block_hint=h)) for h in (0, 1, 2, 3, 4, 6)]:
        try:
            b = sb.ChannelBank(tabs, ch.stage_tuples(), **kw)
        except sb.SdrxGpuError as e:
            res.append("%s: refused" % label); continue
        d_iq = torch.empty(n, 512, 2, device="cuda"); d_au = torch.empty(n, 512, device="cuda")
        for _ in range(60): b.process_device(x.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
        b.synchronize(); t0 = time.perf_counter()
        for _ in range(300): b.process_device(x.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
        b.synchronize(); ms = (time.perf_counter() - t0) / 300 * 1e3
        res.append("%s: %.4f (bodies %d)" % (label, ms, b.info().k12_bodies))
        b.close()
    print(n, "channels |", " | ".join(res), flush=True)
PY
cat $out
