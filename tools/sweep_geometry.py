"""Times the stage-1/2 kernel and the tail of the 10 MS/s chain for every compiled geometry (waves per workgroup x channels
per wave x bodies per lane) over bank sizes 1 ... 1 023: the data the geometry model of sdrxgpu_bank_create was fitted to
(profiles/r02_geometry_sweep.jsonl).  Run on a GPU box from the repo root: python tools/sweep_geometry.py"""
import sys, time, json, numpy as np
sys.path.insert(0,'.')
import torch
from sdrx_amd import bank as sb, chains, synth
ch=chains.chain("10"); L=ch.block_len
x=torch.from_numpy(synth.synth_iq(9,0,L,0.25).view(np.float32).reshape(L,2).copy()).cuda()
def t(tabs, n_ch, **kw):
    iq=torch.empty((n_ch,512,2),dtype=torch.float32,device='cuda'); au=torch.empty((n_ch,512),dtype=torch.float32,device='cuda')
    try: bk=sb.ChannelBank(tabs,ch.stage_tuples(),**kw)
    except sb.SdrxGpuError: return None
    bk.profile_enable(1)
    for _ in range(25): bk.process_device(x.data_ptr(), L, iq.data_ptr(), au.data_ptr(), 512)
    torch.cuda.synchronize(); bk.profile_reset()
    for _ in range(30): bk.process_device(x.data_ptr(), L, iq.data_ptr(), au.data_ptr(), 512)
    torch.cuda.synchronize()
    ms,n=bk.profile_read(sb.K_STAGE12); k12=ms/max(n,1)
    ms,n=bk.profile_read(sb.K_TAIL); kt=ms/max(n,1)
    i=bk.info(); bk.close(); return k12, kt, i.fused_tail
out=open("gpurun_out/geometry_sweep.jsonl","w")
for n_ch in (1, 8, 16, 32, 64, 96, 128, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1000, 1023):
    offs=chains.offsets_cycled(n_ch, ch.N); tabs=sb.translators_for(ch,offs)
    for w,c in ((16,4),(16,2),(16,1),(8,4),(8,2),(8,1),(4,4),(4,2),(4,1)):
        for hint in (1,2,3,4,6,12):
            for fused in ((1,) if hint!=12 else (1,2)):
                r=t(tabs,n_ch,block_hint=hint if hint!=12 else 0xFFFFFFFF,tuning={"k12_waves":w,"k12_cpw":c,"pair_lanes":1,"fused_tail":fused})
                if r: out.write(json.dumps({"n_ch":n_ch,"W":w,"CPW":c,"nb":hint,"fused":int(r[2]),"k12":r[0],"tail":r[1]})+"\n")
    out.flush(); print(n_ch, flush=True)
