import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import torch
from sdrx_amd import bank as sb, chains, synth
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"): sb.LIB_PATH = sys.argv[1]
ch = chains.chain("10")
tabs = sb.translators_for(ch, chains.offsets_cycled(10000, ch.N))
bank = sb.ChannelBank(tabs, ch.stage_tuples())
x = torch.as_tensor(synth.synth_iq(3, 0, ch.block_len, 0.25)).cuda()
d_iq = torch.empty((10000, 512), dtype=torch.complex64, device="cuda"); d_au = torch.empty((10000, 512), dtype=torch.float32, device="cuda")
bank.profile_enable(True)
for _ in range(12): bank.process_device(x.data_ptr(), ch.block_len, d_iq.data_ptr(), d_au.data_ptr(), 512)
bank.synchronize()
print(sys.argv[1:], [round(bank.profile_read(k)[0] / bank.profile_read(k)[1], 4) for k in (0, 1, 2, 3)])
