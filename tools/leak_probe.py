#!/usr/bin/env python3
"""Which pairing makes hipFree a no-op at bank destroy?  Runs the create / host-pointer call / destroy loop of
tests/test_gpu_entries.py against lab builds of the library (tools/leak_probe.sh builds them into tmp_exp/leak/):

  product       uploads on the bank's own stream (hipMemcpyAsync + hipStreamSynchronize), no device-wide sync
  null_upload   rounds 2-3: blocking hipMemcpy (NULL stream) uploads, nothing else                      -> expected to leak
  null_devsync  the same + hipDeviceSynchronize at the end of bank_create (round 3's workaround)
  null_nullsync the same + hipStreamSynchronize(NULL) at the end of bank_create

Each variant runs in a process of its own (the library is chosen at import: SDRXGPU_LIB_PATH).  Prints MB lost per bank."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import torch
from sdrx_amd import bank as sb, chains, synth
ch = chains.chain("10")
n_ch = 2000
tabs = sb.translators_for(ch, chains.offsets_cycled(n_ch, ch.N))
x = synth.synth_iq(3, 0, ch.block_len, 0.25)
entry = sys.argv[2]
d_iq = torch.empty(n_ch, 512, 2, device="cuda"); d_au = torch.empty(n_ch, 512, device="cuda")
d_x = torch.from_numpy(x.view("float32").reshape(-1, 2).copy()).cuda()
def one():
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": 1})
    if entry == "host":
        bank.process(x)
    elif entry == "device":
        bank.process_device(d_x.data_ptr(), ch.block_len, d_iq.data_ptr(), d_au.data_ptr(), 512); bank.synchronize()
    bank.close()
one()
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
n = 6
for _ in range(n):
    one()
torch.cuda.synchronize()
print("LOST_MB_PER_BANK %.1f" % ((free0 - torch.cuda.mem_get_info()[0]) / n / 2 ** 20))
"""

for variant in ("product", "null_upload", "null_devsync", "null_nullsync"):
    lib = os.path.join(ROOT, "tmp_exp", "leak", f"libsdrxgpu_{variant}.so")
    if not os.path.exists(lib):
        print(variant, "not built (tools/leak_probe.sh)")
        continue
    for entry in ("host", "device", "none"):
        r = subprocess.run([sys.executable, "-c", CHILD, ROOT, entry], env=dict(os.environ, SDRXGPU_LIB_PATH=lib), capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("LOST")]
        print(f"{variant:14s} first call: {entry:6s} -> {line[0] if line else 'FAILED ' + r.stderr[-300:]}", flush=True)
