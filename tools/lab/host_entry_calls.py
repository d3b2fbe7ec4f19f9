"""Lab: a few calls of the drop-in entry (sdrxgpu_bank_process, page-locked host buffers) of the 10 000-channel bank, for a
trace: <n_calls> calls with IQ + audio, a pause, <n_calls> with audio only.  tuning as k=v arguments (host_part_ch=a,b,c).
  python3 tools/lab/host_entry_calls.py [n_calls] [k=v ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from sdrx_amd import bank as sb, chains, synth
args = sys.argv[1:]
n_calls = int(args.pop(0)) if args and args[0].isdigit() else 6
tun = {}
for a in args:
    k, v = a.split("=")
    tun[k] = tuple(int(t) for t in v.split(",")) if "," in v or k == "host_part_ch" else int(v)
ch = chains.chain("10"); L = ch.block_len; n = int(os.environ.get("N_CH", "10000"))
tabs = sb.translators_for(ch, chains.offsets_cycled(n, ch.N))
x = synth.synth_iq(9, 0, L, 0.25); o_iq = np.zeros((n, 512), np.complex64); o_au = np.zeros((n, 512), np.float32)
for a in (x, o_iq, o_au):
    sb.host_register(a)
b = sb.ChannelBank(tabs, ch.stage_tuples(), tuning=tun or None)
print("parts: IQ + audio", b.host_parts(), "audio only", b.host_parts(True), flush=True)
if os.environ.get("SIDE_JSON"):
    import json
    json.dump({"parts": b.host_parts(), "parts_audio_only": b.host_parts(True), "n_channels": n, "tuning": {k: list(v) if isinstance(v, tuple) else v for k, v in tun.items()}}, open(os.environ["SIDE_JSON"], "w"))
for io in (dict(iq_out=o_iq, audio_out=o_au), dict(iq_out=None, audio_out=o_au)):
    for _ in range(3):
        b.process(x, **io)
    t0 = time.perf_counter()
    for _ in range(n_calls):
        b.process(x, **io)
    print("%s: %.3f ms per call" % ("IQ + audio" if io["iq_out"] is not None else "audio only", (time.perf_counter() - t0) / n_calls * 1e3), flush=True)
    time.sleep(0.05)
b.close()
