"""Lab: SHA-256 of the outputs (IQ, audio, AGC state) of a fixed set of banks and streams -- run under two builds of the library
(SDRXGPU_LIB_PATH) and compare the lines: a change that claims "same bits" has to leave every line as it was.
  python tools/lab/hash_outputs.py > a.txt;  SDRXGPU_LIB_PATH=.../lib_other.so python tools/lab/hash_outputs.py > b.txt;  diff a.txt b.txt"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from sdrx_amd import bank as sb, chains, synth


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:24]


cases = []
for rate, n_ch, kw in (("10", 64, {}), ("10", 64, {"lf_agc": True}), ("10", 300, {"tuning": {"tail_agc": 2}}), ("10", 300, {"tuning": {"tail_agc": 1}}),
                       ("10", 1024, {}), ("10", 2304, {}), ("2.4", 40, {"lf_agc": True}), ("6", 128, {}), ("10", 96, {"squelch": 9.0}),
                       ("10", 48, {"fm": True, "lf_agc": True}), ("10", 70, {"agc": (0.02, 0.5)}), ("10", 70, {"agc": (0.3, 0.3), "lf_agc": True})):
    cases.append((rate, n_ch, kw))
rng = np.random.default_rng(5)
for rate, n_ch, kw in cases:
    ch = chains.chain(rate)
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    car = [t for t in tabs[1:4] if t is not None]
    amp = [0.9, 0.05, 0.3][:len(car)]
    args = dict(kw)
    agc = lf = None
    if "agc" in args:                       # (attack, decay) other than sdrx's: also decay > attack and decay == attack
        att, dec = args.pop("agc")
        agc, lf = sb.sdrx_agc_params()
        agc.attack, agc.decay, lf.attack, lf.decay = att, dec, dec, att
        args["agc"], args["agc_lf"] = agc, lf
    if args.pop("fm", False):
        args["modulation"] = [i % 2 for i in range(n_ch)]
    sq = "squelch" in args
    if sq:
        pb, ru, rd = chains.squelch_tables()
        args["squelch"] = (np.full(n_ch, args["squelch"], np.float32), pb, ru, rd)
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), **args)
    outs = []
    pos = 0
    for n in (L, L // 2 + 625, 1, L, 3 * ch.m_total + 1, L):
        x = synth.synth_iq(77 + n_ch, pos, n, 0.02 if n_ch == 70 else 0.25, car, amp, ch.N, ch.fs // 1000)
        pos += n
        if sq and n != L:
            continue                        # (squelch banks take whole chunks)
        iq, au = bank.process(x)
        outs += [iq, au]
    st = bank.agc_state()
    bank.close()
    print(rate, n_ch, sorted((k, str(v)[:40]) for k, v in kw.items()), digest(*outs), digest(*st), flush=True)
