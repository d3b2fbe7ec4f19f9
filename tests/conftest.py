import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")

# Parity tolerance from BASELINE.json's north_star: 1e-5 relative.  The comparator is the one
# SURVEY.md 7 fixes: |got-ref| / max(|ref|, rms(ref over the channel)) -- a per-sample |ref|
# denominator alone is meaningless for noise samples that happen to be ~0.  For scale: the
# reference's own AVX2 and portable builds differ from each other by up to 5.1e-6 under this
# comparator (tests/golden/SUMMARY.json, avx2_max_rel_dev).
RTOL = 1e-5


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(got, ref):
    got = np.asarray(got)
    ref = np.asarray(ref)
    if ref.size == 0:
        return 0.0
    rms = np.sqrt(np.mean(np.abs(ref) ** 2))
    den = np.maximum(np.abs(ref), rms if rms > 0 else 1.0)
    return float(np.max(np.abs(got - ref) / den))


def rel_err_rows(got, ref):
    """Row-wise (per channel) comparator; returns the worst row."""
    return max(rel_err(g, r) for g, r in zip(got, ref))


@pytest.fixture(scope="session")
def oracle():
    from oracle import ffi
    ffi.build()
    return ffi.load("oracle")


def load_golden(name, subdir=None):
    import json
    z = np.load(os.path.join(GOLD if subdir is None else os.path.join(GOLD, subdir), name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    d["rate"] = str(d["rate"])
    d["input_sha256"] = str(d["input_sha256"])
    return d


def regen_blocks(g):
    """Regenerate a golden case's input blocks from its stored parameters (inputs are never stored)."""
    import hashlib
    from sdrx_amd import chains, synth
    ch = chains.chain(g["rate"])
    meta = g["meta"]
    lens = [int(v) for v in g["block_lens"]]
    total = sum(lens)
    kind = meta.get("kind")
    if kind == "impulse":
        x = np.zeros(total, dtype=np.complex64)
        x[meta["at"]] = 1.0 + 0.5j
    elif kind == "dc":
        x = np.full(total, complex(*meta["value"]), dtype=np.complex64)
    elif kind == "tone":
        offs = [int(o) for o in g["offsets"]]
        t = g["tables"][offs.index(meta["off"])]
        L = lens[0]
        one = (np.float32(meta["amp"]) * np.conj(np.tile(t, L // ch.N + 1)[:L])).astype(np.complex64)
        x = np.concatenate([one] * len(lens))
    else:
        offs = [int(o) for o in g["offsets"]]
        car = []
        for o in meta.get("carrier_offs", []):
            if o in offs:
                car.append(g["tables"][offs.index(o)])
            else:   # carrier table not among the case's channels: rebuild it (libm-dependent, checked by sha below)
                from oracle import ffi
                car.append(ffi.load("oracle").make_translator(ch.N, ch.z, o))
        x = synth.synth_iq(meta["seed"], 0, total, meta["noise_amp"], car, meta.get("carrier_amp", []), ch.N,
                           meta.get("tri_period", ch.fs // 1000), meta.get("tri_depth", 0.5))
    assert hashlib.sha256(x.tobytes()).hexdigest() == g["input_sha256"], "regenerated input differs from the fixture's"
    blocks, p = [], 0
    for n in lens:
        blocks.append(x[p:p + n])
        p += n
    return ch, blocks


def margin(name, **errs):
    """Record the worst relative error a parity test observed (printed with -s; collected into
    gpurun_out/parity_margins.jsonl on the GPU box and summarised under profiles/)."""
    import json
    rec = {"test": name, **{k: float(v) for k, v in errs.items()}}
    print("margin", json.dumps(rec))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_margins.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
