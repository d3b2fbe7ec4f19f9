"""Conjugate channel pairs under a strong carrier over a quiet floor: the regime in which the matrix-pipe
kernel's `P -+ Q` formulation is at its weakest -- the channel at -k steps is the difference of two sums that both carry the
carrier on +k.  Fixtures: tests/golden/pairs/, outputs of the reference compiled here (oracle/gen_fixtures.py), carrier of mean
amplitude A in {0.3, 0.6, 0.9, 0.99} on +37 ("one") or 0.8 A / 0.2 A on +37 / -37 ("both"), noise in {0.25, 0.05, 0.01}.

Three numbers per case and output kind, written to gpurun_out/r04_parity_pairs.jsonl (-> profiles/r04_parity_margins.json):
fused path (k_stage123m) / matrix path (k_stage12m) / vector path vs reference, the reference's own AVX2 build vs its portable build (from the fixture).

What the fixtures say first (CPU test tests/test_oracle.py): under a strong carrier EVERY weak channel's error floor is set by
the carrier -- in the reference itself.  Its Release build (AVX2 + FMA) is 2e-5 to 5e-5 away from its portable build at a floor
of 0.01, up to 9e-6 at 0.05 and within 3.4e-6 at 0.25: no implementation that is not the portable build instruction for
instruction can hold 1e-5 against it at the quiet end, and none does (measured: the vector kernels, which keep the reference's
operation order, are at 0.9 to 1.4 times the reference's own spread; the matrix path at 0.8 to 1.75 times, after its K steps
were put in outside-in order -- it was up to 2.15 times before).

What is asserted, for both kernel paths:
* within 1e-5 of the reference's portable build wherever the reference's own two builds agree to 0.5e-5 (all cases at a floor of
  0.25, carriers up to 0.6 at a floor of 0.05);
* never further from the portable build than VECTOR_VS_AVX2 resp. MATRIX_VS_AVX2 times the reference's own spread of the case
  ("Numerical contract").
"""
import glob
import json
import os

import numpy as np
import pytest

from conftest import GOLD, ROOT, RTOL, load_golden, regen_blocks, rel_err, margin
from sdrx_amd import bank as sb

pytestmark = pytest.mark.gpu

CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "pairs", "g[89]_*.npz")))
# how far a path may be from the reference's portable build, in units of the reference's own AVX2-vs-portable spread of the
# case (measured: vector kernels <= 1.41, matrix path <= 1.75, fused path <= 3.15), and the spread up to which 1e-5 itself must
# hold -- for the two default paths; the fused path (opt-in: SDRXGPU_FLAG_FUSED_STAGE3) only where the spread is a third of that
VECTOR_VS_AVX2 = 1.7
MATRIX_VS_AVX2 = 2.1
FUSED_VS_AVX2 = 3.5
AGREE = 0.5e-5


def _run(ch, tabs, blocks, matrix_pipe):
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"matrix_pipe": matrix_pipe})
    assert bank.info().matrix_pipe == matrix_pipe - 1
    outs = [bank.process(x) for x in blocks]
    bank.close()
    return np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1)


def test_fixture_inventory():
    assert len(CASES) == 30


@pytest.mark.parametrize("name", CASES)
def test_pair_under_a_strong_carrier(name):
    g = load_golden(name, "pairs")
    ch, blocks = regen_blocks(g)
    offs = [int(o) for o in g["offsets"]]
    tabs = [g["tables"][i] if o != 0 else None for i, o in enumerate(offs)]
    rec = {"case": name, "A": g["meta"]["A"], "noise": g["meta"]["noise_amp"], "variant": g["meta"]["variant"], "offsets": offs}
    worst = {}
    nearer = {}
    for path, mp in (("fused", 3), ("matrix", 2), ("vector", 1)):
        iq, au = _run(ch, tabs, blocks, mp)
        e_iq = [rel_err(iq[i], g["iq"][i]) for i in range(len(offs))]
        e_au = [rel_err(au[i], g["audio_lf"][i]) for i in range(len(offs))]
        rec[path + "_vs_ref_iq"], rec[path + "_vs_ref_audio"] = e_iq, e_au
        worst[path] = (max(e_iq), max(e_au))
        # the fourth number (round 5): per channel the distance to the NEARER of the reference's two builds -- the portable one and
        # the Release one (AVX2 + FMA: what sdrx ships, CMakeLists.txt:26-28,107), whose outputs the fixtures now hold too
        n_iq = [min(a, rel_err(iq[i], g["avx2_iq"][i])) for i, a in enumerate(e_iq)]
        n_au = [min(a, rel_err(au[i], g["avx2_audio_lf"][i])) for i, a in enumerate(e_au)]
        rec[path + "_vs_nearer_build_iq"], rec[path + "_vs_nearer_build_audio"] = n_iq, n_au
        rec[path + "_vs_avx2_build_iq"] = [rel_err(iq[i], g["avx2_iq"][i]) for i in range(len(offs))]
        rec[path + "_vs_avx2_build_audio"] = [rel_err(au[i], g["avx2_audio_lf"][i]) for i in range(len(offs))]
        nearer[path] = (max(n_iq), max(n_au))
    rec["ref_avx2_vs_portable_iq"] = [float(v) for v in g["avx2_rel_dev_iq"]]
    rec["ref_avx2_vs_portable_audio"] = [float(v) for v in g["avx2_rel_dev_audio_lf"]]
    a_iq, a_au = max(rec["ref_avx2_vs_portable_iq"]), max(rec["ref_avx2_vs_portable_audio"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_parity_pairs.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    margin(f"pairs[{name}]", fused_iq=worst["fused"][0], fused_audio=worst["fused"][1], matrix_iq=worst["matrix"][0], matrix_audio=worst["matrix"][1],
           vector_iq=worst["vector"][0], vector_audio=worst["vector"][1], ref_avx2_iq=a_iq, ref_avx2_audio=a_au,
           matrix_nearer_iq=nearer["matrix"][0], matrix_nearer_audio=nearer["matrix"][1],
           vector_nearer_iq=nearer["vector"][0], vector_nearer_audio=nearer["vector"][1])
    for path, times in (("vector", VECTOR_VS_AVX2), ("matrix", MATRIX_VS_AVX2), ("fused", FUSED_VS_AVX2)):
        for got, spread in ((worst[path][0], a_iq), (worst[path][1], a_au)):
            assert got <= max(RTOL, times * spread), (path, got, spread)
            if spread <= (AGREE / 1.7 if path == "fused" else AGREE):
                assert got <= RTOL, (path, got, spread)
