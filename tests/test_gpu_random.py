"""Randomised (fixed seeds) end-to-end parity: random chain, channel set, call-length split, LF-AGC / ftfir / audio FIR
flags, against the oracle's per-channel chain (MSD -> AGC -> |.| -> LfAGC [-> FIR]) on the same stream."""
import numpy as np
import pytest

from conftest import RTOL, margin, rel_err
from sdrx_amd import bank as sb, chains, synth

pytestmark = pytest.mark.gpu

RATES = ["10", "2.4", "0.96", "1.2", "1.44", "1.6", "1.92", "6", "2.56"]


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_configuration_vs_oracle(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    ch = chains.chain(RATES[seed % len(RATES)])
    n_ch = int(rng.integers(1, 40))
    offs = [int(o) for o in rng.integers(-ch.N // 2, ch.N // 2, n_ch)]
    if seed % 3 == 0:
        offs[0] = 0                                           # a channel at the tuner centre (no translator)
    ftfir = bool(seed % 5 == 1) and ch.N % ch.stages[0].m == 0
    lf = bool(seed % 2)
    fir = chains.audio_bpf_taps() if seed % 4 == 2 else None
    hint = [0, 0xFFFFFFFF][seed % 2]
    tabs = sb.translators_for(ch, offs)
    L = ch.block_len
    total = int(L * rng.uniform(1.2, 2.6))
    carriers = [t for t in tabs[:2] if t is not None]
    x = synth.synth_iq(seed, 0, total, 0.25, carriers, [0.2] * len(carriers), ch.N, 5000)
    # random split into calls of 0 .. one block
    cuts, p = [0], 0
    while p < total:
        p = min(total, p + int(rng.choice([0, 1, ch.m_total - 1, ch.m_total, 1000, L // 3, L])))
        cuts.append(p)
    bank = sb.ChannelBank(tabs, ch.stage_tuples(), lf_agc=lf, ftfir=ftfir, audio_fir=fir, block_hint=hint)
    iq_parts, au_parts = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        iq, au = bank.process(x[a:b])
        iq_parts.append(iq)
        au_parts.append(au)
    bank.close()
    iq = np.concatenate(iq_parts, axis=1)
    au = np.concatenate(au_parts, axis=1)
    assert iq.shape[1] == total // ch.m_total
    w_iq = w_au = 0.0
    for c in rng.choice(n_ch, size=min(n_ch, 4), replace=False):
        msd = oracle.msd(tabs[c], ch.stage_tuples(), ftfir)
        y = msd.decimate(x)
        agc = oracle.agc_pair(lf)
        a = agc.adjust_lf(oracle.demod_am(agc.adjust(y)))
        if fir is not None:
            a = oracle.fir(fir, 0.0).filter(a)
        w_iq, w_au = max(w_iq, rel_err(iq[c], y)), max(w_au, rel_err(au[c], a))
    margin(f"random[{seed}] rate={ch.rate} lf={int(lf)} ftfir={int(ftfir)} fir={int(fir is not None)}", iq=w_iq, audio=w_au)
    # An ACTIVE LF AGC amplifies rounding (two builds of the reference itself drift apart by 1.7e-5 within 5 blocks and 4e-3
    # within 16, tests/test_oracle.py::test_active_lf_agc_is_ill_conditioned_in_the_reference_itself; worst seen here over
    # these 1.2 - 2.6 blocks: 1.0624e-5, seed 11, the same on every box and in every round since round 2): that measured worst
    # + 20 % for those cases only -- 1.28e-5, not a round 2e-5
    assert w_iq <= RTOL and w_au <= (1.28e-5 if lf else RTOL), (seed, w_iq, w_au)
