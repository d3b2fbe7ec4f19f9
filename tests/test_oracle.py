"""CPU tests: the oracle (oracle/sdrx_oracle.c) against the golden vectors produced by the
compiled reference (oracle/gen_fixtures.py), plus independent known-answer checks."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLD, RTOL, load_golden, regen_blocks, rel_err, rel_err_rows
from sdrx_amd import chains, synth

ALL_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "*.npz")))


def _bits(a):
    return a if a.dtype == np.int64 else a.view(np.uint32)


def _run_chain(lib, ch, table, blocks, use_ftfir=False, lf_agc=False, fir_taps=None):
    msd = lib.msd(table, ch.stage_tuples(), use_ftfir)
    agc = lib.agc_pair(lf_agc)
    fir = lib.fir(fir_taps, 0.0) if fir_taps is not None else None
    out = {k: [] for k in ("iq", "agc_iq", "audio", "audio_lf", "audio_fir", "out_lens")}
    for x in blocks:
        y = msd.decimate(x)
        out["iq"].append(y)
        out["out_lens"].append(len(y))
        ya = agc.adjust(y)
        out["agc_iq"].append(ya)
        a = lib.demod_am(ya)
        out["audio"].append(a)
        al = agc.adjust_lf(a)
        out["audio_lf"].append(al)
        if fir is not None:
            out["audio_fir"].append(fir.filter(al))
    res = {k: np.concatenate(v) for k, v in out.items() if k != "out_lens" and len(v)}
    res["out_lens"] = np.array(out["out_lens"], dtype=np.int64)
    return res


def test_fixture_inventory():
    assert len(ALL_CASES) >= 16
    for must in ("g1_fs2400_c1", "g2_fs10000", "g3_fs10000_noise", "g4_fs10000_impulse", "g5_fs10000_ftfir",
                 "g6_fs10000_ragged"):
        assert must in ALL_CASES


@pytest.mark.parametrize("name", ALL_CASES)
def test_oracle_bit_exact_vs_reference_golden(oracle, name):
    """The restatement reproduces the reference's portable build bit for bit on every fixture."""
    g = load_golden(name)
    ch, blocks = regen_blocks(g)
    kw = {}
    if "ftfir" in name:
        kw["use_ftfir"] = True
    if "lfagc" in name:
        kw["lf_agc"] = True
    if "audio_fir" in g:
        kw["fir_taps"] = chains.audio_bpf_taps()
    for i, off in enumerate(int(o) for o in g["offsets"]):
        # libm check: the table built on THIS box equals the committed one
        t_here = oracle.make_translator(ch.N, ch.z, off)
        assert np.array_equal(t_here.view(np.uint32), g["tables"][i].view(np.uint32)), "cexpf differs on this box"
        table = g["tables"][i] if off != 0 else None
        r = _run_chain(oracle, ch, table, blocks, **kw)
        for k in ("iq", "agc_iq", "audio", "audio_lf", "audio_fir", "out_lens"):
            if k in g:
                assert np.array_equal(_bits(r[k]), _bits(g[k][i])), f"{name} off={off} {k}: rel={rel_err(r[k], g[k][i])}"


PAIR_CASES = sorted(("pairs", os.path.basename(p)[:-4]) for p in glob.glob(os.path.join(GOLD, "pairs", "*.npz"))) + \
             sorted(("pairs6", os.path.basename(p)[:-4]) for p in glob.glob(os.path.join(GOLD, "pairs6", "*.npz")))


@pytest.mark.parametrize("folder,name", PAIR_CASES)
def test_oracle_bit_exact_vs_reference_on_the_pair_stress_fixtures(oracle, folder, name):
    """tests/golden/pairs (10 MS/s chain) and pairs6 (6 MS/s chain, round 5): conjugate channel pairs under a carrier of up to 0.99 of
    full scale over a floor down to 0.01 (the reference's decimated IQ and audio): the restatement reproduces the portable build bit
    for bit there too."""
    g = load_golden(name, folder)
    ch, blocks = regen_blocks(g)
    for i, off in enumerate(int(o) for o in g["offsets"]):
        r = _run_chain(oracle, ch, g["tables"][i] if off != 0 else None, blocks)
        for k in ("iq", "audio_lf", "out_lens"):
            assert np.array_equal(_bits(r[k]), _bits(g[k][i])), f"{name} off={off} {k}"


def test_the_reference_builds_disagree_by_more_than_the_tolerance_over_a_quiet_floor():
    """Scale of the problem, second half: under a strong carrier every weak channel's error floor is set by the carrier, in the
    reference itself -- its AVX2 (Release) and portable builds are 2e-5 to 5e-5 apart at a noise floor of 0.01 and within 1e-5
    at 0.25.  A path that sums in any order but the portable build's cannot hold 1e-5 against it there (tests/test_gpu_pairs_stress.py)."""
    import json
    s = json.load(open(os.path.join(GOLD, "pairs", "SUMMARY.json")))
    s6 = json.load(open(os.path.join(GOLD, "pairs6", "SUMMARY.json")))
    assert len(s) == 30 and len(s6) == 6 and len(PAIR_CASES) == 36
    # the 6 MS/s chain (three stages, a 53-tap first one): the same picture -- a few 1e-6 for the pair channels, beyond 1e-5 for the
    # carrier's neighbours under a carrier of 0.9
    assert max(v["avx2_max_rel_dev"] for k, v in s6.items() if k.startswith("g8")) < 6e-6
    assert s6["g9_fs6_nb_a030_n250"]["avx2_max_rel_dev"] < 5e-6 < 1e-5 < s6["g9_fs6_nb_a090_n250"]["avx2_max_rel_dev"]
    quiet = [v["avx2_max_rel_dev"] for k, v in s.items() if k.startswith("g8") and k.endswith("_n010")]
    loud = [v["avx2_max_rel_dev"] for k, v in s.items() if k.startswith("g8") and k.endswith("_n250")]
    assert min(quiet) > 1.5e-5 and max(loud) < 5e-6
    # channels NEXT to the carrier (g9: it stays in band for them until the last stage): the two builds are 5e-6 apart under a
    # carrier of 0.3 over a floor of 0.25 -- the synthetic input of the parity tests -- and more than 1e-5 under one of 0.9
    assert 3e-6 < s["g9_nb_a030_n250"]["avx2_max_rel_dev"] < 1e-5 < s["g9_nb_a090_n250"]["avx2_max_rel_dev"]


def test_reference_builds_agree_within_tolerance():
    """Scale of the problem: the reference's AVX2 and portable builds differ by < 1e-5 (but not 0)."""
    import json
    s = json.load(open(os.path.join(GOLD, "SUMMARY.json")))
    worst = max(v["avx2_max_rel_dev"] for v in s.values())
    assert 0 < worst < 1e-5


def test_synth_numpy_equals_c(oracle):
    ch = chains.chain("10")
    tabs = [oracle.make_translator(ch.N, ch.z, o) for o in (37, -415)]
    for seed, start, n in ((1234, 0, 5000), (7, 319990, 4099), (2 ** 40 + 3, 10 ** 9, 1201)):
        a = oracle.synth_iq(seed, start, n, 0.25, tabs, [0.3, 0.05], ch.N, 10000)
        b = synth.synth_iq(seed, start, n, 0.25, tabs, [0.3, 0.05], ch.N, 10000)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        a = oracle.synth_iq(seed, start, n, 0.25)
        b = synth.synth_iq(seed, start, n, 0.25)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    x = synth.synth_iq(1, 0, 100000)
    assert np.all(np.abs(x.real) <= 0.25) and np.all(np.abs(x.imag) <= 0.25)
    assert abs(np.mean(x.real)) < 2e-3


def _float64_chain(ch, table, x):
    """Independent float64 restatement: mix, then for each stage full convolution + decimation phase.

    Output k of a stage (0-based) has input n = m*(k+1)-1 as its newest sample (SURVEY 8a 'state')."""
    y = x.astype(np.complex128)
    if table is not None:
        n = np.arange(len(y))
        y = y * table.astype(np.complex128)[n % len(table)]
    for s in ch.stages:
        h = s.h.astype(np.float64)
        full = np.convolve(y, h)          # full[n] = sum_i h[i] y[n-i]; symmetric taps so order is moot
        idx = np.arange(s.m - 1, len(y), s.m)
        y = full[idx]
    return y


@pytest.mark.parametrize("rate,off", [("10", 0), ("10", 37), ("10", -415), ("2.4", 37), ("6", 7), ("1.44", -11)])
def test_oracle_matches_float64_convolution(oracle, rate, off):
    """Known-answer: the streaming folded-FIR cascade equals plain convolution + decimation."""
    ch = chains.chain(rate)
    n = ch.m_total * 40 + 123
    x = synth.synth_iq(99, 0, n, 0.5)
    table = oracle.make_translator(ch.N, ch.z, off) if off else None
    got = oracle.msd(table, ch.stage_tuples()).decimate(x)
    ref = _float64_chain(ch, table, x)
    assert len(got) == n // ch.m_total == len(ref)
    assert rel_err(got, ref) < 5e-6


def test_dc_gain_and_impulse_known_answers(oracle):
    ch = chains.chain("10")
    # every stage is normalised to sum(h)=1 (reference octave/sincflt.m:13-15) -> DC passes with gain 1
    for s in ch.stages:
        assert abs(float(np.sum(s.h.astype(np.float64))) - 1.0) < 1e-6
    dc = np.full(ch.block_len, 0.5 - 0.25j, dtype=np.complex64)
    y = oracle.msd(None, ch.stage_tuples()).decimate(dc)
    assert len(y) == 512
    assert np.max(np.abs(y[40:] - (0.5 - 0.25j))) < 1e-6
    # receptive field: an impulse at n influences outputs only while it is inside the 8856-sample window
    imp = np.zeros(ch.block_len, dtype=np.complex64)
    imp[1000] = 1.0
    y = oracle.msd(None, ch.stage_tuples()).decimate(imp)
    nz = np.nonzero(np.abs(y) > 0)[0]
    first_k = 1000 // 625          # first output whose newest sample 625(k+1)-1 >= 1000
    assert nz.min() >= first_k
    assert nz.max() <= (1000 + 8856) // 625


def test_ragged_equals_one_shot(oracle):
    """decimate() accepts any in_len; cutting the stream anywhere must not change the output (msd.hpp:73-75)."""
    ch = chains.chain("10")
    t = oracle.make_translator(ch.N, ch.z, -93)
    x = synth.synth_iq(3, 0, 50000, 0.3)
    one = oracle.msd(t, ch.stage_tuples()).decimate(x)
    m = oracle.msd(t, ch.stage_tuples())
    parts, p = [], 0
    for n in (0, 1, 624, 625, 626, 9999, 1, 0, 50000):
        n = min(n, len(x) - p)
        parts.append(m.decimate(x[p:p + n]))
        p += n
    got = np.concatenate(parts)
    assert np.array_equal(got.view(np.uint32), one.view(np.uint32))


def test_bank_equals_single_channel_loop(oracle):
    ch = chains.chain("10")
    offs = [0, 5, -7, 300]
    tabs = [oracle.make_translator(ch.N, ch.z, o) if o else None for o in offs]
    x = [synth.synth_iq(11, b * 20000, 20000, 0.25) for b in range(2)]
    bank = oracle.bank(tabs, ch.stage_tuples())
    bank_mt = oracle.bank(tabs, ch.stage_tuples())
    singles = [(oracle.msd(t, ch.stage_tuples()), oracle.agc_pair(False)) for t in tabs]
    for blk in x:
        iq, au = bank.process(blk, threads=1)
        iq2, au2 = bank_mt.process(blk, threads=3)
        assert np.array_equal(iq.view(np.uint32), iq2.view(np.uint32))
        assert np.array_equal(au.view(np.uint32), au2.view(np.uint32))
        for c, (m, a) in enumerate(singles):
            y = m.decimate(blk)
            assert np.array_equal(y.view(np.uint32), iq[c].view(np.uint32))
            aud = a.adjust_lf(oracle.demod_am(a.adjust(y)))
            assert np.array_equal(aud.view(np.uint32), au[c].view(np.uint32))


def test_ftfir_relation_to_mixer_path(oracle):
    """ftfir output = 2 * exp(+j*2*pi*off*(m1-1)*z/N) * mixer-path output."""
    ch = chains.chain("2.4")
    off = 37
    t = oracle.make_translator(ch.N, ch.z, off)
    x = synth.synth_iq(1234, 0, ch.block_len, 0.25, [t], [0.3], ch.N, 2400)
    a = oracle.msd(t, ch.stage_tuples(), False).decimate(x)
    b = oracle.msd(t, ch.stage_tuples(), True).decimate(x)
    ratio = b[100:] / a[100:]
    pred = 2.0 * np.exp(2j * np.pi * off * (ch.stages[0].m - 1) * ch.z / ch.N)
    assert np.max(np.abs(ratio - pred)) < 2e-3


def test_squelch_restatement_vs_numpy_fft(oracle):
    """The squelch restatement (sdrx.cpp:533-573; direct DFT in place of FFTW -> parity unpinned) against an
    independent numpy.fft evaluation of the same formula, plus the state machine with its 3 dB hysteresis."""
    from oracle import ffi as offi
    ch = chains.chain("10")
    pb, ru, rd = chains.squelch_tables()
    win = offi.hamming_window(oracle)
    assert np.array_equal(win.view(np.uint32), chains.hamming_window().view(np.uint32))
    assert abs(float(ru[100]) - (1 - np.exp(-1.0))) < 1e-6 and abs(float(rd[100]) - np.exp(-1.0)) < 1e-6
    t = oracle.make_translator(ch.N, ch.z, 37)
    sc = offi.SquelchChannel(oracle, 9.0, win, pb, ru, rd)
    msd = oracle.msd(t, ch.stage_tuples())
    states = []
    for b, amp in enumerate([0.0, 0.2, 0.2, 0.0, 0.0]):
        x = synth.synth_iq(5, b * ch.block_len, ch.block_len, 0.25, [t] if amp else [], [amp] if amp else [], ch.N, 10000)
        y = msd.decimate(x)
        audio, snr, st = sc.chunk(y)
        P = np.abs(np.fft.fft(y.astype(np.complex128) * win)) ** 2
        sig = (P[3:91].sum() + P[512 - 90:512 - 2].sum()) / 176
        w2 = pb.astype(np.float64) ** 2
        hi = (P[112:157] * w2[112:157]).sum() / 45
        lo = (P[512 - 156:512 - 111] * w2[512 - 156:512 - 111]).sum() / 45
        assert abs(snr - 10 * np.log10(sig / ((hi + lo) / 2))) < 1e-3
        states.append((st, bool(np.any(audio != 0))))
    # closed/silent, opens after the first carrier chunk (audio still silent in it), open with audio,
    # closes after the first noise chunk (that chunk still has audio), then ramp-down chunk, i.e. audio once more
    assert states == [(0, False), (1, False), (1, True), (0, True), (0, True)]


# ---- stereo mix-down restatements (sdrx.cpp:469-497, 646-656, 1024-1044; fir.hpp:131-183) ----
def test_fir2_bit_exact_vs_reference_fixture(oracle):
    """FIR2 restated == the reference's class (portable build) on the committed stereo vector, ragged calls, in place."""
    z = np.load(os.path.join(GOLD, "aux", "fir2_stereo.npz"))
    f2 = oracle.fir2(chains.audio_bpf_taps(), float(z["gain_db"]))
    cuts = [int(c) for c in z["cuts"]]
    y = np.concatenate([f2.filter(z["x"][a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(y.view(np.uint32), z["y"].view(np.uint32))
    assert float(np.abs(z["y_avx2"] - z["y"]).max()) < 1e-6           # the reference's own two builds
    # FIR2 is FIR on each slot (what the device runs on its two planar rows)
    for k in (0, 1):
        f1 = oracle.fir(chains.audio_bpf_taps(), float(z["gain_db"]))
        assert np.array_equal(f1.filter(z["x"][:, k]).view(np.uint32), z["y"][:, k].view(np.uint32))


def test_audio_pos_table(oracle):
    """get_audio_pos by hand from sdrx.cpp:1024-1044: five positions spread symmetrically, centre for the odd middle."""
    assert [oracle.audio_pos(c, 1) for c in range(1)] == [0]
    assert [oracle.audio_pos(c, 2) for c in range(2)] == [-2, 2]
    assert [oracle.audio_pos(c, 3) for c in range(3)] == [-2, 0, 2]
    assert [oracle.audio_pos(c, 5) for c in range(5)] == [-2, -1, 0, 1, 2]
    assert [oracle.audio_pos(c, 7) for c in range(7)] == [-2, -2, -1, 0, 1, 2, 2]
    assert [oracle.audio_pos(c, 10) for c in range(10)] == [-2, -2, -1, -1, 0, 0, 1, 1, 2, 2]
    assert oracle.audio_pos(12, 10) == 0                                 # out of range: pos stays 0
    for n in (4, 9, 64, 1001):
        p = [oracle.audio_pos(c, n) for c in range(n)]
        assert p == [-q for q in reversed(p)] and min(p) == -2 and max(p) == 2


def test_pan_mix_and_s16_known_answers(oracle):
    f = np.float32
    au = np.array([[1.0, 0.5], [2.0, -1.0], [0.25, 0.0], [4.0, 1.0], [8.0, 3.0]], dtype=np.float32)
    pos = [-2, -1, 0, 1, 2]
    w0 = [f(0.8), f(0.6), f(0.5), f(0.4), f(0.2)]
    w1 = [f(0.2), f(0.4), f(0.5), f(0.6), f(0.8)]
    want = np.zeros((2, 2), np.float32)
    for c in range(5):                                   # serial float32 accumulation in channel order
        for i in range(2):
            want[i, 0] = f(want[i, 0] + f(w0[c] * au[c, i]))
            want[i, 1] = f(want[i, 1] + f(w1[c] * au[c, i]))
    got = oracle.pan_mix(au, pos)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(oracle.pan_mix(au[:1], [7]), np.array([[0.5, 0.5], [0.25, 0.25]], np.float32))   # default: centre
    x = np.array([1.5, -1.5, 1.0, -1.0, 0.5, -0.5, 0.99999, 3.0e-5, -3.0e-5, 0.0], dtype=np.float32)
    assert oracle.to_s16(x).tolist() == [32767, -32767, 32767, -32767, 16383, -16383, 32766, 0, 0, 0]


def test_fm_demod_restatement_vs_float64(oracle):
    """Demod FM branch (sdrx.cpp:181-194): phase step of consecutive normalised samples; first output is
    atan2(0, 0) = 0 with prev_sample = (0, 0) (sdrx.cpp:175); the state carries across calls."""
    rng = np.random.default_rng(0)
    n = 2000
    ph = np.cumsum(rng.uniform(-0.5, 0.5, n))
    x = (rng.uniform(0.5, 2.0, n) * np.exp(1j * ph)).astype(np.complex64)
    prev = np.zeros(2, np.float32)
    y = np.concatenate([oracle.demod_fm(prev, x[:777]), oracle.demod_fm(prev, x[777:])])
    want = np.angle(x[1:].astype(np.complex128) * np.conj(x[:-1].astype(np.complex128)))
    assert y[0] == 0.0
    assert float(np.abs(y[1:] - want).max()) < 5e-7
    xn = x[-1].astype(np.complex128) / abs(x[-1].astype(np.complex128))
    assert abs(prev[0] - xn.real) < 1e-6 and abs(prev[1] - xn.imag) < 1e-6


def test_active_lf_agc_is_ill_conditioned_in_the_reference_itself(oracle):
    """Why long-stream AUDIO parity with --lf-agc cannot be held to 1e-5 by anybody: two builds of the same code that
    differ only in rounding (FMA contraction, vector width) -- the reference's own portable and AVX2 builds where they
    exist here, and always the oracle's plain and -O3 -march builds -- agree on the decimated IQ to ~1e-6 for ever, and on
    the audio to ~1e-6 for ever WITHOUT the LF AGC (sdrx's default), but drift apart on noise-only channels once the LF
    AGC is active (attack 1.0 on a power estimate of the envelope: a feedback loop that amplifies rounding).  So
    tests/test_gpu_long_stream.py holds the audio to 1e-5 over 200 blocks with the LF AGC inactive and, with it active,
    checks each block starting from the state the GPU was in (sdrxgpu_bank_get_agc_state)."""
    from oracle import ffi
    ch = chains.chain("10")
    offs = [0, 599, 37, -77]
    tabs = [oracle.make_translator(ch.N, ch.z, o) if o else None for o in offs]
    car = [tabs[1]]
    L = ch.block_len
    pairs = [("oracle -O2 vs -O3 -march", oracle, ffi.load("oracle_fast"), True)]
    if ffi.have_ref():
        pairs.append(("reference portable vs AVX2", ffi.load("ref_portable"), ffi.load("ref_avx2"), False))
    for label, A, B, is_oracle in pairs:
        for lf in (False, True):
            mk = (lambda lib: lib.bank(tabs, ch.stage_tuples(), use_lf_agc=lf)) if is_oracle else (lambda lib: lib.bank(tabs, ch.stage_tuples(), False, lf))
            a, b = mk(A), mk(B)
            worst_iq = worst_au = 0.0
            for k in range(16):
                x = synth.synth_iq(2024, k * L, L, 0.25, car, [0.3], ch.N, 10000)
                (ia, aa), (ib, ab) = a.process(x, threads=4), b.process(x, threads=4)
                worst_iq = max(worst_iq, rel_err_rows(ib, ia))
                worst_au = max(worst_au, rel_err_rows(ab, aa))
            print(f"{label}, lf_agc={lf}: 16 blocks, iq {worst_iq:.2e}, audio {worst_au:.2e}")
            assert worst_iq <= RTOL
            if lf:
                assert worst_au > 10 * RTOL      # the two builds of ONE code base are > 1e-4 apart
            else:
                assert worst_au <= RTOL
