"""Synthetic IQ generator (numpy twin of ``sdrx_oracle_synth_iq``).

Values lie in [-1, 1) like Airspy FLOAT32_IQ (reference src/airspy_dev.cpp:314).  Uniform
noise comes from a counter-based SplitMix64 (integer-exact, so C and numpy agree bit for
bit); optional AM carriers sit exactly on channel centres because they are built from the
channels' own translator tables (``conj(t[n mod N])``), which keeps libm out of the
generator.  All float32 operations are single, unfused, elementwise operations in a fixed
order, so the result is identical to the C twin compiled with ``-ffp-contract=off``.
"""
import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64_at(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_iq(seed: int, n_start: int, n: int, noise_amp: float = 0.25, carriers=(), carrier_amp=(),
             N: int = 1, tri_period: int = 10000, tri_depth: float = 0.5) -> np.ndarray:
    """Return ``n`` complex64 samples with absolute indices ``n_start .. n_start+n-1``.

    ``tri_depth``: modulation depth of the carriers' triangle AM (0.5 = the C twin's fixed value; the pair-stress
    fixtures of tests/golden/pairs use shallower ones so that a carrier of mean amplitude A peaks at <= 1)."""
    abs_n = np.arange(n_start, n_start + n, dtype=np.uint64)
    scale = np.float32(1.0 / 8388608.0)
    amp = np.float32(noise_amp)
    vr = (_splitmix64_at(seed, np.uint64(2) * abs_n) >> np.uint64(40)).astype(np.int64) - 8388608
    vi = (_splitmix64_at(seed, np.uint64(2) * abs_n + np.uint64(1)) >> np.uint64(40)).astype(np.int64) - 8388608
    re = (vr.astype(np.float32) * scale) * amp
    im = (vi.astype(np.float32) * scale) * amp
    if len(carriers):
        P = int(tri_period)
        v = (abs_n % np.uint64(P)).astype(np.int64)
        d = np.abs(2 * v - P)
        tri_step = np.float32(2.0) / np.float32(P)
        tri = d.astype(np.float32) * tri_step - np.float32(1.0)
        mod = np.float32(1.0) + np.float32(tri_depth) * tri
        tn = (abs_n % np.uint64(N)).astype(np.int64)
        for t, a in zip(carriers, carrier_amp):
            t = np.asarray(t, dtype=np.complex64)
            g = np.float32(a) * mod
            re = re + g * t.real[tn]
            im = im - g * t.imag[tn]
    out = np.empty(n, dtype=np.complex64)
    out.real = re
    out.imag = im
    return out
