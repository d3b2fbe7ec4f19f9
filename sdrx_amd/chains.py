"""Sample-rate -> (N, z, decimator stage list), as sdrx sets them up.

Mirrors the ``switch (settings.rate)`` of reference ``src/sdrx.cpp:1588-1672``: for each IQ
sample rate the translator length ``N``, the offset multiplier ``z`` and the list of
``MSD::Stage{m, h}``.  The tap values themselves are data (``sdrx_amd/data/taps.json``,
float32 bit patterns dumped from the reference's filter headers by
``oracle/extract_taps.cpp``); the library never embeds them -- they are handed to
``sdrxgpu_bank_create`` by the caller exactly as sdrx.cpp hands them to ``MSD``.
"""
import json
import os
from dataclasses import dataclass
from typing import List

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "taps.json")

CH_IQ_BUF_SIZE = 512          # reference src/sdrx.cpp:129: outputs per channel per block
CH_IQ_SAMPLING_FQ = 16000     # reference src/sdrx.cpp:128


@dataclass
class Stage:
    m: int
    h: np.ndarray  # float32 taps, odd length, symmetric


@dataclass
class Chain:
    rate: str
    fs: int
    N: int
    z: int
    stages: List[Stage]

    @property
    def m_total(self) -> int:
        m = 1
        for s in self.stages:
            m *= s.m
        return m

    @property
    def block_len(self) -> int:
        """Device block = 32 ms = fs*4/125 samples (reference src/airspy_dev.cpp:110-114)."""
        return self.fs * 4 // 125

    def stage_tuples(self):
        return [(s.m, s.h) for s in self.stages]


def _hex_to_f32(hex_list):
    return np.array([int(h, 16) for h in hex_list], dtype=np.uint32).view(np.float32)


_cache = None


def _load():
    global _cache
    if _cache is None:
        with open(_DATA) as f:
            _cache = json.load(f)
    return _cache


def chain(rate: str) -> Chain:
    """rate: '10', '2.4', '6', '0.96', '1.2', '1.44', '1.6', '1.92', '2.56' (MS/s)."""
    d = _load()["chains"][rate]
    return Chain(rate=rate, fs=d["fs"], N=d["N"], z=d["z"],
                 stages=[Stage(m=s["m"], h=_hex_to_f32(s["taps"])) for s in d["stages"]])


def audio_bpf_taps() -> np.ndarray:
    """197-tap 250-3300 Hz audio band-pass (reference src/coeffs.hpp:1812, used at sdrx.cpp:880)."""
    return _hex_to_f32(_load()["audio_bpf"]["taps"])


def squelch_tables():
    """(passband_shape[512], ramp_up[512], ramp_down[512]) -- reference src/coeffs.hpp:2018, :2541, :3064."""
    d = _load()["squelch"]
    return _hex_to_f32(d["passband_shape"]), _hex_to_f32(d["ramp_up"]), _hex_to_f32(d["ramp_down"])


def hamming_window(n: int = 512) -> np.ndarray:
    """FFT window of the squelch, reference src/sdrx.cpp:896-900: 0.54f - 0.46f*cos((2.0f*M_PI*n)/FFT_SIZE),
    cos evaluated in double (M_PI is a double), result narrowed to float."""
    k = np.arange(n, dtype=np.float64)
    return (np.float32(0.54) - (np.float32(0.46).astype(np.float64) * np.cos((2.0 * np.pi * k) / n))).astype(np.float32)


def offsets_for_config(name: str):
    """Channel offset sets (in 8.33 kHz steps from the tuner) of the BASELINE.json configs."""
    if name == "c1":       # 1 channel, 2.4 MS/s
        return [37]
    if name == "c2":       # 64 channels, 8.33 kHz grid, tuner centre excluded
        return [o for o in range(-32, 33) if o != 0]
    if name == "c3":       # 1024 channels, mixed 8.33/25 kHz (offsets = 0 mod 3 are the 25 kHz names)
        return list(range(-512, 512))
    raise ValueError(name)


def offsets_cycled(n_channels: int, N: int = 1200):
    """n channel *instances* cycling through all N distinct offsets -N/2 .. N/2-1 (never deduplicated)."""
    return [((i % N) - N // 2) for i in range(n_channels)]
