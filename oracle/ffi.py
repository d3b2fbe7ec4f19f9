"""ctypes bindings for the parity oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  It binds two families of shared objects that share one C signature set:

* ``oracle/liboracle.so`` / ``liboracle_fast.so`` -- our C restatement (``sdrx_oracle_*``);
* ``oracle/_ref/libsdrx_ref_{portable,avx2}.so`` -- the reference's own headers compiled by
  ``oracle/Makefile`` (``sdrx_ref_*``), present only where ``make ref`` has been run.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = os.environ.get("SDRX_REF", "/root/reference")

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def _fp(a):
    return a.ctypes.data_as(_f32p)


def build(ref=None):
    """Compile liboracle*.so (always) and oracle/_ref (only where the reference is mounted)."""
    targets = ["oracle"]
    if ref is None:
        ref = os.path.isdir(os.path.join(REF_SRC, "src"))
    if ref:
        targets.append("ref")
    subprocess.check_call(["make", "-s", "-C", HERE] + targets)


def have_ref():
    return os.path.exists(os.path.join(HERE, "_ref", "libsdrx_ref_portable.so"))


class _Lib:
    """One loaded implementation (oracle restatement or compiled reference)."""

    def __init__(self, path, prefix):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path)
        p = prefix
        L = self.lib
        self._fn = {}
        sig = {
            "make_translator": (None, [C.c_int, C.c_int, C.c_int, _f32p]),
            "msd_create": (C.c_void_p, [_f32p, C.c_size_t, C.c_int, _u32p, _u32p, C.POINTER(_f32p), C.c_int]),
            "msd_clone": (C.c_void_p, [C.c_void_p]),
            "msd_m": (C.c_size_t, [C.c_void_p]),
            "msd_decimate": (None, [C.c_void_p, _f32p, C.c_size_t, _f32p, C.POINTER(C.c_size_t)]),
            "msd_destroy": (None, [C.c_void_p]),
            "demod_am": (None, [_f32p, C.c_size_t, _f32p]),
            "fir_create": (C.c_void_p, [_f32p, C.c_uint]),
            "fir_set_gain": (None, [C.c_void_p, C.c_float]),
            "fir_filter": (None, [C.c_void_p, _f32p, C.c_uint, _f32p]),
            "fir_destroy": (None, [C.c_void_p]),
            "fir2_create": (C.c_void_p, [_f32p, C.c_uint]),
            "fir2_set_gain": (None, [C.c_void_p, C.c_float]),
            "fir2_filter": (None, [C.c_void_p, _f32p, C.c_uint, _f32p]),
            "fir2_destroy": (None, [C.c_void_p]),
            "bank_create": (C.c_void_p, [C.c_int, C.POINTER(_f32p), C.POINTER(C.c_size_t), C.c_int, _u32p, _u32p,
                                         C.POINTER(_f32p), C.c_int, C.c_int]),
            "bank_process": (None, [C.c_void_p, _f32p, C.c_size_t, _f32p, _f32p, C.c_size_t,
                                    C.POINTER(C.c_size_t), C.c_int]),
            "bank_destroy": (None, [C.c_void_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, f"{p}_{name}")
            fn.restype = res
            fn.argtypes = args
            self._fn[name] = fn
        if prefix == "sdrx_oracle":
            L.sdrx_oracle_synth_iq.restype = None
            L.sdrx_oracle_synth_iq.argtypes = [C.c_uint64, C.c_uint64, C.c_size_t, C.c_float, C.c_int,
                                               C.POINTER(_f32p), C.c_int, _f32p, C.c_uint32, _f32p]

    # ---- helpers -------------------------------------------------------
    @staticmethod
    def _stage_args(stages):
        """stages: list of (m, taps ndarray float32)."""
        n = len(stages)
        m = (C.c_uint32 * n)(*[int(s[0]) for s in stages])
        nt = (C.c_uint32 * n)(*[len(s[1]) for s in stages])
        keep = [np.ascontiguousarray(s[1], dtype=np.float32) for s in stages]
        tp = (_f32p * n)(*[_fp(k) for k in keep])
        return n, m, nt, tp, keep

    def make_translator(self, N, z, off):
        out = np.empty(2 * N, dtype=np.float32)
        self._fn["make_translator"](N, z, off, _fp(out))
        return out.view(np.complex64)

    # ---- MSD -----------------------------------------------------------
    def msd(self, translator, stages, use_ftfir=False):
        return _MSD(self, translator, stages, use_ftfir)

    def demod_am(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        out = np.empty(iq.shape[0], dtype=np.float32)
        self._fn["demod_am"](_fp(iq.view(np.float32)), iq.shape[0], _fp(out))
        return out

    def demod_fm(self, prev, iq):
        """Demod FM branch; prev = float32[2] state (updated in place)."""
        assert self.prefix == "sdrx_oracle"
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        out = np.empty(iq.shape[0], dtype=np.float32)
        self.lib.sdrx_oracle_demod_fm.argtypes = [_f32p, _f32p, C.c_size_t, _f32p]
        self.lib.sdrx_oracle_demod_fm.restype = None
        self.lib.sdrx_oracle_demod_fm(_fp(prev), _fp(iq.view(np.float32)), iq.shape[0], _fp(out))
        return out

    def fir(self, taps, gain_db=None):
        return _FIR(self, taps, gain_db)

    def fir2(self, taps, gain_db=None):
        return _FIR(self, taps, gain_db, stereo=True)

    # ---- channel names -> offsets (restatement only: static code of sdrx.cpp) -----
    def parse_fq(self, s, aeronautical=True):
        assert self.prefix == "sdrx_oracle"
        f = self.lib.sdrx_oracle_parse_fq
        f.restype, f.argtypes = C.c_uint32, [C.c_char_p, C.c_int]
        return int(f(s.encode(), int(aeronautical)))

    def channel_to_offset(self, name, tuner_fq):
        assert self.prefix == "sdrx_oracle"
        f = self.lib.sdrx_oracle_channel_to_offset
        f.restype, f.argtypes = C.c_int, [C.c_char_p, C.c_int32]
        return int(f(name.encode(), tuner_fq))

    def tuner_fq(self, names):
        assert self.prefix == "sdrx_oracle"
        f = self.lib.sdrx_oracle_tuner_fq
        f.restype, f.argtypes = C.c_uint32, [C.POINTER(C.c_char_p), C.c_size_t]
        return int(f((C.c_char_p * len(names))(*[n.encode() for n in names]), len(names)))

    def verify_bandwidth(self, names, sample_rate_hz):
        assert self.prefix == "sdrx_oracle"
        f = self.lib.sdrx_oracle_verify_bandwidth
        f.restype, f.argtypes = C.c_int, [C.POINTER(C.c_char_p), C.c_size_t, C.c_uint32]
        return bool(f((C.c_char_p * len(names))(*[n.encode() for n in names]), len(names), sample_rate_hz))

    # ---- stereo mix-down (restatement only: static code of sdrx.cpp) -----
    def audio_pos(self, channel_no, num_channels):
        assert self.prefix == "sdrx_oracle"
        return int(self.lib.sdrx_oracle_audio_pos(C.c_uint(channel_no), C.c_uint(num_channels)))

    def pan_mix(self, audio, pos):
        """audio [n_ch, n] float32, pos [n_ch] int -> interleaved stereo [n, 2] float32."""
        assert self.prefix == "sdrx_oracle"
        a = np.ascontiguousarray(audio, dtype=np.float32)
        p = np.ascontiguousarray(pos, dtype=np.int32)
        out = np.empty((a.shape[1], 2), dtype=np.float32)
        self.lib.sdrx_oracle_pan_mix(_fp(a), C.c_size_t(a.shape[1]), p.ctypes.data_as(C.POINTER(C.c_int)),
                                     C.c_int(a.shape[0]), C.c_size_t(a.shape[1]), _fp(out))
        return out

    def to_s16(self, x):
        assert self.prefix == "sdrx_oracle"
        a = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty(a.shape, dtype=np.int16)
        self.lib.sdrx_oracle_to_s16(_fp(a), C.c_size_t(a.size), out.ctypes.data_as(C.POINTER(C.c_int16)))
        return out

    def bank(self, translators, stages, use_ftfir=False, use_lf_agc=False):
        return _Bank(self, translators, stages, use_ftfir, use_lf_agc)

    def agc_pair(self, use_lf_agc=False):
        return _AGCPair(self, use_lf_agc)

    def synth_iq(self, seed, n_start, n, noise_amp=0.25, carriers=(), carrier_amp=(), N=1, tri_period=10000):
        assert self.prefix == "sdrx_oracle"
        out = np.empty(2 * n, dtype=np.float32)
        nc = len(carriers)
        keep = [np.ascontiguousarray(t, dtype=np.complex64) for t in carriers]
        tp = (_f32p * max(nc, 1))(*[_fp(k.view(np.float32)) for k in keep]) if nc else None
        amp = np.ascontiguousarray(carrier_amp, dtype=np.float32)
        self.lib.sdrx_oracle_synth_iq(seed, n_start, n, noise_amp, nc, tp, N, _fp(amp) if nc else None,
                                      tri_period, _fp(out))
        return out.view(np.complex64)


class _MSD:
    def __init__(self, lib, translator, stages, use_ftfir):
        self.lib = lib
        n, m, nt, tp, self._keep = lib._stage_args(stages)
        if translator is None or len(translator) == 0:
            tptr, tn = None, 0
        else:
            self._t = np.ascontiguousarray(translator, dtype=np.complex64)
            tptr, tn = _fp(self._t.view(np.float32)), len(self._t)
        self.h = lib._fn["msd_create"](tptr, tn, n, m, nt, tp, int(use_ftfir))
        assert self.h
        self.m = lib._fn["msd_m"](self.h)

    def decimate(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        out = np.empty(iq.shape[0] // self.m + 2, dtype=np.complex64)
        n_out = C.c_size_t(0)
        self.lib._fn["msd_decimate"](self.h, _fp(iq.view(np.float32)), iq.shape[0],
                                     _fp(out.view(np.float32)), C.byref(n_out))
        return out[: n_out.value].copy()

    def __del__(self):
        if getattr(self, "h", None):
            self.lib._fn["msd_destroy"](self.h)
            self.h = None


class _FIR:
    """FIR (mono) or FIR2 (interleaved stereo, x is [n, 2] or flat with 2 floats per frame)."""

    def __init__(self, lib, taps, gain_db, stereo=False):
        self.lib = lib
        self.k = "fir2" if stereo else "fir"
        t = np.ascontiguousarray(taps, dtype=np.float32)
        self.h = lib._fn[self.k + "_create"](_fp(t), len(t))
        if gain_db is not None:
            lib._fn[self.k + "_set_gain"](self.h, gain_db)

    def filter(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        self.lib._fn[self.k + "_filter"](self.h, _fp(x), x.size, _fp(out))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.lib._fn[self.k + "_destroy"](self.h)
            self.h = None


class _AGCStruct(C.Structure):
    _fields_ = [(k, C.c_float) for k in
                ("attack", "decay", "reference", "min_gain", "max_gain", "alpha", "gain", "power")] + [("active", C.c_int)]


class _AGCPair:
    """AGC + LfAGC configured as sdrx does per channel (reference src/sdrx.cpp:1702-1710)."""

    def __init__(self, lib, use_lf_agc):
        self.lib = lib
        L = lib.lib
        self.is_ref = lib.prefix == "sdrx_ref"
        if self.is_ref:
            L.sdrx_ref_agc_create_sdrx.restype = C.c_void_p
            L.sdrx_ref_agc_create_sdrx.argtypes = [C.c_int]
            L.sdrx_ref_agc_adjust.argtypes = [C.c_void_p, _f32p, C.c_size_t, _f32p]
            L.sdrx_ref_lfagc_adjust.argtypes = [C.c_void_p, _f32p, C.c_size_t, _f32p]
            L.sdrx_ref_agc_gain.restype = C.c_float
            L.sdrx_ref_agc_gain.argtypes = [C.c_void_p]
            L.sdrx_ref_agc_destroy.argtypes = [C.c_void_p]
            self.h = L.sdrx_ref_agc_create_sdrx(int(use_lf_agc))
        else:
            self.agc = _AGCStruct()
            self.lf = _AGCStruct()
            L.sdrx_oracle_agc_init_sdrx.argtypes = [C.POINTER(_AGCStruct), C.POINTER(_AGCStruct), C.c_int]
            L.sdrx_oracle_agc_adjust.argtypes = [C.POINTER(_AGCStruct), _f32p, C.c_size_t, _f32p]
            L.sdrx_oracle_lfagc_adjust.argtypes = [C.POINTER(_AGCStruct), _f32p, C.c_size_t, _f32p]
            L.sdrx_oracle_agc_init_sdrx(C.byref(self.agc), C.byref(self.lf), int(use_lf_agc))

    def adjust(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        out = np.empty_like(iq)
        if self.is_ref:
            self.lib.lib.sdrx_ref_agc_adjust(self.h, _fp(iq.view(np.float32)), len(iq), _fp(out.view(np.float32)))
        else:
            self.lib.lib.sdrx_oracle_agc_adjust(C.byref(self.agc), _fp(iq.view(np.float32)), len(iq),
                                                _fp(out.view(np.float32)))
        return out

    def adjust_lf(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        if self.is_ref:
            self.lib.lib.sdrx_ref_lfagc_adjust(self.h, _fp(x), len(x), _fp(out))
        else:
            self.lib.lib.sdrx_oracle_lfagc_adjust(C.byref(self.lf), _fp(x), len(x), _fp(out))
        return out

    def __del__(self):
        if self.is_ref and getattr(self, "h", None):
            self.lib.lib.sdrx_ref_agc_destroy(self.h)
            self.h = None


class _Bank:
    """n channels: MSD -> AGC -> |.| -> LfAGC, the loop of sdrx.cpp:342-345 + 457-463."""

    def __init__(self, lib, translators, stages, use_ftfir, use_lf_agc):
        self.lib = lib
        n_ch = len(translators)
        n, m, nt, tp, self._keep = lib._stage_args(stages)
        self._t = [np.ascontiguousarray(t, dtype=np.complex64) if t is not None and len(t) else None for t in translators]
        tptrs = (_f32p * n_ch)(*[_fp(t.view(np.float32)) if t is not None else None for t in self._t])
        tn = (C.c_size_t * n_ch)(*[len(t) if t is not None else 0 for t in self._t])
        self.h = lib._fn["bank_create"](n_ch, tptrs, tn, n, m, nt, tp, int(use_ftfir), int(use_lf_agc))
        assert self.h
        self.n_ch = n_ch
        self.m = 1
        for s in stages:
            self.m *= int(s[0])

    def process(self, iq, threads=1, want_iq=True, want_audio=True, out_stride=None):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        stride = out_stride or (iq.shape[0] // self.m + 1)
        iq_out = np.zeros((self.n_ch, stride), dtype=np.complex64) if want_iq else None
        audio = np.zeros((self.n_ch, stride), dtype=np.float32) if want_audio else None
        n_out = C.c_size_t(0)
        self.lib._fn["bank_process"](self.h, _fp(iq.view(np.float32)), iq.shape[0],
                                     _fp(iq_out.view(np.float32)) if want_iq else None,
                                     _fp(audio) if want_audio else None, stride, C.byref(n_out), threads)
        k = n_out.value
        return (iq_out[:, :k] if want_iq else None), (audio[:, :k] if want_audio else None)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib._fn["bank_destroy"](self.h)
            self.h = None


class _ChannelStruct(C.Structure):
    _fields_ = [("agc", _AGCStruct), ("agc_lf", _AGCStruct), ("sql_level", C.c_float),
                ("sql_state", C.c_int), ("sql_state_prev", C.c_int), ("snr", C.c_float),
                ("fm", C.c_int), ("fm_prev", C.c_float * 2)]


class SquelchChannel:
    """One channel of alsa_write_cb (sdrx.cpp:455-573): squelch-gated audio + squelch update per 512-sample chunk."""

    def __init__(self, lib, sql_level, window, passband_shape, ramp_up, ramp_down, use_lf_agc=False, fm=False):
        assert lib.prefix == "sdrx_oracle"
        self.L = lib.lib
        self.c = _ChannelStruct()
        self._fm = fm
        self.L.sdrx_oracle_channel_init.argtypes = [C.POINTER(_ChannelStruct), C.c_float, C.c_int]
        self.L.sdrx_oracle_channel_chunk.argtypes = [C.POINTER(_ChannelStruct), _f32p, _f32p, _f32p, _f32p, _f32p, _f32p]
        self.L.sdrx_oracle_channel_init(C.byref(self.c), sql_level, int(use_lf_agc))
        self.c.fm = int(fm)
        self.tabs = [np.ascontiguousarray(t, dtype=np.float32) for t in (window, passband_shape, ramp_up, ramp_down)]

    def chunk(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        assert iq.shape[0] == 512
        audio = np.empty(512, dtype=np.float32)
        self.L.sdrx_oracle_channel_chunk(C.byref(self.c), _fp(iq.view(np.float32)), *[_fp(t) for t in self.tabs], _fp(audio))
        return audio, float(self.c.snr), int(self.c.sql_state)


def hamming_window(lib):
    w = np.empty(512, dtype=np.float32)
    lib.lib.sdrx_oracle_hamming_window.argtypes = [_f32p]
    lib.lib.sdrx_oracle_hamming_window(_fp(w))
    return w


_cache = {}


def load(kind="oracle"):
    """kind: 'oracle' | 'oracle_fast' | 'ref_portable' | 'ref_avx2'."""
    if kind in _cache:
        return _cache[kind]
    if kind == "oracle":
        lib = _Lib(os.path.join(HERE, "liboracle.so"), "sdrx_oracle")
    elif kind == "oracle_fast":
        lib = _Lib(os.path.join(HERE, "liboracle_fast.so"), "sdrx_oracle")
    elif kind in ("ref_portable", "ref_avx2"):
        lib = _Lib(os.path.join(HERE, "_ref", f"libsdrx_{kind}.so"), "sdrx_ref")
    else:
        raise ValueError(kind)
    _cache[kind] = lib
    return lib
