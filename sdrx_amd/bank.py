"""Host-side mirror of the reference's channel-worker interface over libsdrxgpu's C ABI.

``ChannelBank`` is what sdrx's set-up loop (reference ``src/sdrx.cpp:1680-1714``) plus its
``data_cb`` channel loop (``src/sdrx.cpp:327-346``) plus the AGC/demod lines of
``alsa_write_cb`` (``src/sdrx.cpp:455-463``) become when every channel of an IQ stream is one
batched GPU pass: construct it with the per-channel translator tables and the ``MSD::Stage``
list, then call ``process(block)`` once per 32 ms block.

There is no CPU path here: if ``libsdrxgpu.so`` is missing, or no gfx950 device is present,
construction raises.  torch is optional plumbing (device tensors in/out); the library itself
knows nothing about torch.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDRXGPU_LIB_PATH: another build of the library (A/B measurements of kernel variants); default = the in-tree build
LIB_PATH = os.environ.get("SDRXGPU_LIB_PATH") or os.path.join(_HERE, "lib", "libsdrxgpu.so")

K_STAGE12, K_TAIL, K_AGC, K_TOTAL = 0, 1, 2, 3
FLAG_LF_AGC_ACTIVE = 0x1
FLAG_PIPELINED = 0x2
FLAG_FTFIR = 0x4
FLAG_FUSED_STAGE3 = 0x8
FLAG_REFERENCE_ORDER = 0x10


class SdrxGpuError(RuntimeError):
    def __init__(self, code, what, detail):
        super().__init__(f"{what}: {detail} (code {code})")
        self.code = code


class _Stage(C.Structure):
    _fields_ = [("m", C.c_uint32), ("n_taps", C.c_uint32), ("taps", C.POINTER(C.c_float))]


class _Channel(C.Structure):
    _fields_ = [("translator", C.c_void_p), ("n_translator", C.c_uint32)]


class AgcParams(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("attack", "decay", "reference", "max_gain", "alpha", "init_gain", "init_power")]


class Tuning(C.Structure):
    """sdrxgpu_tuning: geometry / kernel-path overrides for tests and tuning runs (0 = automatic)."""
    _fields_ = ([(k, C.c_uint32) for k in ("struct_size", "k12_waves", "k12_cpw", "tail_tile", "tail_generic", "fused_tail", "host_parts",
                                             "pair_lanes", "matrix_pipe", "m_slot_groups", "m_seg_groups", "tail_agc", "tail_agc_cpg", "y2_keep_mb")] +
                [("host_part_ch", C.c_uint32 * 7)])

    def __init__(self, **kw):
        kw.setdefault("struct_size", C.sizeof(Tuning))
        super().__init__(**kw)


FUSED_AUTO, FUSED_OFF, FUSED_REQUIRE = 0, 1, 2


class _Cfg(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32),
                ("n_stages", C.c_uint32), ("stages", C.POINTER(_Stage)),
                ("n_channels", C.c_uint32), ("channels", C.POINTER(_Channel)),
                ("agc", AgcParams), ("lf_agc", AgcParams),
                ("device", C.c_int32), ("stream", C.c_void_p),
                ("max_in_len", C.c_uint32), ("block_hint", C.c_uint32),
                ("audio_fir_taps", C.POINTER(C.c_float)), ("n_audio_fir_taps", C.c_uint32), ("audio_fir_gain_db", C.c_float),
                ("sql_level", C.POINTER(C.c_float)), ("passband_shape", C.POINTER(C.c_float)),
                ("ramp_up", C.POINTER(C.c_float)), ("ramp_down", C.POINTER(C.c_float)),
                ("pan_pos", C.POINTER(C.c_int8)), ("mix_fir_taps", C.POINTER(C.c_float)),
                ("n_mix_fir_taps", C.c_uint32), ("mix_fir_gain_db", C.c_float),
                ("modulation", C.POINTER(C.c_uint8)), ("tuning", C.POINTER(Tuning))]


class BankInfo(C.Structure):
    _fields_ = [("n_channels", C.c_uint32), ("m_total", C.c_uint32), ("n_translator", C.c_uint32),
                ("history_len", C.c_uint32), ("samples_in", C.c_uint64), ("samples_out", C.c_uint64),
                ("device_bytes", C.c_uint64), ("fused_tail", C.c_uint32), ("pair_lanes", C.c_uint32),
                ("k12_waves", C.c_uint32), ("k12_cpw", C.c_uint32), ("k12_bodies", C.c_uint32),
                ("matrix_pipe", C.c_uint32), ("m_slots", C.c_uint32), ("tail_agc", C.c_uint32)]


class MultiBankInfo(C.Structure):
    _fields_ = [("n_shards", C.c_uint32), ("n_channels", C.c_uint32), ("m_total", C.c_uint32), ("matrix_pipe", C.c_uint32),
                ("samples_in", C.c_uint64), ("samples_out", C.c_uint64), ("device_bytes", C.c_uint64)]


_lib = None


def _preload_hip_runtime():
    """One process, one HIP runtime.

    libsdrxgpu.so needs ``libamdhip64.so.7``.  PyTorch-ROCm wheels bundle their own copy (same
    SONAME) under ``torch/lib``; if libsdrxgpu pulled in the system copy first and torch were
    imported later, the process would hold two HIP runtimes and the second one finds no GPU.  So
    when torch is installed its bundled runtime is loaded first (by path, without importing torch)
    and libsdrxgpu binds to it by SONAME.  ``SDRXGPU_HIP_RUNTIME=system`` forces /opt/rocm's.
    """
    if os.environ.get("SDRXGPU_HIP_RUNTIME", "auto") == "system":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return None
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(path):
            return C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        return None
    return None


def load_library():
    """Load libsdrxgpu.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdrxGpuError(-2, "libsdrxgpu", f"{LIB_PATH} not built; run __graft_entry__.build() / ./build_lib.sh")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    L.sdrxgpu_strerror.restype = C.c_char_p
    L.sdrxgpu_strerror.argtypes = [C.c_int]
    L.sdrxgpu_last_error.restype = C.c_char_p
    L.sdrxgpu_version.restype = C.c_char_p
    L.sdrxgpu_agc_params_sdrx.restype = None
    L.sdrxgpu_agc_params_sdrx.argtypes = [C.POINTER(AgcParams), C.POINTER(AgcParams)]
    L.sdrxgpu_make_translator.argtypes = [C.c_int, C.c_int, C.c_int, vp]
    L.sdrxgpu_parse_channel.argtypes = [C.c_char_p, C.POINTER(C.c_uint32)]
    L.sdrxgpu_tuner_fq.argtypes = [C.POINTER(C.c_char_p), sz, C.POINTER(C.c_uint32)]
    L.sdrxgpu_channel_to_offset.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_int)]
    L.sdrxgpu_channels_fit_bandwidth.argtypes = [C.POINTER(C.c_char_p), sz, C.c_uint32, C.POINTER(C.c_int)]
    L.sdrxgpu_bank_create.argtypes = [C.POINTER(_Cfg), C.POINTER(vp)]
    L.sdrxgpu_bank_destroy.argtypes = [vp]
    L.sdrxgpu_bank_reset.argtypes = [vp]
    L.sdrxgpu_bank_process.argtypes = [vp, vp, sz, vp, vp, sz, C.POINTER(sz)]
    L.sdrxgpu_bank_process_device.argtypes = [vp, vp, sz, vp, vp, sz, C.POINTER(sz)]
    L.sdrxgpu_bank_process_pinned.argtypes = [vp, vp, sz, vp, vp, sz, C.POINTER(sz)]
    L.sdrxgpu_bank_flush.argtypes = [vp, vp, vp, sz, C.POINTER(sz)]
    L.sdrxgpu_bank_ingest_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    L.sdrxgpu_host_register.argtypes = [vp, sz]
    L.sdrxgpu_host_unregister.argtypes = [vp]
    L.sdrxgpu_bank_synchronize.argtypes = [vp]
    L.sdrxgpu_bank_ingest_buffer_after.argtypes = [vp, sz, C.POINTER(vp)]
    L.sdrxgpu_bank_get_info.argtypes = [vp, C.POINTER(BankInfo)]
    L.sdrxgpu_bank_get_host_parts.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32), sz, C.POINTER(sz)]
    L.sdrxgpu_bank_get_gains.argtypes = [vp, vp, vp]
    L.sdrxgpu_bank_get_agc_state.argtypes = [vp, vp, vp, vp, vp]
    L.sdrxgpu_bank_get_squelch.argtypes = [vp, vp, vp]
    L.sdrxgpu_audio_pos.argtypes = [C.c_uint, C.c_uint]
    L.sdrxgpu_bank_get_mix.argtypes = [vp, vp, vp, sz, C.POINTER(sz)]
    L.sdrxgpu_bank_mix_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz)]
    L.sdrxgpu_bank_profile_enable.argtypes = [vp, C.c_int]
    L.sdrxgpu_bank_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.sdrxgpu_bank_profile_reset.argtypes = [vp]
    u32p = C.POINTER(C.c_uint32)
    L.sdrxgpu_shard_range.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p]
    L.sdrxgpu_multibank_create.argtypes = [C.POINTER(_Cfg), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(vp)]
    L.sdrxgpu_multibank_destroy.argtypes = [vp]
    L.sdrxgpu_multibank_reset.argtypes = [vp]
    L.sdrxgpu_multibank_process.argtypes = [vp, vp, sz, vp, vp, sz, C.POINTER(sz)]
    L.sdrxgpu_multibank_process_pinned.argtypes = [vp, vp, sz, C.POINTER(vp), C.POINTER(vp), sz, C.POINTER(sz)]
    L.sdrxgpu_multibank_synchronize.argtypes = [vp]
    L.sdrxgpu_multibank_get_info.argtypes = [vp, C.POINTER(MultiBankInfo)]
    L.sdrxgpu_multibank_shard.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_int32), u32p, u32p]
    L.sdrxgpu_multibank_get_agc_state.argtypes = [vp, vp, vp, vp, vp]
    L.sdrxgpu_multibank_get_gains.argtypes = [vp, vp, vp]
    L.sdrxgpu_multibank_get_squelch.argtypes = [vp, vp, vp]
    _lib = L
    return L


EXPORTED_SYMBOLS = [
    "sdrxgpu_agc_params_sdrx", "sdrxgpu_make_translator", "sdrxgpu_parse_channel", "sdrxgpu_tuner_fq",
    "sdrxgpu_channel_to_offset", "sdrxgpu_channels_fit_bandwidth", "sdrxgpu_bank_create", "sdrxgpu_bank_destroy",
    "sdrxgpu_bank_reset", "sdrxgpu_bank_process", "sdrxgpu_bank_process_device", "sdrxgpu_bank_process_pinned", "sdrxgpu_bank_flush", "sdrxgpu_bank_ingest_buffer", "sdrxgpu_bank_ingest_buffer_after",
    "sdrxgpu_host_register", "sdrxgpu_host_unregister", "sdrxgpu_bank_synchronize", "sdrxgpu_bank_get_info", "sdrxgpu_bank_get_host_parts", "sdrxgpu_bank_get_gains", "sdrxgpu_bank_get_agc_state", "sdrxgpu_bank_get_squelch", "sdrxgpu_bank_profile_enable",
    "sdrxgpu_audio_pos", "sdrxgpu_bank_get_mix", "sdrxgpu_bank_mix_buffers",
    "sdrxgpu_bank_profile_read", "sdrxgpu_bank_profile_reset", "sdrxgpu_strerror", "sdrxgpu_last_error",
    "sdrxgpu_version",
    "sdrxgpu_shard_range", "sdrxgpu_multibank_create", "sdrxgpu_multibank_destroy", "sdrxgpu_multibank_reset", "sdrxgpu_multibank_process",
    "sdrxgpu_multibank_process_pinned", "sdrxgpu_multibank_synchronize", "sdrxgpu_multibank_get_info", "sdrxgpu_multibank_shard",
    "sdrxgpu_multibank_get_agc_state", "sdrxgpu_multibank_get_gains", "sdrxgpu_multibank_get_squelch",
]


class DeviceView:
    """A raw device address as a float32 array for ``torch.as_tensor(view, device='cuda')`` (zero copy)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": "<f4",
                                         "data": (int(ptr), False), "version": 2}


def _check(rc, what):
    if rc != 0:
        L = load_library()
        raise SdrxGpuError(rc, what, f"{L.sdrxgpu_strerror(rc).decode()}: {L.sdrxgpu_last_error().decode()}")


def host_register(arr: np.ndarray):
    """Page-lock a numpy array that is passed to ``ChannelBank.process`` repeatedly (DMA instead of staged copies)."""
    _check(load_library().sdrxgpu_host_register(arr.ctypes.data, arr.nbytes), "sdrxgpu_host_register")


def host_unregister(arr: np.ndarray):
    _check(load_library().sdrxgpu_host_unregister(arr.ctypes.data), "sdrxgpu_host_unregister")


def make_translator(N: int, z: int, ch_offset: int) -> np.ndarray:
    """Translator table by the reference formula (``src/sdrx.cpp:1686-1689``); host libm, no GPU needed."""
    out = np.empty(N, dtype=np.complex64)
    _check(load_library().sdrxgpu_make_translator(N, z, ch_offset, out.ctypes.data), "sdrxgpu_make_translator")
    return out


def _names(names):
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    return arr, len(names)


def parse_channel(name: str) -> int:
    """Aeronautical channel name -> Hz (``parse_fq(str, true)``, sdrx.cpp:957-1021); raises on an invalid name."""
    fq = C.c_uint32()
    _check(load_library().sdrxgpu_parse_channel(name.encode(), C.byref(fq)), "sdrxgpu_parse_channel")
    return fq.value


def tuner_fq(names) -> int:
    """Tuner centre sdrx picks for these channels (sdrx.cpp:1398-1408)."""
    arr, n = _names(names)
    fq = C.c_uint32()
    _check(load_library().sdrxgpu_tuner_fq(arr, n, C.byref(fq)), "sdrxgpu_tuner_fq")
    return fq.value


def channel_to_offset(name: str, tuner: int) -> int:
    """Signed 8.33 kHz steps from the tuner centre (``channel_to_offset``, sdrx.cpp:1449-1478)."""
    off = C.c_int()
    _check(load_library().sdrxgpu_channel_to_offset(name.encode(), tuner, C.byref(off)), "sdrxgpu_channel_to_offset")
    return off.value


def channels_fit_bandwidth(names, sample_rate_hz: int) -> bool:
    """The 80 % rule of ``verify_requested_bandwidth`` (sdrx.cpp:1424-1443)."""
    arr, n = _names(names)
    fits = C.c_int()
    _check(load_library().sdrxgpu_channels_fit_bandwidth(arr, n, sample_rate_hz, C.byref(fits)), "sdrxgpu_channels_fit_bandwidth")
    return bool(fits.value)


def translators_for(chain, offsets):
    """One table per offset; ``None`` (empty translator) at the tuner centre, as sdrx.cpp:1685 does."""
    cache = {}
    out = []
    for off in offsets:
        if off == 0:
            out.append(None)
            continue
        if off not in cache:
            cache[off] = make_translator(chain.N, chain.z, off)
        out.append(cache[off])
    return out


def audio_pos(channel_no, num_channels):
    """get_audio_pos of sdrx.cpp:1024-1044."""
    return int(load_library().sdrxgpu_audio_pos(channel_no, num_channels))


def sdrx_agc_params():
    agc, lf = AgcParams(), AgcParams()
    load_library().sdrxgpu_agc_params_sdrx(C.byref(agc), C.byref(lf))
    return agc, lf


class ChannelBank:
    """All channels of one IQ stream on one GPU.

    translators : list (one per channel) of complex64 arrays of equal length N, or None for a
                  channel at the tuner centre -- what ``MSD``'s constructor takes (msd.hpp:54).
    stages      : list of (m, float32 taps) -- the ``std::vector<MSD::Stage>`` of sdrx.cpp:1588-1672.
    """

    def __init__(self, translators, stages, **kw):
        L = load_library()
        self._L = L
        cfg = self._make_cfg(translators, stages, **kw)
        h = C.c_void_p()
        _check(L.sdrxgpu_bank_create(C.byref(cfg), C.byref(h)), "sdrxgpu_bank_create")
        self._h = h

    def _make_cfg(self, translators, stages, lf_agc=False, agc=None, agc_lf=None, device=-1, stream=None,
                  max_in_len=0, block_hint=0, pipelined=False, ftfir=False, fused_stage3=False, reference_order=False,
                  audio_fir=None, audio_fir_gain_db=0.0, squelch=None, pan_pos=None, mix_fir=None, mix_fir_gain_db=0.0,
                  modulation=None, tuning=None):
        """sdrxgpu_bank_cfg for these arguments; the arrays it points into stay alive as attributes of self."""
        n_ch = len(translators)
        self._taps = [np.ascontiguousarray(h, dtype=np.float32) for _, h in stages]
        st = (_Stage * len(stages))()
        for i, (m, _) in enumerate(stages):
            st[i].m = int(m)
            st[i].n_taps = len(self._taps[i])
            st[i].taps = self._taps[i].ctypes.data_as(C.POINTER(C.c_float))
        self._tabs = [np.ascontiguousarray(t, dtype=np.complex64) if t is not None and len(t) else None for t in translators]
        chs = (_Channel * n_ch)()
        for i, t in enumerate(self._tabs):
            chs[i].translator = t.ctypes.data if t is not None else None
            chs[i].n_translator = len(t) if t is not None else 0
        d_agc, d_lf = sdrx_agc_params()
        cfg = _Cfg()
        cfg.struct_size = C.sizeof(_Cfg)
        cfg.flags = (FLAG_LF_AGC_ACTIVE if lf_agc else 0) | (FLAG_PIPELINED if pipelined else 0) | (FLAG_FTFIR if ftfir else 0) | (FLAG_FUSED_STAGE3 if fused_stage3 else 0) | (FLAG_REFERENCE_ORDER if reference_order else 0)
        cfg.n_stages = len(stages)
        cfg.stages = st
        cfg.n_channels = n_ch
        cfg.channels = chs
        cfg.agc = agc or d_agc
        cfg.lf_agc = agc_lf or d_lf
        cfg.device = device
        cfg.stream = stream
        cfg.max_in_len = max_in_len
        cfg.block_hint = block_hint
        if audio_fir is not None:
            self._fir = np.ascontiguousarray(audio_fir, dtype=np.float32)
            cfg.audio_fir_taps = self._fir.ctypes.data_as(C.POINTER(C.c_float))
            cfg.n_audio_fir_taps = len(self._fir)
            cfg.audio_fir_gain_db = audio_fir_gain_db
        if squelch is not None:
            # squelch = (sql_level per channel, passband_shape, ramp_up, ramp_down)
            self._sq = [np.ascontiguousarray(a_, dtype=np.float32) for a_ in squelch]
            assert len(self._sq[0]) == n_ch and all(len(t) == 512 for t in self._sq[1:])
            fp = C.POINTER(C.c_float)
            cfg.sql_level, cfg.passband_shape, cfg.ramp_up, cfg.ramp_down = [t.ctypes.data_as(fp) for t in self._sq]
        if modulation is not None:
            # per channel: 0 / "AM", 1 / "FM"
            self._mod = np.ascontiguousarray([{"AM": 0, "FM": 1}.get(m_, m_) for m_ in modulation], dtype=np.uint8)
            assert len(self._mod) == n_ch
            cfg.modulation = self._mod.ctypes.data_as(C.POINTER(C.c_uint8))
        if pan_pos is not None:
            # stereo mix-down: pan position per channel (-2..2), optional common FIR2 taps + volume in dB
            self._pan = np.ascontiguousarray(pan_pos, dtype=np.int8)
            assert len(self._pan) == n_ch
            cfg.pan_pos = self._pan.ctypes.data_as(C.POINTER(C.c_int8))
            if mix_fir is not None:
                self._mixfir = np.ascontiguousarray(mix_fir, dtype=np.float32)
                cfg.mix_fir_taps = self._mixfir.ctypes.data_as(C.POINTER(C.c_float))
                cfg.n_mix_fir_taps = len(self._mixfir)
                cfg.mix_fir_gain_db = mix_fir_gain_db
        if tuning is not None:
            # dict of sdrxgpu_tuning fields, e.g. {"k12_waves": 8, "k12_cpw": 4} (tests / tuning runs only)
            self._tuning = Tuning(**tuning)
            cfg.tuning = C.pointer(self._tuning)
        self.n_channels = n_ch
        self.m = 1
        for m, _ in stages:
            self.m *= int(m)
        self._keep = (st, chs)
        return cfg

    # ---- lifetime ----
    def close(self):
        if getattr(self, "_h", None):
            self._L.sdrxgpu_bank_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reset(self):
        _check(self._L.sdrxgpu_bank_reset(self._h), "sdrxgpu_bank_reset")

    def synchronize(self):
        _check(self._L.sdrxgpu_bank_synchronize(self._h), "sdrxgpu_bank_synchronize")

    def info(self) -> BankInfo:
        i = BankInfo()
        _check(self._L.sdrxgpu_bank_get_info(self._h, C.byref(i)), "sdrxgpu_bank_get_info")
        return i

    def host_parts(self, audio_only=False):
        """Channel bounds [0, ..., n_channels] of the parts ``process`` (host pointers) runs the bank in: the cut for calls that
        take the decimated IQ, or (``audio_only``) the one for calls that take the audio alone."""
        b = (C.c_uint32 * 16)()
        n = C.c_size_t(0)
        _check(self._L.sdrxgpu_bank_get_host_parts(self._h, int(audio_only), b, 16, C.byref(n)), "sdrxgpu_bank_get_host_parts")
        return [int(b[i]) for i in range(n.value + 1)]

    def gains(self):
        a = np.empty(self.n_channels, dtype=np.float32)
        b = np.empty(self.n_channels, dtype=np.float32)
        _check(self._L.sdrxgpu_bank_get_gains(self._h, a.ctypes.data, b.ctypes.data), "sdrxgpu_bank_get_gains")
        return a, b

    def agc_state(self):
        """(agc gain, agc power, lf gain, lf power), each [n_ch]: the whole recurrence state of the path."""
        a = [np.empty(self.n_channels, dtype=np.float32) for _ in range(4)]
        _check(self._L.sdrxgpu_bank_get_agc_state(self._h, *[v.ctypes.data for v in a]), "sdrxgpu_bank_get_agc_state")
        return tuple(a)

    def squelch_state(self):
        """(snr_db[n_ch], state[n_ch]) after the last chunk (sdrx.cpp:567-573)."""
        snr = np.empty(self.n_channels, dtype=np.float32)
        st = np.empty(self.n_channels, dtype=np.uint8)
        _check(self._L.sdrxgpu_bank_get_squelch(self._h, snr.ctypes.data, st.ctypes.data), "sdrxgpu_bank_get_squelch")
        return snr, st

    def mix(self):
        """(float32 [n, 2], int16 [n, 2]): the stereo mix of the last process call (sdrx.cpp:469-497, 646-656)."""
        n = C.c_size_t(0)
        _check(self._L.sdrxgpu_bank_get_mix(self._h, None, None, 0, C.byref(n)), "sdrxgpu_bank_get_mix")
        f = np.empty((n.value, 2), dtype=np.float32)
        q = np.empty((n.value, 2), dtype=np.int16)
        _check(self._L.sdrxgpu_bank_get_mix(self._h, f.ctypes.data, q.ctypes.data, n.value, C.byref(n)), "sdrxgpu_bank_get_mix")
        return f, q

    # ---- the hot path ----
    def process(self, iq, out_stride=None, want_iq=True, want_audio=True, iq_out=None, audio_out=None):
        """Host buffers in, host buffers out (the drop-in call).  Returns (iq[ch, n_out], audio[ch, n_out]).
        ``iq_out`` / ``audio_out``: caller-owned [n_ch, out_stride] arrays to fill (e.g. page-locked ring chunks)."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        stride = int(out_stride or (iq.shape[0] // self.m + 1))
        if iq_out is not None or audio_out is not None:
            stride = (iq_out if iq_out is not None else audio_out).shape[1]
            want_iq, want_audio = iq_out is not None, audio_out is not None
            audio = audio_out
        else:
            iq_out = np.zeros((self.n_channels, stride), dtype=np.complex64) if want_iq else None
            audio = np.zeros((self.n_channels, stride), dtype=np.float32) if want_audio else None
        n_out = C.c_size_t(0)
        _check(self._L.sdrxgpu_bank_process(self._h, iq.ctypes.data, iq.shape[0],
                                            iq_out.ctypes.data if want_iq else None,
                                            audio.ctypes.data if want_audio else None, stride, C.byref(n_out)),
               "sdrxgpu_bank_process")
        k = n_out.value
        return (iq_out[:, :k] if want_iq else None), (audio[:, :k] if want_audio else None)

    def process_device(self, d_in_ptr, in_len, d_iq_out_ptr, d_audio_out_ptr, out_stride):
        """Raw device pointers; asynchronous on the bank's stream.  Returns outputs per channel."""
        n_out = C.c_size_t(0)
        _check(self._L.sdrxgpu_bank_process_device(self._h, d_in_ptr, in_len, d_iq_out_ptr, d_audio_out_ptr,
                                                   out_stride, C.byref(n_out)), "sdrxgpu_bank_process_device")
        return n_out.value

    def process_pinned(self, host_in_ptr, in_len, d_iq_out_ptr, d_audio_out_ptr, out_stride):
        """Page-locked host block in, device pointers out, nothing waits (H2D on the bank's copy stream into its
        free input ring).  The host block must stay unchanged until the next call on this bank or ``synchronize``."""
        n_out = C.c_size_t(0)
        _check(self._L.sdrxgpu_bank_process_pinned(self._h, host_in_ptr, in_len, d_iq_out_ptr, d_audio_out_ptr,
                                                   out_stride, C.byref(n_out)), "sdrxgpu_bank_process_pinned")
        return n_out.value

    def flush(self, d_iq_out_ptr, d_audio_out_ptr, out_stride):
        """Pipelined banks: outputs of the block whose tail stages are still pending."""
        n_out = C.c_size_t(0)
        _check(self._L.sdrxgpu_bank_flush(self._h, d_iq_out_ptr, d_audio_out_ptr, out_stride, C.byref(n_out)), "sdrxgpu_bank_flush")
        return n_out.value

    def ingest_buffer(self):
        p, cap = C.c_void_p(), C.c_size_t()
        _check(self._L.sdrxgpu_bank_ingest_buffer(self._h, C.byref(p), C.byref(cap)), "sdrxgpu_bank_ingest_buffer")
        return p.value, cap.value

    def ingest_buffer_after(self, in_len):
        """Where the block after a next call of ``in_len`` samples may be written (the bank's other input ring)."""
        p = C.c_void_p()
        _check(self._L.sdrxgpu_bank_ingest_buffer_after(self._h, in_len, C.byref(p)), "sdrxgpu_bank_ingest_buffer_after")
        return p.value

    # ---- timing (HIP events on the bank's stream) ----
    def profile_enable(self, on=True):
        _check(self._L.sdrxgpu_bank_profile_enable(self._h, int(on)), "sdrxgpu_bank_profile_enable")

    def profile_reset(self):
        _check(self._L.sdrxgpu_bank_profile_reset(self._h), "sdrxgpu_bank_profile_reset")

    def profile_read(self, kernel):
        ms, n = C.c_double(), C.c_uint64()
        _check(self._L.sdrxgpu_bank_profile_read(self._h, kernel, C.byref(ms), C.byref(n)), "sdrxgpu_bank_profile_read")
        return ms.value, n.value


def shard_range(n_items: int, index: int, n_shards: int):
    """[lo, hi) of shard ``index``: sdrxgpu_shard_range, the rule a multibank cuts its channel list by (pure arithmetic, no GPU)."""
    lo, hi = C.c_uint32(), C.c_uint32()
    _check(load_library().sdrxgpu_shard_range(n_items, index, n_shards, C.byref(lo), C.byref(hi)), "sdrxgpu_shard_range")
    return lo.value, hi.value


class MultiChannelBank(ChannelBank):
    """One channel list over several GPUs of one node in ONE process (``sdrxgpu_multibank``): contiguous channel ranges per
    device, per-device state, one H2D copy of the block per device from the caller's buffer, outputs straight into each
    device's slice of the caller's arrays; one call = all devices, complete on return.  Same constructor as
    ``ChannelBank`` plus ``devices`` (HIP ordinals, one shard each; the same ordinal may repeat; empty = all visible)."""

    def __init__(self, translators, stages, devices=(), **kw):
        L = load_library()
        self._L = L
        cfg = self._make_cfg(translators, stages, **kw)
        devs = (C.c_int32 * max(1, len(devices)))(*[int(d) for d in devices])
        h = C.c_void_p()
        _check(L.sdrxgpu_multibank_create(C.byref(cfg), devs if len(devices) else None, len(devices), C.byref(h)), "sdrxgpu_multibank_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.sdrxgpu_multibank_destroy(self._h)
            self._h = None

    def reset(self):
        _check(self._L.sdrxgpu_multibank_reset(self._h), "sdrxgpu_multibank_reset")

    def synchronize(self):
        _check(self._L.sdrxgpu_multibank_synchronize(self._h), "sdrxgpu_multibank_synchronize")

    def info(self) -> MultiBankInfo:
        i = MultiBankInfo()
        _check(self._L.sdrxgpu_multibank_get_info(self._h, C.byref(i)), "sdrxgpu_multibank_get_info")
        return i

    def shards(self):
        """[(device, ch_lo, ch_hi, BankInfo)] per shard."""
        out = []
        for k in range(self.info().n_shards):
            b, dev, lo, hi = C.c_void_p(), C.c_int32(), C.c_uint32(), C.c_uint32()
            _check(self._L.sdrxgpu_multibank_shard(self._h, k, C.byref(b), C.byref(dev), C.byref(lo), C.byref(hi)), "sdrxgpu_multibank_shard")
            bi = BankInfo()
            _check(self._L.sdrxgpu_bank_get_info(b, C.byref(bi)), "sdrxgpu_bank_get_info")
            out.append((dev.value, lo.value, hi.value, bi))
        return out

    def agc_state(self):
        a = [np.empty(self.n_channels, dtype=np.float32) for _ in range(4)]
        _check(self._L.sdrxgpu_multibank_get_agc_state(self._h, *[v.ctypes.data for v in a]), "sdrxgpu_multibank_get_agc_state")
        return tuple(a)

    def process(self, iq, out_stride=None, want_iq=True, want_audio=True, iq_out=None, audio_out=None):
        """Host buffers in, host buffers out, all devices, complete on return (``sdrxgpu_multibank_process``)."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        stride = int(out_stride or (iq.shape[0] // self.m + 1))
        if iq_out is not None or audio_out is not None:
            stride = (iq_out if iq_out is not None else audio_out).shape[1]
            want_iq, want_audio = iq_out is not None, audio_out is not None
            audio = audio_out
        else:
            iq_out = np.zeros((self.n_channels, stride), dtype=np.complex64) if want_iq else None
            audio = np.zeros((self.n_channels, stride), dtype=np.float32) if want_audio else None
        n_out = C.c_size_t(0)
        _check(self._L.sdrxgpu_multibank_process(self._h, iq.ctypes.data, iq.shape[0], iq_out.ctypes.data if want_iq else None,
                                                 audio.ctypes.data if want_audio else None, stride, C.byref(n_out)),
               "sdrxgpu_multibank_process")
        k = n_out.value
        return (iq_out[:, :k] if want_iq else None), (audio[:, :k] if want_audio else None)

    def process_pinned(self, host_in_ptr, in_len, d_iq_out_ptrs, d_audio_out_ptrs, out_stride):
        """Page-locked block in, one device pointer per shard out, asynchronous (``sdrxgpu_multibank_process_pinned``)."""
        n = self.info().n_shards
        mk = lambda ps: (C.c_void_p * n)(*ps) if ps is not None else None   # noqa: E731
        n_out = C.c_size_t(0)
        _check(self._L.sdrxgpu_multibank_process_pinned(self._h, host_in_ptr, in_len, mk(d_iq_out_ptrs), mk(d_audio_out_ptrs),
                                                        out_stride, C.byref(n_out)), "sdrxgpu_multibank_process_pinned")
        return n_out.value

    # single-device entries that a multibank does not have
    def _no(self, *a, **k):
        raise SdrxGpuError(-5, "MultiChannelBank", "not available across devices (use the shards' banks)")

    def gains(self):
        a = np.empty(self.n_channels, dtype=np.float32)
        b = np.empty(self.n_channels, dtype=np.float32)
        _check(self._L.sdrxgpu_multibank_get_gains(self._h, a.ctypes.data, b.ctypes.data), "sdrxgpu_multibank_get_gains")
        return a, b

    def squelch_state(self):
        snr = np.empty(self.n_channels, dtype=np.float32)
        st = np.empty(self.n_channels, dtype=np.uint8)
        _check(self._L.sdrxgpu_multibank_get_squelch(self._h, snr.ctypes.data, st.ctypes.data), "sdrxgpu_multibank_get_squelch")
        return snr, st

    mix = process_device = flush = ingest_buffer = ingest_buffer_after = host_parts = _no
    profile_enable = profile_reset = profile_read = _no
