"""CPU: the index arithmetic of the fused stage-2/3 path.  tools/emu_fused.py is the model (per-lane slots, emission schedule,
dump, assembly of every stage-3 output from the entries of its segments, carry across calls) and is itself checked against a
direct convolution on ragged call sequences; sdrx_amd/csrc/fused_index.hpp is what the kernels compile -- held to the model here,
forward (locate) and backward (entry_output), over random stream positions and call lengths."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import emu_fused as emu  # noqa: E402


@pytest.fixture(scope="module")
def fx(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fx") / "libfx.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", os.path.join(ROOT, "tests", "cpp", "fused_index_c.cpp"), "-o", so])
    lib = C.CDLL(so)
    ip = C.POINTER(C.c_int)
    lib.fx_locate.argtypes = [C.c_int] * 4 + [ip]
    lib.fx_overlaps.argtypes = [C.c_int] * 4
    lib.fx_entry_output.argtypes = [C.c_int] * 5 + [ip, ip]
    lib.fx_emits.argtypes = [C.c_int, C.c_int, ip]
    return lib


def test_the_model_reproduces_a_direct_convolution():
    emu.main()


def test_constants_agree(fx):
    assert [fx.fx_sigma(c) for c in range(5)] == list(emu.SIGMA) and [fx.fx_delta(c) for c in range(5)] == list(emu.DELTA)


def test_emission_schedule_agrees(fx):
    for K0 in list(range(0, 400)) + [64000, 128000, 10 ** 9 + 7]:
        for n in range(0, 250, 10):
            slot = C.c_int()
            got = fx.fx_emits(K0 % 150, n, C.byref(slot))
            want, p = emu.emits(K0, n)
            assert bool(got) == bool(want) and (not want or slot.value == p), (K0, n)


def test_locate_and_its_inverse_agree_with_the_model(fx):
    rng = np.random.default_rng(5)
    cases = [(0, 64000), (64000, 64000), (128000, 64000), (3, 1), (7, 24), (100, 239), (100, 240), (100, 241), (12345, 4000), (149, 150)]
    cases += [(int(rng.integers(0, 10 ** 7)), int(rng.integers(1, 70000))) for _ in range(12)]
    for K0, n_y1 in cases:
        k25, r0 = K0 % 25, K0 // 25
        r0m6 = r0 % 6
        n_seg = -(-n_y1 // emu.LSEG)
        forward = {}
        R = r0
        while 25 * R - 94 < K0 + n_y1:
            for s in range(max(0, (25 * R - 94 - K0) // emu.LSEG), min(n_seg - 1, (25 * R + 24 - K0) // emu.LSEG if 25 * R + 24 - K0 >= 0 else -1) + 1):
                seg_lo, seg_hi = K0 + emu.LSEG * s, min(K0 + emu.LSEG * (s + 1), K0 + n_y1)
                if 25 * R + 24 < seg_lo or 25 * R - 94 >= seg_hi:
                    continue
                want, nc_want = emu.locate(K0, s, R)
                nc = C.c_int()
                assert fx.fx_overlaps(k25, n_y1, s, R - r0) == 1
                assert fx.fx_locate(k25, r0m6, s, R - r0, C.byref(nc)) == want and nc.value == nc_want, (K0, n_y1, s, R)
                assert (s, want) not in forward, "two outputs in one entry"
                forward[(s, want)] = R - r0
            R += 1
        # outputs that are still in flight in segments they do not complete in (heads) were not enumerated above for s beyond
        # the completing segment's clamp; enumerate them through the model's tail loop instead
        for s in range(n_seg):
            for entry in range(emu.NENT):
                rr, nc = C.c_int(), C.c_int()
                ok = fx.fx_entry_output(k25, r0m6, n_y1, s, entry, C.byref(rr), C.byref(nc))
                if (s, entry) in forward:
                    assert ok and rr.value == forward[(s, entry)], (K0, n_y1, s, entry)
                elif ok:
                    # must then be an output whose window overlaps the segment and that the forward map puts here
                    Rabs = rr.value + r0
                    assert emu.locate(K0, s, Rabs)[0] == entry
                    seg_lo, seg_hi = K0 + emu.LSEG * s, min(K0 + emu.LSEG * (s + 1), K0 + n_y1)
                    assert 25 * Rabs - 94 < seg_hi and 25 * Rabs + 24 >= seg_lo
