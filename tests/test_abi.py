"""CPU tests of the boundary: the C-ABI library builds, loads and exports every symbol
include/sdrx_gpu.h declares; host-only entry points work without a GPU; and without a GPU the
product path fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from sdrx_amd import bank as sb
from sdrx_amd import chains


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(sb.LIB_PATH):
        subprocess.check_call([os.path.join(ROOT, "build_lib.sh")])
    return sb.load_library()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sdrx_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdrxgpu_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    decl = _declared_symbols()
    assert len(decl) >= 15
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/sdrx_gpu.h but not exported by libsdrxgpu.so"
    assert sorted(sb.EXPORTED_SYMBOLS) == decl


def test_no_oracle_or_torch_in_product_library(lib):
    """The product library must not link the oracle (or torch): a parity claim through the checker is void."""
    out = subprocess.check_output(["ldd", sb.LIB_PATH]).decode()
    assert "oracle" not in out and "torch" not in out
    assert "libamdhip64" in out
    for root, _, files in os.walk(os.path.join(ROOT, "sdrx_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "liboracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_strerror_and_version(lib):
    assert lib.sdrxgpu_strerror(0) == b"ok"
    assert b"gfx950" in lib.sdrxgpu_strerror(-2)
    assert b"gfx950" in lib.sdrxgpu_version()


def test_null_handles_are_refused_not_dereferenced(lib):
    import ctypes as C
    n = C.c_size_t(7)
    assert lib.sdrxgpu_bank_get_host_parts(None, 0, None, 0, C.byref(n)) == -1
    assert lib.sdrxgpu_bank_synchronize(None) == -1 and lib.sdrxgpu_bank_destroy(None) == -1
    assert lib.sdrxgpu_multibank_synchronize(None) == -1 and lib.sdrxgpu_multibank_destroy(None) == -1
    assert b"null" in lib.sdrxgpu_last_error()


def test_make_translator_matches_oracle(lib, oracle):
    """Host-side table builder == the oracle's restatement of sdrx.cpp:1686-1689 (same libm)."""
    for rate, off in (("10", 1), ("10", -415), ("10", 480), ("2.4", 37), ("1.44", -11), ("0.96", 200)):
        ch = chains.chain(rate)
        a = sb.make_translator(ch.N, ch.z, off)
        b = oracle.make_translator(ch.N, ch.z, off)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_agc_defaults_match_sdrx_setup(lib):
    agc, lf = sb.sdrx_agc_params()
    assert (agc.attack, agc.decay, agc.reference, agc.max_gain, agc.alpha) == (1.0, np.float32(0.01), 1.0, 300.0, np.float32(0.01))
    assert (agc.init_gain, agc.init_power) == (1.0, 0.25)
    assert (lf.max_gain, lf.alpha, lf.init_power) == (200.0, np.float32(0.005), 0.0625)


def test_create_fails_loudly_without_gpu_or_with_bad_args(lib):
    ch = chains.chain("10")
    # bad arguments are rejected before any device is touched
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank([None], [(5, np.ones(4, dtype=np.float32))] * 4)   # even tap count (msd.hpp:170)
    assert e.value.code == -1
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank([None], ch.stage_tuples()[:2])                      # 2 stages
    assert e.value.code == -5
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(sb.SdrxGpuError) as e:
            sb.ChannelBank([None], ch.stage_tuples())
        assert e.value.code == -2          # SDRXGPU_ERR_NO_DEVICE: no CPU fallback exists


def test_shard_range_is_the_rule_the_ranks_use(lib):
    """sdrxgpu_shard_range (how a multibank cuts its channel list over devices) == sdrx_amd.dist.shard_range (how bench.py's
    ranks cut theirs): contiguous, complete, the first n % shards one longer.  Pure arithmetic, no GPU."""
    from sdrx_amd import dist as sdist
    for n in (0, 1, 5, 8, 9, 64, 1024, 8192, 10000):
        for w in (1, 2, 3, 7, 8):
            rs = [sb.shard_range(n, i, w) for i in range(w)]
            assert rs == [sdist.shard_range(n, i, w) for i in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert sb.shard_range(10000, 3, 8) == (3750, 5000)
    with pytest.raises(sb.SdrxGpuError):
        sb.shard_range(10, 8, 8)
    with pytest.raises(sb.SdrxGpuError):
        sb.shard_range(10, 0, 0)


def test_multibank_create_refuses_before_any_device_and_fails_loudly_without_one(lib):
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, [1, -1, 2])
    with pytest.raises(sb.SdrxGpuError) as e:       # a sum over all channels does not shard
        sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0), pan_pos=[0, 1, -1])
    assert e.value.code == -5
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0), pipelined=True)
    assert e.value.code == -5
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0), stream=0x1000)
    assert e.value.code == -1
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(sb.SdrxGpuError) as e:
            sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=(0, 0))
        assert e.value.code == -2


def test_tuning_struct_of_another_layout_is_refused(lib):
    """sdrxgpu_tuning carries its size: a caller built against a header with other fields is told so instead of having its
    values read as other fields."""
    ch = chains.chain("10")
    tabs = sb.translators_for(ch, [1])
    with pytest.raises(sb.SdrxGpuError) as e:
        sb.ChannelBank(tabs, ch.stage_tuples(), tuning={"struct_size": 40, "matrix_pipe": 1})
    assert e.value.code == -1 and "struct_size" in str(e.value)
