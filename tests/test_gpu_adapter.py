"""GPU test of the C++ drop-in adapter (include/sdrx_gpu_adapter.hpp): a C++ program shaped like
sdrx's data_cb (reference src/sdrx.cpp:327-346) drives MSD::decimate / DS::addJob through the adapter
and compares every output with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from sdrx_amd import chains

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def adapter_bin(tmp_path_factory):
    from oracle import ffi
    ffi.build(ref=False)
    out = tmp_path_factory.mktemp("cpp")
    exe = str(out / "adapter_data_cb")
    subprocess.check_call(["g++", "-std=c++20", "-O2", "-pthread", os.path.join(ROOT, "tests", "cpp", "adapter_data_cb.cpp"),
                           "-o", exe, "-L" + os.path.join(ROOT, "sdrx_amd", "lib"), "-lsdrxgpu",
                           "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "sdrx_amd", "lib"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    ring = str(out / "adapter_ring")
    subprocess.check_call(["g++", "-std=c++20", "-O2", "-pthread", os.path.join(ROOT, "tests", "cpp", "adapter_ring.cpp"),
                           "-o", ring, "-L" + os.path.join(ROOT, "sdrx_amd", "lib"), "-lsdrxgpu",
                           "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "sdrx_amd", "lib"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    ch = chains.chain("10")
    taps = str(out / "taps.txt")
    with open(taps, "w") as f:
        f.write(f"{ch.N} {ch.z} {ch.block_len} {len(ch.stages)}\n")
        for s in ch.stages:
            f.write(f"{s.m} {len(s.h)} " + " ".join(f"{int(u):08x}" for u in s.h.view(np.uint32)) + "\n")
    return exe, taps, ring


@pytest.mark.parametrize("mode", ["serial", "ds", "pool", "serial_direct", "ds_direct", "pool_direct", "broken_direct"])
def test_data_cb_loop_through_adapter(adapter_bin, mode):
    exe, taps, _ = adapter_bin
    r = subprocess.run([exe, taps, "33", "3", mode], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("PASS")


@pytest.mark.parametrize("mode,shards", [("serial", 2), ("ds_direct", 2), ("pool", 3), ("pool_direct", 2), ("pool", -1), ("ds", -1)])
def test_data_cb_loop_through_adapter_cut_over_devices(adapter_bin, mode, shards):
    """The same loop with sdrxgpu::devices() = {0, 0[, 0]}: the bank is a sdrxgpu_multibank (contiguous channel ranges, one
    worker thread and one H2D / D2H per device).  Against the oracle in every mode; the pool modes also compare the cut bank
    with an uncut one bit for bit.  shards = -1: sdrxgpu::use_all_devices() -- the C ABI's "every visible device"
    (n_devices == 0), one shard on this box."""
    exe, taps, _ = adapter_bin
    r = subprocess.run([exe, taps, "33", "3", mode, str(shards)], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("PASS") and f"shards={shards}" in r.stdout


def test_zero_copy_into_a_page_locked_chunk_ring(adapter_bin):
    """tests/cpp/adapter_ring.cpp: an 8-chunk ring with the reference CRB's semantics (src/crb.hpp:64-189), chunks page-locked,
    acquire -> data_cb loop through the adapter in chunk_stride() mode -> commit for 24 blocks, a consumer thread one chunk
    behind that stalls in the middle: every chunk against the oracle, none written after its commit, blocks dropped on a full
    ring (src/sdrx.cpp:360-363) and the stream correct afterwards."""
    _, taps, ring = adapter_bin
    r = subprocess.run([ring, taps, "24"], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("PASS")
