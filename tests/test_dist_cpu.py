"""CPU tests of the N > 1 path: world_size-2 `gloo` processes run the same sharding / block fan-out
code bench.py uses on GPUs (sdrx_amd/dist.py), with the oracle standing in for the per-rank bank, and
must reproduce the unsharded result exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from sdrx_amd import chains, synth
from sdrx_amd import dist as sdist


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 1024, 8192, 10000):
        for world in (1, 2, 3, 4, 8):
            spans = [sdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sdist.shard_list(list(range(10)), 1, 3) == [4, 5, 6]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_ch, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ffi
    orc = ffi.load("oracle")
    ch = chains.chain("2.4")
    offs_all = [((i * 7) % 41) - 20 for i in range(n_ch)]
    offs = sdist.shard_list(offs_all, rank, world)
    tabs = [orc.make_translator(ch.N, ch.z, o) if o else None for o in offs]
    bank = orc.bank(tabs, ch.stage_tuples())
    L = ch.block_len
    buf = torch.empty(L, 2, dtype=torch.float32)
    fan = sdist.BlockFanout(dist, buf, src=0)
    outs = []
    for b in range(2):
        src = None
        if rank == 0:      # only the stream owner has the samples
            src = torch.from_numpy(synth.synth_iq(3, b * L, L, 0.25).view(np.float32).reshape(L, 2).copy())
        cur = fan.step(src)
        x = cur.numpy().reshape(-1).view(np.complex64)
        iq, au = bank.process(x, threads=1)
        outs.append((iq, au))
    dist.barrier()
    t = sdist.max_over_ranks(dist, 1.0 + rank)
    assert t == float(world)          # the slowest rank's time
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), iq=np.concatenate([o[0] for o in outs], axis=1),
             au=np.concatenate([o[1] for o in outs], axis=1), lo=sdist.shard_range(n_ch, rank, world)[0])
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharded_equals_unsharded(oracle, tmp_path):
    n_ch, world = 7, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_ch, str(tmp_path)), nprocs=world, join=True)
    ch = chains.chain("2.4")
    offs_all = [((i * 7) % 41) - 20 for i in range(n_ch)]
    tabs = [oracle.make_translator(ch.N, ch.z, o) if o else None for o in offs_all]
    bank = oracle.bank(tabs, ch.stage_tuples())
    L = ch.block_len
    ref = [bank.process(synth.synth_iq(3, b * L, L, 0.25), threads=1) for b in range(2)]
    ref_iq = np.concatenate([r[0] for r in ref], axis=1)
    ref_au = np.concatenate([r[1] for r in ref], axis=1)
    got_iq = np.zeros_like(ref_iq)
    got_au = np.zeros_like(ref_au)
    covered = 0
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        lo = int(z["lo"])
        got_iq[lo:lo + z["iq"].shape[0]] = z["iq"]
        got_au[lo:lo + z["au"].shape[0]] = z["au"]
        covered += z["iq"].shape[0]
    assert covered == n_ch
    assert np.array_equal(got_iq.view(np.uint32), ref_iq.view(np.uint32))
    assert np.array_equal(got_au.view(np.uint32), ref_au.view(np.uint32))


@pytest.mark.timeout(300)
def test_bench_gpus_2_starts_two_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks as child processes and relay rank 0's
    JSON line (here: the CPU self-test of the launcher + double-buffered fan-out, gloo, no GPU work)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch", "--steps", "12"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12
    assert d["selftest"]["blocks_received_wrong"] == 0
    assert {"metric", "value", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"} <= set(d)


def test_pipelined_fanout_single_rank_passes_blocks_through():
    """world == 1: nothing is communicated, block k comes back from get(k)."""
    fan = sdist.PipelinedFanout(None, [torch.empty(4, 2), torch.empty(4, 2)])
    blocks = [torch.full((4, 2), float(i)) for i in range(5)]
    fan.post(0, blocks[0])
    for k in range(5):
        cur = fan.get(k)
        if k + 1 < 5:
            fan.post(k + 1, blocks[k + 1])
        assert torch.equal(cur, blocks[k])
        fan.done(k)
