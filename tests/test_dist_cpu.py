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


def test_plans_of_the_baseline_configurations():
    """bench.py --config: which channels and which streams every rank of an 8-GPU job runs."""
    c3 = chains.offsets_for_config("c3")
    # configs[3]: 8 192 channels = 1 024 per GPU, one stream, block broadcast
    plans = [sdist.plan("c4", r, 8, 1200) for r in range(8)]
    assert all(p.broadcast and p.seeds == [None] and p.offsets == [c3] for p in plans)
    assert plans[0].channels_total == 8192 and sum(p.channels_per_gpu for p in plans) == 8192
    # configs[4]: 16 independent streams x 1 024 channels, 2 per GPU, seeds 1..16 each used once, nothing broadcast
    plans = [sdist.plan("c5", r, 8, 1200) for r in range(8)]
    assert all(not p.broadcast and p.offsets == [c3, c3] for p in plans)
    assert sorted(s for p in plans for s in p.seeds) == list(range(1, 17))
    assert plans[0].channels_total == 16 * 1024 and all(p.channels_per_gpu == 2048 for p in plans)
    # the headline job: C instances per GPU, a contiguous range of the cycled offset list each, never deduplicated
    plans = [sdist.plan("headline", r, 4, 1200, 10000) for r in range(4)]
    whole = chains.offsets_cycled(40000, 1200)
    assert [o for p in plans for o in p.offsets[0]] == whole and plans[0].channels_total == 40000
    with pytest.raises(ValueError):
        sdist.plan("c6", 0, 1, 1200)


def test_block_checksum_sees_every_kind_of_damage():
    x = synth.synth_iq(3, 0, 4096, 0.25)
    want = sdist.block_checksum(x)
    assert sdist.block_checksum(x.copy()) == want == sdist.block_checksum(torch.from_numpy(x.view(np.float32).reshape(-1, 2).copy()))
    y = x.copy(); y[100] = y[100] + np.complex64(1e-7)
    z = x.copy(); z[[5, 6]] = z[[6, 5]]                      # two samples exchanged
    w = np.roll(x, 1)
    assert len({want, sdist.block_checksum(y), sdist.block_checksum(z), sdist.block_checksum(w), sdist.block_checksum(x[:-1])}) == 5


def _worker_c5(rank, world, port, out_dir):
    """configs[4] under gloo, the oracle as each bank: every rank runs the two streams its plan names; nothing is communicated
    but the barrier and the timing."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ffi
    orc = ffi.load("oracle")
    ch = chains.chain("2.4")
    plan = sdist.plan("c5", rank, world, ch.N)
    L = ch.block_len
    res = {}
    for offs, seed in zip(plan.offsets, plan.seeds):
        offs = offs[500:505]                                 # (five channels of the 1 024 keep the CPU test short)
        bank = orc.bank([orc.make_translator(ch.N, ch.z, o) if o else None for o in offs], ch.stage_tuples())
        outs = [bank.process(synth.synth_iq(seed, b * L, L, 0.25), threads=1) for b in range(2)]
        res[f"iq{seed}"] = np.concatenate([o[0] for o in outs], axis=1)
    dist.barrier()
    np.savez(os.path.join(out_dir, f"c5_rank{rank}.npz"), **res)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_independent_streams_equal_a_single_process(oracle, tmp_path):
    world = 2
    mp.spawn(_worker_c5, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ch = chains.chain("2.4")
    L = ch.block_len
    offs = chains.offsets_for_config("c3")[500:505]
    tabs = [oracle.make_translator(ch.N, ch.z, o) if o else None for o in offs]
    seen = set()
    for r in range(world):
        z = np.load(tmp_path / f"c5_rank{r}.npz")
        for key in z.files:
            seed = int(key[2:])
            seen.add(seed)
            bank = oracle.bank(tabs, ch.stage_tuples())
            ref = np.concatenate([bank.process(synth.synth_iq(seed, b * L, L, 0.25), threads=1)[0] for b in range(2)], axis=1)
            assert np.array_equal(z[key].view(np.uint32), ref.view(np.uint32))
    assert seen == {1, 2, 3, 4}


@pytest.mark.timeout(300)
@pytest.mark.parametrize("fanout", ["host", "rccl"])
def test_bench_fanouts_under_two_gloo_ranks(fanout):
    """The two ways the one stream's block reaches every rank (bench.py --fanout): host = rank 0's blocks in a ring in shared host
    memory that every rank maps and reads itself (default: no communicator in the data path, SURVEY 8e's "independent pinned-host
    H2D copies"), rccl = double-buffered broadcast.  On CPU: the transport and the input check of every block on every rank."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch", "--steps", "9", "--config", "c4",
                        "--fanout", fanout], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["fanout"] == fanout
    assert d["input_check"]["blocks_wrong"] == 0 and d["input_check"]["blocks_checked"] == 18
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("sdrx_selftest_ring_")]      # nothing left behind


@pytest.mark.timeout(300)
@pytest.mark.parametrize("config", ["c4", "c5"])
def test_bench_configs_run_under_two_gloo_ranks(config):
    """`python bench.py --gpus 2 --config c4|c5 --selftest-launch`: launcher, plan, block transport (broadcast resp. the ranks'
    own streams), the input check of every block on every rank and the JSON contract, on CPU."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch", "--steps", "6", "--config", config],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["config"]["name"] == config
    assert d["config"]["channels_total"] == (2048 if config == "c4" else 4096)
    assert d["input_check"]["blocks_wrong"] == 0 and d["input_check"]["blocks_checked"] == (12 if config == "c4" else 24)
    assert ("configs[3]" if config == "c4" else "configs[4]") in d["config"]["workload"]


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
