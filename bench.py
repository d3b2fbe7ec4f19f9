#!/usr/bin/env python3
"""bench.py -- throughput of the sdrx channelizer hot path on MI355X.

A "step" = one 32 ms block of synthetic 10 MS/s IQ (320 000 complex<float>, already resident in
HBM) through every channel of this GPU's bank: NCO mixer -> 4-stage FIR decimation -> IF-AGC ->
AM envelope (-> LF-AGC), i.e. the reference's data_cb channel loop (src/sdrx.cpp:327-346) plus
alsa_write_cb's per-sample lines (455-463).  Metric (BASELINE.json): IQ Msamples/s ingested x
channels demodulated = million channel-samples per second, whole job.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--channels C] [--config headline|c4|c5]

--config: headline (default) = north_star's target, C channel instances per GPU from one stream; c4 = BASELINE configs[3]
(8 192 channels = 1 024 per GPU from one stream, block broadcast); c5 = BASELINE configs[4] (16 independent streams x 1 024
channels = 2 banks per GPU, blocks from page-locked host memory, no broadcast).  Same JSON contract for all three;
config.workload names the configuration.  Outside the timed region every rank checks the blocks it received against
the generator (input_check in the JSON line; a wrong block makes the run fail).

N > 1: one rank per GPU over RCCL.  Started as `python bench.py --gpus N` this process touches no GPU: it
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's JSON line
and exits with the child's code.  Started under torch.distributed.run (RANK / WORLD_SIZE in the environment,
as the driver does) it is a rank.  Channels shard across ranks (weak scaling, C channel instances per GPU, no
data-path collective); rank 0 owns the IQ stream and block k+1 reaches the other GPUs by one RCCL broadcast
(2.56 MB per 32 ms of signal, the only traffic over xGMI) into the banks' other input ring WHILE block k is
being processed.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
FP32_PEAK_TF = 157.3      # MI355X fp32 vector peak (MI355X_MICROARCH.md; v_pk_fma_f32 / v_fma_f32 measure 116-123 TF in
                          # tools/ubench_valu.hip with >= 2 waves per SIMD: profiles/r02_ubench_valu.txt)
SETTLE_STEPS = 120


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--channels", type=int, default=10000, help="channel instances per GPU")
    ap.add_argument("--config", choices=("headline", "c4", "c5"), default="headline",
                    help="headline: --channels instances per GPU from one stream; c4 / c5: BASELINE configs[3] / configs[4] (see module docstring)")
    ap.add_argument("--rate", default="10")
    ap.add_argument("--block-hint", type=int, default=0)
    ap.add_argument("--fused", type=int, default=0, help="tuning.fused_tail: 0 auto, 1 separate tail kernel, 2 require the fused path")
    ap.add_argument("--pair", type=int, default=0, help="tuning.pair_lanes: 0 auto, 1 one channel per lane, 2 require two channels per lane")
    ap.add_argument("--matrix", type=int, default=0, help="tuning.matrix_pipe: 0 auto, 1 vector kernels only, 2 require k_stage12m (mixer + stage 1 on the f32 matrix pipe; k_tail_s behind it), 3 require k_stage123m (+ stages 2-3 in the same kernel, no stage-2 scratch; k_tail_c behind it)")
    ap.add_argument("--mgs", type=int, default=0, help="tuning.m_slot_groups (k_stage12m workgroup shape), 0 auto")
    ap.add_argument("--mgt", type=int, default=0, help="tuning.m_seg_groups, 0 auto")
    ap.add_argument("--tail-tile", type=int, default=0, help="tuning.tail_tile (final outputs per workgroup of the separate tail kernel), 0 auto")
    ap.add_argument("--tail-agc", type=int, default=0, help="tuning.tail_agc: 0 auto, 1 k_tail_s + k_agc, 2 require k_tail_agc (tail stages + AGC in one kernel)")
    ap.add_argument("--tail-agc-cpg", type=int, default=0, help="tuning.tail_agc_cpg (channels per workgroup of k_tail_agc), 0 auto")
    ap.add_argument("--y2-keep-mb", type=int, default=0, help="tuning.y2_keep_mb (k_stage12m's scratch stores: 0 auto = the last 192 MB of a launch ordinary, the rest streaming; 1 = all streaming; 4294967295 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--pipelined", action="store_true",
                    help="SDRXGPU_FLAG_PIPELINED: call k returns block k-1 (measured: no gain, the two kernels compete for CUs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the RCCL fan-out even with one rank (rehearsal of the N > 1 path on one GPU)")
    ap.add_argument("--fanout", choices=("host", "rccl", "pipelined", "inline"), default="host",
                    help="how the one IQ stream's block reaches every rank: host (default) = rank 0's blocks lie in a shared page-locked "
                         "ring in host memory and every rank copies its own H2D on its bank's copy stream (sdrxgpu_bank_process_pinned; "
                         "SURVEY 8e: 'simply 8 independent pinned-host H2D copies'; no RCCL communicator at all, gloo carries the "
                         "barrier and the max-over-ranks); rccl (= pipelined) = RCCL broadcast of block k+1 over xGMI into the banks' "
                         "other input ring beside the kernels of block k; inline = that broadcast in line before block k's kernels (A/B)")
    ap.add_argument("--ranks-on-one-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses GPU 0 (host fan-out only: RCCL refuses two ranks on one device)")
    ap.add_argument("--multibank", type=int, default=0,
                    help="D > 0: ONE process, the channel list cut over D devices by sdrxgpu_multibank (--channels per device; devices "
                         "0..D-1, or D shards on GPU 0 when the box has fewer): the in-process form of the multi-GPU path a drop-in sdrx uses")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="CPU-only check of the launcher and the fan-out: gloo ranks, no GPU, no channelizer work")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# parent mode: start the ranks
# ------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(args):
    """`--gpus N` without a launcher around us: be the launcher.  This process never initialises a GPU (nothing
    below imports torch.cuda state or loads libsdrxgpu); the ranks are fresh child processes."""
    # (--standalone: torchrun's own rendezvous picks and binds a free port itself -- no window between finding a port and using it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line_json = None
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            line_json = line.strip()
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if line_json:
        print(line_json, flush=True)
    sys.exit(rc if rc else (0 if line_json else 1))


# ------------------------------------------------------------------------------------------------
# stdout carries ONE line
# ------------------------------------------------------------------------------------------------
_REAL_STDOUT = None


def quiet_stdout():
    """Libraries print banners on stdout (RCCL's version block, gloo's "connected to N peer ranks"); the contract is one JSON
    line there.  From here on file descriptor 1 goes where stderr goes; emit() writes to the real one."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(line: str):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


# ------------------------------------------------------------------------------------------------
# accounting
# ------------------------------------------------------------------------------------------------
def algorithmic_bytes_per_channel_sample(m_total: int) -> float:
    """8 B read (one complex<float> in) + (8 + 4)/M B written (decimated IQ + audio)."""
    return 8.0 + 12.0 / m_total


def flops_per_channel_sample(chain, n_stages):
    """fp32 flop per input sample of one channel, as the reference's folded FIR does them (msd.hpp:309-332): mixer =
    one complex multiply (6); a stage of T taps, half = (T-1)/2: (half pair sums + (half+1) products + (half+1)
    accumulations) x 2 parts per output, one output per prod(m) input samples."""
    total, dec = 6.0, 1
    for st in chain.stages[:n_stages]:
        half = (len(st.h) - 1) // 2
        dec *= st.m
        total += 2.0 * (half + 2 * (half + 1)) / dec
    return total


def profile_facts(n_ch, fused, pair, matrix=False, fused3=False):
    """HBM bytes per k_stage12 launch and the clock the chip sustains, from the rocprofv3 --pmc passes of this build
    (tools_pmc.sh -> profiles/*_traffic.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    None when no profile of this channel count / kernel path is committed."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            if (d.get("channels") == n_ch and bool(d.get("fused_tail", False)) == bool(fused) and bool(d.get("pair_lanes", False)) == bool(pair)
                    and bool(d.get("matrix_pipe", False)) == bool(matrix) and bool(d.get("fused3", False)) == bool(fused3)):
                return d, os.path.basename(f)
        except (OSError, ValueError):
            pass
    return None, None


def cpu_baseline(chain, seconds=12.0):
    """The oracle's C restatement of the reference CPU path (kind "port": oracle/sdrx_oracle.c built -O3 -march=native,
    bit-identical to the compiled reference on every golden vector), timed on this box's host cores on a bounded sample
    of the same workload.  When the compiled reference itself travelled here (oracle/_ref, git-ignored) it is timed
    too and reported under "reference_build".  This is the ONLY place bench.py touches oracle/."""
    from oracle import ffi
    from sdrx_amd import chains, synth
    ffi.build(ref=False)
    libs = [("port", ffi.load("oracle_fast"))]
    if ffi.have_ref() and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsdrx_ref_avx2.so")):
        try:
            libs.append(("reference", ffi.load("ref_avx2")))
        except OSError:
            pass
    n_thr = max(1, min(len(os.sched_getaffinity(0)), 64))
    n_ch = 16 * n_thr
    offs = chains.offsets_cycled(n_ch, chain.N)
    hlp = ffi.load("oracle")
    tabs = [hlp.make_translator(chain.N, chain.z, o) if o else None for o in offs]
    x = synth.synth_iq(1, 0, chain.block_len, 0.25)

    def run(lib, threads, n_channels, budget):
        b = lib.bank(tabs[:n_channels], chain.stage_tuples())
        b.process(x, threads=threads)           # warm-up block
        t0 = time.perf_counter()
        blocks = 0
        while True:
            b.process(x, threads=threads)
            blocks += 1
            if time.perf_counter() - t0 >= budget and blocks >= 2:
                break
        dt = time.perf_counter() - t0
        return n_channels * chain.block_len * blocks / dt, blocks

    share = seconds / len(libs)
    res = {}
    for kind, lib in libs:
        multi, nb_m = run(lib, n_thr, n_ch, share * 0.6)
        single, nb_s = run(lib, 1, 16, share * 0.4)
        res[kind] = {"value": multi / 1e6, "unit": "Mch-samples/s", "cores": n_thr, "kind": kind,
                     "sample": f"{n_ch} channels x {nb_m} blocks of {chain.block_len} samples, one thread per channel range "
                               f"(reference -t/DS model); single thread on 16 channels x {nb_s} blocks: {single / 1e6:.1f} Mch-samples/s",
                     "single_core_value": single / 1e6}
    out = res["port"]
    if "reference" in res:
        out["reference_build"] = res["reference"]
    return out


# ------------------------------------------------------------------------------------------------
# launcher / fan-out self-test (CPU, gloo)
# ------------------------------------------------------------------------------------------------
def selftest_launch(args):
    """What `--gpus N` does around the channelizer, without a GPU: N gloo ranks, the configuration's plan (which channels,
    which streams on which rank), the double-buffered fan-out of the IQ blocks (headline / c4) or the ranks' own streams
    (c5), every received block checked against the generator, the barrier / max-over-ranks timing and rank 0's JSON
    line.  No channelizer work (value = 0)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from sdrx_amd import chains
    from sdrx_amd import dist as sdist
    from sdrx_amd import synth
    rank, _, world = sdist.env_rank_world()
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo")
    ch = chains.chain(args.rate)
    plan = sdist.plan(args.config, rank, world, ch.N, args.channels)
    L = 4096
    steps = args.steps

    def block(seed, k):
        return torch.from_numpy(synth.synth_iq(seed, k * L, L, 0.25).view(np.float32).reshape(L, 2).copy())

    bad = checked = 0
    dist.barrier()
    t0 = time.perf_counter()
    if plan.broadcast and args.fanout == "host":
        # host fan-out: rank 0 lays the stream's blocks into a ring in shared host memory, every rank maps it and reads its blocks
        # there by itself (on a GPU: its own H2D copy); nothing else crosses ranks.  Only rank 0 generates; the others check what
        # they READ against what the generator makes.
        n_ring = 4
        shm_path = "/dev/shm/sdrx_selftest_ring_%s" % os.environ.get("MASTER_PORT", "0")
        if rank == 0:
            mm = np.memmap(shm_path, dtype=np.float32, mode="w+", shape=(n_ring, L, 2))
            for b in range(n_ring):
                mm[b] = block(7, b).numpy()
            mm.flush()
        dist.barrier()
        if rank != 0:
            mm = np.memmap(shm_path, dtype=np.float32, mode="r+", shape=(n_ring, L, 2))
        dist.barrier()
        if rank == 0:
            os.unlink(shm_path)
        for k in range(steps):
            bad += int(sdist.block_checksum(np.array(mm[k % n_ring])) != sdist.block_checksum(block(7, k % n_ring)))
            checked += 1
    elif plan.broadcast:
        bufs = [torch.empty(L, 2), torch.empty(L, 2)]
        fan = sdist.PipelinedFanout(dist, bufs, src=0)
        fan.post(0, block(7, 0) if rank == 0 else None)
        for k in range(steps):
            cur = fan.get(k)
            if k + 1 < steps:
                fan.post(k + 1, block(7, k + 1) if rank == 0 else None)
            # every rank regenerates the block it should have received and compares checksums
            bad += int(sdist.block_checksum(cur) != sdist.block_checksum(block(7, k)))
            checked += 1
            fan.done(k)
    else:
        # every bank has a stream of its own, generated where it is consumed: nothing crosses ranks.  What is checked: the
        # seeds of the job's streams are all different and each arrives in its bank's buffer unchanged
        rings = [torch.empty(L, 2) for _ in plan.seeds]
        for k in range(steps):
            for i, seed in enumerate(plan.seeds):
                rings[i].copy_(block(seed, k))
                bad += int(sdist.block_checksum(rings[i]) != sdist.block_checksum(block(seed, k)))
                checked += 1
        seeds = [None] * world
        dist.all_gather_object(seeds, plan.seeds)
        flat = [x for sub in seeds for x in sub]
        bad += int(len(set(flat)) != len(flat) or sorted(flat) != list(range(1, 2 * world + 1)))
    dist.barrier()
    dt = sdist.max_over_ranks(dist, time.perf_counter() - t0)
    t = torch.tensor([bad, checked, plan.channels_per_gpu], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"metric": "IQ Msamples/s ingested x channels demodulated (channel-samples/s)", "value": 0.0,
                          "unit": "Mch-samples/s", "n_gpus": world, "steps": steps, "warmup": 0, "ms_per_step": dt / steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "selftest (gloo, CPU; launcher, plan, block transport and input check only) of: " + plan.workload,
                                     "name": plan.name, "channels_total": int(t[2].item()), "parallelism": plan.parallelism + f" ({world} gloo ranks)",
                                     "fanout": args.fanout if plan.broadcast else None},
                          "input_check": {"blocks_checked": int(t[1].item()), "blocks_wrong": int(t[0].item())},
                          "selftest": {"blocks_received_wrong": int(t[0].item())}}), flush=True)
    dist.destroy_process_group()
    sys.exit(1 if int(t[0].item()) else 0)


# ------------------------------------------------------------------------------------------------
# rank mode
# ------------------------------------------------------------------------------------------------
def time_bank(bank, fan, ring, rank, dist, L, d_iq, d_au, steps, warmup, n_ring, sb, sdist, torch, events=True, expect=None, check=None,
              pipelined=False, host_ring=None, red_dev="cuda"):
    """An input-check pass (expect: the generator's checksum of every ring block, known to every rank; check: dict that
    takes the counts), SETTLE + W untimed steps, then exactly `steps` timed steps bracketed by synchronize + barrier.
    events: HIP events around the dominant kernel inside the timed region (the headline configuration: roofline.achieved
    comes from them; they cost ~5 us per step, which a 1 ms step does not notice and a 0.05 ms step does)."""
    state = {"k": 0, "calls": 0}

    def block_for(k):
        return ring[k % n_ring] if rank == 0 else None

    if host_ring is None:
        fan.post(0, block_for(0))

    # Input check, outside the timed region: every block of the ring travels the way the timed blocks do (rank 0's ring ->
    # fan-out -> this rank's bank input ring) and every rank compares the checksum of what ARRIVED with the generator's.
    if expect is not None and host_ring is not None:
        # host fan-out: the block is H2D-copied by this rank itself from the shared page-locked ring; what is checked is what
        # arrived in the bank's input ring (where sdrxgpu_bank_ingest_buffer said the block would land)
        for _ in range(n_ring):
            k = state["k"]
            where = bank.ingest_buffer()[0]
            assert bank.process_pinned(host_ring[k % n_ring], L, d_iq.data_ptr(), d_au.data_ptr(), 512) == 512
            bank.synchronize()
            got = sdist.block_checksum(torch.as_tensor(sb.DeviceView(where, (L, 2)), device="cuda").cpu())
            check["blocks_checked"] += 1
            check["blocks_wrong"] += int(got != expect[k % n_ring])
            state["k"] = k + 1
            state["calls"] += 1
    elif expect is not None:
        for _ in range(n_ring):
            k = state["k"]
            cur = fan.get(k)
            fan.post(k + 1, block_for(k + 1))
            got = sdist.block_checksum(cur.cpu())           # (synchronises the current stream: the block has arrived)
            check["blocks_checked"] += 1
            check["blocks_wrong"] += int(got != expect[k % n_ring])
            fan.done(k)
            state["k"] = k + 1

    def step_host():
        # host fan-out: this rank's own H2D copy of block k from the shared page-locked ring, on the bank's copy stream into the
        # input ring the previous block's kernels are not reading; nothing waits, nothing is exchanged between ranks
        k = state["k"]
        n = bank.process_pinned(host_ring[k % n_ring], L, d_iq.data_ptr(), d_au.data_ptr(), 512)
        assert n == 512
        state["k"] = k + 1
        state["calls"] += 1

    def step_fan():
        k = state["k"]
        cur = fan.get(k)                      # N > 1: the compute stream waits for block k's broadcast (issued a step ago)
        if not pipelined:
            fan.post(k + 1, block_for(k + 1))     # block k+1 starts moving into the bank's other input ring
        n = bank.process_device(cur.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
        if pipelined:
            # a pipelined bank reads the input of call k-1 until call k has been enqueued (include/sdrx_gpu.h): only now is
            # ring (k-1) % 2 free for block k+1
            if k > 0:
                fan.done(k - 1)
            fan.post(k + 1, block_for(k + 1))
        else:
            fan.done(k)
        assert n == 512 or (state["calls"] == 0 and n == 0)      # a pipelined bank returns block k-1 at call k
        state["k"] = k + 1
        state["calls"] += 1

    step = step_host if host_ring is not None else step_fan

    # Clock settling, untimed and in front of the W warm-up steps: from idle the chip needs ~50 ms of this load before
    # its clock is steady (measured per step: 1.24, 1.35, 1.61, 1.52, ... 1.22 at step 12, 1.15 from step ~40 on).
    # A fixed number of steps, the same on every rank (each step is a collective when N > 1).
    for _ in range(SETTLE_STEPS + warmup):
        step()
    bank.synchronize()          # (the bank's own streams: copy stream and, pipelined, its second chain)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    bank.profile_reset()
    if events:
        bank.profile_enable(2)  # HIP events around the dominant kernel only (roofline.achieved comes from the timed region)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    bank.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt = sdist.max_over_ranks(dist, dt, device=red_dev)
    k = {}
    if events:
        ms, n = bank.profile_read(sb.K_STAGE12)
        k["k_stage12"] = (ms / max(n, 1), int(n))
    # the smaller kernels (events=False: all kernels): a few more steps after the timed region, every kernel bracketed
    # (three steps first that are not counted: the barrier above let the chip's clock drop, and five steps of a 0.1 ms bank averaged
    # over that ramp read 50 % high)
    for _ in range(min(3, steps)):
        step()
    bank.profile_reset()
    bank.profile_enable(1)
    for _ in range(min(10, steps)):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    for name, kid in (("k_stage12", sb.K_STAGE12), ("k_tail", sb.K_TAIL), ("k_agc", sb.K_AGC)):
        if name in k:
            continue
        ms, n = bank.profile_read(kid)
        k[name] = (ms / max(n, 1), int(n))
    bank.profile_enable(0)
    if host_ring is None:
        fan.get(state["k"])      # drain the broadcast that is still in flight
    bank.synchronize()
    torch.cuda.synchronize()
    return dt, k


def run_c5(args, plan, ch, rank, local_rank, world, dist, torch, np, sb, sdist, synth):
    """BASELINE configs[4]: every rank drives 2 banks of 1 024 channels, each from an IQ stream of its own that arrives from
    page-locked host memory (sdrxgpu_bank_process_pinned: the H2D copy of a bank's next block runs on its copy stream beside
    the other bank's kernels).  No broadcast, nothing between GPUs; a step = one block through BOTH banks of every rank."""
    L = ch.block_len
    n_host = 4                                                     # distinct page-locked blocks per stream, used in turn
    banks, hosts, outs, expect = [], [], [], []
    for offs, seed in zip(plan.offsets, plan.seeds):
        tabs = sb.translators_for(ch, offs)
        banks.append(sb.ChannelBank(tabs, ch.stage_tuples(), device=local_rank, tuning=args.tuning))
        blocks = [synth.synth_iq(seed, b * L, L, 0.25).view(np.float32).reshape(L, 2).copy() for b in range(n_host)]
        expect.append([sdist.block_checksum(b_) for b_ in blocks])
        hosts.append([torch.from_numpy(b_).pin_memory() for b_ in blocks])
        outs.append((torch.empty(len(offs), 512, 2, device="cuda"), torch.empty(len(offs), 512, device="cuda")))
    state = {"k": 0}

    def step():
        k = state["k"]
        for i, bk in enumerate(banks):
            n = bk.process_pinned(hosts[i][k % n_host].data_ptr(), L, outs[i][0].data_ptr(), outs[i][1].data_ptr(), 512)
            assert n == 512
        state["k"] = k + 1

    # input check, outside the timed region: what arrived in each bank's input ring against the generator
    check = {"blocks_checked": 0, "blocks_wrong": 0}
    for _ in range(n_host):
        k = state["k"]
        where = [bk.ingest_buffer()[0] for bk in banks]            # where the NEXT block of each bank will be written
        step()
        for i, bk in enumerate(banks):
            bk.synchronize()
            got = torch.as_tensor(sb.DeviceView(where[i], (L, 2)), device="cuda").cpu()
            check["blocks_checked"] += 1
            check["blocks_wrong"] += int(sdist.block_checksum(got) != expect[i][k % n_host])
    for _ in range(SETTLE_STEPS + args.warmup):
        step()
    for bk in banks:
        bk.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    banks[0].profile_reset()
    banks[0].profile_enable(2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for bk in banks:
        bk.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = sdist.max_over_ranks(dist, time.perf_counter() - t0, device=args.red_dev)
    k12_ms, k12_n = banks[0].profile_read(sb.K_STAGE12)
    k12_ms /= max(k12_n, 1)
    banks[0].profile_enable(0)
    infos = [bk.info() for bk in banks]
    for i, bk in enumerate(banks):
        assert bool(torch.isfinite(outs[i][0]).all()) and bool(torch.isfinite(outs[i][1]).all())
        bk.close()
    n_wrong = check["blocks_wrong"]
    if dist is not None:
        tw = torch.tensor([check["blocks_checked"], check["blocks_wrong"]], dtype=torch.int64, device=args.red_dev)
        dist.all_reduce(tw)
        check = {"blocks_checked": int(tw[0].item()), "blocks_wrong": int(tw[1].item())}
        n_wrong = check["blocks_wrong"]
    check["how"] = "every rank: checksum of its banks' blocks as they arrived in the banks' input rings (H2D from page-locked memory) against the generator's"
    if rank == 0:
        n_bank = len(plan.offsets[0])
        units = float(n_bank) * L
        f12 = flops_per_channel_sample(ch, 3 if infos[0].fused_tail else 2)
        ach = f12 * units / (k12_ms / 1e3) / 1e12 if k12_n else None
        value = float(plan.channels_total) * L * args.steps / dt / 1e6
        out = {
            "metric": "IQ Msamples/s ingested x channels demodulated (channel-samples/s)", "value": value, "unit": "Mch-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": plan.workload + f"; {args.rate} MS/s, 32 ms blocks of {L} complex<float>, mixer + 4-stage FIR "
                                   f"decimation /{ch.m_total} + IF-AGC + AM envelope",
                       "name": plan.name, "channels_per_gpu": plan.channels_per_gpu, "channels_total": plan.channels_total,
                       "streams_per_gpu": len(plan.seeds), "sample_rate": ch.fs, "block_len": L, "parallelism": plan.parallelism,
                       "kernel_path": ("mixer + stage 1 on the f32 matrix pipe (k_stage12m)" if infos[0].matrix_pipe else
                                       "vector kernels" + (", two channels per lane" if infos[0].pair_lanes else "")) +
                                      ("; stage 3 fused behind stages 1-2 (k_tail4)" if infos[0].fused_tail else "; separate tail (k_tail_s)"),
                       "includes": "PCIe H2D of every block (the configuration is about that overlap; inputs are NOT resident in HBM "
                                   "when a step starts -- unlike the headline line)",
                       "realtime_factor": value * 1e6 / (plan.channels_total * ch.fs),
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"), "settle_steps": SETTLE_STEPS},
            "roofline": {"bound": "mfma" if infos[0].matrix_pipe else "valu_fp32",
                         "kernel": ("k_stage123m" if infos[0].matrix_pipe == 2 else "k_stage12m" if infos[0].matrix_pipe else
                                    "k_stage12p" if infos[0].pair_lanes else "k_stage12") + " (one bank's launch while the other bank's kernels share the chip)",
                         "achieved": ach, "peak": FP32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP32_PEAK_TF if ach else None,
                         "flops_per_unit": f12, "traffic": None, "launches": int(k12_n),
                         "whole_step_tflops": flops_per_channel_sample(ch, len(ch.stages)) * units * len(plan.seeds) / (dt / args.steps) / 1e12},
            "input_check": check,
            "cpu_baseline": cpu_baseline(ch, args.cpu_seconds) if (world == 1 and not args.no_cpu_baseline) else None,
        }
        emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    sys.exit("input check failed: %d block(s) arrived different from what the generator made" % n_wrong if n_wrong else 0)


def whole_configs_on_one_gpu(args, ch, device, torch, np, sb, chains, synth, sdist):
    """BASELINE configs[3] (8 192 channels of one stream as 8 shards of 1 024) and configs[4] (16 streams x 1 024 channels, one
    thread per stream) at full size with every shard / bank on ONE GPU: the N-device code paths (sdrxgpu_multibank's worker
    threads and per-shard copies; 16 banks driven concurrently through sdrxgpu_bank_process_pinned) timed where only one device
    exists.  Not a scaling measurement: eight devices and eight PCIe links at once stay unmeasured."""
    import threading
    L = ch.block_len
    out = {}
    n_it = max(30, args.steps)
    # ---- configs[3]: the channel list of dist.plan("c4", r, 8), r = 0..7, cut over 8 shards on this device
    offs = [o for r in range(8) for o in sdist.plan("c4", r, 8, ch.N).offsets[0]]
    tabs = sb.translators_for(ch, offs)
    ring = np.stack([synth.synth_iq(1234, b * L, L, 0.25).view(np.float32).reshape(L, 2) for b in range(4)])
    sb.host_register(ring)
    try:
        for label, tuning in (("as the shards run when they share a device (k_tail_s + k_agc)", None),
                              ("with tuning.tail_agc = 2: every shard runs k_tail_agc, the kernel a shard alone on its GPU takes", {"tail_agc": 2})):
            mb = sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=[device] * 8, tuning=tuning)
            sh = mb.shards()
            outs = [(torch.empty(hi - lo, 512, 2, device="cuda"), torch.empty(hi - lo, 512, device="cuda")) for _, lo, hi, _ in sh]
            p_iq, p_au = [o[0].data_ptr() for o in outs], [o[1].data_ptr() for o in outs]
            for k in range(args.warmup + 40):
                mb.process_pinned(ring[k % 4].ctypes.data, L, p_iq, p_au, 512)
            mb.synchronize()
            t0 = time.perf_counter()
            for k in range(n_it):
                mb.process_pinned(ring[k % 4].ctypes.data, L, p_iq, p_au, 512)
            mb.synchronize()
            dt = time.perf_counter() - t0
            assert all(bool(torch.isfinite(o[1]).all()) for o in outs)
            fam = int(mb.info().matrix_pipe)
            ta = [int(s_[3].tail_agc) for s_ in sh]
            mb.close()
            out["8192 channels of one stream as 8 shards x 1024 on ONE GPU (configs[3] whole; sdrxgpu_multibank, devices = 8 x [0]), " + label] = {
                "value": len(offs) * L * n_it / dt / 1e6, "unit": "Mch-samples/s", "ms_per_step": dt / n_it * 1e3,
                "realtime_factor": L * n_it / dt / ch.fs, "kernel_family": ["vector kernels", "k_stage12m", "k_stage123m"][fam],
                "tail_agc_channels_per_workgroup_by_shard": ta,
                "includes": "8 H2D copies of every block (one per shard) from the caller's page-locked buffer; 8 worker threads",
                "not_measured": "eight devices / eight PCIe links at once (no multi-GPU node was available)"}
    finally:
        sb.host_unregister(ring)
    # ---- configs[4]: 16 streams (seeds 1..16) x configs[2]'s 1 024 channels, one bank and one driving thread per stream
    o = chains.offsets_for_config("c3")
    tabs = sb.translators_for(ch, o)
    n_streams = 16
    banks = [sb.ChannelBank(tabs, ch.stage_tuples(), device=device) for _ in range(n_streams)]
    hosts = [torch.from_numpy(synth.synth_iq(1 + i, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).pin_memory() for i in range(n_streams)]
    outs = [(torch.empty(len(o), 512, 2, device="cuda"), torch.empty(len(o), 512, device="cuda")) for _ in range(n_streams)]
    torch.cuda.synchronize()
    gate = threading.Barrier(n_streams + 1)
    errs = []

    def drive(i, n):
        try:
            gate.wait()
            for _ in range(n):
                banks[i].process_pinned(hosts[i].data_ptr(), L, outs[i][0].data_ptr(), outs[i][1].data_ptr(), 512)
            banks[i].synchronize()
        except Exception as e:      # noqa: BLE001
            errs.append(repr(e))

    def run_threads(n):
        th = [threading.Thread(target=drive, args=(i, n)) for i in range(n_streams)]
        for t in th:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    run_threads(args.warmup + 20)
    dt = run_threads(n_it)
    assert not errs, errs
    assert all(bool(torch.isfinite(o_[1]).all()) for o_ in outs)
    ta = [int(bk.info().tail_agc) for bk in banks]
    for bk in banks:
        bk.close()
    out["16 streams x 1024 channels as 16 banks on ONE GPU, one driving thread per stream (configs[4] whole)"] = {
        "value": n_streams * len(o) * L * n_it / dt / 1e6, "unit": "Mch-samples/s", "ms_per_step": dt / n_it * 1e3,
        "realtime_factor": L * n_it / dt / ch.fs, "tail_agc_channels_per_workgroup_by_bank": ta,
        "includes": "PCIe H2D of every stream's block (sdrxgpu_bank_process_pinned), 16 host threads",
        "not_measured": "eight devices / eight PCIe links at once (no multi-GPU node was available)"}
    return out


def run_multibank(args, torch, np, sb, chains, synth, sdist):
    """--multibank D: ONE process, one channel list of D x --channels instances cut over D devices by sdrxgpu_multibank
    (include/sdrx_gpu.h): per-device state, one H2D copy of the block per device from the caller's page-locked buffer, one worker
    thread per device -- the in-process form of the channel shard, what a drop-in sdrx (one process, one device thread calling
    data_cb, reference src/sdrx.cpp:306-364; workers src/ds.hpp:43-77) gets through the adapter's sdrxgpu::devices().  Devices
    0..D-1, or D shards on GPU 0 when the box has fewer (rehearsal: then n_gpus = 1 and the shards share the one card).
    Two numbers: the throughput entry (page-locked block in, per-shard device rows out, asynchronous) as `value`, and the
    drop-in entry (host pointers in and out, complete on return) under host_pointer_entry."""
    D = args.multibank
    n_dev = torch.cuda.device_count()
    devices = list(range(D)) if D <= n_dev else [0] * D
    ch = chains.chain(args.rate)
    L = ch.block_len
    n_ch = args.channels * D
    offs = chains.offsets_cycled(n_ch, ch.N)
    tabs = sb.translators_for(ch, offs)
    tuning = {"matrix_pipe": args.matrix} if args.matrix else None
    mb = sb.MultiChannelBank(tabs, ch.stage_tuples(), devices=devices, tuning=tuning)
    shards = mb.shards()
    outs = [(torch.empty(hi - lo, 512, 2, device=f"cuda:{dev}"), torch.empty(hi - lo, 512, device=f"cuda:{dev}")) for dev, lo, hi, _ in shards]
    p_iq, p_au = [o[0].data_ptr() for o in outs], [o[1].data_ptr() for o in outs]
    n_ring = 8
    car = [sb.make_translator(ch.N, ch.z, o) for o in (5, -20)]
    ring = np.stack([synth.synth_iq(1234, b * L, L, 0.25, car, [0.3, 0.1], ch.N, ch.fs // 1000).view(np.float32).reshape(L, 2) for b in range(n_ring)])
    sb.host_register(ring)
    ptrs = [ring[b].ctypes.data for b in range(n_ring)]
    state = {"k": 0}

    def step():
        assert mb.process_pinned(ptrs[state["k"] % n_ring], L, p_iq, p_au, 512) == 512
        state["k"] += 1

    for _ in range(SETTLE_STEPS + args.warmup):
        step()
    mb.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    mb.synchronize()
    dt = time.perf_counter() - t0
    for o in outs:
        assert bool(torch.isfinite(o[0]).all()) and bool(torch.isfinite(o[1]).all())
    # the drop-in entry: host pointers in and out (the ring chunk page-locked), all devices, complete on return
    chunk = np.zeros((n_ch, 512), np.complex64)
    audio = np.zeros((n_ch, 512), np.float32)
    x_host = np.ascontiguousarray(ring[0]).view(np.complex64).reshape(L)
    for a_ in (chunk, audio):
        sb.host_register(a_)
    for _ in range(3):
        mb.process(x_host, iq_out=chunk, audio_out=audio)
    n_it = 10
    t1 = time.perf_counter()
    for _ in range(n_it):
        mb.process(x_host, iq_out=chunk, audio_out=audio)
    dth = (time.perf_counter() - t1) / n_it
    info = mb.info()
    fam = int(info.matrix_pipe)
    mb.close()
    value = float(n_ch) * L * args.steps / dt / 1e6
    emit(json.dumps({
        "metric": "IQ Msamples/s ingested x channels demodulated (channel-samples/s)", "value": value, "unit": "Mch-samples/s",
        "n_gpus": len(set(devices)), "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.channels} channel instances per shard x {D} shards = {n_ch} channels from one IQ stream, 8.33 kHz grid; "
                               f"{args.rate} MS/s, 32 ms blocks of {L} complex<float>, mixer + 4-stage FIR decimation /{ch.m_total} + IF-AGC + AM envelope",
                   "name": "multibank", "channels_total": n_ch, "shards": D, "devices": devices,
                   "parallelism": f"ONE process: sdrxgpu_multibank, {D} shard(s) on device(s) {sorted(set(devices))}, one worker thread + one H2D "
                                  "copy of the block per shard from the caller's page-locked buffer, no communicator",
                   "kernel_family": ["vector kernels", "k_stage12m", "k_stage123m"][fam],
                   "includes": "PCIe H2D of every block, once per shard (sdrxgpu_multibank_process_pinned)",
                   "realtime_factor": value * 1e6 / (n_ch * ch.fs), "settle_steps": SETTLE_STEPS},
        "host_pointer_entry": {"what": "sdrxgpu_multibank_process: host pointers in and out (page-locked chunk), every shard's rows D2H into its "
                                       "slice of the chunk, complete on return", "ms_per_block": dth * 1e3, "value": n_ch * L / dth / 1e6,
                               "unit": "Mch-samples/s"},
        "roofline": None, "cpu_baseline": None}))
    sys.exit(0)


def main():
    args = parse_args()
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        spawn_ranks(args)                   # does not return
    if args.selftest_launch:
        selftest_launch(args)               # does not return

    # HIP maps streams onto 4 hardware queues by default; a process that has had more streams than that (this one: torch's,
    # and two per bank) can end up with two banks' compute streams on ONE queue, and their kernels then run one after the
    # other instead of side by side (measured, tools/two_streams.py: 0.24 ms per step of two 1 024-channel banks with
    # distinct queues, 0.33 when they collide; 8 queues: 0.26 either way).  Must be set before the HIP runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    quiet_stdout()
    import numpy as np
    import torch  # before libsdrxgpu: it then binds to the HIP runtime torch ships (one runtime per process)
    from sdrx_amd import bank as sb
    from sdrx_amd import chains, synth
    from sdrx_amd import dist as sdist

    rank, local_rank, world = sdist.env_rank_world()
    if world > 1:
        args.gpus = world
    if args.multibank:
        run_multibank(args, torch, np, sb, chains, synth, sdist)                                 # does not return
    if args.fanout == "pipelined":
        args.fanout = "rccl"
    if args.ranks_on_one_gpu:
        if args.fanout != "host":
            sys.exit("--ranks-on-one-gpu needs --fanout host (RCCL refuses two ranks on one device)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # host fan-out: nothing of the data path goes through a communicator, so there is none: gloo (CPU) carries the barrier, the
    # max-over-ranks and the input-check counts.  rccl / inline: backend nccl (= RCCL), the block travels by broadcast.
    backend = "gloo" if args.fanout == "host" else "nccl"
    args.red_dev = "cpu" if backend == "gloo" else "cuda"
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "gloo":
            # one node by contract: the loopback interface always exists, the container's hostname may not resolve
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ch = chains.chain(args.rate)
    plan = sdist.plan(args.config, rank, world, ch.N, args.channels)
    args.tuning = None
    if args.fused or args.pair or args.matrix or args.mgs or args.mgt or args.tail_tile or args.tail_agc or args.tail_agc_cpg or args.y2_keep_mb:
        args.tuning = {"fused_tail": args.fused, "pair_lanes": args.pair, "matrix_pipe": args.matrix, "m_slot_groups": args.mgs, "m_seg_groups": args.mgt,
                       "tail_tile": args.tail_tile, "tail_agc": args.tail_agc, "tail_agc_cpg": args.tail_agc_cpg, "y2_keep_mb": args.y2_keep_mb}
    if plan.name == "c5":
        run_c5(args, plan, ch, rank, local_rank, world, dist, torch, np, sb, sdist, synth)       # does not return
    n_ch = plan.channels_per_gpu
    L = ch.block_len
    # one explicit (non-default) torch stream carries the bank's kernels and the ring copies; the RCCL broadcast of
    # the NEXT block runs on the fan-out's own stream beside them
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)

    # synthetic IQ stream: a ring of distinct 32 ms blocks in HBM on rank 0 (uniform noise + 2 AM carriers)
    n_ring = 8
    ring = None
    car = [sb.make_translator(ch.N, ch.z, o) for o in (5, -20)]
    # (every rank generates the ring: rank 0 to send it, the others to know what they should receive)
    host = np.stack([synth.synth_iq(1234, b * L, L, 0.25, car, [0.3, 0.1], ch.N, ch.fs // 1000).view(np.float32).reshape(L, 2)
                     for b in range(n_ring)])
    expect = [sdist.block_checksum(host[b]) for b in range(n_ring)]
    check = {"blocks_checked": 0, "blocks_wrong": 0}
    host_ring = None
    use_host = args.fanout == "host" and (world > 1 or args.force_dist)
    if use_host:
        # Host fan-out: rank 0 owns the stream and lays its blocks into a ring in SHARED host memory (what the device thread's
        # double buffer is in sdrx, airspy_dev.cpp:410-414, made visible to the other ranks); every rank maps it, page-locks its
        # mapping and copies block k H2D itself.  The other ranks never see `host`: they read what rank 0 wrote.
        shm_path = "/dev/shm/sdrx_bench_ring_%s" % os.environ.get("MASTER_PORT", "0")
        if rank == 0:
            mm = np.memmap(shm_path, dtype=np.float32, mode="w+", shape=(n_ring, L, 2))
            mm[:] = host
            mm.flush()
        if dist is not None:
            dist.barrier()
        if rank != 0:
            mm = np.memmap(shm_path, dtype=np.float32, mode="r+", shape=(n_ring, L, 2))
        shared = np.asarray(mm)
        sb.host_register(shared)
        host_ring = [shared[b].ctypes.data for b in range(n_ring)]
        if dist is not None:
            dist.barrier()
        if rank == 0:
            os.unlink(shm_path)              # (the mappings keep the memory; nothing is left behind if a rank dies)
    elif rank == 0:
        ring = torch.from_numpy(host).cuda()

    tuning = args.tuning
    fused_seen = {}
    pair_seen = {}
    matrix_seen = {}
    tailagc_seen = {}

    def path_of(n):
        """the kernels a bank of n channels of this run actually launched per block (from sdrxgpu_bank_info)"""
        m, f, pr, ta = matrix_seen[n], fused_seen[n], pair_seen[n], tailagc_seen[n]
        k12 = "k_stage123m" if m == 2 else "k_stage12m" if m == 1 else "k_stage12p" if pr else "k_stage12"
        if m == 2:
            return k12 + " + k_tail_c + k_agc"
        if f:
            return k12 + " (stage 3 fused) + k_tail4 + k_agc"
        return k12 + (f" + k_tail_agc (tail stages and AGC/demod in one kernel, {ta} channel(s) per workgroup)" if ta else " + k_tail_s + k_agc")

    def run(offsets, steps, warmup, events=True, tuning_override=None, checked=False):
        tabs = sb.translators_for(ch, offsets)
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), device=local_rank, stream=tstream.cuda_stream, block_hint=args.block_hint,
                              pipelined=args.pipelined, tuning=tuning_override or tuning)
        fused_seen[len(offsets)] = int(bank.info().fused_tail)
        pair_seen[len(offsets)] = int(bank.info().pair_lanes)
        matrix_seen[len(offsets)] = int(bank.info().matrix_pipe)
        tailagc_seen[len(offsets)] = int(bank.info().tail_agc)
        # N > 1: the broadcasts land directly in the bank's two input rings, alternately (their addresses repeat
        # for whole blocks), so process_device finds each block in place; any other address would be copied in
        p0, cap = bank.ingest_buffer()
        p1 = bank.ingest_buffer_after(L)
        recv = [torch.as_tensor(sb.DeviceView(p, (L, 2)), device="cuda") for p in (p0, p1)]
        fan = sdist.PipelinedFanout(dist if not use_host else None, recv, src=0, force=args.force_dist, inline=args.fanout == "inline")
        d_iq = torch.empty(len(offsets), 512, 2, dtype=torch.float32, device="cuda")
        d_au = torch.empty(len(offsets), 512, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        dt, k = time_bank(bank, fan, ring, rank, dist, L, d_iq, d_au, steps, warmup, n_ring, sb, sdist, torch, events=events,
                          expect=expect if checked else None, check=check, pipelined=args.pipelined, host_ring=host_ring, red_dev=args.red_dev)
        if args.pipelined:
            assert bank.flush(d_iq.data_ptr(), d_au.data_ptr(), 512) == 512
            torch.cuda.synchronize()
        assert bool(torch.isfinite(d_au).all()) and bool(torch.isfinite(d_iq).all())
        bank.close()
        return dt, k

    # this rank's shard of the job's channel list (instances cycle through all N distinct offsets; never deduplicated)
    offs = plan.offsets[0]
    assert len(offs) == n_ch
    dt, k = run(offs, args.steps, args.warmup, checked=True)
    n_wrong = check["blocks_wrong"]
    if dist is not None:
        tw = torch.tensor([check["blocks_checked"], check["blocks_wrong"]], dtype=torch.int64, device=args.red_dev)
        dist.all_reduce(tw)
        check = {"blocks_checked": int(tw[0].item()), "blocks_wrong": int(tw[1].item())}
        n_wrong = check["blocks_wrong"]
    check["how"] = ("every rank: checksum of each of the ring's blocks as it arrived in the bank's input ring (rank 0's ring -> "
                    "fan-out -> ingest buffer) against the generator's, before the settling steps")
    fused = fused_seen[n_ch]
    matrix_main, pair_main = matrix_seen[n_ch], pair_seen[n_ch]      # (of THIS run: later runs of other paths overwrite the dictionaries)
    tailagc_main = tailagc_seen[n_ch]
    path_main = path_of(n_ch)

    extra = {}
    if world == 1 and not args.no_extra_configs and plan.name == "headline":
        # BASELINE.json's other single-GPU configurations, same step definition (parity-test sizes; short runs)
        for name, key in (("64 channels, 8.33 kHz grid (configs[1])", "c2"), ("1024 channels, mixed 8.33/25 kHz (configs[2])", "c3")):
            o = chains.offsets_for_config(key)
            dte, ke = run(o, max(10, args.steps), args.warmup, events=False)    # (kernel times: 5 instrumented steps after the timed ones)
            extra[name] = {"value": len(o) * L * max(10, args.steps) / dte / 1e6, "unit": "Mch-samples/s",
                           "ms_per_step": dte / max(10, args.steps) * 1e3, "realtime_factor": L * max(10, args.steps) / dte / ch.fs,
                           "kernel_ms_avg": {n: v[0] for n, v in ke.items()}, "fused_tail": fused_seen[len(o)],
                           "kernel_path": path_of(len(o)),
                           "kernel_ms_note": ("k_tail = k_tail_agc, which contains the AGC recurrence (no k_agc launch)" if tailagc_seen[len(o)] else None)}
            # HBM bytes per step from the committed rocprofv3 --pmc passes of a bank of this size and kernel path (tools_pmc.sh)
            pf, pf_src = profile_facts(len(o), fused_seen[len(o)], pair_seen[len(o)], bool(matrix_seen[len(o)]), matrix_seen[len(o)] == 2)
            if pf:
                extra[name]["hbm_traffic_step_bytes"] = sum(pf[kk]["hbm_bytes_per_launch"] for kk in ("k_stage12", "k_tail", "k_agc") if kk in pf)
                extra[name]["traffic_source"] = "committed profile profiles/" + pf_src + " (rocprofv3 --pmc passes of this build, not this run)"

    if world == 1 and not args.no_extra_configs and plan.name == "headline" and tuning is None and matrix_main == 1:
        # the same workload with SDRXGPU_FLAG_FUSED_STAGE3 (k_stage123m + k_tail_c: no stage-2 scratch; opt-in because of its rounding,
        # include/sdrx_gpu.h)
        key3 = "the workload with SDRXGPU_FLAG_FUSED_STAGE3 (stages 2-3 in the stage-1 kernel, no stage-2 scratch; opt-in)"
        try:
            dtb, kb = run(offs, max(30, args.steps), args.warmup, tuning_override={"matrix_pipe": 3})
            nb_steps = max(30, args.steps)
            extra[key3] = {"value": n_ch * L * nb_steps / dtb / 1e6, "unit": "Mch-samples/s", "ms_per_step": dtb / nb_steps * 1e3,
                           "realtime_factor": L * nb_steps / dtb / ch.fs, "kernel_ms_avg": {n: v[0] for n, v in kb.items()}}
        except sb.SdrxGpuError as e:          # (e.g. --block-hint N: k_stage123m takes whole segments per lane only)
            extra[key3] = {"refused": str(e)}

    if world == 1 and not args.no_extra_configs and plan.name == "headline" and not use_host:
        # BASELINE configs[3], one GPU's share (= --config c4 at N = 1): configs[2]'s 1 024 channels with every block arriving the
        # way an N-GPU run's blocks do -- H2D by this rank from a page-locked ring (the host fan-out) -- and, for comparison, the
        # same bank with SDRXGPU_FLAG_PIPELINED (call k returns block k-1; two free-running chains)
        o = chains.offsets_for_config("c3")
        hb_ring = np.ascontiguousarray(host)
        sb.host_register(hb_ring)
        n_it = max(100, args.steps)
        saved, saved_p = host_ring, args.pipelined
        try:
            host_ring = [hb_ring[b_].ctypes.data for b_ in range(n_ring)]
            dte, ke = run(o, n_it, args.warmup, events=False)
            extra["1024 channels per GPU, every block H2D by the rank itself from a page-locked ring (configs[3] per GPU = --config c4 --fanout host at N = 1)"] = {
                "value": len(o) * L * n_it / dte / 1e6, "unit": "Mch-samples/s", "ms_per_step": dte / n_it * 1e3,
                "realtime_factor": L * n_it / dte / ch.fs, "kernel_ms_avg": {n: v[0] for n, v in ke.items()}, "kernel_path": path_of(len(o)),
                "includes": "PCIe H2D of every block (sdrxgpu_bank_process_pinned)"}
            host_ring, args.pipelined = None, True
            dte, ke = run(o, n_it, args.warmup, events=False)
            extra["1024 channels (configs[2]) with SDRXGPU_FLAG_PIPELINED: call k returns block k-1"] = {
                "value": len(o) * L * n_it / dte / 1e6, "unit": "Mch-samples/s", "ms_per_step": dte / n_it * 1e3,
                "realtime_factor": L * n_it / dte / ch.fs, "call_semantics": "pipelined: call k returns block k-1 (bit-identical outputs, one block of latency)"}
        finally:
            host_ring, args.pipelined = saved, saved_p
            sb.host_unregister(hb_ring)

    if world == 1 and not args.no_extra_configs and plan.name == "headline":
        # BASELINE configs[4], one GPU's share: 2 independent 10 MS/s streams x 1024 channels, each stream's block
        # arriving from pinned host memory: sdrxgpu_bank_process_pinned enqueues the H2D copy into the bank's free input
        # ring on the bank's copy stream and the kernels behind it, and returns without waiting
        n_streams, o = 2, chains.offsets_for_config("c3")
        tabs = sb.translators_for(ch, o)
        banks = [sb.ChannelBank(tabs, ch.stage_tuples(), device=local_rank) for _ in range(n_streams)]
        hosts = [torch.from_numpy(synth.synth_iq(1 + i, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).pin_memory() for i in range(n_streams)]
        outs = [(torch.empty(len(o), 512, 2, device="cuda"), torch.empty(len(o), 512, device="cuda")) for _ in range(n_streams)]

        def multi_step():
            for i in range(n_streams):
                banks[i].process_pinned(hosts[i].data_ptr(), L, outs[i][0].data_ptr(), outs[i][1].data_ptr(), 512)

        # (warm-up long enough for the clock to settle: ~75 ms of this load -- timed 20 steps after the light configurations
        # above, the loop read 0.33-0.38 ms where it runs at 0.24)
        for _ in range(args.warmup + 300):
            multi_step()
        for bk in banks:
            bk.synchronize()
        n_it = max(300, args.steps)
        t0 = time.perf_counter()
        for _ in range(n_it):
            multi_step()
        for bk in banks:
            bk.synchronize()
        dtm = time.perf_counter() - t0
        extra["2 streams x 1024 channels, H2D from pinned host overlapped (configs[4] per GPU)"] = {
            "value": n_streams * len(o) * L * n_it / dtm / 1e6, "unit": "Mch-samples/s", "ms_per_step": dtm / n_it * 1e3,
            "realtime_factor": L * n_it / dtm / ch.fs, "includes": "PCIe H2D of every block (sdrxgpu_bank_process_pinned)"}
        for bk in banks:
            bk.close()

    if world == 1 and not args.no_extra_configs and plan.name == "headline":
        # BASELINE configs[3] and configs[4] WHOLE, on this one GPU (what an 8-GPU node runs on eight): every shard / bank of
        # the configuration exists and works, they share one card and one PCIe link
        extra.update(whole_configs_on_one_gpu(args, ch, local_rank, torch, np, sb, chains, synth, sdist))

    host_path = None
    if world == 1 and not args.no_extra_configs and plan.name == "headline":
        # the drop-in entry (host pointers in and out, outputs complete on return): PCIe-inclusive, NOT `value`
        tabs = sb.translators_for(ch, offs)
        hb = sb.ChannelBank(tabs, ch.stage_tuples(), device=local_rank, tuning=tuning)
        x_host = synth.synth_iq(9, 0, L, 0.25)
        o_iq = np.zeros((n_ch, 512), dtype=np.complex64)
        o_au = np.zeros((n_ch, 512), dtype=np.float32)
        res = {}
        for label, pin, kw in (("pageable", False, {}), ("page_locked", True, {}), ("page_locked_audio_only", True, {"iq": False})):
            if pin:
                for a_ in (x_host, o_iq, o_au):
                    sb.host_register(a_)
            io = dict(iq_out=o_iq if kw.get("iq", True) else None, audio_out=o_au)
            for _ in range(2):
                hb.process(x_host, **io)
            n_it = 8
            t0 = time.perf_counter()
            for _ in range(n_it):
                hb.process(x_host, **io)
            dth = (time.perf_counter() - t0) / n_it
            res[label] = {"ms_per_block": dth * 1e3, "value": n_ch * L / dth / 1e6}
            if pin:
                for a_ in (x_host, o_iq, o_au):
                    sb.host_unregister(a_)
        cuts = {"calls_with_iq": hb.host_parts(False), "audio_only_calls": hb.host_parts(True)}
        hb.close()
        for label, nbytes in (("pageable", n_ch * 512 * 12), ("page_locked", n_ch * 512 * 12), ("page_locked_audio_only", n_ch * 512 * 4)):
            # bytes that left the device per second of the WHOLE call (kernels included): the link itself gives 55 - 56 GB/s,
            # profiles/r06_host_entry_timeline.txt (rocprofv3 --memory-copy-trace of this entry)
            res[label]["d2h_gbs"] = nbytes / (res[label]["ms_per_block"] * 1e-3) / 1e9
        host_path = {"what": f"sdrxgpu_bank_process: {L * 8 / 1e6:.2f} MB H2D + {n_ch * 512 * 12 / 1e6:.1f} MB D2H per block "
                             f"({n_ch * 512 * 4 / 1e6:.1f} MB audio-only), outputs complete on return; the bank runs in channel "
                             f"parts cut by the rounds of the stage-1/2 kernel (sdrxgpu_bank_get_host_parts), so that the D2H of one part overlaps the kernels of the next",
                     "channel_parts": cuts, "d2h_gbs_is": "output bytes / whole call time (the copies alone run at 55 - 56 GB/s: the committed copy trace)",
                     "unit": "Mch-samples/s", **res}

    if rank == 0:
        total_cs = float(n_ch) * world * L * args.steps
        value = total_cs / dt / 1e6
        bpu = algorithmic_bytes_per_channel_sample(ch.m_total)
        k12_ms, k12_n = k["k_stage12"]
        units = float(n_ch) * L                                   # channel-samples per launch
        fused3 = matrix_main == 2                           # k_stage123m: stages 1-3 in the dominant kernel, k_tail_c behind it
        f12 = flops_per_channel_sample(ch, 3 if (fused or fused3) else 2)     # what the dominant kernel computes per channel-sample
        f_all = flops_per_channel_sample(ch, len(ch.stages))
        ach_tf = f12 * units / (k12_ms / 1e3) / 1e12 if k12_n else None
        matrix = bool(matrix_main)
        prof, prof_src = profile_facts(n_ch, fused, pair_main, matrix, fused3)
        n_y2, n_y3 = L // (ch.stages[0].m * ch.stages[1].m), L // (ch.stages[0].m * ch.stages[1].m * ch.stages[2].m)
        if fused3:    # k_tail_c: the stage-3 entries k_stage123m left (16 of 8 bytes per segment of 240 stage-1 outputs) in, decimated IQ out
            tail_bytes = n_ch * (8.0 * 16 * ((L // ch.stages[0].m + 239) // 240) + 8.0 * 512)
        elif fused:   # k_tail4: [hist3 | y3] + the range-boundary buffer in, decimated IQ out (k_agc reads the caller's rows)
            tail_bytes = n_ch * (8.0 * (n_y3 + 80) + 8.0 * 40 * ((n_y2 + 767) // 768) + 8.0 * 512)
        else:         # k_tail_s: the stage-2 scratch in, decimated IQ out
            tail_bytes = n_ch * (8.0 * n_y2 + 8.0 * 512)
        kern = {
            "k_stage12": {"ms": k12_ms, "bound": "mfma" if matrix else "valu_fp32", "flops_per_unit": f12, "achieved_tflops": ach_tf,
                          "frac": ach_tf / FP32_PEAK_TF if ach_tf else None,
                          "does": ("k_stage123m: mixer + stage 1 as f32 MFMAs (16x16x4, conjugate pairs share four real products), stages 2-3 as "
                                   "one 119-tap transposed filter on the vector ALU, no stage-2 scratch" if fused3 else
                                   "k_stage12m: mixer + stage 1 as f32 MFMAs (16x16x4, conjugate pairs share four real products), "
                                   "stage 2 on the vector ALU" if matrix else
                                   "mixer + stages 1-2" + (" + stage 3 of the workgroup's own range (fused)" if fused else ""))},
            "k_tail": {"ms": k["k_tail"][0], "bound": "hbm", "algorithmic_bytes": tail_bytes,
                       "achieved_gbs": tail_bytes / (k["k_tail"][0] / 1e3) / 1e9 if k["k_tail"][1] else None,
                       "frac": tail_bytes / (k["k_tail"][0] / 1e3) / 1e9 / HBM_PEAK_GBS if k["k_tail"][1] else None,
                       "frac_is": "the bytes the kernel must move (algorithmic_bytes: its inputs once + its outputs) over the 8 TB/s HBM spec peak; "
                                  "measured traffic: roofline.traffic_step / the committed PMC profile",
                       "does": ("k_tail_c: stage-3 outputs put together from k_stage123m's entries + stage 4" if fused3 else
                                "k_tail4: range-boundary stage-3 outputs + stage 4" if fused else
                                "k_tail_agc: stages 3-4 from the stage-2 scratch AND the AGC / demodulator, one kernel" if tailagc_main else
                                "k_tail_s: stages 3-4 from the stage-2 scratch")},
            "k_agc": ({"ms": None, "note": "no launch: the recurrence runs inside k_tail_agc (k_tail above)"} if tailagc_main else
                      {"ms": k["k_agc"][0], "bound": "latency (512 dependent steps per channel)",
                       "ns_per_step": k["k_agc"][0] * 1e6 / 512 if k["k_agc"][1] else None}),
        }
        out = {
            "metric": "IQ Msamples/s ingested x channels demodulated (channel-samples/s)",
            "value": value, "unit": "Mch-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": plan.workload + f"; {args.rate} MS/s, 32 ms blocks of {L} complex<float>, "
                            f"mixer + 4-stage FIR decimation /{ch.m_total} + IF-AGC + AM envelope",
                "name": plan.name,
                "channels_per_gpu": n_ch, "channels_total": plan.channels_total, "sample_rate": ch.fs, "block_len": L,
                "outputs_per_channel_per_block": 512,
                "parallelism": ("one process on one GPU, no communicator" if dist is None else
                                plan.parallelism.replace("RCCL rank", "rank") if use_host else plan.parallelism) +
                               ((", every rank copies block k+1 H2D itself from rank 0's shared page-locked ring on its bank's copy stream "
                                 "while block k is processed (no communicator, no traffic between GPUs; gloo for the barrier)") if use_host else
                                (", block k+1 broadcast by RCCL over xGMI into the banks' other input ring while block k is processed"
                                 if (world > 1 or args.force_dist) else "")),
                "fanout": (args.fanout if (world > 1 or args.force_dist) else None),
                "includes": ("PCIe H2D of every block (host fan-out: the block is in host memory when a step starts)" if use_host else None),
                "kernel_path": path_main + (f"; {pair_main - 1} conjugate pairs share the mixer" if pair_main else "") +
                               ("; mixer + stage 1 on the f32 matrix pipe" if matrix_main else ""),
                "kernel_flags": {"fused_tail": bool(fused), "pair_lanes": bool(pair_main), "matrix_pipe": bool(matrix_main), "fused3": matrix_main == 2,
                                 "tail_agc": int(tailagc_main)},      # (what tools_pmc.sh files a profile under; profile_facts() finds it by these)
                "realtime_factor": value * 1e6 / (n_ch * world * ch.fs),
                "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),   # (set before the HIP runtime started: two banks' streams need queues of their own)
                "settle_steps": SETTLE_STEPS,    # untimed, before the W warm-up steps (clock settling, see time_bank)
                "call_semantics": "drop-in: call k returns block k" if not args.pipelined else
                                  "pipelined: call k returns block k-1 (bit-identical outputs, one block of latency)",
            },
            "roofline": {
                # the ceiling that binds: fp32 vector issue.  achieved = the flop the reference's
                # algorithm does per channel-sample in this kernel x channel-samples per launch / the kernel's mean
                # duration (HIP events on the bank's stream, inside the timed region)
                # matrix path: the f32 MFMA and the vector ALU share the FP32 datapath (measured additive): one
                # ceiling of 157.3 TFLOP/s for both, the dense f32 MFMA peak of MI355X_MICROARCH.md
                "bound": "mfma" if matrix else "valu_fp32", "kernel": "k_stage123m" if fused3 else "k_stage12m" if matrix else "k_stage12",
                "achieved": ach_tf, "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                "frac": ach_tf / FP32_PEAK_TF if ach_tf else None,
                "flops_per_unit": f12, "flops_per_unit_whole_step": f_all,
                "whole_step_tflops": f_all * units / (dt / args.steps) / 1e12,
                # HBM bytes per launch of the dominant kernel / per step (all three kernels), from the COMMITTED rocprofv3 --pmc
                # passes of this build (tools_pmc.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes) -- not measured in this run
                "traffic": prof.get("k_stage12_hbm_bytes_per_launch") if prof else None,
                "traffic_step": (sum(prof[kk]["hbm_bytes_per_launch"] for kk in ("k_stage12", "k_tail", "k_agc") if kk in prof)
                                 if prof and "k_tail" in prof else None),
                "traffic_source": ("committed profile profiles/" + prof_src + " (rocprofv3 --pmc passes of this build, not this run)") if prof_src else None,
                "sustained_clock_ghz": prof.get("sustained_clock_ghz") if prof else None,
                "launches": int(k12_n),
                "kernels": kern,
                "kernel_ms_note": "k_stage12: HIP events inside the timed region; k_tail / k_agc: from 10 extra steps run right after it (3 uncounted ones first)",
                # north_star's figure, kept under its own name: ALGORITHMIC bytes (every channel-sample counted
                # as 8.02 B although the IQ block is shared by all channels on chip) over the kernel time.  It exceeds the
                # HBM peak by construction and is not a fraction of any ceiling.
                "algorithmic_hbm": {"bytes_per_unit": bpu, "bytes_per_launch": bpu * units,
                                    "gbs_over_k_stage12": bpu * units / (k12_ms / 1e3) / 1e9 if k12_n else None,
                                    "gbs_over_step": bpu * units / (dt / args.steps) / 1e9,
                                    "hbm_peak_gbs": HBM_PEAK_GBS,
                                    "ratio_to_hbm_peak_over_step": bpu * units / (dt / args.steps) / 1e9 / HBM_PEAK_GBS},
            },
        }
        out["input_check"] = check
        if extra:
            out["other_configs"] = extra
        if host_path:
            out["host_pointer_entry"] = host_path
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ch, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if n_wrong:
        sys.exit("input check failed: %d block(s) arrived different from what the generator made" % n_wrong)


if __name__ == "__main__":
    main()
