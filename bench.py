#!/usr/bin/env python3
"""bench.py -- throughput of the sdrx channelizer hot path on MI355X.

A "step" = one 32 ms block of synthetic 10 MS/s IQ (320 000 complex<float>, already resident in
HBM) through every channel of this GPU's bank: NCO mixer -> 4-stage FIR decimation -> IF-AGC ->
AM envelope (-> LF-AGC), i.e. the reference's data_cb channel loop (src/sdrx.cpp:327-346) plus
alsa_write_cb's per-sample lines (455-463).  Metric (BASELINE.json): IQ Msamples/s ingested x
channels demodulated = million channel-samples per second, whole job.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--channels C]

N > 1 is launched by the driver through torch.distributed.run, one rank per GPU: channels shard
across ranks (weak scaling, C channel instances per GPU, no data-path collective); rank 0 owns the
IQ stream and each step's block reaches the other GPUs by one RCCL broadcast (the only traffic
over xGMI, 2.56 MB per 32 ms of signal).
"""
import argparse
import json
import os
import sys
import time

import torch  # first: libsdrxgpu then binds to the HIP runtime torch ships (one runtime per process)
import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sdrx_amd import bank as sb  # noqa: E402
from sdrx_amd import chains, synth  # noqa: E402
from sdrx_amd import dist as sdist  # noqa: E402

FORCE_DIST = os.environ.get("SDRX_BENCH_FORCE_DIST") == "1"
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)


def algorithmic_bytes_per_channel_sample(m_total: int) -> float:
    """8 B read (one complex<float> in) + (8 + 4)/M B written (decimated IQ + audio)."""
    return 8.0 + 12.0 / m_total


def cpu_baseline(chain, seconds=12.0):
    """Reference CPU path timed on this box's host cores on a bounded sample of the same workload.

    Uses the compiled reference (oracle/_ref, kind 'reference') when that build travelled here, else
    the oracle's -O3 build of the C restatement (kind 'port').  This is the ONLY place bench.py touches oracle/."""
    from oracle import ffi
    kind = "port"
    lib = None
    if ffi.have_ref() and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libsdrx_ref_avx2.so")):
        try:
            lib = ffi.load("ref_avx2")
            kind = "reference"
        except OSError:
            lib = None
    if lib is None:
        ffi.build(ref=False)
        lib = ffi.load("oracle_fast")
    n_thr = max(1, min(len(os.sched_getaffinity(0)), 64))
    n_ch = 16 * n_thr
    offs = chains.offsets_cycled(n_ch, chain.N)
    hlp = ffi.load("oracle") if kind == "port" else lib
    tabs = [hlp.make_translator(chain.N, chain.z, o) if o else None for o in offs]
    x = synth.synth_iq(1, 0, chain.block_len, 0.25)

    def run(threads, n_channels, budget):
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

    multi, nb_m = run(n_thr, n_ch, seconds * 0.6)
    single, nb_s = run(1, 16, seconds * 0.4)
    return {
        "value": multi / 1e6, "unit": "Mch-samples/s", "cores": n_thr, "kind": kind,
        "sample": f"{n_ch} channels x {nb_m} blocks of {chain.block_len} samples, one thread per channel range "
                  f"(reference -t/DS model); single thread on 16 channels x {nb_s} blocks: {single / 1e6:.1f} Mch-samples/s",
        "single_core_value": single / 1e6,
    }


SETTLE_STEPS = 120


def time_bank(bank, fan, ring, rank, world, dist, L, d_iq, d_au, steps, warmup, n_ring):
    """W untimed warm-up steps, then exactly `steps` timed steps bracketed by synchronize + barrier."""
    def step(i):
        src = ring[i % n_ring] if rank == 0 else None
        cur = fan.step(src, force=FORCE_DIST)  # N > 1: RCCL broadcast over xGMI of the shared IQ block
        n = bank.process_device(cur.data_ptr(), L, d_iq.data_ptr(), d_au.data_ptr(), 512)
        assert n == 512 or (i == 0 and n == 0)      # a pipelined bank returns block k-1 at call k

    # Clock settling, untimed and in front of the W warm-up steps: from idle the chip needs ~50 ms of this load before
    # its clock is steady (measured per step: 1.24, 1.35, 1.61, 1.52, ... 1.22 at step 12, 1.15 from step ~40 on).
    # A fixed number of steps, the same on every rank (each step is a collective when N > 1).
    for i in range(SETTLE_STEPS):
        step(i)
    for i in range(warmup):
        step(SETTLE_STEPS + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    warmup += SETTLE_STEPS           # (ring index of the timed steps below)
    bank.profile_reset()
    bank.profile_enable(2)      # HIP events around the dominant kernel only (roofline.achieved comes from the timed region)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt = sdist.max_over_ranks(dist, dt, device="cuda")
    k = {}
    ms, n = bank.profile_read(sb.K_STAGE12)
    k["k_stage12"] = (ms / max(n, 1), int(n))
    # the smaller kernels: a few more steps after the timed region, every kernel bracketed
    bank.profile_reset()
    bank.profile_enable(1)
    extra = min(5, steps)
    for i in range(extra):
        step(warmup + steps + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    for name, kid in (("k_tail", sb.K_TAIL), ("k_agc", sb.K_AGC)):
        ms, n = bank.profile_read(kid)
        k[name] = (ms / max(n, 1), int(n))
    bank.profile_enable(0)
    return dt, k


def measured_traffic(n_ch):
    """HBM bytes per k_stage12 launch from the rocprofv3 --pmc passes of this build (tools_pmc.sh ->
    profiles/*_traffic.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  null when no
    profile of this channel count is committed."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("channels") == n_ch:
                return d.get("k_stage12_hbm_bytes_per_launch"), os.path.basename(f)
        except (OSError, ValueError):
            pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--channels", type=int, default=10000, help="channel instances per GPU")
    ap.add_argument("--rate", default="10")
    ap.add_argument("--block-hint", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true")
    ap.add_argument("--pipelined", action="store_true",
                    help="SDRXGPU_FLAG_PIPELINED: call k returns block k-1 (measured: no gain, the two kernels compete for CUs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank, local_rank, world = sdist.env_rank_world()
    if world != args.gpus and world > 1:
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("SDRX_BENCH_FORCE_DIST") == "1"     # rehearse the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist and world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    ch = chains.chain(args.rate)
    n_ch = args.channels
    L = ch.block_len
    # one explicit (non-default) torch stream carries everything: the ring copy, the RCCL broadcast and the
    # bank's kernels are then ordered by stream order alone
    tstream = torch.cuda.Stream()
    torch.cuda.set_stream(tstream)

    # synthetic IQ stream: a ring of distinct 32 ms blocks in HBM on rank 0 (uniform noise + 2 AM carriers)
    n_ring = 8
    blk = torch.empty(L, 2, dtype=torch.float32, device="cuda")
    ring = None
    if rank == 0:
        car = [sb.make_translator(ch.N, ch.z, o) for o in (5, -20)]
        host = np.stack([synth.synth_iq(1234, b * L, L, 0.25, car, [0.3, 0.1], ch.N, ch.fs // 1000).view(np.float32).reshape(L, 2)
                         for b in range(n_ring)])
        ring = torch.from_numpy(host).cuda()
    def run(offsets, steps, warmup):
        tabs = sb.translators_for(ch, offsets)
        bank = sb.ChannelBank(tabs, ch.stage_tuples(), device=local_rank, stream=tstream.cuda_stream, block_hint=args.block_hint,
                              pipelined=args.pipelined)
        # N > 1: the broadcast lands directly in the bank's ingest buffer (its address is the same for every whole
        # block), so process_device finds the block in place; should the address ever differ the call copies
        ing_ptr, ing_cap = bank.ingest_buffer()
        recv = torch.as_tensor(sb.DeviceView(ing_ptr, (L, 2)), device="cuda") if ing_cap >= L else blk
        fan = sdist.BlockFanout(dist, recv, src=0)
        d_iq = torch.empty(len(offsets), 512, 2, dtype=torch.float32, device="cuda")
        d_au = torch.empty(len(offsets), 512, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        dt, k = time_bank(bank, fan, ring, rank, world, dist, L, d_iq, d_au, steps, warmup, n_ring)
        if args.pipelined:
            assert bank.flush(d_iq.data_ptr(), d_au.data_ptr(), 512) == 512
            torch.cuda.synchronize()
        assert bool(torch.isfinite(d_au).all()) and bool(torch.isfinite(d_iq).all())
        bank.close()
        return dt, k

    # this rank's shard of the job's channel list (instances cycle through all N distinct offsets; never deduplicated)
    offs = sdist.shard_list(chains.offsets_cycled(n_ch * world, ch.N), rank, world)
    assert len(offs) == n_ch
    dt, k = run(offs, args.steps, args.warmup)

    extra = {}
    if world == 1 and not args.no_extra_configs:
        # BASELINE.json's other single-GPU configurations, same step definition (parity-test sizes; short runs)
        for name, key in (("64 channels, 8.33 kHz grid (configs[1])", "c2"), ("1024 channels, mixed 8.33/25 kHz (configs[2])", "c3")):
            o = chains.offsets_for_config(key)
            dte, ke = run(o, max(10, args.steps), args.warmup)
            extra[name] = {"value": len(o) * L * max(10, args.steps) / dte / 1e6, "unit": "Mch-samples/s",
                           "ms_per_step": dte / max(10, args.steps) * 1e3, "realtime_factor": L * max(10, args.steps) / dte / ch.fs,
                           "kernel_ms_avg": {n: v[0] for n, v in ke.items()}}

    if world == 1 and not args.no_extra_configs:
        # BASELINE configs[4], one GPU's share: 2 independent 10 MS/s streams x 1024 channels, each stream's block
        # arriving from pinned host memory by an async H2D copy on the bank's own stream (copy overlaps the other
        # stream's kernels)
        n_streams, o = 2, chains.offsets_for_config("c3")
        tabs = sb.translators_for(ch, o)
        sts = [torch.cuda.Stream() for _ in range(n_streams)]
        banks = [sb.ChannelBank(tabs, ch.stage_tuples(), device=local_rank, stream=st.cuda_stream) for st in sts]
        hosts = [torch.from_numpy(synth.synth_iq(1 + i, 0, L, 0.25).view(np.float32).reshape(L, 2).copy()).pin_memory() for i in range(n_streams)]
        ing = [torch.as_tensor(sb.DeviceView(bk.ingest_buffer()[0], (L, 2)), device="cuda") for bk in banks]
        outs = [(torch.empty(len(o), 512, 2, device="cuda"), torch.empty(len(o), 512, device="cuda")) for _ in range(n_streams)]

        def multi_step():
            for i in range(n_streams):
                with torch.cuda.stream(sts[i]):
                    ing[i].copy_(hosts[i], non_blocking=True)
                    banks[i].process_device(ing[i].data_ptr(), L, outs[i][0].data_ptr(), outs[i][1].data_ptr(), 512)

        for _ in range(args.warmup):
            multi_step()
        torch.cuda.synchronize()
        n_it = max(10, args.steps)
        t0 = time.perf_counter()
        for _ in range(n_it):
            multi_step()
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        extra["2 streams x 1024 channels, H2D from pinned host overlapped (configs[4] per GPU)"] = {
            "value": n_streams * len(o) * L * n_it / dtm / 1e6, "unit": "Mch-samples/s", "ms_per_step": dtm / n_it * 1e3,
            "realtime_factor": L * n_it / dtm / ch.fs, "includes": "PCIe H2D of every block"}
        for bk in banks:
            bk.close()

    host_path = None
    if world == 1 and not args.no_extra_configs:
        # the drop-in entry (host pointers in and out, outputs complete on return): PCIe-inclusive, NOT `value`
        tabs = sb.translators_for(ch, offs)
        hb = sb.ChannelBank(tabs, ch.stage_tuples(), device=local_rank)
        x_host = synth.synth_iq(9, 0, L, 0.25)
        o_iq = np.zeros((n_ch, 512), dtype=np.complex64)
        o_au = np.zeros((n_ch, 512), dtype=np.float32)
        res = {}
        for label, pin in (("pageable", False), ("page_locked", True)):
            if pin:
                for a_ in (x_host, o_iq, o_au):
                    sb.host_register(a_)
            for _ in range(2):
                hb.process(x_host, iq_out=o_iq, audio_out=o_au)
            n_it = 8
            t0 = time.perf_counter()
            for _ in range(n_it):
                hb.process(x_host, iq_out=o_iq, audio_out=o_au)
            dth = (time.perf_counter() - t0) / n_it
            res[label] = {"ms_per_block": dth * 1e3, "value": n_ch * L / dth / 1e6}
            if pin:
                for a_ in (x_host, o_iq, o_au):
                    sb.host_unregister(a_)
        hb.close()
        host_path = {"what": f"sdrxgpu_bank_process: {L * 8 / 1e6:.2f} MB H2D + {n_ch * 512 * 12 / 1e6:.1f} MB D2H per block, synchronous",
                     "unit": "Mch-samples/s", **res}

    if rank == 0:
        total_cs = float(n_ch) * world * L * args.steps
        value = total_cs / dt / 1e6
        bpu = algorithmic_bytes_per_channel_sample(ch.m_total)
        k12_ms, k12_n = k["k_stage12"]
        achieved = bpu * n_ch * L / (k12_ms / 1e3) / 1e9 if k12_n else None
        traffic, traffic_src = measured_traffic(n_ch)
        out = {
            "metric": "IQ Msamples/s ingested x channels demodulated (channel-samples/s)",
            "value": value, "unit": "Mch-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{n_ch} channel instances per GPU from one {args.rate} MS/s IQ stream, 8.33 kHz grid "
                            f"(offsets cycle through all {ch.N} translator steps), 32 ms blocks of {L} complex<float>, "
                            f"mixer + 4-stage FIR decimation /{ch.m_total} + IF-AGC + AM envelope",
                "channels_per_gpu": n_ch, "channels_total": n_ch * world, "sample_rate": ch.fs, "block_len": L,
                "outputs_per_channel_per_block": 512,
                "parallelism": f"channel-sharded x{world}" + (", RCCL broadcast of the IQ block per step" if world > 1 else ""),
                "realtime_factor": value * 1e6 / (n_ch * world * ch.fs),
                "settle_steps": SETTLE_STEPS,    # untimed, before the W warm-up steps (clock settling, see time_bank)
                "call_semantics": "drop-in: call k returns block k" if not args.pipelined else
                                  "pipelined: call k returns block k-1 (bit-identical outputs, one block of latency)",
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_stage12 (mixer + stages 1-2)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bpu * n_ch * L,
                "kernel_ms_avg": {n: v[0] for n, v in k.items()},
                "kernel_ms_note": "k_stage12: HIP events inside the timed region; k_tail / k_agc: from 5 extra steps run right after it",
                "launches": int(k12_n),
                "note": "achieved = ALGORITHMIC bytes (8.02 B per channel-sample) / kernel time; the IQ block is shared by "
                        "all channels on chip, so physical HBM traffic is far smaller and the kernel is bound by packed-fp32 "
                        "VALU issue, not by HBM (DESIGN.md)",
            },
        }
        if extra:
            out["other_configs"] = extra
        if host_path:
            out["host_pointer_entry"] = host_path
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ch, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
