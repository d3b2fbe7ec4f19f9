"""Multi-GPU plumbing: one process per GPU, channels sharded across ranks.

The reference's only parallelism is one worker thread per channel (``DS``, reference
``src/ds.hpp:32-78``; dispatch in ``data_cb``, ``src/sdrx.cpp:327-340``): channels never talk to
each other.  Across GPUs that becomes: every rank owns a contiguous range of the channel list
(its own ``ChannelBank``, its own per-channel state, its own slice of the output chunk) and needs
the *whole* IQ block.  The only data that crosses xGMI is therefore the block itself -- one
broadcast from the rank that owns the stream (2.56 MB per 32 ms of 10 MS/s signal) -- and there is
no collective in the compute path.  ``torch.distributed`` (backend ``nccl`` = RCCL on GPUs,
``gloo`` on CPU for tests) carries the broadcast, the barrier and the max-over-ranks timing.
"""
import os


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [lo, hi) of rank's items; the first n_items % world ranks take one extra."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_list(items, rank: int, world: int):
    lo, hi = shard_range(len(items), rank, world)
    return items[lo:hi]


class Plan:
    """What one rank of a bench / job configuration runs: its banks (one offset list each), where each bank's IQ
    stream comes from (broadcast from rank 0, or a stream of its own with this seed) and how the line is labelled."""

    def __init__(self, name, workload, offsets, seeds, broadcast, channels_total, parallelism):
        self.name = name
        self.workload = workload
        self.offsets = offsets            # one list of channel offsets per bank of this rank
        self.seeds = seeds                # per bank: seed of its own IQ stream, or None (the broadcast stream)
        self.broadcast = broadcast        # True: rank 0 owns the one stream, every block is broadcast
        self.channels_total = channels_total
        self.parallelism = parallelism

    @property
    def channels_per_gpu(self):
        return sum(len(o) for o in self.offsets)


def plan(config: str, rank: int, world: int, N: int, n_ch: int = 10000) -> "Plan":
    """BASELINE.json's configurations as per-rank plans (reference dispatch src/sdrx.cpp:327-340,
    src/ds.hpp:43-77: one independent worker per channel, so any partition of the channel list is valid).

    headline: n_ch channel instances per GPU from ONE stream (weak scaling), offsets cycling through all N steps,
              rank r takes instances [r n_ch, (r+1) n_ch); block broadcast.
    c4:       configs[3] -- 8 192 channels = 1 024 per GPU (configs[2]'s offset set -512..511 on every GPU) from one
              10 MS/s stream, the block broadcast to every GPU.  With world ranks: 1 024 x world channels.
    c5:       configs[4] -- 16 independent streams x 1 024 channels on 8 GPUs = 2 banks per GPU, each bank with a stream
              of its own (seeds 1..16: rank r has 2r+1 and 2r+2) arriving from page-locked host memory; NO broadcast,
              no traffic between GPUs at all."""
    from . import chains
    if config == "headline":
        offs = shard_list(chains.offsets_cycled(n_ch * world, N), rank, world)
        return Plan("headline", f"{n_ch} channel instances per GPU from one IQ stream, 8.33 kHz grid (offsets cycle through all "
                    f"{N} translator steps)", [offs], [None], True, n_ch * world,
                    f"channel-sharded over {world} RCCL rank(s)")
    if config == "c4":
        o = chains.offsets_for_config("c3")
        return Plan("c4", f"BASELINE configs[3]: {len(o) * world} channels from one 10 MS/s IQ stream, {len(o)} per GPU (configs[2]'s "
                    "mixed 8.33/25 kHz offset set -512..511 on every GPU), per-GPU channel partition, the IQ block broadcast",
                    [o], [None], True, len(o) * world, f"channel-sharded over {world} RCCL rank(s)")
    if config == "c5":
        o = chains.offsets_for_config("c3")
        seeds = [2 * rank + 1, 2 * rank + 2]
        return Plan("c5", f"BASELINE configs[4]: {2 * world} independent 10 MS/s IQ streams x {len(o)} channels each, 2 streams "
                    "(banks) per GPU, every block H2D from page-locked host memory on the bank's copy stream beside the "
                    "other bank's kernels (sdrxgpu_bank_process_pinned)", [o, o], seeds, False, 2 * len(o) * world,
                    f"{2 * world} independent streams, 2 per GPU; no collective, no traffic between GPUs")
    raise ValueError(f"unknown configuration {config!r} (headline, c4, c5)")


def block_checksum(block) -> int:
    """Order-sensitive 64-bit checksum of a block's bytes (numpy array or CPU torch tensor): what a rank compares with
    the generator's to know that the block it received is the block that was sent."""
    import numpy as np
    a = np.ascontiguousarray(block.numpy() if hasattr(block, "numpy") else block).view(np.uint32).reshape(-1).astype(np.uint64)
    w = (np.arange(a.size, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    with np.errstate(over="ignore"):
        return int(np.bitwise_xor.reduce(a * w + (a << np.uint64(17))))


class BlockFanout:
    """Rank ``src`` owns the IQ stream; every step its block is broadcast to all ranks.

    ``buf`` is this rank's receive buffer (a torch tensor: device tensor under nccl, CPU tensor under
    gloo).  With world == 1 nothing is communicated."""

    def __init__(self, dist, buf, src=0):
        self.dist = dist
        self.buf = buf
        self.src = src
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0

    def step(self, block=None, force=False):
        """On rank ``src`` pass the block (same shape/dtype as ``buf``); returns the buffer holding it.
        ``force``: run the broadcast even with a single rank (rehearsal of the RCCL path on one GPU)."""
        if self.world == 1 and not (force and self.dist is not None):
            return block
        if self.rank == self.src:
            self.buf.copy_(block)
        self.dist.broadcast(self.buf, src=self.src)
        return self.buf


class PipelinedFanout:
    """Double-buffered fan-out: the broadcast of block k+1 runs beside the processing of block k.

    ``bufs`` are the two receive buffers of this rank, used alternately (on GPUs: the bank's two input rings,
    ``ChannelBank.ingest_buffer()`` and ``ingest_buffer_after()``, so that a block lands where the kernels read
    it).  Per step k::

        cur = fan.get(k)            # block k; the current stream waits for its broadcast, the host does not
        fan.post(k + 1, nxt)        # rank src: start moving block k+1 (async, on the side stream)
        ... enqueue the processing of cur ...
        fan.done(k)                 # buffer k % 2 may be refilled once what was enqueued so far has run

    With one rank nothing is communicated unless ``force`` (rehearsal of the RCCL path on one GPU).  ``gloo`` / CPU
    tensors take the same calls (no streams; ``async_op`` broadcasts).

    ``consumer_stream``: the stream on which the buffers are READ (the bank's stream: ``ChannelBank(stream=...)``).
    ``done(k)`` records "buffer k % 2 is free" there and the refill waits for that record, so it must be the stream
    the bank's kernels run on -- a bank on a stream of its own while ``done`` recorded on another would let the
    broadcast of block k+2 overwrite a ring block k's kernels still read.  Default: the stream that is current when
    the fan-out is built (bench.py builds it with the bank's stream current); ``get`` and ``done`` refuse to run
    with any other stream current."""

    def __init__(self, dist, bufs, src=0, force=False, inline=False, consumer_stream=None):
        self.dist = dist
        self.inline = inline        # A/B only: the broadcast of block k in line on the compute stream, at get(k)
        self.bufs = list(bufs)
        self.src = src
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.active = dist is not None and (self.world > 1 or force)
        self.cuda = bool(self.bufs) and self.bufs[0].is_cuda
        self._work = [None, None]
        self._local = [None, None]
        if self.active and self.cuda:
            import torch
            self._torch = torch
            self.comm = torch.cuda.Stream()
            self.consumer = consumer_stream if consumer_stream is not None else torch.cuda.current_stream()
            self.free = [torch.cuda.Event(), torch.cuda.Event()]
            for e in self.free:
                e.record(self.consumer)

    def post(self, k, block=None):
        """Start moving block k into buffer k % 2; ``block`` is needed on rank ``src`` only."""
        i = k & 1
        if not self.active or self.inline:
            self._local[i] = block
            return
        buf = self.bufs[i]
        if self.cuda:
            torch = self._torch
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(self.free[i])          # whoever read this buffer last is done with it
                if self.rank == self.src:
                    buf.copy_(block, non_blocking=True)
                self._work[i] = self.dist.broadcast(buf, src=self.src, async_op=True)
        else:
            if self.rank == self.src:
                buf.copy_(block)
            self._work[i] = self.dist.broadcast(buf, src=self.src, async_op=True)

    def get(self, k):
        """Block k (posted earlier).  On GPUs the current stream waits for the broadcast, not the host."""
        i = k & 1
        if not self.active:
            return self._local[i]
        if self.inline:
            if self.rank == self.src:
                self.bufs[i].copy_(self._local[i])
            self.dist.broadcast(self.bufs[i], src=self.src)
            return self.bufs[i]
        self._check_stream()
        w, self._work[i] = self._work[i], None
        if w is not None:
            w.wait()
        return self.bufs[i]

    def _check_stream(self):
        if self.active and self.cuda and self._torch.cuda.current_stream() != self.consumer:
            raise RuntimeError("PipelinedFanout: get()/done() must run with the consumer stream current "
                               "(the stream the bank reads the buffers on)")

    def done(self, k):
        """Everything that reads buffer k % 2 has been enqueued on the current stream."""
        if self.active and self.cuda:
            self._check_stream()
            self.free[k & 1].record(self.consumer)


def max_over_ranks(dist, seconds: float, device="cpu") -> float:
    """Wall time of the slowest rank (the bench contract's timing)."""
    if dist is None or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
