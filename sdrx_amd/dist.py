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


def max_over_ranks(dist, seconds: float, device="cpu") -> float:
    """Wall time of the slowest rank (the bench contract's timing)."""
    if dist is None or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
