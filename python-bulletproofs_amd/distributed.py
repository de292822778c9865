"""Sharded MSM across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The MSM sum_i e_i * P_i is a sum over independent pairs, so it shards with no data-path
exchange: every rank runs the full single-GPU MSM on its own contiguous shard and
produces ONE 64-byte partial point.  The only exchange step is an all_gather of those
64-byte partials (EC addition is not an RCCL reduction operator, so "reduce" =
all_gather of raw bytes + a local fold with bpmi_ec_sum); it is latency-bound (64 B per
rank), link bandwidth is irrelevant.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world, rank):
    """Contiguous shard [lo, hi) of n items for `rank`; sizes differ by at most one."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_bytes(data: bytes, group=None):
    """All ranks contribute `data` (same length everywhere); returns the list of all
    contributions, in rank order.  One collective."""
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine, group=group)
    return [bytes(t.cpu().numpy().tobytes()) for t in outs]


class ShardedMSM:
    """msm(pts_bytes, scalar_bytes, n) -> 64 bytes and fold(points_bytes, k) -> 64 bytes are
    the two engine operations used; by default they are the HIP engine's."""

    def __init__(self, engine=None, group=None, msm=None, fold=None):
        if engine is None and (msm is None or fold is None):
            from .engine import default_engine
            engine = default_engine()
        self.msm = msm or engine.msm_bytes
        self.fold = fold or engine.ec_sum_bytes
        self.msm_dev = getattr(engine, "msm_dev", None)
        self.group = group

    def combine(self, partial: bytes) -> bytes:
        """partial = this rank's 64-byte partial result -> the global result on every rank."""
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return partial
        parts = all_gather_bytes(partial, self.group)
        return self.fold(b"".join(parts), len(parts))

    def multiexp_replicated(self, pts: bytes, scalars: bytes, n: int) -> bytes:
        """Every rank holds the full input; each computes its shard, then combine()."""
        if not dist.is_initialized():
            return self.msm(pts, scalars, n)
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        lo, hi = shard_bounds(n, world, rank)
        part = self.msm(pts[64 * lo: 64 * hi], scalars[32 * lo: 32 * hi], hi - lo)
        return self.combine(part)

    def multiexp_local_dev(self, d_pts, d_scalars, n_local: int) -> bytes:
        """Each rank already holds its own shard in device memory (weak scaling)."""
        return self.combine(self.msm_dev(d_pts, d_scalars, n_local))


def shard_verdicts(verify_one, items, group=None):
    """Batch verification sharded by proof: rank r verifies items[lo:hi]; verdict bytes
    (1 = accepted) are exchanged with one all_gather.  `verify_one(item) -> bool`."""
    if not dist.is_initialized():
        return [bool(verify_one(it)) for it in items]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (len(items) + world - 1) // world
    lo, hi = min(rank * per, len(items)), min((rank + 1) * per, len(items))
    mine = bytearray(per)
    for j, it in enumerate(items[lo:hi]):
        mine[j] = 1 if verify_one(it) else 0
    parts = all_gather_bytes(bytes(mine), group)
    flat = b"".join(parts)
    return [bool(flat[i]) for i in range(len(items))]
