"""Worker for tests/test_distributed_cpu.py: world_size-2 gloo run of the sharding logic
with the ORACLE standing in for the HIP engine (tests may use the oracle; the product's
default is the HIP engine)."""
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import torch.distributed as dist  # noqa: E402

import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.distributed import ShardedMSM, shard_bounds, shard_verdicts  # noqa: E402
from oracle import cbind  # noqa: E402
from oracle.ec import secp256k1, point_from_le64, point_to_le64, INF  # noqa: E402


def oracle_fold(buf, k):
    acc = INF
    for p in cbind.unpack_points(buf, k):
        acc = acc + p
    return point_to_le64(acc)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    Q = secp256k1.q
    rnd = random.Random(77)           # same inputs on every rank
    for n in (1, 2, 5, 64, 301):
        pts = cbind.ec_mul_batch([secp256k1.G] * n, [rnd.randrange(1, Q) for _ in range(n)])
        es = [rnd.randrange(Q) for _ in range(n)]
        pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
        sm = ShardedMSM(msm=lambda p, s, k: cbind.msm_bytes(p, s, k, 1), fold=oracle_fold)
        got = sm.multiexp_replicated(pb, sb, n)
        want = cbind.msm_bytes(pb, sb, n, 1)
        assert got == want, (rank, n)
        # weak-scaling form: each rank owns a different shard
        lo, hi = shard_bounds(n, world, rank)
        part = cbind.msm_bytes(pb[64 * lo: 64 * hi], sb[32 * lo: 32 * hi], hi - lo, 1)
        assert sm.combine(part) == want, (rank, n)
    verdicts = shard_verdicts(lambda k: k % 3 != 0, list(range(11)))
    assert verdicts == [k % 3 != 0 for k in range(11)], verdicts
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
