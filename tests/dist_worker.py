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


def batch_mode():
    """Each rank batch-verifies its own half of the proofs; the partial values fold to the
    identity iff the whole batch is valid."""
    from test_batch_verify_cpu import make_batch, oracle_msm
    from bulletproofs_amd.rangeproofs.batch import BatchRangeVerifier
    from bulletproofs_amd.utils.utils import ModP
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    b = make_batch(5)
    sm = ShardedMSM(msm=oracle_msm, fold=oracle_fold)
    lo, hi = shard_bounds(5, world, rank)
    for corrupt in (False, True):
        bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm)
        for k in range(lo, hi):
            pr = b["proofs"][k]
            if corrupt and k == 4:
                pr.mu = pr.mu + ModP(1, secp256k1.q)
            bv.add(b["Vs"][k], pr)
        try:
            ok = bv.verify(sharded=sm)
        except Exception as e:
            ok = str(e)
        assert ok == ("Proof invalid" if corrupt else True), (rank, corrupt, ok)
    dist.barrier()
    if rank == 0:
        print("DIST_BATCH_OK world=%d" % world)
    dist.destroy_process_group()


def main():
    if os.environ.get("BPMI_DIST_MODE") == "batch":
        return batch_mode()
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
