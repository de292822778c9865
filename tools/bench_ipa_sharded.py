#!/usr/bin/env python3
"""C3 (inner-product argument prover, n = 2^20) sharded over the GPUs of one node:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         tools/bench_ipa_sharded.py [log2 n]
Every rank owns the cyclic shard i = rank (mod N) of g, h, a, b in its own HBM; per round one
all_gather of 128 bytes per rank (bulletproofs_amd.distributed.ShardedFastNIProver2).  Without
torchrun it runs on one GPU.  BPMI_DIST_BACKEND=gloo lets several ranks share one GPU (functional
check only)."""
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.distributed import ShardedFastNIProver2  # noqa: E402
from bulletproofs_amd.ec import secp256k1  # noqa: E402
from bulletproofs_amd.engine import default_engine  # noqa: E402
from bulletproofs_amd.utils import elliptic_hash  # noqa: E402

Q = secp256k1.q


def sha_scalars(idx, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little")
                    for i in idx)


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        backend = os.environ.get("BPMI_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend)
    eng = default_engine()
    n = 1 << logn
    idx = range(rank, n, world)                      # cyclic shard: global indices of this rank
    nl = len(idx)

    def device_points(seed):
        d_k = eng.upload(sha_scalars(idx, seed))
        d_G = eng.upload(secp256k1.G.to_le64() * nl)
        d_p = eng.alloc(64 * nl)
        eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, nl, d_p.ptr))
        eng.sync()
        d_G.free()
        d_k.free()
        return d_p

    d_g, d_h = device_points(3), device_points(4)
    d_a, d_b = eng.upload(sha_scalars(idx, 5)), eng.upload(sha_scalars(idx, 6))
    u = elliptic_hash(b"bench-u")
    grp = type("G", (), {"q": Q})()
    times = []
    digest = None
    for rep in range(3):
        st = eng.ipa_create_dev(d_g, d_h, d_a, d_b, nl, u.to_le64())
        if world > 1:
            dist.barrier()
        eng.sync()
        t = time.perf_counter()
        proof = ShardedFastNIProver2(None, None, u, None, None, None, grp, b"YmVuY2g=&", engine=eng, state=st).prove()
        eng.sync()
        dt = time.perf_counter() - t
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64)
            if dist.get_backend() == "nccl":
                tt = tt.cuda()
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        times.append(dt)
        digest = hashlib.sha256(proof.transcript).hexdigest()[:16]
    if rank == 0:
        print(json.dumps({"config": "C3 IPA prover n=2^%d, cyclic shards" % logn, "n_gpus_or_ranks": world,
                          "seconds": min(times), "elements_per_s": n / min(times), "rounds": len(proof.Ls),
                          "transcript_sha256_16": digest}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
