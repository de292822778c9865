#!/usr/bin/env python3
"""Config C5: batch-verify 2^14 x 64-bit range proofs, sharded by proof across the GPUs of
one node (one process per GPU):

  python tools/bench_batch_verify.py [--log-batch 14] [--distinct 64]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         tools/bench_batch_verify.py --log-batch 14

Every rank proves `distinct` proofs (seeded, identical on all ranks), cycles them to fill
its shard of the batch, runs the host-side transcript checks and scalar preparation for its
shard -- by default on serialised proofs, in a pool of worker processes (add_wire) --
evaluates ONE MSM (rangeproofs/batch.py), and the 64-byte partial values are combined with
one all_gather + fold.  Prints one JSON line on rank 0."""
import argparse
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-batch", type=int, default=14)
    ap.add_argument("--distinct", type=int, default=64)
    ap.add_argument("--native", type=int, default=1,
                    help="1 (default): per-proof work in libbpmi (see --prepare; --workers = host threads of the host variant); "
                         "0: in Python (worker processes)")
    ap.add_argument("--prepare", choices=("device", "host"), default="device",
                    help="with --native 1: per-proof preparation on the GPU (bpmi_rp_batch_prepare_dev, default) or on host threads")
    ap.add_argument("--workers", type=int, default=-1,
                    help="host worker processes per rank for the wire path (0: serial add() on proof objects; "
                         "default: min(32, host cores / ranks))")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import bulletproofs_amd  # noqa: F401
    from bulletproofs_amd.distributed import ShardedMSM, shard_bounds
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.engine import Engine, set_default_engine
    from bulletproofs_amd.rangeproofs import BatchRangeVerifier, NIRangeProver
    from bulletproofs_amd.utils import ModP, commitment, elliptic_hash, mod_hash
    eng = Engine(device=local)
    set_default_engine(eng)
    Q = secp256k1.q
    n = 64
    gs = [elliptic_hash(str(i).encode() + b"gs") for i in range(n)]
    hs = [elliptic_hash(str(i).encode() + b"hs") for i in range(n)]
    g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
    proofs = []
    t0 = time.perf_counter()
    for j in range(args.distinct):
        v = ModP(int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big"), Q)
        gamma = mod_hash(b"gamma%d" % j, Q)
        proofs.append((commitment(g, h, v, gamma), NIRangeProver(v, n, g, h, gs, hs, gamma, u, secp256k1, b"seed%d" % j).prove()))
    t_prove = time.perf_counter() - t0
    total = 1 << args.log_batch
    lo, hi = shard_bounds(total, world, rank)
    sharded = ShardedMSM(engine=eng)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    usable = len(os.sched_getaffinity(0))
    try:                                              # cgroup v2 CPU quota of the container, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(usable, int(quota) // int(period)))
    except Exception:
        pass
    workers = args.workers if args.workers >= 0 else max(1, min(32, usable // world))
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    t_pool = 0.0
    if workers:
        # the proofs arrive as bytes (rangeproofs/codec.py); the pool is part of the service, not of a batch
        from bulletproofs_amd.rangeproofs.codec import proof_to_bytes
        wire = [proof_to_bytes(pr) for _, pr in proofs]
        Vs_in = [proofs[k % args.distinct][0] for k in range(lo, hi)]
        blobs_in = [wire[k % args.distinct] for k in range(lo, hi)]
        t0 = time.perf_counter()
        if not args.native:
            bv.start_workers(workers)
        t_pool = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
    if workers and args.native:
        # the service shape of bench.py's extra: one page-locked receive buffer + offsets, commitments packed, a warm batch first
        # (staging buffers, workspaces), then the timed one
        import ctypes
        from itertools import accumulate
        joined = b"".join(blobs_in)
        offs = (ctypes.c_uint64 * (len(blobs_in) + 1))(0, *accumulate(map(len, blobs_in)))
        hb = eng.host_alloc(len(joined))
        hb.view[:] = joined
        v_packed = b"".join(V.to_le64() for V in Vs_in)
        bv.add_wire_native(v_packed, hb, threads=workers, offsets=offs, prepare=args.prepare)
        assert bv.verify(sharded=sharded if world > 1 else None)
        bv.reset()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    if workers and args.native:
        bv.add_wire_native(v_packed, hb, threads=workers, offsets=offs, prepare=args.prepare)
    elif workers:
        bv.add_wire(Vs_in, blobs_in)
    else:
        for k in range(lo, hi):
            V, pr = proofs[k % args.distinct]
            bv.add(V, pr)
    t_host = time.perf_counter() - t0
    ok = bv.verify(sharded=sharded if world > 1 else None)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    if rank == 0:
        print(json.dumps({"metric": "range-proof verifies/sec (batched, 64-bit proofs)", "value": total / elapsed,
                          "unit": "verifies/s", "n_gpus": world, "batch": total, "seconds": elapsed,
                          "host_prep_s_rank0": t_host, "msm_points_rank0": 3 + 2 * n + 19 * (hi - lo), "ok": ok,
                          "proves_per_s_one_gpu": args.distinct / t_prove, "host_workers_per_rank": workers,
                          "input": "wire bytes (GPU batch decompression inside the timed region)" if workers else "proof objects",
                          "per_proof_work": (("GPU (bpmi_rp_batch_prepare_dev)" if args.prepare == "device" else "native host code, %d threads" % workers)
                                             if (workers and args.native) else ("python, %d processes" % workers if workers else "python, in-process")),
                          "timed": "one warm batch, one at a time (bench.py's extra also reports two in flight)" if (workers and args.native) else "one cold batch",
                          "worker_pool_startup_s": t_pool, "host_cores": os.cpu_count(),
                          "host_cores_usable": usable}))
    bv.stop_workers()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
