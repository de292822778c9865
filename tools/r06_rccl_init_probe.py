#!/usr/bin/env python3
"""Does an initialised RCCL process group slow this process's OWN kernels down?  (Round 6: the headline loop runs 1.01-1.02 ms per step
instead of 0.95 as soon as a one-rank RCCL group exists, WITHOUT any collective per step, and the stages of a synchronous MSM are 7-12 %
longer each.)  One process, one GPU: a throughput-bound library call (bpmi_ec_mul_batch_dev, 2^18 scalar multiplications) and a synchronous
MSM of 2^20 pairs timed (i) before the group exists, (ii) after init_process_group("nccl", device_id=...), (iii) after the first
collective, (iv) after destroy_process_group.   RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29560 python tools/r06_rccl_init_probe.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29560")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.engine import Engine

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if os.environ.get("PROBE_GROUP_FIRST") == "1":          # the bench's order: the group exists before the engine's streams do
    dist.init_process_group("nccl", device_id=dev)
    ctl = dist.new_group(backend="gloo")
    print("RCCL group (+ a gloo control group) initialised BEFORE the engine and its streams", flush=True)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
eng = Engine(device=0, stream=stream.cuda_stream)
eng.set_option("async_lanes", 1)
n = 1 << 20
rng = np.random.default_rng(3)
ks = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
ks[:, 7] &= 0x7FFFFFFF
d_k = eng.upload(ks.tobytes())
d_G = eng.upload(secp256k1.G.to_le64() * n)
d_p = eng.alloc(64 * n)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr))
eng.sync()


def measure(tag):
    m = 1 << 18
    for _ in range(3):
        eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, m, d_p.ptr + 0))
    eng.sync()
    t = time.perf_counter()
    for _ in range(5):
        eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, m, d_p.ptr + 0))
    eng.sync()
    mul_ms = (time.perf_counter() - t) / 5 * 1e3
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr))
    eng.sync()
    for _ in range(30):
        eng.msm_dev(d_p, d_k, n)
    t = time.perf_counter()
    for _ in range(20):
        eng.msm_dev(d_p, d_k, n)
    msm_ms = (time.perf_counter() - t) / 20 * 1e3

    def burst(k, D=2):
        for j in range(min(k, D)):
            eng.msm_dev_enqueue(j % D, d_p, d_k, n)
        for j in range(k):
            eng.msm_finish(j % D)
            if j + D < k:
                eng.msm_dev_enqueue(j % D, d_p, d_k, n)
    burst(40)
    t = time.perf_counter()
    burst(60)
    pipe_ms = (time.perf_counter() - t) / 60 * 1e3
    print("%-52s 2^18 scalar multiplications %.3f ms | one synchronous MSM of 2^20 pairs %.3f ms | two in flight %.4f ms per MSM" % (tag, mul_ms, msm_ms, pipe_ms), flush=True)


if os.environ.get("PROBE_GROUP_FIRST") == "1":
    measure("group first, engine second")
    dist.barrier()
    torch.cuda.synchronize()
    measure("after a barrier")
    dist.destroy_process_group()
    measure("group destroyed")
    sys.exit(0)
measure("before any process group")
dist.init_process_group("gloo")
measure("gloo group initialised")
dist.destroy_process_group()
dist.init_process_group("nccl", device_id=dev)
measure("RCCL group initialised (device_id: eager)")
x = torch.ones(16, device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
measure("after the first collective")
dist.destroy_process_group()
measure("group destroyed")
