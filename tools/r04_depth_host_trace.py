#!/usr/bin/env python3
"""Host-side clock of the asynchronous MSM pipeline at n = 2^20: when every bpmi_msm_dev_enqueue is called and returns, when every
bpmi_msm_finish returns (us, relative), for D = 2 and 3 MSMs in flight:  python tools/r04_depth_host_trace.py [name=value ...]"""
import os, sys, time, hashlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.engine import default_engine
Q = secp256k1.q
eng = default_engine()
for kv in sys.argv[1:]:
    name, value = kv.split("=")
    eng.set_option(name, int(value))
N = 1 << 20
pre = b"bpmi/scalar"


def sha_scalars(n, seed):
    p = pre + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(p + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))


d_k = eng.upload(sha_scalars(N, 1)); d_G = eng.upload(secp256k1.G.to_le64() * N); d_p = eng.alloc(64 * N)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, N, d_p.ptr)); eng.sync()
d_s = eng.upload(sha_scalars(N, 2))
eng.set_option("async_lanes", 1)
for D in (2, 3):
    K = 60
    log = []
    for j in range(D - 1):
        eng.msm_dev_enqueue(j % D, d_p, d_s, N)
    t0 = time.perf_counter()
    for j in range(K):
        a = time.perf_counter()
        if j + D - 1 < K:
            eng.msm_dev_enqueue((j + D - 1) % D, d_p, d_s, N)
        b = time.perf_counter()
        eng.msm_finish(j % D)
        c = time.perf_counter()
        log.append((a - t0, b - t0, c - t0))
    print("D = %d: %.4f ms per step over the last 40" % (D, (log[-1][2] - log[-41][2]) / 40 * 1e3))
    base = log[30][0]
    for j in range(30, 39):
        a, b, c = log[j]
        print("  step %2d  enqueue called %8.1f  returned %8.1f (%6.1f us)  finish returned %8.1f  (step %6.1f us)" %
              (j, (a - base) * 1e6, (b - base) * 1e6, (b - a) * 1e6, (c - base) * 1e6, (c - log[j - 1][2]) * 1e6))
# batches: B MSMs queued, then B finished (the GPU idles between batches only for the host's turn-around)
for B in (2, 3):
    K = 60
    t0 = time.perf_counter()
    for j in range(0, K, B):
        for s in range(B):
            eng.msm_dev_enqueue(s, d_p, d_s, N)
        for s in range(B):
            eng.msm_finish(s)
    print("batches of %d: %.4f ms per MSM" % (B, (time.perf_counter() - t0) / K * 1e3))
