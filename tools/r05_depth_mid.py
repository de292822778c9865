#!/usr/bin/env python3
"""Round 5: MSMs in flight at mid sizes: one (synchronous), two, three (the engine's three asynchronous slots).  python tools/r05_depth_mid.py [n ...]"""
import hashlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()


def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))


sizes = [int(a) for a in sys.argv[1:]] or [1 << 14, 1 << 15, 1 << 16, 1 << 17]
nmax = max(sizes)
d_k = eng.upload(sha_scalars(nmax, 1)); d_G = eng.upload(secp256k1.G.to_le64() * nmax); d_p = eng.alloc(64 * nmax)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, nmax, d_p.ptr)); eng.sync()
d_s = eng.upload(sha_scalars(nmax, 2))
reps = 300
for n in sizes:
    ref = eng.msm_dev(d_p, d_s, n)
    best = {}
    for rnd in range(3):
        for _ in range(5): eng.msm_dev(d_p, d_s, n)
        t = time.perf_counter()
        for _ in range(reps): eng.msm_dev(d_p, d_s, n)
        best[1] = min(best.get(1, 1e9), (time.perf_counter() - t) / reps)
        eng.set_option("async_lanes", 1)
        for depth in (2, 3):
            for k in range(depth - 1): eng.msm_dev_enqueue(k, d_p, d_s, n)
            t = time.perf_counter()
            for j in range(reps):
                if j + depth - 1 < reps: eng.msm_dev_enqueue((j + depth - 1) % depth, d_p, d_s, n)
                r = eng.msm_finish(j % depth)
            dt = (time.perf_counter() - t) / reps
            assert r == ref
            best[depth] = min(best.get(depth, 1e9), dt)
        eng.set_option("async_lanes", 0)
    print("## n=%7d one at a time %.4f ms  two in flight %.4f  three in flight %.4f" % (n, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3), flush=True)
