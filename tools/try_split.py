#!/usr/bin/env python3
"""A/B of option "split" (one MSM as two window groups on two lanes): python tools/try_split.py [log2 n ...]"""
import os, sys, time, hashlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.engine import default_engine
Q = secp256k1.q
eng = default_engine()

def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))

for logn in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << logn
    d_k = eng.upload(sha_scalars(n, 1)); d_G = eng.upload(secp256k1.G.to_le64() * n); d_p = eng.alloc(64 * n)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync()
    d_s = eng.upload(sha_scalars(n, 2))
    res = {}
    for split in (0, 1, 0, 1):
        eng.set_option("split", split)
        for _ in range(3):
            out = eng.msm_dev(d_p, d_s, n)
        t = time.perf_counter()
        for _ in range(20):
            out = eng.msm_dev(d_p, d_s, n)
        dt = (time.perf_counter() - t) / 20
        res.setdefault(split, []).append((round(dt * 1e3, 4), out[:8].hex()))
    print(logn, res, flush=True)
    eng.set_option("split", 0)
    for b in (d_k, d_G, d_p, d_s): b.free()
