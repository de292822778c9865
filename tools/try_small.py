#!/usr/bin/env python3
"""Small-MSM kernel against the bucket pipeline: python tools/try_small.py  (ms per MSM)"""
import os, sys, time, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.engine import default_engine
Q = secp256k1.q
eng = default_engine()
rnd = random.Random(3)
N = 1 << 14
d_k = eng.upload(b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(N)))
d_G = eng.upload(secp256k1.G.to_le64() * N); d_p = eng.alloc(64 * N)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, N, d_p.ptr)); eng.sync()
d_s = eng.upload(b"".join(rnd.randrange(Q).to_bytes(32, "little") for _ in range(N)))
for logn in range(1, 15):
    n = 1 << logn
    row = []
    for small in (-1, 1 << 16):
        eng.set_option("small_n", small)
        outs = [eng.msm_dev(d_p, d_s, n) for _ in range(3)]
        t = time.perf_counter()
        for _ in range(30):
            out = eng.msm_dev(d_p, d_s, n)
        row.append(((time.perf_counter() - t) / 30 * 1e3, out[:6].hex()))
    print("n=2^%-2d  bucket %.3f ms   small %.3f ms   same=%s" % (logn, row[0][0], row[1][0], row[0][1] == row[1][1]), flush=True)
eng.set_option("small_n", 0)
