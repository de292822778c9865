#!/usr/bin/env python3
"""A plain loop of synchronous MSMs for a kernel trace: python tools/r05_msm_loop.py n reps [opt=v ...]"""
import hashlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()
n, reps = int(sys.argv[1]), int(sys.argv[2])
for kv in sys.argv[3:]:
    k, v = kv.split("="); eng.set_option(k, int(v))


def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))


d_k = eng.upload(sha_scalars(n, 1)); d_G = eng.upload(secp256k1.G.to_le64() * n); d_p = eng.alloc(64 * n)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync()
d_s = eng.upload(sha_scalars(n, 2))
for _ in range(20): eng.msm_dev(d_p, d_s, n)
t = time.perf_counter()
for _ in range(reps): eng.msm_dev(d_p, d_s, n)
print("n=%d %s: %.4f ms per MSM" % (n, " ".join(sys.argv[3:]), (time.perf_counter() - t) / reps * 1e3))
