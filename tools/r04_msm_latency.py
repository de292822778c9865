#!/usr/bin/env python3
"""ms per MSM, one at a time, at 2^10 .. 2^20 with the given options:  python tools/r04_msm_latency.py [name=value ...]"""
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
row = []
for logn in (10, 12, 14, 16, 18, 20):
    n = 1 << logn
    for _ in range(8):
        eng.msm_dev(d_p, d_s, n)
    reps = 80 if logn < 20 else 30
    t = time.perf_counter()
    for _ in range(reps):
        eng.msm_dev(d_p, d_s, n)
    row.append("2^%d %.3f" % (logn, (time.perf_counter() - t) / reps * 1e3))
print("%-22s ms per MSM one at a time: %s" % (" ".join(sys.argv[1:]) or "(defaults)", "  ".join(row)), flush=True)
