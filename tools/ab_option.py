#!/usr/bin/env python3
"""A/B of one engine option on the same box, same process: the MSM at n = 2^logn, synchronous and with two in flight, alternating
the option's values several times.   python tools/ab_option.py <option> <v0> <v1> [logn ...]"""
import hashlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()
opt, v0, v1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
logs = [int(a) for a in sys.argv[4:]] or [20]


def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))


nmax = 1 << max(logs)
d_k = eng.upload(sha_scalars(nmax, 1)); d_G = eng.upload(secp256k1.G.to_le64() * nmax); d_p = eng.alloc(64 * nmax)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, nmax, d_p.ptr)); eng.sync()
d_s = eng.upload(sha_scalars(nmax, 2))
for lg in logs:
    n = 1 << lg
    reps = 200 if lg <= 17 else 60
    ref = None
    for rnd in range(4):
        for v in (v0, v1):
            eng.set_option(opt, v)
            r = eng.msm_dev(d_p, d_s, n)
            ref = ref or r
            assert r == ref
            for _ in range(3): eng.msm_dev(d_p, d_s, n)
            t = time.perf_counter()
            for _ in range(reps): eng.msm_dev(d_p, d_s, n)
            ds = (time.perf_counter() - t) / reps
            eng.set_option("async_lanes", 1)
            eng.msm_dev_enqueue(0, d_p, d_s, n)
            t = time.perf_counter()
            for j in range(reps):
                if j + 1 < reps: eng.msm_dev_enqueue((j + 1) & 1, d_p, d_s, n)
                eng.msm_finish(j & 1)
            dp = (time.perf_counter() - t) / reps
            eng.set_option("async_lanes", 0)
            print("n=2^%d %s=%d  sync %.4f ms  two in flight %.4f ms" % (lg, opt, v, ds * 1e3, dp * 1e3), flush=True)
