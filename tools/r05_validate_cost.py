#!/usr/bin/env python3
"""Cost of the on-curve check (option validate_points) at n = 2^20 and 2^16: bpmi_msm_dev at level 2 against level 1 (device pointers:
the check kernel is the only difference), and bpmi_msm from host memory at level 1 against 0 (upload included)."""
import hashlib, os, sys, time, ctypes
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


nmax = 1 << 20
d_k = eng.upload(sha_scalars(nmax, 1)); d_G = eng.upload(secp256k1.G.to_le64() * nmax); d_p = eng.alloc(64 * nmax)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, nmax, d_p.ptr)); eng.sync()
sb = sha_scalars(nmax, 2)
d_s = eng.upload(sb)
pb = d_p.download()
for n in (1 << 16, 1 << 20):
    reps = 100 if n <= (1 << 16) else 30
    for rnd in range(3):
        for level in (1, 2):
            eng.set_option("validate_points", level)
            for _ in range(5): eng.msm_dev(d_p, d_s, n)
            t = time.perf_counter()
            for _ in range(reps): eng.msm_dev(d_p, d_s, n)
            print("n=%8d bpmi_msm_dev validate_points=%d  %.4f ms" % (n, level, (time.perf_counter() - t) / reps * 1e3), flush=True)
        for level in (0, 1):
            eng.set_option("validate_points", level)
            for _ in range(3): eng.msm_bytes(pb, sb, n)
            t = time.perf_counter()
            for _ in range(max(5, reps // 5)): eng.msm_bytes(pb, sb, n)
            print("n=%8d bpmi_msm (host pointers) validate_points=%d  %.4f ms" % (n, level, (time.perf_counter() - t) / max(5, reps // 5) * 1e3), flush=True)
eng.set_option("validate_points", 1)
