#!/usr/bin/env python3
"""Round 5 probe: one MSM as two window groups on two lanes (option "split"), synchronous calls only, at mid sizes."""
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


sizes = [int(a) for a in sys.argv[1:]] or [1 << 15, 1 << 16, 1 << 17, 311427, 1 << 18]
nmax = max(sizes)
d_k = eng.upload(sha_scalars(nmax, 1)); d_G = eng.upload(secp256k1.G.to_le64() * nmax); d_p = eng.alloc(64 * nmax)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, nmax, d_p.ptr)); eng.sync()
d_s = eng.upload(sha_scalars(nmax, 2))
for n in sizes:
    ref = None
    for rnd in range(3):
        for split in (0, 1):
            eng.set_option("split", split)
            r = eng.msm_dev(d_p, d_s, n)
            ref = ref or r
            assert r == ref
            for _ in range(5): eng.msm_dev(d_p, d_s, n)
            t = time.perf_counter()
            for _ in range(200): eng.msm_dev(d_p, d_s, n)
            print("n=%7d split=%d sync %.4f ms" % (n, split, (time.perf_counter() - t) / 200 * 1e3), flush=True)
eng.set_option("split", 0)
