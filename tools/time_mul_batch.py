#!/usr/bin/env python3
"""bpmi_ec_mul_batch_dev at n = 2^logn: the GLV fixed-window path against the bit-serial ladder.   python tools/time_mul_batch.py [logn]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
eng = default_engine()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
ks = b"".join(hashlib.sha256(b"%d" % i).digest() for i in range(n))
d_k = eng.upload(ks); d_G = eng.upload(secp256k1.G.to_le64() * n); d_p = eng.alloc(64 * n)
for opt in (1, 0, 1):
    eng.set_option("mul_batch_glv", opt)
    for rep in range(3):
        eng.sync(); t = time.perf_counter()
        eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync()
        print("mul_batch_glv=%d n=%d: %.3f ms" % (opt, n, (time.perf_counter() - t) * 1e3), flush=True)
