#!/usr/bin/env python3
"""Two independent MSMs of n = 2^logn that are needed TOGETHER (a round of the inner-product argument): two synchronous calls,
the two lanes unchained (what msm_run_pair does), the two lanes with chained accumulations, at several chunk lengths.
  python tools/pair_modes.py [logn]"""
import hashlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
ks = b"".join(hashlib.sha256(b"%d" % i).digest() for i in range(n))
d_k = eng.upload(ks); d_G = eng.upload(secp256k1.G.to_le64() * n); d_p = eng.alloc(64 * n)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync()


def pair(mode):
    if mode == "sync":
        eng.msm_dev(d_p, d_k, n); eng.msm_dev(d_p, d_k, n)
    else:
        eng.msm_dev_enqueue(0, d_p, d_k, n); eng.msm_dev_enqueue(1, d_p, d_k, n)
        eng.msm_finish(0); eng.msm_finish(1)


only = os.environ.get("PAIR_ONLY")          # e.g. "lanes chained": one mode only (for a kernel trace)
for chunk in ((0,) if only else (0, 64, 86, 128)):
    for mode, lanes in (("sync", 0), ("lanes", 0), ("lanes chained", 1)):
        if only and mode != only:
            continue
        eng.set_option("chunk", chunk); eng.set_option("async_lanes", lanes)
        t = time.perf_counter()
        while time.perf_counter() - t < 0.08: pair(mode)
        reps = 40
        t = time.perf_counter()
        for _ in range(reps): pair(mode)
        print("n=2^%d chunk=%3d %-14s %.3f ms per pair" % (n.bit_length() - 1, chunk, mode, (time.perf_counter() - t) / reps * 1e3), flush=True)
