#!/usr/bin/env python3
"""Per-round wall times of the inner-product-argument prover at n = 2^logn (config C3) on one GPU:
round_LR (two MSMs + dots), host Fiat-Shamir, fold -- where the 20 rounds spend their milliseconds.
  python tools/c3_round_times.py [logn]"""
import hashlib
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.ec import Point, secp256k1  # noqa: E402
from bulletproofs_amd.engine import default_engine  # noqa: E402
from bulletproofs_amd.utils import Transcript, elliptic_hash  # noqa: E402

Q = secp256k1.q
eng = default_engine()
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for kv in sys.argv[2:]:
    name, value = kv.split("=")
    eng.set_option(name, int(value))
n = 1 << logn


def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))


def device_points(n, seed):
    d_k, d_G, d_p = eng.upload(sha_scalars(n, seed)), eng.upload(secp256k1.G.to_le64() * n), eng.alloc(64 * n)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr))
    eng.sync()
    return d_p


d_g, d_h = device_points(n, 3), device_points(n, 4)
d_a, d_b = eng.upload(sha_scalars(n, 5)), eng.upload(sha_scalars(n, 6))
u = elliptic_hash(b"bench-u")
for rep in range(3):
    st = eng.ipa_create_dev(d_g, d_h, d_a, d_b, n, u.to_le64())
    tr = Transcript(b"bench")
    rows = []
    t_all = time.perf_counter()
    while len(st) > 1:
        m = len(st)
        t0 = time.perf_counter()
        Lb, Rb = st.round_LR()
        t1 = time.perf_counter()
        tr.add_list_points([Point.from_le64(Lb), Point.from_le64(Rb)])
        x = tr.get_modp(Q)
        tr.add_number(x)
        xi = x.inv()
        t2 = time.perf_counter()
        st.fold(x.x, xi.x)
        eng.sync()
        t3 = time.perf_counter()
        rows.append((m, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    st.finish()
    total = (time.perf_counter() - t_all) * 1e3
    st.close()
    if rep == 2:
        print("len      round_LR ms   host hash ms   fold+sync ms")
        for m, a, b, c in rows:
            print("%8d  %9.3f  %9.3f  %9.3f" % (m, a, b, c))
        print("total %.3f ms; round_LR %.3f, hash %.3f, fold %.3f" % (total, sum(r[1] for r in rows), sum(r[2] for r in rows), sum(r[3] for r in rows)))
