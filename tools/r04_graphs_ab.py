#!/usr/bin/env python3
"""Option "graphs" on / off, same process: one MSM at a time at 2^12 .. 2^20 (ms per MSM), the inner-product prover at 2^20 and
2^14 (ms per proof through bpmi_ipa_prove_rounds), results compared.   python tools/r04_graphs_ab.py"""
import os, sys, time, random, hashlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.utils import elliptic_hash
Q = secp256k1.q
eng = default_engine()
for kv in sys.argv[1:]:
    name, value = kv.split("=")
    eng.set_option(name, int(value))
rnd = random.Random(3)
N = 1 << 20


def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))


d_k = eng.upload(sha_scalars(N, 1)); d_G = eng.upload(secp256k1.G.to_le64() * N); d_p = eng.alloc(64 * N)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, N, d_p.ptr)); eng.sync()
d_s = eng.upload(sha_scalars(N, 2))
d_q = eng.alloc(64 * N)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_p.ptr, d_k.ptr, N, d_q.ptr)); eng.sync()
u = elliptic_hash(b"bench-u")
for logn in (12, 13, 14, 16, 18, 20):
    n = 1 << logn
    row = []
    for graphs in (0, 1, 0, 1):
        eng.set_option("graphs", graphs)
        for _ in range(5):
            out = eng.msm_dev(d_p, d_s, n)
        reps = 60 if logn < 20 else 20
        t = time.perf_counter()
        for _ in range(reps):
            out = eng.msm_dev(d_p, d_s, n)
        row.append(((time.perf_counter() - t) / reps * 1e3, out[:8].hex()))
    print("MSM n=2^%-2d  graphs off %.3f / %.3f ms   on %.3f / %.3f ms   same=%s" % (logn, row[0][0], row[2][0], row[1][0], row[3][0], len({r[1] for r in row}) == 1), flush=True)
for logn in (20, 14):
    n = 1 << logn
    row = []
    for graphs in (0, 1, 0, 1):
        eng.set_option("graphs", graphs)
        best, tr = 1e9, None
        for rep in range(4):
            st = eng.ipa_create_dev(d_p, d_q, d_s, d_k, n, u.to_le64())
            t = time.perf_counter()
            tr, xs, Ls, Rs = st.prove_rounds(b"bench")
            a, b = st.finish()
            best = min(best, time.perf_counter() - t)
            st.close()
        row.append((best * 1e3, hashlib.sha256(tr).hexdigest()[:12]))
    print("IPA n=2^%-2d  graphs off %.3f / %.3f ms   on %.3f / %.3f ms   same=%s" % (logn, row[0][0], row[2][0], row[1][0], row[3][0], len({r[1] for r in row}) == 1), flush=True)
eng.set_option("graphs", 0)
