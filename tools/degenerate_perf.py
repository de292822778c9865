"""Timing of the MSM on degenerate scalar distributions at n = 2^logn (default 20; robustness check:
no input shape may fall off a performance cliff).   python tools/degenerate_perf.py [logn]"""
import hashlib, os, sys, time, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
rnd = random.Random(1)
ks = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(n))
d_G = eng.upload(secp256k1.G.to_le64() * n); d_k = eng.upload(ks); d_p = eng.alloc(64 * n)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync()
le = lambda v: (v % Q).to_bytes(32, "little")
shapes = {
    "uniform": b"".join(le(rnd.getrandbits(256)) for _ in range(n)),
    "acommit {0,1}|{0,q-1}": b"".join(le(rnd.randrange(2)) for _ in range(n // 2)) + b"".join(le(rnd.randrange(2) - 1) for _ in range(n // 2)),
    "all same scalar": le(rnd.getrandbits(256)) * n,
    "all ones": le(1) * n,
    "20-bit scalars": b"".join(le(rnd.getrandbits(20)) for _ in range(n)),
    "64-bit scalars": b"".join(le(rnd.getrandbits(64)) for _ in range(n)),
    "two values": b"".join(le([12345678901234567890123, Q - 5][rnd.randrange(2)]) for _ in range(n)),
    # the vector commitments of a range proof (rangeproof_prover.py:43-50): bits, bits - 1 and a few blinding factors -- one
    # giant bucket, then thousands of EMPTY buckets before the next entry of every window
    "range-proof bits + blinding": b"".join(le(rnd.getrandbits(256) if i % 4096 == 0 else (rnd.randrange(2) if i < n // 2 else rnd.randrange(2) - 1)) for i in range(n)),
    "1 in 1024 non-zero": b"".join(le(rnd.getrandbits(256) if i % 1024 == 0 else 0) for i in range(n)),
}
for name, sc in shapes.items():
    d_s = eng.upload(sc)
    eng.msm_dev(d_p, d_s, n)
    t = time.perf_counter()
    for _ in range(3):
        eng.msm_dev(d_p, d_s, n)
    dt = (time.perf_counter() - t) / 3
    print("n=2^%d  %-28s %8.3f ms" % (n.bit_length() - 1, name, dt * 1e3), flush=True)
    d_s.free()
