import hashlib, os, sys, time, random
sys.path.insert(0, "/root/repo")
import bulletproofs_amd
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()
n = 1 << 16
rnd = random.Random(1)
ks = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(n))
d_G = eng.upload(secp256k1.G.to_le64() * n); d_k = eng.upload(ks); d_p = eng.alloc(64 * n)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync()
le = lambda v: (v % Q).to_bytes(32, "little")
shapes = {
 "uniform": b"".join(le(rnd.getrandbits(256)) for _ in range(n)),
 "bits": b"".join(le(rnd.randrange(2) if i < n // 2 else rnd.randrange(2) - 1) for i in range(n)),
 "bits+blinding": b"".join(le(rnd.getrandbits(256) if i % 4096 == 0 else (rnd.randrange(2) if i < n // 2 else rnd.randrange(2) - 1)) for i in range(n)),
}
for name, sc in shapes.items():
    d_s = eng.upload(sc)
    for _ in range(5): eng.msm_dev(d_p, d_s, n)
    t = time.perf_counter()
    for _ in range(20): eng.msm_dev(d_p, d_s, n)
    dt = (time.perf_counter() - t) / 20
    eng.profile(True); eng.profile_reset()
    for _ in range(8): eng.msm_dev(d_p, d_s, n)
    pr = eng.profile_read(); eng.profile(False)
    print("%-16s %.3f ms" % (name, dt * 1e3), {k.replace("msm_", ""): round(v[0] / v[1], 4) for k, v in pr.items() if v[1]}, flush=True)
