import hashlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bulletproofs_amd
from bulletproofs_amd.ec import secp256k1, Point, PackedPoints, PackedScalars
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.pippenger import PipSECP256k1
import random
Q = secp256k1.q
eng = default_engine()
nm = 8192
def gens(k, seed):
    ks = b"".join(random.Random(seed * 7 + i).randrange(1, Q).to_bytes(32, "little") for i in range(k))
    out = eng.ec_mul_batch_bytes(secp256k1.G.to_le64() * k, ks, k)
    return [Point.from_le64(out[64 * i: 64 * i + 64]) for i in range(k)]
gs, hs = PackedPoints(gens(nm, 1)), PackedPoints(gens(nm, 2))
h = gens(1, 3)
base = PackedPoints.join(gs, hs, h)
rnd = random.Random(5)
bits = [rnd.randrange(2) for _ in range(nm)]
aL = bits; aR = [(b - 1) % Q for b in bits]
sA = PackedScalars(aL + aR + [12345])
sS = PackedScalars([rnd.randrange(Q) for _ in range(2 * nm + 1)])
for name, fn in (("A alone", lambda: PipSECP256k1.multiexp(base, sA)), ("S alone", lambda: PipSECP256k1.multiexp(base, sS)),
                 ("A,S pair", lambda: PipSECP256k1.multiexp2(base, sA, base, sS)), ("T pair", lambda: PipSECP256k1.multiexp2(h + h, [5, 7], h + h, [9, 11]))):
    fn()
    eng.profile(1); eng.profile_reset()
    t = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t) / 5
    pr = eng.profile_read(); eng.profile(False)
    print(name, "%.3f ms" % (dt * 1e3), {k: round(v[0] / 5, 3) for k, v in pr.items() if v[1]})
