#!/usr/bin/env python3
"""cProfile of config C4 (aggregated 128 x 64-bit range proof) through the host layer."""
import cProfile, hashlib, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.rangeproofs import AggregNIRangeProver, AggregRangeVerifier
from bulletproofs_amd.utils import ModP, commitment, elliptic_hash, mod_hash
Q = secp256k1.q
m, n = 128, 64
from bulletproofs_amd.engine import default_engine
eng = default_engine()
# generators k_i * G (fast to derive; the timing does not depend on how they were made)
import random
rnd = random.Random(1)
from bulletproofs_amd.ec import Point
def gens(k, seed):
    ks = b"".join(random.Random(seed * 7 + i).randrange(1, Q).to_bytes(32, "little") for i in range(k))
    out = eng.ec_mul_batch_bytes(secp256k1.G.to_le64() * k, ks, k)
    return [Point.from_le64(out[64 * i: 64 * i + 64]) for i in range(k)]
from bulletproofs_amd.ec import PackedPoints
gs, hs = PackedPoints(gens(n * m, 1)), PackedPoints(gens(n * m, 2))      # lists of Points with their wire form attached (packed once)
g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
vs = [ModP(int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big"), Q) for j in range(m)]
gammas = [mod_hash(b"gamma%d" % j, Q) for j in range(m)]
Vs = [commitment(g, h, vs[j], gammas[j]) for j in range(m)]
proof = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, b"seed").prove()
for what in ("prove", "verify"):
    for _ in range(2):                                   # unprofiled wall time first
        t = time.perf_counter()
        if what == "prove":
            proof = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, b"seed").prove()
        else:
            assert AggregRangeVerifier(Vs, g, h, gs, hs, u, proof).verify()
        print(what, "unprofiled ms", round((time.perf_counter() - t) * 1e3, 3))
    p = cProfile.Profile(); t = time.perf_counter(); p.enable()
    if what == "prove":
        proof = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, b"seed").prove()
    else:
        assert AggregRangeVerifier(Vs, g, h, gs, hs, u, proof).verify()
    p.disable(); print(what, "ms", (time.perf_counter() - t) * 1e3)
    pstats.Stats(p).sort_stats("tottime").print_stats(12)
    eng.profile(1); eng.profile_reset()
    t = time.perf_counter()
    if what == "prove":
        proof = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, b"seed").prove()
    else:
        assert AggregRangeVerifier(Vs, g, h, gs, hs, u, proof).verify()
    dt = time.perf_counter() - t
    pr = eng.profile_read(); eng.profile(False)
    print(what, "with GPU stage timers: %.3f ms; stage ms:" % (dt * 1e3), {k: round(v[0], 3) for k, v in pr.items() if v[1]}, "sum %.3f" % sum(v[0] for v in pr.values()))
