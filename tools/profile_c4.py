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
gs, hs = gens(n * m, 1), gens(n * m, 2)
g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
vs = [ModP(int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big"), Q) for j in range(m)]
gammas = [mod_hash(b"gamma%d" % j, Q) for j in range(m)]
Vs = [commitment(g, h, vs[j], gammas[j]) for j in range(m)]
proof = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, b"seed").prove()
for what in ("prove", "verify"):
    p = cProfile.Profile(); t = time.perf_counter(); p.enable()
    if what == "prove":
        proof = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, b"seed").prove()
    else:
        assert AggregRangeVerifier(Vs, g, h, gs, hs, u, proof).verify()
    p.disable(); print(what, "ms", (time.perf_counter() - t) * 1e3)
    pstats.Stats(p).sort_stats("tottime").print_stats(12)
