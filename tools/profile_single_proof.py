import sys, time, cProfile, pstats, hashlib
sys.path.insert(0, "/root/repo")
import bulletproofs_amd
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.rangeproofs import NIRangeProver, RangeVerifier
from bulletproofs_amd.utils import ModP, commitment, elliptic_hash, mod_hash
Q = secp256k1.q
n = 64
gs = [elliptic_hash(str(i).encode() + b"gs") for i in range(n)]
hs = [elliptic_hash(str(i).encode() + b"hs") for i in range(n)]
g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
def one(j):
    v = ModP(int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big"), Q)
    gamma = mod_hash(b"gamma%d" % j, Q)
    V = commitment(g, h, v, gamma)
    pr = NIRangeProver(v, n, g, h, gs, hs, gamma, u, secp256k1, b"seed%d" % j).prove()
    return V, pr
for j in range(3): V, pr = one(j)
t = time.perf_counter()
for j in range(20): V, pr = one(j)
print("prove ms", (time.perf_counter() - t) / 20 * 1e3)
t = time.perf_counter()
for j in range(20): RangeVerifier(V, g, h, gs, hs, u, pr).verify()
print("verify ms", (time.perf_counter() - t) / 20 * 1e3)
p = cProfile.Profile(); p.enable()
for j in range(20): V, pr = one(j)
p.disable(); pstats.Stats(p).sort_stats("tottime").print_stats(14)
p = cProfile.Profile(); p.enable()
for j in range(20): RangeVerifier(V, g, h, gs, hs, u, pr).verify()
p.disable(); pstats.Stats(p).sort_stats("tottime").print_stats(10)
