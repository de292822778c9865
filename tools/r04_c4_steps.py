#!/usr/bin/env python3
"""Where config C4's proving time goes, call by call: every function the prover calls (native or Python) wrapped with a wall-clock
timer, five proofs, the median of each.  python tools/r04_c4_steps.py"""
import hashlib, os, statistics, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa
from bulletproofs_amd import _native, engine as E
from bulletproofs_amd.ec import PackedPoints, PackedScalars, Point, secp256k1
from bulletproofs_amd.pippenger import PipSECP256k1
from bulletproofs_amd.rangeproofs import AggregNIRangeProver, common
from bulletproofs_amd.innerproduct import inner_product_prover as ipp
from bulletproofs_amd.utils import ModP, elliptic_hash, mod_hash
import random
Q = secp256k1.q
m, n = 128, 64
eng = E.default_engine()


def gens(k, seed):
    ks = b"".join(random.Random(seed * 7 + i).randrange(1, Q).to_bytes(32, "little") for i in range(k))
    out = eng.ec_mul_batch_bytes(secp256k1.G.to_le64() * k, ks, k)
    return [Point.from_le64(out[64 * i: 64 * i + 64]) for i in range(k)]


gs, hs = PackedPoints(gens(n * m, 1)), PackedPoints(gens(n * m, 2))
g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
vs = [ModP(int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big"), Q) for j in range(m)]
gammas = [mod_hash(b"gamma%d" % j, Q) for j in range(m)]
times = {}


def wrap(obj, name, label=None):
    f = getattr(obj, name)
    label = label or name

    def g_(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            times.setdefault(label, []).append(time.perf_counter() - t)
    setattr(obj, name, g_)


wrap(common, "_mod_hash_ints")
wrap(PipSECP256k1, "multiexp2")
wrap(PipSECP256k1, "multiexp")
lib = _native.load()
for nm_ in ("bpmi_rp_poly_coeffs", "bpmi_rp_final_vectors"):
    wrap(lib, nm_)
wrap(ipp.NIProver, "prove", "NIProver.prove (inner-product argument, whole)")
wrap(E.IpaState, "prove_rounds", "  of it: bpmi_ipa_prove_rounds")
wrap(E.Engine, "ipa_create", "  of it: ipa_create (uploads)")
wrap(PackedPoints, "join", "PackedPoints.join")
wrap(PackedScalars, "join", "PackedScalars.join")
tot = []
for rep in range(7):
    for v in times.values():
        v.append(None)
    t = time.perf_counter()
    AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, b"seed").prove()
    tot.append(time.perf_counter() - t)
print("C4 prove, 7 proofs: median %.3f ms (min %.3f)" % (statistics.median(tot) * 1e3, min(tot) * 1e3))
acc = 0.0
for label, v in times.items():
    per, cur = [], 0.0
    for x in v[1:] + [None]:
        if x is None:
            per.append(cur); cur = 0.0
        else:
            cur += x
    per = per[2:]                       # the first two proofs warm up
    med = statistics.median(per) * 1e3
    if not label.startswith("  "):
        acc += med
    print("  %-52s %7.3f ms per proof" % (label, med))
print("  %-52s %7.3f ms" % ("everything else (Python between the calls)", statistics.median(tot[2:]) * 1e3 - acc))
