#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.json by running the
REFERENCE'S OWN, UNMODIFIED Python (imported from /root/reference) on fixed
seeds.  Run in the build container only (the reference does not exist on the
GPU box):

    python tests/golden/make_golden.py            # all fixtures (~3 min)
    python tests/golden/make_golden.py multiexp   # one family

The reference imports the third-party C extension `fastecdsa`, which is absent
from /root/reference and not installable here.  This
script therefore puts a four-file adapter package named `fastecdsa` into a
temporary directory (NOT the repo) that re-exports oracle/ec.py's restatement
of the secp256k1 affine group law under the names the reference imports.
Everything above that boundary -- Pippenger, commitments, transcript, mod_hash,
IPA, range proofs -- is the reference's own code, so the vectors pin all of it;
oracle/ec.py itself is pinned by public secp256k1 known answers
(tests/test_oracle_ec.py).

Fixtures store seeds and outputs (hex), never bulk inputs: inputs are
re-derived from the seeds by elliptic_hash / mod_hash, which fixture
`hash_codec.json` pins.
"""
import json
import os
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"

sys.dont_write_bytecode = True  # /root/reference is read-only
sys.setrecursionlimit(100000)
sys.path.insert(0, REPO)


def _install_adapter():
    d = tempfile.mkdtemp(prefix="bp_golden_")
    pkg = os.path.join(d, "fastecdsa")
    os.makedirs(pkg)
    files = {
        "__init__.py": "",
        "point.py": "from oracle.ec import Point\n",
        "curve.py": "from oracle.ec import Curve, secp256k1\n",
        "util.py": "from oracle.ec import mod_sqrt\n",
    }
    for name, text in files.items():
        with open(os.path.join(pkg, name), "w") as f:
            f.write(text)
    sys.path.insert(0, d)
    sys.path.insert(0, REFERENCE)


_install_adapter()

from oracle import ec as OE  # noqa: E402

# count group operations exactly as the reference performs them
_ADD = OE.Point.__add__
OPS = [0]


def _counted_add(self, other):
    OPS[0] += 1
    return _ADD(self, other)


OE.Point.__add__ = _counted_add

from src.pippenger import PipSECP256k1, Pippenger  # noqa: E402
from src.pippenger.group import MultIntModP  # noqa: E402
from src.pippenger.modp import ModP as PModP  # noqa: E402
from src.utils.utils import ModP, mod_hash, point_to_b64, point_to_bytes, inner_product  # noqa: E402
from src.utils.transcript import Transcript  # noqa: E402
from src.utils.elliptic_curve_hash import elliptic_hash  # noqa: E402
from src.utils.commitments import vector_commitment, commitment  # noqa: E402
from src.innerproduct.inner_product_prover import NIProver, FastNIProver2  # noqa: E402
from src.innerproduct.inner_product_verifier import Verifier1, Verifier2  # noqa: E402
from src.rangeproofs import (  # noqa: E402
    NIRangeProver, RangeVerifier, AggregNIRangeProver, AggregRangeVerifier,
)

CURVE = OE.secp256k1
Q = CURVE.q


def hx(v):
    return "%x" % int(v)


def pt(P):
    return [hx(P.x), hx(P.y)]


def seed(i):
    return bytes([i]) * 10


def gens(n, s):
    return [elliptic_hash(str(i).encode() + s, CURVE) for i in range(n)]


def scal(n, s):
    return [mod_hash(str(i).encode() + s, Q) for i in range(n)]


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote %s (%d bytes)" % (name, os.path.getsize(path)))


# ---------------------------------------------------------------------------
def gen_hash_codec():
    msgs = [b"", b"test", b"a", b"bulletproofs", seed(1), seed(7), b"\x00" * 32,
            b"0" + seed(3), b"17" + seed(3), b"alpha" + b"AQEBAQEBAQEBAQ==&"]
    out = {"mod_hash_q": [[m.hex(), hx(mod_hash(m, Q).x)] for m in msgs]}
    out["mod_hash_small"] = [[m.hex(), p, mod_hash(m, p).x] for m in msgs for p in (17, 1009)]
    out["elliptic_hash"] = [[m.hex(), pt(elliptic_hash(m, CURVE))] for m in msgs]
    G = CURVE.G
    pts = [G, 2 * G, 3 * G, (Q - 1) * G, elliptic_hash(b"test", CURVE)]
    out["point_to_b64"] = [[pt(P), point_to_b64(P).decode()] for P in pts]
    out["point_to_bytes_identity"] = point_to_bytes(OE.Point.IDENTITY_ELEMENT).hex()
    tr = Transcript(seed(5))
    tr.add_point(G)
    tr.add_number(mod_hash(b"x", Q))
    tr.add_list_points([2 * G, 3 * G])
    out["transcript"] = {"seed": seed(5).hex(), "digest": tr.digest.decode(),
                         "challenge": hx(tr.get_modp(Q).x)}
    dump("hash_codec.json", out)


def gen_multiexp():
    cases = []

    def run(label, n, gs, es, es_desc):
        OPS[0] = 0
        R = PipSECP256k1.multiexp(gs, es)
        cases.append({"label": label, "n": n, "scalars": es_desc, "result": pt(R), "ops": OPS[0]})

    sg, ss_ = seed(11), seed(12)
    for n in (0, 1, 2, 3, 16, 33, 128, 129, 513, 1024, 4096):
        t = time.time()
        gs = gens(n, sg)
        run("random", n, gs, scal(n, ss_), "mod_hash(str(i)+seed12)")
        print("  multiexp random n=%d %.1fs" % (n, time.time() - t))
    for n in (16, 128, 513):
        gs = gens(n, sg)
        bits = [mod_hash(str(i).encode() + seed(13), Q).x for i in range(n)]
        # A-commitment shape (rangeproof_prover.py:42-47): first half in {0,1},
        # second half in {q-1, 0}
        es = [b & 1 for b in bits[: n // 2]] + [((b & 1) - 1) % Q for b in bits[n // 2:]]
        run("acommit", n, gs, es, "lo half bit0 of mod_hash(str(i)+seed13); hi half (bit0-1)%q")
    for n in (16, 33):
        gs = gens(n, sg)
        es = scal(n, ss_)
        # duplicates and cancelling pairs
        gs2 = list(gs)
        gs2[1] = gs2[0]
        gs2[3] = -gs2[2]
        es2 = list(es)
        es2[3] = es2[2]          # e*P + e*(-P) cancels
        gs2[5] = gs2[4]
        es2[5] = -es2[4]         # e*P + (-e)*P cancels (ModP.__neg__)
        run("dup_neg", n, gs2, es2, "g1=g0, g3=-g2 & e3=e2, g5=g4 & e5=-e4")
        es3 = [int(e.x) + Q * (i % 3) if i % 2 else -int(e.x) for i, e in enumerate(es)]
        run("unreduced", n, gs, es3, "odd i: e+q*(i%3), even i: -e  (python ints)")
        run("all_zero", n, gs, [0] * n, "0")
        run("all_same", n, gs, [es[0]] * n, "e0 for all")
        run("same_point", n, [gs[0]] * n, es, "g0 for all")
    dump("multiexp.json", {"seed_points": sg.hex(), "seed_scalars": ss_.hex(), "cases": cases})


def gen_multiexp_big():
    """ONE reference run at BASELINE config C2's size, n = 2^16 ("feasible once": 23.7 M group operations,
    ~13 min and ~11 GB of subset tables in the reference's own algorithm).  Not part of the default families: run it as
    `python tests/golden/make_golden.py multiexp_big`."""
    n = 1 << 16
    sg, ss_ = seed(11), seed(12)
    t = time.time()
    gs = gens(n, sg)
    es = scal(n, ss_)
    print("  inputs %.1fs" % (time.time() - t), flush=True)
    OPS[0] = 0
    R = PipSECP256k1.multiexp(gs, es)
    dump("multiexp_big.json", {"seed_points": sg.hex(), "seed_scalars": ss_.hex(), "n": n, "scalars": "mod_hash(str(i)+seed12)",
                               "result": pt(R), "ops": OPS[0], "reference_seconds": round(time.time() - t, 1)})


def gen_modp_group():
    # Pippenger over (Z/p)* -- the generic operator API + mult counter
    p = 1000003
    out = []
    for n in (1, 4, 16, 40):
        PModP.reset()
        G = MultIntModP(p, p - 1)
        pip = Pippenger(G)
        gs = [PModP(2 + 3 * i, p) for i in range(n)]
        es = [(12345 * (i + 1) ** 3) % (p - 1) for i in range(n)]
        PModP.reset()
        r = pip.multiexp(gs, es)
        out.append({"p": p, "n": n, "result": r.x, "num_of_mult": PModP.num_of_mult})
    dump("modp_group.json", {"cases": out})


def _proof2_json(p2):
    return {"a": hx(p2.a.x), "b": hx(p2.b.x), "xs": [hx(x.x) for x in p2.xs],
            "Ls": [pt(L) for L in p2.Ls], "Rs": [pt(R) for R in p2.Rs],
            "transcript": p2.transcript.decode(), "start_transcript": p2.start_transcript}


def gen_ipa():
    cases = []
    for k in range(9):
        n = 2 ** k
        s = [seed(20 + j) for j in range(6)]
        g, h = gens(n, s[0]), gens(n, s[1])
        u = elliptic_hash(s[2], CURVE)
        a, b = scal(n, s[3]), scal(n, s[4])
        c = inner_product(a, b)
        P = vector_commitment(g, h, a, b) + c * u
        p2 = FastNIProver2(g, h, u, P, a, b, CURVE).prove()
        V2 = Verifier2(g, h, u, P, p2)
        ok2 = V2.verify()
        ss = V2.get_ss(p2.xs)
        P1 = vector_commitment(g, h, a, b)
        p1 = NIProver(g, h, u, P1, c, a, b, CURVE, s[5]).prove()
        ok1 = Verifier1(g, h, u, P1, c, p1).verify()
        cases.append({
            "n": n, "seeds": [x.hex() for x in s], "P": pt(P), "c": hx(c.x),
            "proof2": _proof2_json(p2), "verify2": ok2, "ss": [hx(x.x) for x in ss],
            "P1": pt(P1),
            "proof1": {"u_new": pt(p1.u_new), "P_new": pt(p1.P_new),
                       "transcript": p1.transcript.decode(), "proof2": _proof2_json(p1.proof2)},
            "verify1": ok1,
        })
        print("  ipa n=%d done" % n)
    dump("ipa.json", {"cases": cases})


def _rp_json(pr):
    ip = pr.innerProof
    return {"taux": hx(pr.taux.x), "mu": hx(pr.mu.x), "t_hat": hx(pr.t_hat.x),
            "T1": pt(pr.T1), "T2": pt(pr.T2), "A": pt(pr.A), "S": pt(pr.S),
            "transcript": pr.transcript.decode(),
            "inner": {"u_new": pt(ip.u_new), "P_new": pt(ip.P_new),
                      "transcript": ip.transcript.decode(), "proof2": _proof2_json(ip.proof2)}}


def _expect_invalid(fn):
    try:
        fn()
    except Exception as e:  # the reference raises a plain Exception
        return str(e)
    return "ACCEPTED"


def gen_rangeproofs():
    single, aggreg = [], []
    for k in range(1, 8):
        n = 2 ** k
        s = [seed(40 + j) for j in range(7)]
        v = ModP(int.from_bytes(mod_hash(b"v" + s[0], Q).x.to_bytes(32, "big")[-(n + 7) // 8:], "big") % 2 ** n, Q)
        gs, hs = gens(n, s[0]), gens(n, s[1])
        g, h, u = (elliptic_hash(s[j], CURVE) for j in (2, 3, 4))
        gamma = mod_hash(s[5], Q)
        V = commitment(g, h, v, gamma)
        t = time.time()
        pr = NIRangeProver(v, n, g, h, gs, hs, gamma, u, CURVE, s[6]).prove()
        ok = RangeVerifier(V, g, h, gs, hs, u, pr).verify()
        case = {"n": n, "v": hx(v.x), "seeds": [x.hex() for x in s], "V": pt(V),
                "proof": _rp_json(pr), "verify": ok}
        if n == 16:
            # cheating mutations of src/tests/test_rangeproofs.py:57-114
            bad_v = ModP(2 ** n + 5, Q)
            prb = NIRangeProver(bad_v, n, g, h, gs, hs, gamma, u, CURVE, s[6]).prove()
            case["cheat_out_of_range"] = _expect_invalid(
                RangeVerifier(commitment(g, h, bad_v, gamma), g, h, gs, hs, u, prb).verify)
            case["cheat_wrong_commitment"] = _expect_invalid(
                RangeVerifier(commitment(g, h, v + 1, gamma), g, h, gs, hs, u, pr).verify)
        single.append(case)
        print("  range n=%d %.1fs" % (n, time.time() - t))
    for (m, n) in ((4, 16), (2, 64), (32, 16)):
        s = [seed(60 + j) for j in range(7)]
        vs = [ModP(mod_hash(str(j).encode() + b"v" + s[0], Q).x % 2 ** n, Q) for j in range(m)]
        vs[-1] = ModP(2 ** n - 1, Q)
        gs, hs = gens(n * m, s[0]), gens(n * m, s[1])
        g, h, u = (elliptic_hash(s[j], CURVE) for j in (2, 3, 4))
        gammas = [mod_hash(str(j).encode() + s[5], Q) for j in range(m)]
        Vs = [commitment(g, h, vs[j], gammas[j]) for j in range(m)]
        t = time.time()
        pr = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, CURVE, s[6]).prove()
        ok = AggregRangeVerifier(Vs, g, h, gs, hs, u, pr).verify()
        case = {"m": m, "n": n, "vs": [hx(v.x) for v in vs], "seeds": [x.hex() for x in s],
                "Vs": [pt(V) for V in Vs], "proof": _rp_json(pr), "verify": ok}
        if (m, n) == (4, 16):
            Vs_bad = list(Vs)
            Vs_bad[1] = commitment(g, h, vs[1] + 1, gammas[1])
            case["cheat_wrong_commitment"] = _expect_invalid(
                AggregRangeVerifier(Vs_bad, g, h, gs, hs, u, pr).verify)
        aggreg.append(case)
        print("  aggreg m=%d n=%d %.1fs" % (m, n, time.time() - t))
    dump("rangeproofs.json", {"single": single, "aggregated": aggreg})


FAMILIES = {
    "hash_codec": gen_hash_codec,
    "multiexp": gen_multiexp,
    "multiexp_big": gen_multiexp_big,
    "modp_group": gen_modp_group,
    "ipa": gen_ipa,
    "rangeproofs": gen_rangeproofs,
}

if __name__ == "__main__":
    which = sys.argv[1:] or [f for f in FAMILIES if f != "multiexp_big"]
    # the reference prints "OK" on every successful Verifier2.verify
    for name in which:
        t0 = time.time()
        FAMILIES[name]()
        print("%s: %.1fs" % (name, time.time() - t0))
