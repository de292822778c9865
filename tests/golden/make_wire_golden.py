#!/usr/bin/env python3
"""tests/golden/wire_formats.json: the package's three wire formats (rangeproofs/codec.py) of the REFERENCE-MADE golden proofs of
rangeproofs.json -- length and SHA-256 of every serialisation, and the full bytes (hex) of the smallest proof in every format.  The
reference has no serialisation (only point_to_bytes / bytes_to_point, src/utils/utils.py:100-131): these vectors pin OUR byte layout
on the reference's own proofs, so that a change of the codec, of the host expander or of the device expander cannot go unnoticed.
Needs no reference and no GPU (the inputs are the committed goldens):    python tests/golden/make_wire_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.rangeproofs.codec import proof_to_bytes  # noqa: E402


def proof_of(want):
    """The Proof object whose fields are a golden's (tests/golden/rangeproofs.json)."""
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.innerproduct.inner_product_verifier import Proof1, Proof2
    from bulletproofs_amd.rangeproofs.common import Proof
    from bulletproofs_amd.utils import ModP
    from bulletproofs_amd.ec import Point

    def P(xy):
        x, y = int(xy[0], 16), int(xy[1], 16)
        return Point.IDENTITY_ELEMENT if x == 0 and y == 0 else Point(x, y, secp256k1)
    q = secp256k1.q
    sc = lambda h: ModP(int(h, 16), q)
    inner, w2 = want["inner"], want["inner"]["proof2"]
    p2 = Proof2(sc(w2["a"]), sc(w2["b"]), [sc(x) for x in w2["xs"]], [P(p) for p in w2["Ls"]], [P(p) for p in w2["Rs"]], w2["transcript"].encode(),
                w2["start_transcript"])
    p1 = Proof1(P(inner["u_new"]), P(inner["P_new"]), p2, inner["transcript"].encode())
    return Proof(sc(want["taux"]), sc(want["mu"]), sc(want["t_hat"]), P(want["T1"]), P(want["T2"]), P(want["A"]), P(want["S"]), p1, want["transcript"].encode())


def entries():
    with open(os.path.join(HERE, "rangeproofs.json")) as f:
        gold = json.load(f)
    out = []
    for family in ("single", "aggregated"):
        for i, c in enumerate(gold[family]):
            pr = proof_of(c["proof"])
            e = {"family": family, "index": i}
            for v in (1, 2, 3):
                b = proof_to_bytes(pr, version=v)
                e["format_%d" % v] = {"bytes": len(b), "sha256": hashlib.sha256(b).hexdigest()}
                if family == "single" and i == 0:
                    e["format_%d" % v]["hex"] = b.hex()
            out.append(e)
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "wire_formats.json"), "w") as f:
        json.dump({"proofs": entries()}, f, indent=1)
    print("wrote wire_formats.json")
