"""Wire format + GPU batch decompression (bpmi_ec_decompress_batch) vs the oracle's
bytes_to_point (restating src/utils/utils.py:119-131), and serialise -> parse -> verify
round trips of whole proofs."""
import random

import pytest

from helpers import Q, gens
from oracle import bp_ref as R
from oracle import cbind
from oracle.ec import INF, secp256k1

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


def test_decompress_batch_vs_oracle(gp):
    pts, _ = gp.rand_points(500, 91)
    pts[3] = INF
    comp = b"".join(R.point_to_bytes(p) if p != INF else bytes(33) for p in pts)
    out, ok = gp.engine().ec_decompress_batch_bytes(comp, len(pts))
    assert ok == bytes([1]) * len(pts)
    assert out == cbind.pack_points(pts)
    for p in pts[:20]:
        if p != INF:
            assert R.bytes_to_point(R.point_to_bytes(p)) == p          # the oracle agrees with itself
    # both parities of the same x
    p = pts[0]
    both = b"\x02" + p.x.to_bytes(32, "big") + b"\x03" + p.x.to_bytes(32, "big")
    out, ok = gp.engine().ec_decompress_batch_bytes(both, 2)
    ys = {int.from_bytes(out[32:64], "little"), int.from_bytes(out[96:128], "little")}
    assert ok == b"\x01\x01" and ys == {p.y, secp256k1.p - p.y}
    assert int.from_bytes(out[32:64], "little") % 2 == 0 and int.from_bytes(out[96:128], "little") % 2 == 1


def test_decompress_rejects_invalid_encodings(gp):
    P = secp256k1.p
    rnd = random.Random(4)
    nonres = next(x for x in (rnd.randrange(P) for _ in range(1000)) if pow((x**3 + 7) % P, (P - 1) // 2, P) != 1)
    good = gp.rand_points(1, 5)[0][0]
    cases = [
        b"\x02" + nonres.to_bytes(32, "big"),                 # x^3 + 7 is not a square
        b"\x04" + good.x.to_bytes(32, "big"),                 # unknown tag
        b"\x02" + (P + 5).to_bytes(32, "big"),                # x >= p
        b"\x00" + (1).to_bytes(32, "big"),                    # identity tag with a non-zero body
        b"\x03" + good.x.to_bytes(32, "big"),                 # valid
        bytes(33),                                            # identity
    ]
    out, ok = gp.engine().ec_decompress_batch_bytes(b"".join(cases), len(cases))
    assert ok == bytes([0, 0, 0, 0, 1, 1])
    assert out[: 64 * 4] == bytes(64 * 4) and out[64 * 5:] == bytes(64)


def test_proof_wire_round_trip(gp):
    from bulletproofs_amd.ec import secp256k1 as curve
    from bulletproofs_amd.rangeproofs import (NIRangeProver, RangeVerifier, batch_verify, proof_to_bytes,
                                               proofs_from_bytes)
    from bulletproofs_amd.utils import ModP, mod_hash, commitment
    n = 16
    gs, hs = gp.to_gpu_list(gens(n, b"wgs")), gp.to_gpu_list(gens(n, b"whs"))
    g, h, u = (gp.to_gpu(R.elliptic_hash(s)) for s in (b"wg", b"wh", b"wu"))
    rnd = random.Random(6)
    Vs, proofs = [], []
    for k in range(5):
        v = ModP(rnd.randrange(2 ** n), Q)
        gamma = mod_hash(b"wgamma%d" % k, Q)
        Vs.append(commitment(g, h, v, gamma))
        proofs.append(NIRangeProver(v, n, g, h, gs, hs, gamma, u, curve, b"wseed%d" % k).prove())
    blobs = [proof_to_bytes(p) for p in proofs]
    assert all(len(b) == len(blobs[0]) or True for b in blobs)
    back = proofs_from_bytes(blobs)
    for a, b in zip(proofs, back):
        assert proof_to_bytes(b) == proof_to_bytes(a)
        assert (a.taux.x, a.mu.x, a.t_hat.x) == (b.taux.x, b.mu.x, b.t_hat.x)
        assert a.A == b.A and a.innerProof.proof2.Ls == b.innerProof.proof2.Ls
        assert a.transcript == b.transcript and a.innerProof.proof2.start_transcript == b.innerProof.proof2.start_transcript
    for V, pr in zip(Vs, back):
        assert RangeVerifier(V, g, h, gs, hs, u, pr).verify() is True
    assert batch_verify(Vs, back, g, h, gs, hs, u) is True
    # malformed blobs
    for bad in (blobs[0][:-1], b"XXXXX" + blobs[0][5:], blobs[0] + b"\x00", b""):
        with pytest.raises(Exception, match="Proof invalid"):
            proofs_from_bytes([bad])
    # a point whose x is not on the curve
    k = blobs[0][5]
    off = 6 + 32 * (5 + k)
    P = secp256k1.p
    nonres = next(x for x in range(2, 1000) if pow((x**3 + 7) % P, (P - 1) // 2, P) != 1)
    evil = blobs[0][:off] + b"\x02" + nonres.to_bytes(32, "big") + blobs[0][off + 33:]
    with pytest.raises(Exception, match="Proof invalid"):
        proofs_from_bytes([evil])
