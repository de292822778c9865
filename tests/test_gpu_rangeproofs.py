"""GPU parity of the range-proof callers (NIRangeProver / RangeVerifier /
AggregNIRangeProver / AggregRangeVerifier of the product package) vs the goldens
produced by the reference; mirrors src/tests/test_rangeproofs.py and
test_aggreg_rangeproofs.py (n = 2..128, m = 2..32, every cheating case)."""
import pytest

from conftest import load_golden
from helpers import P, Q, gens, hx
from oracle import bp_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


def check_range_proof(gp, pr, want):
    assert hx(pr.taux.x % Q) == want["taux"] and hx(pr.mu.x % Q) == want["mu"] and hx(pr.t_hat.x % Q) == want["t_hat"]
    for name in ("T1", "T2", "A", "S"):
        assert gp.same_point(getattr(pr, name), P(want[name])), name
    assert pr.transcript.decode() == want["transcript"]
    ip = pr.innerProof
    assert gp.same_point(ip.u_new, P(want["inner"]["u_new"])) and gp.same_point(ip.P_new, P(want["inner"]["P_new"]))
    assert ip.transcript.decode() == want["inner"]["transcript"]
    w2 = want["inner"]["proof2"]
    p2 = ip.proof2
    assert hx(p2.a.x) == w2["a"] and hx(p2.b.x) == w2["b"] and [hx(x.x) for x in p2.xs] == w2["xs"]
    assert all(gp.same_point(a, P(b)) for a, b in zip(p2.Ls, w2["Ls"]))
    assert all(gp.same_point(a, P(b)) for a, b in zip(p2.Rs, w2["Rs"]))
    assert p2.transcript.decode() == w2["transcript"] and p2.start_transcript == w2["start_transcript"]


def inputs(gp, c, m):
    s = [bytes.fromhex(x) for x in c["seeds"]]
    n = c["n"]
    gs, hs = gp.to_gpu_list(gens(n * m, s[0])), gp.to_gpu_list(gens(n * m, s[1]))
    g, h, u = (gp.to_gpu(R.elliptic_hash(s[j])) for j in (2, 3, 4))
    return s, n, gs, hs, g, h, u


@pytest.mark.parametrize("k", range(7))
def test_single_rangeproof_goldens(gp, k):
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.rangeproofs import NIRangeProver, RangeVerifier
    from bulletproofs_amd.utils import ModP, mod_hash, commitment
    c = load_golden("rangeproofs.json")["single"][k]
    s, n, gs, hs, g, h, u = inputs(gp, c, 1)
    v = ModP(int(c["v"], 16), Q)
    gamma = mod_hash(s[5], Q)
    V = commitment(g, h, v, gamma)
    assert gp.same_point(V, P(c["V"]))
    pr = NIRangeProver(v, n, g, h, gs, hs, gamma, u, secp256k1, s[6]).prove()
    check_range_proof(gp, pr, c["proof"])
    assert RangeVerifier(V, g, h, gs, hs, u, pr).verify() is True
    if "cheat_wrong_commitment" in c:
        with pytest.raises(Exception, match="Proof invalid"):
            RangeVerifier(commitment(g, h, v + 1, gamma), g, h, gs, hs, u, pr).verify()
        bad = ModP(2 ** n + 5, Q)
        prb = NIRangeProver(bad, n, g, h, gs, hs, gamma, u, secp256k1, s[6]).prove()
        with pytest.raises(Exception, match="Proof invalid"):
            RangeVerifier(commitment(g, h, bad, gamma), g, h, gs, hs, u, prb).verify()
        good = pr.transcript
        items = good.split(b"&")
        items[3] = b"12345"
        pr.transcript = b"&".join(items)
        with pytest.raises(Exception, match="Proof invalid"):
            RangeVerifier(V, g, h, gs, hs, u, pr).verify()
        pr.transcript = good


@pytest.mark.parametrize("k", range(3))
def test_aggregated_rangeproof_goldens(gp, k):
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.rangeproofs import AggregNIRangeProver, AggregRangeVerifier
    from bulletproofs_amd.utils import ModP, mod_hash, commitment
    c = load_golden("rangeproofs.json")["aggregated"][k]
    m = c["m"]
    s, n, gs, hs, g, h, u = inputs(gp, c, m)
    vs = [ModP(int(v, 16), Q) for v in c["vs"]]
    gammas = [mod_hash(str(j).encode() + s[5], Q) for j in range(m)]
    Vs = [commitment(g, h, vs[j], gammas[j]) for j in range(m)]
    assert all(gp.same_point(a, P(b)) for a, b in zip(Vs, c["Vs"]))
    pr = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, s[6]).prove()
    check_range_proof(gp, pr, c["proof"])
    assert AggregRangeVerifier(Vs, g, h, gs, hs, u, pr).verify() is True
    if "cheat_wrong_commitment" in c:
        bad = list(Vs)
        bad[1] = commitment(g, h, vs[1] + 1, gammas[1])
        with pytest.raises(Exception, match="Proof invalid"):
            AggregRangeVerifier(bad, g, h, gs, hs, u, pr).verify()
