"""Pins the Python restatement (oracle/bp_ref.py) to the vectors produced by the
reference's own code (tests/golden/make_golden.py)."""
import pytest

from conftest import load_golden
from helpers import P, Q, gens, hx, scal, seed
from oracle import bp_ref as R
from oracle.ec import INF, secp256k1

G = secp256k1.G


def test_hash_codec_golden():
    g = load_golden("hash_codec.json")
    for m, want in g["mod_hash_q"]:
        assert hx(R.mod_hash(bytes.fromhex(m), Q).x) == want
    for m, p, want in g["mod_hash_small"]:
        assert R.mod_hash(bytes.fromhex(m), p).x == want
    for m, want in g["elliptic_hash"]:
        assert R.elliptic_hash(bytes.fromhex(m)) == P(want)
    for xy, want in g["point_to_b64"]:
        assert R.point_to_b64(P(xy)).decode() == want
        assert R.b64_to_point(want.encode()) == P(xy)
    assert R.point_to_bytes(INF).hex() == g["point_to_bytes_identity"]
    t = g["transcript"]
    tr = R.Transcript(bytes.fromhex(t["seed"]))
    tr.add_point(G)
    tr.add_number(R.mod_hash(b"x", Q))
    tr.add_list_points([2 * G, 3 * G])
    assert tr.digest.decode() == t["digest"] and hx(tr.get_modp(Q).x) == t["challenge"]
    # survey-recorded values
    assert R.mod_hash(b"test", Q).x == 18346318663465204740897678531052866794504627225558141060753557145136529746486
    assert R.point_to_b64(G) == b"Anm+Zn753LusVaBilc6HCwcCm/zbLc4o2VnygVsW+BeY"


def multiexp_case_inputs(case, sg, ss):
    """Rebuild the (points, scalars) of a multiexp golden case from its label."""
    n, label = case["n"], case["label"]
    gs, es = gens(n, sg), scal(n, ss)
    if label == "random":
        return gs, es
    if label == "acommit":
        bits = [R.mod_hash(str(i).encode() + seed(13), Q).x for i in range(n)]
        return gs, [b & 1 for b in bits[: n // 2]] + [((b & 1) - 1) % Q for b in bits[n // 2:]]
    if label == "dup_neg":
        gs2, es2 = list(gs), list(es)
        gs2[1] = gs2[0]
        gs2[3] = -gs2[2]
        es2[3] = es2[2]
        gs2[5] = gs2[4]
        es2[5] = -es2[4]
        return gs2, es2
    if label == "unreduced":
        return gs, [int(e.x) + Q * (i % 3) if i % 2 else -int(e.x) for i, e in enumerate(es)]
    if label == "all_zero":
        return gs, [0] * n
    if label == "all_same":
        return gs, [es[0]] * n
    if label == "same_point":
        return [gs[0]] * n, es
    raise KeyError(label)


def test_multiexp_golden_restatement_and_opcounts():
    g = load_golden("multiexp.json")
    sg, ss = bytes.fromhex(g["seed_points"]), bytes.fromhex(g["seed_scalars"])
    for case in g["cases"]:
        if case["n"] > 1024:
            continue  # n=4096 is checked against the C oracle (test_oracle_c.py)
        gs, es = multiexp_case_inputs(case, sg, ss)
        grp = R.EC()
        got = R.Pippenger(grp).multiexp(gs, es)
        assert got == P(case["result"]), case["label"]
        # same s/t/b schedule => the same number of group operations as the reference
        assert grp.ops == case["ops"], (case["label"], case["n"])
        if case["n"] <= 129:
            assert R.multiexp_naive(gs, es) == got


def test_multiexp_errors():
    with pytest.raises(Exception, match="Different number of group elements and exponents"):
        R.PipSECP256k1.multiexp([G], [1, 2])
    assert R.PipSECP256k1.multiexp([], []) == INF


def test_modp_group_golden():
    g = load_golden("modp_group.json")
    for c in g["cases"]:
        p, n = c["p"], c["n"]
        pip = R.Pippenger(R.MultIntModP(p, p - 1))
        gs = [R.CountedModP(2 + 3 * i, p) for i in range(n)]
        es = [(12345 * (i + 1) ** 3) % (p - 1) for i in range(n)]
        R.CountedModP.reset()
        r = pip.multiexp(gs, es)
        assert r.x == c["result"] and R.CountedModP.num_of_mult == c["num_of_mult"]
        want = 1
        for gi, e in zip(gs, es):
            want = want * pow(gi.x, e, p) % p
        assert r.x == want


def check_proof2(p2, want):
    assert hx(p2.a.x) == want["a"] and hx(p2.b.x) == want["b"]
    assert [hx(x.x) for x in p2.xs] == want["xs"]
    assert p2.Ls == [P(v) for v in want["Ls"]] and p2.Rs == [P(v) for v in want["Rs"]]
    assert p2.transcript.decode() == want["transcript"]
    assert p2.start_transcript == want["start_transcript"]


@pytest.mark.parametrize("k", range(9))
def test_ipa_golden(k):
    c = load_golden("ipa.json")["cases"][k]
    n = c["n"]
    s = [bytes.fromhex(x) for x in c["seeds"]]
    g, h = gens(n, s[0]), gens(n, s[1])
    u = R.elliptic_hash(s[2])
    a, b = scal(n, s[3]), scal(n, s[4])
    ip = R.inner_product(a, b)
    assert hx(ip.x) == c["c"]
    mexp = R.PipSECP256k1.multiexp if n <= 64 else R.multiexp_naive
    Pt = R.vector_commitment(g, h, a, b, mexp) + ip * u
    assert Pt == P(c["P"])
    p2 = R.ipa2_prove(g, h, u, Pt, a, b, multiexp=mexp)
    check_proof2(p2, c["proof2"])
    assert [hx(x.x) for x in R.get_ss(p2.xs, n)] == c["ss"]
    assert R.ipa2_verify(g, h, u, Pt, p2, mexp) is True
    P1 = P(c["P1"])
    p1 = R.ipa1_prove(g, h, u, P1, ip, a, b, seed=s[5], multiexp=mexp)
    assert p1.u_new == P(c["proof1"]["u_new"]) and p1.P_new == P(c["proof1"]["P_new"])
    assert p1.transcript.decode() == c["proof1"]["transcript"]
    check_proof2(p1.proof2, c["proof1"]["proof2"])
    assert R.ipa1_verify(g, h, u, P1, ip, p1, mexp) is True
    if n == 16:  # cheating: wrong P / wrong c  (src/tests/test_innerprod.py:33-98,138-224)
        with pytest.raises(Exception, match="Proof invalid"):
            R.ipa2_verify(g, h, u, 2 * Pt, p2, mexp)
        with pytest.raises(Exception, match="Proof invalid"):
            R.ipa1_verify(g, h, u, P1, ip + R.Zq(1, Q), p1, mexp)


def check_range_proof(pr, want):
    assert hx(pr.taux.x) == want["taux"] and hx(pr.mu.x) == want["mu"] and hx(pr.t_hat.x) == want["t_hat"]
    for name in ("T1", "T2", "A", "S"):
        assert getattr(pr, name) == P(want[name]), name
    assert pr.transcript.decode() == want["transcript"]
    ip = pr.innerProof
    assert ip.u_new == P(want["inner"]["u_new"]) and ip.P_new == P(want["inner"]["P_new"])
    assert ip.transcript.decode() == want["inner"]["transcript"]
    check_proof2(ip.proof2, want["inner"]["proof2"])


def range_inputs(c, m):
    s = [bytes.fromhex(x) for x in c["seeds"]]
    n = c["n"]
    gs, hs = gens(n * m, s[0]), gens(n * m, s[1])
    g, h, u = (R.elliptic_hash(s[j]) for j in (2, 3, 4))
    return s, n, gs, hs, g, h, u


@pytest.mark.parametrize("k", range(7))
def test_rangeproof_golden(k):
    c = load_golden("rangeproofs.json")["single"][k]
    s, n, gs, hs, g, h, u = range_inputs(c, 1)
    v = R.Zq(int(c["v"], 16), Q)
    gamma = R.mod_hash(s[5], Q)
    V = R.commitment(g, h, v, gamma)
    assert V == P(c["V"])
    mexp = R.multiexp_naive
    pr = R.range_prove(v, n, g, h, gs, hs, gamma, u, seed=s[6], multiexp=mexp)
    check_range_proof(pr, c["proof"])
    assert R.range_verify(V, g, h, gs, hs, u, pr, mexp) is True
    if "cheat_wrong_commitment" in c:
        assert c["cheat_wrong_commitment"] == "Proof invalid" == c["cheat_out_of_range"]
        with pytest.raises(Exception, match="Proof invalid"):
            R.range_verify(R.commitment(g, h, v + 1, gamma), g, h, gs, hs, u, pr, mexp)
        bad = R.Zq(2**n + 5, Q)
        prb = R.range_prove(bad, n, g, h, gs, hs, gamma, u, seed=s[6], multiexp=mexp)
        with pytest.raises(Exception, match="Proof invalid"):
            R.range_verify(R.commitment(g, h, bad, gamma), g, h, gs, hs, u, prb, mexp)


@pytest.mark.parametrize("k", range(3))
def test_aggregated_rangeproof_golden(k):
    from oracle import cbind
    c = load_golden("rangeproofs.json")["aggregated"][k]
    m = c["m"]
    s, n, gs, hs, g, h, u = range_inputs(c, m)
    vs = [R.Zq(int(v, 16), Q) for v in c["vs"]]
    gammas = [R.mod_hash(str(j).encode() + s[5], Q) for j in range(m)]
    Vs = [R.commitment(g, h, vs[j], gammas[j]) for j in range(m)]
    assert Vs == [P(v) for v in c["Vs"]]
    mexp = cbind.msm  # C oracle keeps the 512-wide case fast; pinned in test_oracle_c.py
    pr = R.aggreg_range_prove(vs, n, g, h, gs, hs, gammas, u, seed=s[6], multiexp=mexp)
    check_range_proof(pr, c["proof"])
    assert R.aggreg_range_verify(Vs, g, h, gs, hs, u, pr, mexp) is True
    if "cheat_wrong_commitment" in c:
        bad = list(Vs)
        bad[1] = R.commitment(g, h, vs[1] + 1, gammas[1])
        with pytest.raises(Exception, match="Proof invalid"):
            R.aggreg_range_verify(bad, g, h, gs, hs, u, pr, mexp)


@pytest.mark.parametrize("k", range(3))
def test_aggregated_rangeproof_golden_with_bulk_ec(k):
    """The `ecops` variant of the restatement (element-wise EC expressions evaluated in bulk by oracle/c; used by the
    configuration-size GPU tests) gives the reference's bytes as well."""
    from oracle import cbind
    c = load_golden("rangeproofs.json")["aggregated"][k]
    m = c["m"]
    s, n, gs, hs, g, h, u = range_inputs(c, m)
    vs = [R.Zq(int(v, 16), Q) for v in c["vs"]]
    gammas = [R.mod_hash(str(j).encode() + s[5], Q) for j in range(m)]
    pr = R.aggreg_range_prove(vs, n, g, h, gs, hs, gammas, u, seed=s[6], multiexp=cbind.msm, ecops=cbind.BulkEC(2))
    check_range_proof(pr, c["proof"])
