"""The reference's own test strategy (random seeds, prove -> verify ==
True, every cheating case raises "Proof invalid"), run against THIS package through nothing
but the reference-shaped call surface -- the drop-in check.  Sizes follow
/root/reference/src/tests/: IPA N = 2^0..2^8, range proofs n = 2..128, aggregated m = 2..32."""
import os
from random import randint

import pytest

pytestmark = pytest.mark.gpu

import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.ec import secp256k1 as CURVE  # noqa: E402
from bulletproofs_amd.innerproduct import FastNIProver2, NIProver, Verifier1, Verifier2  # noqa: E402
from bulletproofs_amd.rangeproofs import (AggregNIRangeProver, AggregRangeVerifier, NIRangeProver,  # noqa: E402
                                          RangeVerifier)
from bulletproofs_amd.utils import (ModP, b64_to_point, bytes_to_point, commitment, elliptic_hash,  # noqa: E402
                                    inner_product, mod_hash, point_to_b64, point_to_bytes, vector_commitment)

q = CURVE.q


def generators(n, seed):
    return [elliptic_hash(str(i).encode() + seed, CURVE) for i in range(n)]


def scalars(n, seed):
    return [mod_hash(str(i).encode() + seed, q) for i in range(n)]


def test_hash_and_codecs():
    x = mod_hash(b"test", 1009)
    assert x.x < 1009 and x == mod_hash(b"test", 1009)
    assert all(mod_hash(os.urandom(10), 17).x != 0 for _ in range(50))
    for _ in range(10):
        P = elliptic_hash(os.urandom(10), CURVE)
        assert CURVE.is_point_on_curve((P.x, P.y))
        assert bytes_to_point(point_to_bytes(P)) == P and b64_to_point(point_to_b64(P)) == P
    e = randint(0, q)
    assert bytes_to_point(point_to_bytes(e * CURVE.G)) == e * CURVE.G


@pytest.mark.parametrize("log_n", range(9))
def test_inner_product_argument_sizes(log_n):
    N = 2 ** log_n
    s = [os.urandom(10) for _ in range(6)]
    g, h, u = generators(N, s[0]), generators(N, s[1]), elliptic_hash(s[2], CURVE)
    a, b = scalars(N, s[3]), scalars(N, s[4])
    c = inner_product(a, b)
    P2 = vector_commitment(g, h, a, b) + c * u
    assert Verifier2(g, h, u, P2, FastNIProver2(g, h, u, P2, a, b, CURVE).prove()).verify()
    P1 = vector_commitment(g, h, a, b)
    assert Verifier1(g, h, u, P1, c, NIProver(g, h, u, P1, c, a, b, CURVE, s[5]).prove()).verify()


def test_verifier_prints_ok_like_the_reference_when_asked(capsys):
    """/root/reference/src/innerproduct/inner_product_verifier.py:146 prints "OK" from a successful Verifier2.verify(); here that is
    opt-in (Verifier2.PRINT_OK), silent by default, and never printed for a proof that fails."""
    N = 8
    s = [os.urandom(10) for _ in range(5)]
    g, h, u = generators(N, s[0]), generators(N, s[1]), elliptic_hash(s[2], CURVE)
    a, b = scalars(N, s[3]), scalars(N, s[4])
    P2 = vector_commitment(g, h, a, b) + inner_product(a, b) * u
    proof = FastNIProver2(g, h, u, P2, a, b, CURVE).prove()
    assert Verifier2(g, h, u, P2, proof).verify()
    assert capsys.readouterr().out == ""
    try:
        Verifier2.PRINT_OK = True
        assert Verifier2(g, h, u, P2, proof).verify()
        assert capsys.readouterr().out == "OK\n"
        with pytest.raises(Exception, match="Proof invalid"):
            Verifier2(g, h, u, 2 * P2, proof).verify()
        assert capsys.readouterr().out == ""
    finally:
        Verifier2.PRINT_OK = False


def test_inner_product_cheating():
    N = 16
    s = [os.urandom(10) for _ in range(6)]
    g, h, u = generators(N, s[0]), generators(N, s[1]), elliptic_hash(s[2], CURVE)
    a, b = scalars(N, s[3]), scalars(N, s[4])
    c = inner_product(a, b)
    P = vector_commitment(g, h, a, b) + c * u
    proof = FastNIProver2(g, h, u, P, a, b, CURVE).prove()
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier2(g, h, u, 2 * P, proof).verify()
    a2 = list(a)
    a2[randint(0, N - 1)] *= 2
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier2(g, h, u, P, FastNIProver2(g, h, u, P, a2, b, CURVE).prove()).verify()
    P1 = vector_commitment(g, h, a, b)
    p1 = NIProver(g, h, u, P1, c, a, b, CURVE, s[5]).prove()
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier1(g, h, u, P1, c + ModP(1, q), p1).verify()


@pytest.mark.parametrize("log_n", range(1, 8))
def test_range_proofs(log_n):
    n = 2 ** log_n
    s = [os.urandom(10) for _ in range(7)]
    v = ModP(randint(0, 2 ** n - 1), q)
    gs, hs = generators(n, s[0]), generators(n, s[1])
    g, h, u = (elliptic_hash(s[j], CURVE) for j in (2, 3, 4))
    gamma = mod_hash(s[5], q)
    V = commitment(g, h, v, gamma)
    proof = NIRangeProver(v, n, g, h, gs, hs, gamma, u, CURVE, s[6]).prove()
    assert RangeVerifier(V, g, h, gs, hs, u, proof).verify()
    if n == 16:
        bad = ModP(randint(2 ** 16, 2 ** 17), q)
        with pytest.raises(Exception, match="Proof invalid"):
            RangeVerifier(commitment(g, h, bad, gamma), g, h, gs, hs, u,
                          NIRangeProver(bad, n, g, h, gs, hs, gamma, u, CURVE, s[6]).prove()).verify()
        with pytest.raises(Exception, match="Proof invalid"):
            RangeVerifier(commitment(g, h, v + 1, gamma), g, h, gs, hs, u, proof).verify()


@pytest.mark.parametrize("m,n", [(2, 16), (4, 16), (8, 16), (32, 16), (4, 64)])
def test_aggregated_range_proofs(m, n):
    s = [os.urandom(10) for _ in range(7)]
    vs = [ModP(randint(0, 2 ** n - 1), q) for _ in range(m)]
    gs, hs = generators(n * m, s[0]), generators(n * m, s[1])
    g, h, u = (elliptic_hash(s[j], CURVE) for j in (2, 3, 4))
    gammas = [mod_hash(s[5] + bytes([j]), q) for j in range(m)]
    Vs = [commitment(g, h, vs[j], gammas[j]) for j in range(m)]
    proof = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, CURVE, s[6]).prove()
    assert AggregRangeVerifier(Vs, g, h, gs, hs, u, proof).verify()
    if m == 4 and n == 16:
        Vs[randint(0, m - 1)] = commitment(g, h, ModP(5, q), gammas[0])
        with pytest.raises(Exception, match="Proof invalid"):
            AggregRangeVerifier(Vs, g, h, gs, hs, u, proof).verify()
