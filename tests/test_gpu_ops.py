"""GPU parity of the batched point / scalar entry points of the C-ABI vs the oracle."""
import random

import pytest

from helpers import Q
from oracle import cbind
from oracle.ec import INF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


def test_ec_mul_batch(gp):
    n = 300
    pts, _ = gp.rand_points(n, 3)
    rnd = random.Random(4)
    es = [rnd.randrange(Q) for _ in range(n)]
    es[:6] = [0, 1, 2, Q - 1, Q - 2, (Q - 1) // 2]
    pts[7] = INF
    got = gp.engine().ec_mul_batch_bytes(cbind.pack_points(pts), cbind.pack_scalars(es), n)
    assert got == cbind.pack_points(cbind.ec_mul_batch(pts, es))


def test_ec_mul_batch_glv_fixed_window_path(gp):
    """n >= 32 768 takes k_ec_odd_multiples + k_ec_mul_batch_glv (GLV halves, 43 signed odd three-bit digits each, affine
    3P / 5P / 7P): against the C oracle's ladder on scalars that make a half-scalar 0, 1, even, odd, negative or maximal
    (0, 1, 2, q - 1, lambda, lambda^2, +-the lattice constants, single bits, 2^b - 1), unreduced ones, identity points, and
    a size that is not a multiple of anything (slices of 196 608 points; 16 points per table thread)."""
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    a1, mb1 = 0x3086D221A7D46BCDE86C90E49284EB15, 0xE4437ED6010E88286F547FA90ABFE4C3
    edge = [0, 1, 2, 3, Q - 1, Q - 2, Q // 2, Q // 2 + 1, lam, Q - lam, lam * lam % Q, (lam + 1) % Q, (lam - 1) % Q, 2 * lam % Q, a1, mb1, Q - a1,
            Q - mb1, Q, Q + 1, (1 << 256) - 1, 1 << 255]
    edge += [1 << b for b in range(256)] + [(1 << b) - 1 for b in range(2, 257)]
    edge += [(t * a1 + d) % Q for t in (1, 7, 1 << 64, (1 << 127) + 3) for d in (-1, 0, 1)]
    n = 32768 + 1111
    pts, _ = gp.rand_points(n, 41)
    rnd = random.Random(42)
    es = [edge[i] if i < len(edge) else rnd.randrange(Q) for i in range(n)]
    for i in (5, 700, n - 1):
        pts[i] = INF
    raw = b"".join(e.to_bytes(32, "little") for e in es)
    eng = gp.engine()
    want = cbind.pack_points(cbind.ec_mul_batch(pts, es))
    got = eng.ec_mul_batch_bytes(cbind.pack_points(pts), raw, n)
    assert got == want
    eng.set_option("mul_batch_glv", 0)                   # the bit-serial ladder on the same inputs
    try:
        assert eng.ec_mul_batch_bytes(cbind.pack_points(pts), raw, n) == want
    finally:
        eng.set_option("mul_batch_glv", 1)


@pytest.mark.parametrize("n", [3 * 65536, 3 * 65536 + 1, 400000])
def test_ec_mul_batch_slices(gp, n):
    """More than one slice: k_i * G for all i equals the C oracle's values at sampled positions (every slice seam) and the
    sum of all products equals (sum k_i) * G (an MSM-free check of every element)."""
    import hashlib
    eng = gp.engine()
    ks = b"".join(hashlib.sha256(b"mulb%d" % i).digest() for i in range(n))
    G64 = cbind.pack_points([gp.G])
    out = eng.ec_mul_batch_bytes(G64 * n, ks, n)
    idx = sorted(set([0, 1, 15, 16, 17, 196607, 196608, 196609, n - 1] + [random.Random(n).randrange(n) for _ in range(40)]))
    idx = [i for i in idx if i < n]
    want = cbind.ec_mul_batch([gp.G] * len(idx), [int.from_bytes(ks[32 * i: 32 * i + 32], "little") for i in idx])
    assert [out[64 * i: 64 * i + 64] for i in idx] == [cbind.pack_points([w]) for w in want]
    total = sum(int.from_bytes(ks[32 * i: 32 * i + 32], "little") for i in range(n)) % Q
    assert eng.ec_sum_bytes(out, n) == cbind.pack_points(cbind.ec_mul_batch([gp.G], [total]))


def test_ec_lincomb2_batch(gp):
    n = 200
    pts, _ = gp.rand_points(2 * n, 5)
    p1, p2 = pts[:n], pts[n:]
    p2[0] = p1[0]            # equal points
    p2[1] = -p1[1]           # opposite points
    p1[2] = INF
    p2[3] = INF
    rnd = random.Random(6)
    for k1, k2 in ((rnd.randrange(Q), rnd.randrange(Q)), (5, 5), (1, Q - 1), (0, 7), (3, 0)):
        got = gp.engine().ec_lincomb2_batch_bytes(cbind.pack_points(p1), cbind.pack_points(p2),
                                                  cbind.pack_scalars([k1]), cbind.pack_scalars([k2]), n)
        assert got == cbind.pack_points(cbind.ec_lincomb2_batch(p1, p2, k1, k2)), (k1, k2)


def test_ec_sum_and_point_operators(gp):
    from bulletproofs_amd.ec import Point, secp256k1
    pts, _ = gp.rand_points(700, 8)
    for n in (1, 2, 3, 255, 256, 257, 700):
        want = INF
        for p in pts[:n]:
            want = want + p
        assert gp.engine().ec_sum_bytes(cbind.pack_points(pts[:n]), n) == cbind.pack_points([want])
    a, b = gp.to_gpu(pts[0]), gp.to_gpu(pts[1])
    assert gp.same_point(a + b, pts[0] + pts[1])
    assert gp.same_point(a + a, 2 * pts[0])
    assert (a + (-a)) == Point.IDENTITY_ELEMENT
    assert gp.same_point(a + Point.IDENTITY_ELEMENT, pts[0])
    assert gp.same_point(12345 * a, 12345 * pts[0])
    assert gp.same_point(a * (Q + 3), 3 * pts[0])
    assert (0 * a) == Point.IDENTITY_ELEMENT and (Q * a) == Point.IDENTITY_ELEMENT
    G = secp256k1.G
    assert (2 * G).x == 0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5


def test_scalar_bulk_ops(gp):
    rnd = random.Random(9)
    for n in (1, 2, 255, 256, 257, 5000, 300000):
        a = [rnd.randrange(Q) for _ in range(n)]
        b = [rnd.randrange(Q) for _ in range(n)]
        a[0], b[0] = Q - 1, Q - 1
        ab, bb = cbind.pack_scalars(a), cbind.pack_scalars(b)
        got = int.from_bytes(gp.engine().sc_dot_bytes(ab, bb, n), "little")
        assert got == sum(x * y for x, y in zip(a, b)) % Q
        if n <= 5000:
            x = rnd.randrange(1, Q)
            xi = pow(x, -1, Q)
            out = gp.engine().sc_fold_bytes(ab, bb, cbind.pack_scalars([x]), cbind.pack_scalars([xi]), n)
            assert out == cbind.pack_scalars([(x * u + xi * v) % Q for u, v in zip(a, b)])


def test_inner_product_and_commitments(gp):
    from bulletproofs_amd.utils import inner_product, commitment, vector_commitment, ModP
    from oracle import bp_ref as R
    rnd = random.Random(10)
    a = [ModP(rnd.randrange(Q), Q) for _ in range(50)]
    b = [ModP(rnd.randrange(Q), Q) for _ in range(50)]
    assert inner_product(a, b).x == sum(x.x * y.x for x, y in zip(a, b)) % Q
    pts, _ = gp.rand_points(8, 12)
    g, h = gp.to_gpu(pts[0]), gp.to_gpu(pts[1])
    assert gp.same_point(commitment(g, h, a[0], b[0]), a[0].x * pts[0] + b[0].x * pts[1])
    vc = vector_commitment(gp.to_gpu_list(pts[:4]), gp.to_gpu_list(pts[4:]), a[:4], b[:4])
    assert gp.same_point(vc, R.multiexp_naive(pts, [v.x for v in a[:4] + b[:4]]))


def test_ec_sum_dev_matches_host_pointer_version(gp):
    eng = gp.engine()
    pts, _ = gp.rand_points(37, 4)
    buf = cbind.pack_points(pts + [INF, pts[3]])
    d = eng.upload(buf)
    try:
        assert eng.ec_sum_dev(d, 39) == eng.ec_sum_bytes(buf, 39)
        assert eng.ec_sum_dev(d, 0) == bytes(64)
    finally:
        d.free()


def test_unreduced_scalars_are_reduced_on_load(gp):
    """A C caller may hand bpmi_msm / bpmi_ec_mul_batch any 256-bit scalar: s = q, q + 5, 2^256 - 1 behave as
    s mod q (the reference's `e % order`, src/pippenger/pippenger.py:26), in every MSM path and in the ladder."""
    eng = gp.engine()
    pts, _ = gp.rand_points(5000, 21)
    rnd = random.Random(22)
    raw = [Q, Q + 5, 2 ** 256 - 1, Q - 1, 0, 2 ** 255, 2 ** 256 - 2] + [rnd.randrange(2 ** 256) for _ in range(5000 - 7)]
    rawb = b"".join(v.to_bytes(32, "little") for v in raw)
    red = [v % Q for v in raw]
    for n in (1, 3, 7, 300, 5000):                     # small-MSM kernel, and the bucket pipeline above 4096
        want = cbind.msm_bytes(cbind.pack_points(pts[:n]), cbind.pack_scalars(red[:n]), n)
        assert eng.msm_bytes(cbind.pack_points(pts[:n]), rawb[: 32 * n], n) == want, n
    eng.set_option("small_n", -1)                       # force the bucket pipeline at small n too
    try:
        assert eng.msm_bytes(cbind.pack_points(pts[:300]), rawb[: 32 * 300], 300) == \
            cbind.msm_bytes(cbind.pack_points(pts[:300]), cbind.pack_scalars(red[:300]), 300)
    finally:
        eng.set_option("small_n", 0)
    got = eng.ec_mul_batch_bytes(cbind.pack_points(pts[:64]), rawb[: 32 * 64], 64)
    assert got == cbind.pack_points(cbind.ec_mul_batch(pts[:64], red[:64]))


@pytest.mark.parametrize("k", [0, 1, 2, 3, 5, 8, 11, 14])
def test_sc_svector_vs_reference_get_ss(gp, k):
    """bpmi_sc_svector against the oracle's restatement of Verifier2.get_ss
    (/root/reference/src/innerproduct/inner_product_verifier.py:91-102), with the final a, b and an optional
    per-generator scale folded in."""
    from oracle import bp_ref as R
    eng = gp.engine()
    rnd = random.Random(100 + k)
    n = 1 << k
    xs = [rnd.randrange(1, Q) for _ in range(k)]
    xi = [pow(x, -1, Q) for x in xs]
    a, b = rnd.randrange(Q), rnd.randrange(Q)
    ss = [int(s.x) for s in R.get_ss([R.Zq(x, Q) for x in xs], n)]
    assert len(ss) == n
    sa, sb = eng.sc_svector_bytes(cbind.pack_scalars(xs), cbind.pack_scalars(xi), k, a, b)
    assert sa == cbind.pack_scalars([a * s for s in ss])
    assert sb == cbind.pack_scalars([b * pow(s, -1, Q) for s in ss])
    scale = [rnd.randrange(Q) for _ in range(n)]
    sa2, sb2 = eng.sc_svector_bytes(cbind.pack_scalars(xs), cbind.pack_scalars(xi), k, a, b, cbind.pack_scalars(scale))
    assert sa2 == sa and sb2 == cbind.pack_scalars([b * pow(s, -1, Q) * c for s, c in zip(ss, scale)])


def test_verifier2_device_path_equals_host_path(gp):
    """Verifier2.verify switches to bpmi_ipa_verify_dev from n = 1024: same verdicts as the host s-vector path
    (forced by raising the threshold) on a valid proof, with h_scale, and on mutations."""
    import copy
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.innerproduct import FastNIProver2, Verifier2
    from bulletproofs_amd.utils import ModP, inner_product, vector_commitment
    n = 2048
    pts, _ = gp.rand_points(2 * n + 1, 77)
    g, h, u = gp.to_gpu_list(pts[:n]), gp.to_gpu_list(pts[n:2 * n]), gp.to_gpu(pts[2 * n])
    rnd = random.Random(78)
    a = [ModP(rnd.randrange(Q), Q) for _ in range(n)]
    b = [ModP(rnd.randrange(Q), Q) for _ in range(n)]
    scale = [rnd.randrange(1, Q) for _ in range(n)]
    for hs in (None, scale):
        hh = h if hs is None else gp.to_gpu_list(cbind.ec_mul_batch(pts[n:2 * n], scale))
        Pt = vector_commitment(g, hh, a, b) + inner_product(a, b) * u
        proof = FastNIProver2(g, h, u, Pt, a, b, secp256k1, h_scale=hs).prove()
        bad = copy.copy(proof)
        bad.b = proof.b + ModP(1, Q)
        for threshold in (1024, 1 << 30):
            old = Verifier2.DEVICE_SVECTOR_MIN_N
            Verifier2.DEVICE_SVECTOR_MIN_N = threshold
            try:
                assert Verifier2(g, h, u, Pt, proof, h_scale=hs).verify() is True
                with pytest.raises(Exception, match="Proof invalid"):
                    Verifier2(g, h, u, Pt, bad, h_scale=hs).verify()
                with pytest.raises(Exception, match="Proof invalid"):
                    Verifier2(g, h, u, Pt + u, proof, h_scale=hs).verify()
            finally:
                Verifier2.DEVICE_SVECTOR_MIN_N = old
