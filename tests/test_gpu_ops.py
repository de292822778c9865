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
