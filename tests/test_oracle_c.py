"""Pins the plain-C oracle (oracle/c/bp_oracle.c) to oracle/ec.py and to the
reference-generated multiexp goldens."""
import random

from conftest import load_golden
from helpers import P, Q, gens, scal, seed
from oracle import bp_ref as R
from oracle import cbind
from oracle.ec import INF, secp256k1
from test_oracle_golden import multiexp_case_inputs

G = secp256k1.G


def test_c_msm_matches_reference_goldens():
    g = load_golden("multiexp.json")
    sg, ss = bytes.fromhex(g["seed_points"]), bytes.fromhex(g["seed_scalars"])
    pts_cache = {}
    for case in g["cases"]:
        gs, es = multiexp_case_inputs(case, sg, ss)
        assert cbind.msm(gs, es) == P(case["result"]), (case["label"], case["n"])
        assert cbind.msm(gs, es, threads=1) == P(case["result"])


def test_c_msm_edge_inputs():
    pts = gens(8, seed(9))
    assert cbind.msm([], []) == INF
    assert cbind.msm([INF, pts[0]], [5, 0]) == INF
    assert cbind.msm([INF, pts[0]], [5, 1]) == pts[0]
    assert cbind.msm([pts[0]] * 3, [Q - 1, 1, 0]) == INF
    assert cbind.msm([pts[0], -pts[0]], [7, 7]) == INF
    assert cbind.msm([pts[0], pts[0]], [7, 7]) == 14 * pts[0]
    # structured known answer: P_i = (k0 + i*d) G  =>  MSM = (sum e_i k_i) G
    n, k0, d = 300, 0xDEADBEEF, 0x1234567
    step, cur, ps, ks = d * G, k0 * G, [], []
    for i in range(n):
        ps.append(cur)
        ks.append(k0 + i * d)
        cur = cur + step
    rnd = random.Random(5)
    es = [rnd.randrange(Q) for _ in range(n)]
    assert cbind.msm(ps, es) == (sum(e * k for e, k in zip(es, ks)) % Q) * G


def test_c_point_and_scalar_ops():
    rnd = random.Random(7)
    pts = gens(12, seed(8))
    es = [rnd.randrange(Q) for _ in range(12)] + [0, 1, Q - 1]
    pts2 = pts + [pts[0], INF, pts[1]]
    assert cbind.ec_mul_batch(pts2, es) == [e * p for e, p in zip(es, pts2)]
    k1, k2 = rnd.randrange(Q), rnd.randrange(Q)
    lo, hi = pts[:6] + [pts[0], pts[1]], pts[6:] + [pts[0], -pts[1]]
    assert cbind.ec_lincomb2_batch(lo, hi, k1, k2) == [k1 * a + k2 * b for a, b in zip(lo, hi)]
    assert cbind.ec_lincomb2_batch(lo, hi, 5, 5) == [5 * a + 5 * b for a, b in zip(lo, hi)]
    for a, b in ((pts[0], pts[1]), (pts[0], pts[0]), (pts[0], -pts[0]), (INF, pts[0]), (pts[0], INF), (INF, INF)):
        assert cbind.ec_add(a, b) == a + b
    a = [rnd.randrange(Q) for _ in range(50)]
    b = [rnd.randrange(Q) for _ in range(50)]
    assert cbind.sc_dot(a, b) == sum(x * y for x, y in zip(a, b)) % Q
    x = rnd.randrange(1, Q)
    xi = pow(x, -1, Q)
    assert cbind.sc_fold(a, b, x, xi) == [(x * u + xi * v) % Q for u, v in zip(a, b)]
    assert cbind.sc_dot([Q - 1] * 9, [Q - 1] * 9) == 9 % Q


def test_c_msm_matches_the_reference_run_at_config_c2_size():
    """ONE run of the reference's own multiexp at n = 2^16 (BASELINE config C2; 23.7 M group operations, 740 s and ~11 GB of subset
    tables here: tests/golden/make_golden.py multiexp_big) pins the plain-C oracle at that size; the GPU suite compares the engine
    with the same vector (tests/test_gpu_configs.py::test_c2_msm_2e16_equals_the_reference_run)."""
    g = load_golden("multiexp_big.json")
    n = g["n"]
    gs, es = gens(n, bytes.fromhex(g["seed_points"])), scal(n, bytes.fromhex(g["seed_scalars"]))
    assert cbind.msm(gs, es) == P(g["result"])
