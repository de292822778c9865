"""GPU parity of the inner-product argument (FastNIProver2 / NIProver / Verifier1 /
Verifier2 of the product package) vs the reference goldens and the oracle; mirrors
/root/reference/src/tests/test_innerprod.py (sizes 2^0..2^8 and every cheating case)."""
import random

import pytest

from conftest import load_golden
from helpers import P, Q, gens, hx, scal
from oracle import bp_ref as R
from oracle import cbind

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


def build_case(gp, c):
    from bulletproofs_amd.ec import secp256k1
    n = c["n"]
    s = [bytes.fromhex(x) for x in c["seeds"]]
    og, oh = gens(n, s[0]), gens(n, s[1])
    ou = R.elliptic_hash(s[2])
    oa, ob = scal(n, s[3]), scal(n, s[4])
    return dict(n=n, s=s, g=gp.to_gpu_list(og), h=gp.to_gpu_list(oh), u=gp.to_gpu(ou),
                a=[gp.gsc(v) for v in oa], b=[gp.gsc(v) for v in ob], curve=secp256k1)


def check_proof2(gp, p2, want):
    assert hx(p2.a.x) == want["a"] and hx(p2.b.x) == want["b"]
    assert [hx(x.x) for x in p2.xs] == want["xs"]
    assert all(gp.same_point(a, P(b)) for a, b in zip(p2.Ls, want["Ls"])) and len(p2.Ls) == len(want["Ls"])
    assert all(gp.same_point(a, P(b)) for a, b in zip(p2.Rs, want["Rs"])) and len(p2.Rs) == len(want["Rs"])
    assert p2.transcript.decode() == want["transcript"]
    assert p2.start_transcript == want["start_transcript"]


@pytest.mark.parametrize("k", range(9))
def test_ipa_reference_goldens(gp, k):
    from bulletproofs_amd.innerproduct import NIProver, FastNIProver2, Verifier1, Verifier2
    from bulletproofs_amd.utils import vector_commitment, inner_product
    c = load_golden("ipa.json")["cases"][k]
    d = build_case(gp, c)
    ip = inner_product(d["a"], d["b"])
    assert hx(ip.x) == c["c"]
    Pt = vector_commitment(d["g"], d["h"], d["a"], d["b"]) + ip * d["u"]
    assert gp.same_point(Pt, P(c["P"]))
    p2 = FastNIProver2(d["g"], d["h"], d["u"], Pt, d["a"], d["b"], d["curve"]).prove()
    check_proof2(gp, p2, c["proof2"])
    v2 = Verifier2(d["g"], d["h"], d["u"], Pt, p2)
    assert [hx(x.x) for x in v2.get_ss(p2.xs)] == c["ss"]
    assert v2.verify() is True
    P1 = gp.to_gpu(P(c["P1"]))
    p1 = NIProver(d["g"], d["h"], d["u"], P1, ip, d["a"], d["b"], d["curve"], d["s"][5]).prove()
    assert gp.same_point(p1.u_new, P(c["proof1"]["u_new"])) and gp.same_point(p1.P_new, P(c["proof1"]["P_new"]))
    assert p1.transcript.decode() == c["proof1"]["transcript"]
    check_proof2(gp, p1.proof2, c["proof1"]["proof2"])
    assert Verifier1(d["g"], d["h"], d["u"], P1, ip, p1).verify() is True


def test_ipa_cheating_cases(gp):
    """src/tests/test_innerprod.py:33-98, 138-268."""
    from bulletproofs_amd.innerproduct import NIProver, FastNIProver2, Verifier1, Verifier2
    from bulletproofs_amd.utils import vector_commitment, inner_product, ModP
    c = load_golden("ipa.json")["cases"][4]          # n = 16
    d = build_case(gp, c)
    g, h, u, a, b, curve = d["g"], d["h"], d["u"], d["a"], d["b"], d["curve"]
    ip = inner_product(a, b)
    Pt = vector_commitment(g, h, a, b) + ip * u
    p2 = FastNIProver2(g, h, u, Pt, a, b, curve).prove()
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier2(g, h, u, 2 * Pt, p2).verify()
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier2(g, h, 2 * u, Pt, p2).verify()
    a_bad = list(a)
    a_bad[3] = a_bad[3] * 2
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier2(g, h, u, Pt, FastNIProver2(g, h, u, Pt, a_bad, b, curve).prove()).verify()
    b_bad = list(b)
    b_bad[5] = b_bad[5] * 2
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier2(g, h, u, Pt, FastNIProver2(g, h, u, Pt, a, b_bad, curve).prove()).verify()
    P1 = vector_commitment(g, h, a, b)
    p1 = NIProver(g, h, u, P1, ip, a, b, curve, d["s"][5]).prove()
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier1(g, h, u, P1, ip + ModP(1, Q), p1).verify()
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier1(g, h, u, 2 * P1, ip, p1).verify()
    # corrupted outer transcript / swapped inner transcript
    good = p1.transcript
    p1.transcript = good.replace(good.split(b"&")[1], b"1234", 1)
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier1(g, h, u, P1, ip, p1).verify()
    p1.transcript = good
    other = NIProver(g, h, u, P1, ip, a, b, curve, b"another seed").prove()
    p1.proof2.transcript = other.proof2.transcript
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier1(g, h, u, P1, ip, p1).verify()


@pytest.mark.parametrize("n,big_m,small_m", [(2, 0, 0), (64, 0, 0), (1024, 0, 0), (8192, 0, 0), (4096, 256, 0), (1 << 14, 1 << 10, 0), (1 << 12, 1 << 12, 0),
                                             (64, 0, 4), (1024, 0, 64), (1024, 0, 512), (8192, 1024, 32), (1 << 15, 0, 0), (1 << 14, 0, 1)])
def test_ipa_rounds_vs_oracle(gp, n, big_m, small_m):
    """Every round's L, R and the folded vectors against the C oracle, through the raw
    C-ABI state object (bpmi_ipa_*), with arbitrary challenges.  `big_m` lowers the base
    length from which generators are folded 16-way at once, so the deferred-MSM path, the
    materialisation kernel and the rounds after it are all exercised at test sizes; `small_m` is the
    logical length at which smaller bases are folded through per-term products (round 4: 0 = the
    default 4096 -- reached by the 2^15 case, K = 8, on the GLV product kernel --, 1 = never, else the
    length: K = 16, 2 and 16 behind a ladder fold), after which L and R run as ONE small-MSM launch."""
    eng = gp.engine()
    eng.set_option("ipa_big_m", big_m)
    eng.set_option("ipa_small_m", small_m)
    pts, _ = gp.rand_points(2 * n + 1, 40 + n)
    g, h, u = pts[:n], pts[n:2 * n], pts[2 * n]
    rnd = random.Random(n)
    a = [rnd.randrange(Q) for _ in range(n)]
    b = [rnd.randrange(Q) for _ in range(n)]
    st = eng.ipa_create(cbind.pack_points(g), cbind.pack_points(h), cbind.pack_scalars(a), cbind.pack_scalars(b),
                        n, cbind.pack_points([u]))
    while len(st) > 1:
        half = len(st) // 2
        L, Rr = st.round_LR()
        cl = cbind.sc_dot(a[:half], b[half:])
        cr = cbind.sc_dot(a[half:], b[:half])
        assert L == cbind.pack_points([cbind.msm(g[half:] + h[:half] + [u], a[:half] + b[half:] + [cl])])
        assert Rr == cbind.pack_points([cbind.msm(g[:half] + h[half:] + [u], a[half:] + b[:half] + [cr])])
        x = rnd.randrange(1, Q)
        xi = pow(x, -1, Q)
        st.fold(x, xi)
        g = cbind.ec_lincomb2_batch(g[:half], g[half:], xi, x)
        h = cbind.ec_lincomb2_batch(h[:half], h[half:], x, xi)
        a = cbind.sc_fold(a[:half], a[half:], x, xi)
        b = cbind.sc_fold(b[:half], b[half:], xi, x)
        if half <= 8:      # bpmi_ipa_export: the folded generators the reference holds at :84
            assert st.export() == (cbind.pack_points(g), cbind.pack_points(h), cbind.pack_scalars(a), cbind.pack_scalars(b))
    assert st.finish() == (a[0], b[0])
    st.close()
    eng.set_option("ipa_big_m", 0)
    eng.set_option("ipa_small_m", 0)


@pytest.mark.parametrize("n,big_m", [(512, 256), (4096, 256), (1 << 13, 1 << 13)])
def test_sixteen_way_generator_fold_three_ladders_one_result(gp, n, big_m):
    """The 16-way fold of the generators behind four deferred rounds, three ways (ctx option fold_wnaf): 2 = width-4 NAF of the
    coefficients' GLV halves over the tables of odd multiples and their beta-x twins (k_ec_multifold_w4g, the default), 1 = width-4
    NAF of the whole coefficients (k_ec_multifold_w4), 0 = the plain NAF ladder (k_ec_multifold).  Same L, R in every round, same
    exported generators, and test_ipa_rounds_vs_oracle pins the default against the oracle."""
    eng = gp.engine()
    eng.set_option("ipa_big_m", big_m)
    eng.set_option("ipa_small_m", 1)
    pts, _ = gp.rand_points(2 * n + 1, 77 + n)
    # identities among the generators: a table entry (0, 0) and its beta-x twin must both read as the identity
    pts[3] = cbind.INF
    pts[n + 5] = cbind.INF
    rnd = random.Random(5 * n)
    a = [rnd.randrange(Q) for _ in range(n)]
    b = [rnd.randrange(Q) for _ in range(n)]
    xs = [rnd.randrange(1, Q) for _ in range(n.bit_length() - 1)]
    seen = []
    for mode in (2, 1, 0):
        eng.set_option("fold_wnaf", mode)
        st = eng.ipa_create(cbind.pack_points(pts[:n]), cbind.pack_points(pts[n:2 * n]), cbind.pack_scalars(a), cbind.pack_scalars(b),
                            n, cbind.pack_points([pts[2 * n]]))
        trace = []
        for x in xs:
            trace.append(st.round_LR())
            st.fold(x, pow(x, -1, Q))
            if len(st) == 8:
                trace.append(st.export())
        trace.append(st.finish())
        st.close()
        seen.append(trace)
    eng.set_option("fold_wnaf", 2)
    eng.set_option("ipa_big_m", 0)
    eng.set_option("ipa_small_m", 0)
    assert seen[0] == seen[1] == seen[2]


@pytest.mark.parametrize("n,big_m", [(4096, 256), (1 << 13, 1 << 13)])
def test_fixed_generators_keep_the_fold_tables_between_proofs(gp, n, big_m):
    """Option ipa_fixed_generators (round 5): the tables of odd multiples the 16-way fold builds from the generators are kept between
    proofs that name the SAME device arrays.  Three proofs over the same generators (different a, b, challenges) give what they give
    with the option off; other arrays, another length, a host-pointer state or switching the option off rebuild the tables."""
    eng = gp.engine()
    pts, _ = gp.rand_points(2 * n + 1, 91 + n)
    g, h, u = cbind.pack_points(pts[:n]), cbind.pack_points(pts[n:2 * n]), cbind.pack_points([pts[2 * n]])
    d_g, d_h, d_h2 = eng.upload(g), eng.upload(h), eng.upload(g)
    rnd = random.Random(n)

    def run(dg, dh, nn, seed, host=False):
        r = random.Random(seed)
        a = cbind.pack_scalars([r.randrange(Q) for _ in range(nn)])
        b = cbind.pack_scalars([r.randrange(Q) for _ in range(nn)])
        if host:
            st = eng.ipa_create(g[: 64 * nn], h[: 64 * nn], a, b, nn, u)
        else:
            d_a, d_b = eng.upload(a), eng.upload(b)
            st = eng.ipa_create_dev(dg, dh, d_a, d_b, nn, u)
        trace = []
        for _ in range(nn.bit_length() - 1):
            trace.append(st.round_LR())
            x = r.randrange(1, Q)
            st.fold(x, pow(x, -1, Q))
        trace.append(st.finish())
        st.close()
        if not host:
            d_a.free(); d_b.free()
        return trace

    try:
        eng.set_option("ipa_big_m", big_m)
        eng.set_option("ipa_small_m", 1)
        want = [run(d_g, d_h, n, s) for s in (1, 2, 3)]
        want_swapped = run(d_g, d_h2, n, 4)
        want_half = run(d_g, d_h, n // 2, 5)
        eng.set_option("ipa_fixed_generators", 1)
        assert [run(d_g, d_h, n, s) for s in (1, 2, 3)] == want               # built once, used three times
        assert run(d_g, d_h2, n, 4) == want_swapped                           # another array: rebuilt
        assert run(d_g, d_h, n, 1) == want[0]                                 # ... and back
        assert run(d_g, d_h, n // 2, 5) == want_half                          # another length
        assert run(None, None, n, 2, host=True) == want[1]                    # host pointers: never kept
        assert run(d_g, d_h, n, 3) == want[2]
        eng.set_option("ipa_fixed_generators", 0)
        assert run(d_g, d_h, n, 2) == want[1]
    finally:
        eng.set_option("ipa_fixed_generators", 0)
        eng.set_option("ipa_big_m", 0)
        eng.set_option("ipa_small_m", 0)
        for d in (d_g, d_h, d_h2):
            d.free()


def test_fixed_generators_second_fold_never_records_tables(gp):
    """with n >= 16 big_m a proof reaches a SECOND 16-way fold, over the already folded, challenge-dependent bases.  Its tables
    must not be recorded under the caller's arrays with length n / 16 -- a later proof over the first n / 16 generators of the same arrays
    (a prefix is a natural call) would skip k_ec_odd_multiples and fold with stale tables.  n = 16 384, big_m = 64: folds at 16 384 and at
    1 024; then a 1 024-element proof on the same pointers, against the option switched off."""
    eng = gp.engine()
    n = 1 << 14
    pts, _ = gp.rand_points(2 * n + 1, 777)
    g, h, u = cbind.pack_points(pts[:n]), cbind.pack_points(pts[n:2 * n]), cbind.pack_points([pts[2 * n]])
    d_g, d_h = eng.upload(g), eng.upload(h)

    def run(nn, seed):
        r = random.Random(seed)
        d_a = eng.upload(cbind.pack_scalars([r.randrange(Q) for _ in range(nn)]))
        d_b = eng.upload(cbind.pack_scalars([r.randrange(Q) for _ in range(nn)]))
        st = eng.ipa_create_dev(d_g, d_h, d_a, d_b, nn, u)
        trace = []
        for _ in range(nn.bit_length() - 1):
            trace.append(st.round_LR())
            x = r.randrange(1, Q)
            st.fold(x, pow(x, -1, Q))
        trace.append(st.finish())
        st.close()
        d_a.free(); d_b.free()
        return trace

    try:
        eng.set_option("ipa_big_m", 64)
        eng.set_option("ipa_small_m", 1)
        want_big, want_small = run(n, 1), run(n // 16, 2)
        eng.set_option("ipa_fixed_generators", 1)
        assert run(n, 1) == want_big                       # two folds; only the first may record its tables
        assert run(n // 16, 2) == want_small               # the prefix: its first fold has length n / 16 and must build its own tables
        assert run(n // 16, 2) == want_small               # ... which ARE kept now
        assert run(n, 1) == want_big
    finally:
        eng.set_option("ipa_fixed_generators", 0)
        eng.set_option("ipa_big_m", 0)
        eng.set_option("ipa_small_m", 0)
        d_g.free(); d_h.free()


@pytest.mark.parametrize("n,small_m", [(1024, 64), (1 << 14, 1024), (1 << 16, 0)])
def test_product_fold_shared_scalars_equals_per_lane_products(gp, n, small_m):
    """The two forms of the product fold -- shared GLV halves in non-adjacent form with two terms per thread (k_ec_fold_glv, the
    default when no per-generator scale rides along) and per-lane products (bpmi_ec_mul_batch + k_ec_sum_strided) -- must leave
    the same generators: identical L, R in every round after the fold, identical final scalars, for K = 16 (twice) and at the
    default length."""
    eng = gp.engine()
    pts, _ = gp.rand_points(2 * n + 1 if n <= (1 << 14) else 4097, 333)
    if n > (1 << 14):                       # many points from few: products of the pool on the GPU
        ks = b"".join(random.Random(5).randrange(1, Q).to_bytes(32, "little") for _ in range(2 * n + 1))
        raw = eng.ec_mul_batch_bytes(b"".join(cbind.pack_points([pts[i % 4097]]) for i in range(2 * n + 1)), ks, 2 * n + 1)
        gb, hb, ub = raw[:64 * n], raw[64 * n: 128 * n], raw[128 * n:]
    else:
        gb, hb, ub = cbind.pack_points(pts[:n]), cbind.pack_points(pts[n:2 * n]), cbind.pack_points([pts[2 * n]])
    rnd = random.Random(n)
    ab = cbind.pack_scalars([rnd.randrange(Q) for _ in range(n)]), cbind.pack_scalars([rnd.randrange(Q) for _ in range(n)])
    xs = [rnd.randrange(1, Q) for _ in range(n.bit_length())]
    runs = []
    try:
        eng.set_option("ipa_small_m", small_m)
        for shared in (1, 0):
            eng.set_option("fold_shared", shared)
            st = eng.ipa_create(gb, hb, ab[0], ab[1], n, ub)
            trace = []
            for x in xs:
                if len(st) <= 1:
                    break
                trace.append(st.round_LR())
                st.fold(x, pow(x, -1, Q))
            trace.append(st.finish())
            st.close()
            runs.append(trace)
    finally:
        eng.set_option("ipa_small_m", 0)
        eng.set_option("fold_shared", 1)
    assert runs[0] == runs[1]


@pytest.mark.parametrize("n,scaled", [(4096, False), (512, False), (2048, True), (2, False)])
def test_one_launch_step_for_short_vectors_equals_the_four_kernels(gp, n, scaled):
    """Option ipa_small_step (off by default: measured slower): fold of a and b, the coefficient tables and the NEXT round's c_L,
    c_R and expanded scalars in one launch must give the rounds the four separate kernels give, with and without a per-generator
    scale, and an export in between must not disturb it."""
    eng = gp.engine()
    pts, _ = gp.rand_points(2 * n + 1, 444)
    rnd = random.Random(n + 9)
    gb, hb, ub = cbind.pack_points(pts[:n]), cbind.pack_points(pts[n:2 * n]), cbind.pack_points([pts[2 * n]])
    ab = cbind.pack_scalars([rnd.randrange(Q) for _ in range(n)]), cbind.pack_scalars([rnd.randrange(Q) for _ in range(n)])
    hsc = cbind.pack_scalars([rnd.randrange(1, Q) for _ in range(n)]) if scaled else None
    xs = [rnd.randrange(1, Q) for _ in range(n.bit_length())]
    runs = []
    try:
        for step in (1, 0):
            eng.set_option("ipa_small_step", step)
            st = eng.ipa_create(gb, hb, ab[0], ab[1], n, ub, hsc)
            trace = []
            for x in xs:
                if len(st) <= 1:
                    break
                trace.append(st.round_LR())
                st.fold(x, pow(x, -1, Q))
                if len(st) in (8, 2):
                    trace.append(st.export())
            trace.append(st.finish())
            st.close()
            runs.append(trace)
    finally:
        eng.set_option("ipa_small_step", 0)
    assert runs[0] == runs[1]


def test_ipa_full_size_round_trip(gp):
    """Size-independent property at a size the oracle cannot reach in seconds: a proof
    produced by FastNIProver2 at n = 2^16 (deferred folding, 32 MSMs) must be accepted by
    Verifier2 (one MSM of 2n+1 points against one of 2 log n), and rejected after any
    single-field mutation."""
    import copy
    from bulletproofs_amd.ec import Point, secp256k1, unpack_points
    from bulletproofs_amd.innerproduct import FastNIProver2, Verifier2
    from bulletproofs_amd.utils import ModP, inner_product, vector_commitment
    eng = gp.engine()
    n = 1 << 16
    rnd = random.Random(2016)
    G64 = secp256k1.G.to_le64()
    ks = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(2 * n + 1))
    pts = unpack_points(eng.ec_mul_batch_bytes(G64 * (2 * n + 1), ks, 2 * n + 1), 2 * n + 1)
    g, h, u = pts[:n], pts[n:2 * n], pts[2 * n]
    a = [ModP(rnd.randrange(Q), Q) for _ in range(n)]
    b = [ModP(rnd.randrange(Q), Q) for _ in range(n)]
    P = vector_commitment(g, h, a, b) + inner_product(a, b) * u
    proof = FastNIProver2(g, h, u, P, a, b, secp256k1).prove()
    assert len(proof.Ls) == 16 and len(proof.xs) == 16
    assert Verifier2(g, h, u, P, proof).verify() is True
    for field in ("a", "b"):
        bad = copy.copy(proof)
        setattr(bad, field, getattr(bad, field) + ModP(1, Q))
        with pytest.raises(Exception, match="Proof invalid"):
            Verifier2(g, h, u, P, bad).verify()
    with pytest.raises(Exception, match="Proof invalid"):
        Verifier2(g, h, u, P + u, proof).verify()


@pytest.mark.parametrize("n,big_m,small_m", [(1, 0, 0), (2, 0, 0), (256, 0, 0), (1024, 256, 0), (2048, 2048, 0), (256, 0, 16), (1 << 14, 0, 0)])
def test_ipa_scaled_generators(gp, n, big_m, small_m):
    """bpmi_ipa_create_scaled: the argument over c_i * h[i] must equal the plain argument over
    the materialised points, in both regimes (factors in the MSM scalars / one batched
    multiplication for bases that get folded), and through the product fold, which multiplies the
    factors into the folded generators (small_m = 16; and the default at the 2^14 generators of an
    aggregated 128 x 64-bit range proof: K = 4 at length 4096)."""
    eng = gp.engine()
    eng.set_option("ipa_big_m", big_m)
    eng.set_option("ipa_small_m", small_m)
    pts, _ = gp.rand_points(2 * n + 1, 70 + n)
    g, h, u = pts[:n], pts[n:2 * n], pts[2 * n]
    rnd = random.Random(n + 1)
    a = [rnd.randrange(Q) for _ in range(n)]
    b = [rnd.randrange(Q) for _ in range(n)]
    scale = [rnd.randrange(1, Q) for _ in range(n)]
    scale[0] = 1
    hs = cbind.ec_mul_batch(h, scale)
    st = eng.ipa_create(cbind.pack_points(g), cbind.pack_points(h), cbind.pack_scalars(a), cbind.pack_scalars(b), n,
                        cbind.pack_points([u]), cbind.pack_scalars(scale))
    ref = eng.ipa_create(cbind.pack_points(g), cbind.pack_points(hs), cbind.pack_scalars(a), cbind.pack_scalars(b), n,
                         cbind.pack_points([u]))
    while len(st) > 1:
        assert st.round_LR() == ref.round_LR()
        x = rnd.randrange(1, Q)
        xi = pow(x, -1, Q)
        st.fold(x, xi)
        ref.fold(x, xi)
        if len(st) <= 4:
            assert st.export() == ref.export()
    if n == 1:
        assert st.export() == ref.export()
    assert st.finish() == ref.finish()
    st.close()
    ref.close()
    eng.set_option("ipa_big_m", 0)
    eng.set_option("ipa_small_m", 0)


def test_ipa_export_too_long_is_a_state_error(gp):
    eng = gp.engine()
    n = 256
    pts, _ = gp.rand_points(2 * n + 1, 5)
    st = eng.ipa_create(cbind.pack_points(pts[:n]), cbind.pack_points(pts[n:2 * n]), cbind.pack_scalars([1] * n),
                        cbind.pack_scalars([2] * n), n, cbind.pack_points([pts[2 * n]]))
    assert len(st.export()[0]) == 64 * n          # nothing folded yet: plain copies
    st.round_LR()
    st.fold(3, pow(3, -1, Q))
    with pytest.raises(RuntimeError, match="export"):
        st.export()                               # 128 deferred generators: refused, not computed
    st.close()


def test_two_rank_sharded_ipa_prover_on_one_gpu():
    """ShardedFastNIProver2: two processes (gloo rendezvous, both on cuda:0, each with its own
    bpmi ctx) prove over cyclic shards; the Proof2 equals the oracle's / the one-GPU prover's."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", BPMI_DIST_MODE="ipa_gpu")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(repo, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DIST_IPA_OK world=2" in r.stdout
