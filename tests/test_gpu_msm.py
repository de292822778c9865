"""GPU parity: Pippenger.multiexp on EC(secp256k1) through the C-ABI vs the reference
goldens, the C oracle on seeded inputs, and size-independent properties at the
BASELINE.json sizes (n = 2^16, 2^20)."""
import ctypes
import random

import pytest

from conftest import load_golden
from helpers import P, Q, seed
from oracle import cbind
from oracle.ec import INF
from test_oracle_golden import multiexp_case_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


def test_multiexp_reference_goldens(gp):
    from bulletproofs_amd.pippenger import PipSECP256k1
    g = load_golden("multiexp.json")
    sg, ss = bytes.fromhex(g["seed_points"]), bytes.fromhex(g["seed_scalars"])
    for case in g["cases"]:
        gs, es = multiexp_case_inputs(case, sg, ss)
        es = [gp.gsc(e) if hasattr(e, "x") else e for e in es]
        got = PipSECP256k1.multiexp(gp.to_gpu_list(gs), es)
        assert gp.same_point(got, P(case["result"])), (case["label"], case["n"])


def test_multiexp_call_surface(gp):
    from bulletproofs_amd.pippenger import PipSECP256k1, Pippenger, EC
    from bulletproofs_amd.ec import Point, secp256k1
    assert PipSECP256k1.multiexp([], []) == Point.IDENTITY_ELEMENT
    with pytest.raises(Exception, match="Different number of group elements and exponents"):
        PipSECP256k1.multiexp([secp256k1.G], [1, 2])
    Gp = secp256k1.G
    assert PipSECP256k1.multiexp([Gp], [0]) == Point.IDENTITY_ELEMENT
    assert PipSECP256k1.multiexp([Gp, Gp], [Q - 1, 1]) == Point.IDENTITY_ELEMENT
    assert PipSECP256k1.multiexp([Gp], [-1]) == -Gp
    assert PipSECP256k1.multiexp([Gp], [Q + 2]) == Gp + Gp
    assert PipSECP256k1.multiexp([Point.IDENTITY_ELEMENT, Gp], [5, 3]) == 3 * Gp
    assert Pippenger(EC(secp256k1)).multiexp([Gp, 2 * Gp], [3, 4]) == 11 * Gp
    inputs = [Gp, 2 * Gp]
    before = [(p.x, p.y) for p in inputs]
    PipSECP256k1.multiexp(inputs, [7, 9])
    assert [(p.x, p.y) for p in inputs] == before           # inputs are never mutated


@pytest.mark.parametrize("n", [1, 2, 7, 63, 64, 65, 255, 1000, 4097, 20000, 65536])
def test_msm_vs_oracle_random(gp, n):
    pts, _ = gp.rand_points(n, 100 + n)
    rnd = random.Random(n)
    es = [rnd.randrange(Q) for _ in range(n)]
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    assert gp.engine().msm_bytes(pb, sb, n) == cbind.msm_bytes(pb, sb, n)


@pytest.mark.parametrize("n", [4096, 4097, 4608, 4609, 5631, 5632, 8448, 8449, 10239, 10240, 15359, 15360, 18999, 19000, 32767, 32768])
def test_msm_at_the_window_table_boundaries(gp, n):
    """Either side of every switch of the geometry, this round's and the earlier ones': one-launch kernel / one-block-per-window kernel
    (4 608), that kernel / the bucket pipeline with 12-bit mixed-width windows (8 448; 5 632 with one block per window), 12 / 13 bits
    (19 000); 13 / 16 at 185 000 is
    in test_gpu_msm_midsize.py.  Scalars with the edge values mixed in (0, 1, q - 1, >= q, 2^255, the half-order boundary, and the
    largest digits of a 15-bit window)."""
    pts, _ = gp.rand_points(n, 900 + n)
    rnd = random.Random(n)
    edge = [0, 1, Q - 1, Q, Q + 1, (1 << 256) - 1, 1 << 255, (Q - 1) // 2, (Q + 1) // 2, (1 << 13) - 1, 1 << 12, (1 << 16) - 1, 1 << 15,
            (1 << 15) - 1, 1 << 14, (0x7FFF << 240) % Q, ((1 << 255) - 1) % Q]
    es = [edge[rnd.randrange(len(edge))] if i % 5 == 0 else rnd.randrange(Q) for i in range(n)]
    pb, sb = cbind.pack_points(pts), b"".join(e.to_bytes(32, "little") for e in es)       # unreduced: both sides reduce mod q
    assert gp.engine().msm_bytes(pb, sb, n) == cbind.msm_bytes(pb, sb, n)


@pytest.mark.parametrize("shape", ["acommit", "all_same", "same_point", "zeros", "cancel", "small", "with_inf", "top"])
def test_msm_degenerate_shapes(gp, shape):
    n = 5000
    pts, _ = gp.rand_points(n, 7)
    rnd = random.Random(11)
    es = [rnd.randrange(Q) for _ in range(n)]
    if shape == "acommit":            # rangeproof_prover.py:42-47: {0,1} then {0,q-1}
        es = [rnd.randrange(2) for _ in range(n // 2)] + [(rnd.randrange(2) - 1) % Q for _ in range(n - n // 2)]
    elif shape == "all_same":
        es = [es[0]] * n
    elif shape == "same_point":
        pts = [pts[0]] * n
    elif shape == "zeros":
        es = [0] * n
    elif shape == "cancel":
        pts = pts[: n // 2] + [-p for p in pts[: n // 2]]
        es = es[: n // 2] * 2
    elif shape == "small":
        es = [rnd.randrange(1 << 20) for _ in range(n)]
    elif shape == "with_inf":
        pts = [INF if i % 3 == 0 else p for i, p in enumerate(pts)]
    elif shape == "top":              # scalars around q/2 and q-1: sign recoding edges
        half = (Q - 1) // 2
        es = [[half, half + 1, half - 1, Q - 1, Q - 2, 1, 2][i % 7] for i in range(n)]
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    assert gp.engine().msm_bytes(pb, sb, n) == cbind.msm_bytes(pb, sb, n)


@pytest.mark.parametrize("n,chunk,c", [(9, 1, 4), (20, 1, 4), (64, 1, 0), (300, 2, 0), (200, 1, 8), (1000, 0, 4),
                                       (1000, 8, 6), (3000, 0, 16), (3000, 3, 9), (1 << 12, 0, 2)])
def test_msm_multiblock_segscan(gp, n, chunk, c):
    """Forces the multi-level partial-record path (many 256-record blocks) and every
    window size: regression for the segscan store epilogue."""
    eng = gp.engine()
    pts, _ = gp.rand_points(n, 5)
    rnd = random.Random(n * 31 + chunk)
    es = [rnd.randrange(Q) for _ in range(n)]
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    try:
        eng.set_option("chunk", chunk)
        eng.set_option("window_bits", c)
        for fused in (1, 0):                 # the wave-level scan inside k_accum_l0 (two records per wave) / two records per thread
            eng.set_option("fused_scan", fused)
            eng.set_option("split", 0)
            for tail in (1, 2):
                eng.set_option("tail", tail)
                assert eng.msm_bytes(pb, sb, n) == cbind.msm_bytes(pb, sb, n)
            eng.set_option("split", 1)           # two window groups on two lanes (carry crosses the groups)
            assert eng.msm_bytes(pb, sb, n) == cbind.msm_bytes(pb, sb, n)
    finally:
        eng.set_option("chunk", 0)
        eng.set_option("window_bits", 0)
        eng.set_option("tail", 0)
        eng.set_option("split", 0)
        eng.set_option("fused_scan", 1)


@pytest.mark.parametrize("shape", ["one_scalar", "two_scalars", "runs_of_chunk_multiples", "runs_of_odd_lengths", "sparse_then_dense", "uniform"])
@pytest.mark.parametrize("n,chunk,c", [(2500, 1, 4), (2500, 2, 8), (6000, 4, 10), (6000, 7, 10), (20000, 3, 8), (70, 1, 4), (130, 2, 4), (40000, 0, 0)])
def test_msm_fused_wave_scan_shapes(gp, shape, n, chunk, c):
    """The segmented scan k_accum_l0 runs over the 64 chunks of a wave (round 4): chunks that are one single run chained through
    many lanes and through whole waves, runs that end exactly at chunk and at wave boundaries, a last wave with idle lanes, every
    bucket written exactly once -- against the C oracle, and identical to the unfused path."""
    eng = gp.engine()
    pts, _ = gp.rand_points(n, 11)
    rnd = random.Random(n * 7 + chunk * 13 + c)
    L = chunk if chunk else 8
    if shape == "one_scalar":
        es = [rnd.randrange(Q)] * n
    elif shape == "two_scalars":
        a, b = rnd.randrange(Q), rnd.randrange(1 << 40)
        es = [a if rnd.random() < 0.7 else b for _ in range(n)]
    elif shape == "runs_of_chunk_multiples":       # every distinct scalar exactly 64 L k times: runs end at chunk AND wave boundaries
        es = []
        while len(es) < n:
            es += [rnd.randrange(Q)] * (64 * L * rnd.randrange(1, 3))
        es = es[:n]
    elif shape == "runs_of_odd_lengths":
        es = []
        while len(es) < n:
            es += [rnd.randrange(Q)] * rnd.choice((1, 2, 3, L, L + 1, 2 * L - 1, 5 * L, 63 * L, 64 * L + 1, 200))
        es = es[:n]
    elif shape == "sparse_then_dense":             # a handful of full-size scalars, then small ones: empty-bucket runs and heavy buckets in one MSM
        es = [rnd.randrange(Q) if i % 97 == 0 else rnd.randrange(4) for i in range(n)]
    else:
        es = [rnd.randrange(Q) for _ in range(n)]
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    want = cbind.msm_bytes(pb, sb, n, 4)
    try:
        eng.set_option("chunk", chunk)
        eng.set_option("window_bits", c)
        eng.set_option("fused_scan", 1)
        assert eng.msm_bytes(pb, sb, n) == want
        eng.set_option("fused_scan", 0)
        assert eng.msm_bytes(pb, sb, n) == want
    finally:
        eng.set_option("chunk", 0)
        eng.set_option("window_bits", 0)
        eng.set_option("fused_scan", 1)


@pytest.mark.parametrize("logn", [16, 20])
def test_msm_full_size_properties(gp, logn):
    """At BASELINE.json's sizes the oracle is too slow to be the only check, so use
    properties that do not depend on n:
      known answer  P_i = k_i*G  =>  MSM(P, e) = (sum e_i k_i mod q) * G
      linearity     MSM(P, e) + MSM(P, f) = MSM(P, e + f)
      determinism   same input twice -> identical bytes
    and, at 2^16, the C oracle on all host cores."""
    eng = gp.engine()
    n = 1 << logn
    rnd = random.Random(logn)
    ks = [rnd.randrange(1, Q) for _ in range(n)]
    Gb = cbind.pack_points([gp.G])
    d_G = eng.upload(Gb * n)
    d_k = eng.upload(cbind.pack_scalars(ks))
    d_pts = eng.alloc(64 * n)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_pts.ptr))
    d_G.free()
    es = [rnd.randrange(Q) for _ in range(n)]
    fs = [rnd.randrange(Q) for _ in range(n)]
    d_e = eng.upload(cbind.pack_scalars(es))
    d_f = eng.upload(cbind.pack_scalars(fs))
    d_ef = eng.upload(cbind.pack_scalars([(a + b) % Q for a, b in zip(es, fs)]))
    r_e = eng.msm_dev(d_pts, d_e, n)
    r_f = eng.msm_dev(d_pts, d_f, n)
    r_ef = eng.msm_dev(d_pts, d_ef, n)
    assert r_e == eng.msm_dev(d_pts, d_e, n)
    eng.set_option("split", 1)
    try:
        assert r_e == eng.msm_dev(d_pts, d_e, n)
    finally:
        eng.set_option("split", 0)
    want = (sum(e * k for e, k in zip(es, ks)) % Q) * gp.G
    assert r_e == cbind.pack_points([want])
    assert eng.ec_sum_bytes(r_e + r_f, 2) == r_ef
    if logn == 16:
        pts_host = d_pts.download()
        assert r_e == cbind.msm_bytes(pts_host, cbind.pack_scalars(es), n)
        # spot-check the generated points themselves
        assert pts_host[:64] == cbind.pack_points([ks[0] * gp.G])
    for d in (d_k, d_pts, d_e, d_f, d_ef):
        d.free()


def test_msm_2e24_periodic_inputs(gp):
    """n = 2^24 (the global-atomic sort path used above 2^23, 2^28 sorted entries, 32-bit
    position arithmetic near its range) with inputs of period 2^16: MSM(P x 256, e x 256)
    must equal 256 * MSM(P, e).  Every bucket receives 256 copies of the same point, so the
    P + P branch of the complete addition formula runs millions of times."""
    eng = gp.engine()
    m, reps = 1 << 16, 256
    rnd = random.Random(24)
    G64 = cbind.pack_points([gp.G])
    ks = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(m))
    pts = eng.ec_mul_batch_bytes(G64 * m, ks, m)
    es = b"".join(rnd.randrange(Q).to_bytes(32, "little") for _ in range(m))
    small = eng.msm_bytes(pts, es, m)
    d_p, d_e = eng.upload(pts * reps), eng.upload(es * reps)
    try:
        big = eng.msm_dev(d_p, d_e, m * reps)            # round 6: sixteen slices of 2^20 pairs, two in flight
        eng.set_option("window_bits", 16)                # forced window bits: ONE MSM, on the global-atomic sort beyond 2^23 pairs
        one = eng.msm_dev(d_p, d_e, m * reps)
    finally:
        eng.set_option("window_bits", 0)
        d_p.free()
        d_e.free()
    want = eng.ec_mul_batch_bytes(small, (reps).to_bytes(32, "little"), 1)
    assert big == want
    assert one == want


def test_msm_sliced_above_2e23(gp):
    """n > 2^23 runs as slices on alternating lanes whose results are added (bpmi_msm_dev; round 6: nine slices of ~2^20 pairs,
    and with slice_n = -1 the round-1 geometry: slices of 2^23).  Size-independent check: with the points tiled from D distinct ones,
    MSM(tiled, e) == MSM(distinct, column sums of e mod q); ragged last slice."""
    import numpy as np
    eng = gp.engine()
    D = 1 << 12
    n = (1 << 23) + (1 << 16) + 5 * D                 # 2 slices; a multiple of D
    pts, _ = gp.rand_points(D, 99)
    small = cbind.pack_points(pts)
    rng = np.random.default_rng(3)
    reps = n // D
    e = rng.integers(0, 1 << 32, size=(reps, D, 8), dtype=np.uint64).astype(np.uint32)
    e[:, :, 7] &= 0x7FFFFFFF
    col = e.astype(np.uint64).sum(axis=0)
    folded = []
    for j in range(D):
        v = 0
        for k in range(7, -1, -1):
            v = (v << 32) + int(col[j, k])
        folded.append(v % Q)
    d_pts = eng.alloc(64 * n)
    tile = small * 64
    for r in range(0, reps, 64):
        d_pts.upload(tile[: 64 * D * min(64, reps - r)], 64 * D * r)
    d_e = eng.upload(e.tobytes())
    try:
        got = eng.msm_dev(d_pts, d_e, n)
        assert got == cbind.msm_bytes(small, cbind.pack_scalars(folded), D)
        assert got == eng.msm_dev(d_pts, d_e, n)
        eng.set_option("slice_n", -1)
        assert got == eng.msm_dev(d_pts, d_e, n)
    finally:
        eng.set_option("slice_n", 0)
        d_pts.free()
        d_e.free()


@pytest.mark.parametrize("c", [10, 12, 13, 15, 16])            # (15: the last window is unsigned, 2B buckets -- digits B and 2B - 1 land there too)
def test_msm_digit_of_magnitude_B_both_signs(gp, c):
    """Signed recoding edge: a digit of magnitude exactly B = 2^(c-1) is always a positive digit,
    but the scalar may have been replaced by q - s, so the ENTRY's sign can be either; the sort's
    16-bit digit code keeps the two signs apart (digit sign in the code, negation flag per scalar)."""
    eng = gp.engine()
    n = 3000
    B = 1 << (c - 1)
    pts, _ = gp.rand_points(n, 17)
    rnd = random.Random(c)
    es = []
    for i in range(n):
        r = 0
        for w in range(rnd.randrange(1, 6)):
            r |= (B if rnd.random() < 0.7 else rnd.randrange(1 << c)) << (c * rnd.randrange(0, 200 // c))
        if i % 3 == 0:
            r |= rnd.choice((B, 2 * B - 1, B - 1, B + 1)) << (c * (254 // c))          # the last window's own edge digits
        r %= Q // 2
        es.append(r if i % 2 else Q - r)              # odd i: digits of r; even i: negated scalar, same digits
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    try:
        eng.set_option("window_bits", c)
        assert eng.msm_bytes(pb, sb, n) == cbind.msm_bytes(pb, sb, n)
    finally:
        eng.set_option("window_bits", 0)


def test_msm2_pairs(gp):
    """bpmi_msm2 / Pippenger.multiexp2: two independent MSMs on the two lanes == each one alone,
    for small / large / empty / unequal sizes."""
    from bulletproofs_amd.pippenger import PipSECP256k1
    eng = gp.engine()
    rnd = random.Random(8)
    pts, _ = gp.rand_points(6000, 21)
    for n0, n1 in ((1, 1), (2, 130), (129, 12), (5000, 3), (6000, 6000), (70, 0)):
        a = [rnd.randrange(Q) for _ in range(n0)]
        b = [rnd.randrange(Q) for _ in range(n1)]
        p0, p1 = pts[:n0], pts[6000 - n1:] if n1 else []
        if n1:
            o0, o1 = eng.msm2_bytes(cbind.pack_points(p0), cbind.pack_scalars(a), n0, cbind.pack_points(p1), cbind.pack_scalars(b), n1)
            assert o0 == cbind.msm_bytes(cbind.pack_points(p0), cbind.pack_scalars(a), n0)
            assert o1 == cbind.msm_bytes(cbind.pack_points(p1), cbind.pack_scalars(b), n1)
        g0, g1 = PipSECP256k1.multiexp2(gp.to_gpu_list(p0), a, gp.to_gpu_list(p1), b)
        assert gp.same_point(g0, cbind.msm(p0, a)) and gp.same_point(g1, cbind.msm(p1, b) if n1 else INF)
    with pytest.raises(Exception, match="Different number"):
        PipSECP256k1.multiexp2(gp.to_gpu_list(pts[:2]), [1], [], [])


def test_msm2_both_sorts_before_either_accumulation(gp):
    """Option pair_phases (an experiment that came out neutral, off by default): a synchronous pair on the bucket pipeline queues both
    MSMs' sorts first and the rest of each afterwards (msm_enqueue phase 1 / phase 2).  Same results as one MSM after the other, for
    unequal sizes, for a pair whose second MSM is on a one-launch kernel (no separate sort: phase 2 does everything), and twice in a
    row (the lanes' workspaces are reused)."""
    eng = gp.engine()
    rnd = random.Random(88)
    n = 40000
    pts, _ = gp.rand_points(n, 23)
    pb = cbind.pack_points(pts)
    a = cbind.pack_scalars([rnd.randrange(Q) for _ in range(n)])
    b = cbind.pack_scalars([rnd.randrange(Q) for _ in range(n)])
    want = {}
    try:
        for mode in (0, 1, 1):
            eng.set_option("pair_phases", mode)
            for n0, n1 in ((n, n), (n, 33000), (36000, 5)):
                got = eng.msm2_bytes(pb, a, n0, pb, b, n1)
                if (n0, n1) not in want:
                    want[(n0, n1)] = (cbind.msm_bytes(pb, a, n0), cbind.msm_bytes(pb, b, n1))
                assert tuple(got) == want[(n0, n1)], (mode, n0, n1)
    finally:
        eng.set_option("pair_phases", 0)


# (40 000, 0): the default table's c = 15 with the unsigned last window; window bits 16, and 13 / 12 with their short top window, always heavy
@pytest.mark.parametrize("n,c", [(40000, 0), (40000, 16), (30000, 13), (14000, 0), (140000, 0)])
@pytest.mark.parametrize("shape", ["all_same", "two_values", "bits01", "bits_and_blinding", "small_range"])
def test_msm_heavy_partitions(gp, shape, n, c):
    """Skewed digit distributions on the LDS-sort path: partitions with more than 12 288 entries
    are counted and scattered by the tile kernels (k_fine_hist_heavy / k_fine_scatter_heavy) instead of
    one block's LDS; mixed with light partitions in the same MSM.  Buckets with thousands of entries also take the
    wave-cooperative chunk-key fill."""
    eng = gp.engine()
    pts, _ = gp.rand_points(n, 23)
    rnd = random.Random(len(shape))
    if shape == "all_same":
        es = [rnd.randrange(Q)] * n
    elif shape == "two_values":
        vals = [12345678901234567890123, Q - 5]
        es = [vals[rnd.randrange(2)] for _ in range(n)]
    elif shape == "bits01":                       # the aL / aR vectors of a range proof
        es = [rnd.randrange(2) for _ in range(n // 2)] + [(rnd.randrange(2) - 1) % Q for _ in range(n - n // 2)]
    elif shape == "bits_and_blinding":            # ... with a few full-size scalars: long runs of EMPTY buckets between entries (bisection in k_accum_l0)
        es = [rnd.randrange(2) for _ in range(n // 2)] + [(rnd.randrange(2) - 1) % Q for _ in range(n - n // 2)]
        for i in range(0, n, 4099):
            es[i] = rnd.randrange(Q)
    else:
        es = [rnd.randrange(1 << 20) if i % 3 else rnd.randrange(Q) for i in range(n)]
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    try:
        eng.set_option("window_bits", c)
        assert eng.msm_bytes(pb, sb, n) == cbind.msm_bytes(pb, sb, n)
    finally:
        eng.set_option("window_bits", 0)


@pytest.mark.parametrize("n", [300, 5000, 40000])
def test_msm_graph_replay_and_result_placement(gp, n):
    """Options "graphs" (the launch sequence of a repeated MSM replayed as a HIP graph: first call captures, later calls replay,
    another input misses the cache) and "direct_result" (the last kernel writes into the slot's page-locked buffer / into the
    workspace with a copy behind) must not change a byte; synchronous, paired and asynchronous entry points."""
    eng = gp.engine()
    pts, _ = gp.rand_points(n, 77)
    rnd = random.Random(n)
    sets = [[rnd.randrange(Q) for _ in range(n)] for _ in range(2)]
    pb = cbind.pack_points(pts)
    sbs = [cbind.pack_scalars(es) for es in sets]
    wants = [cbind.msm_bytes(pb, sb, n, 4) for sb in sbs]
    d_p = eng.upload(pb)
    d_s = [eng.upload(sb) for sb in sbs]
    try:
        for direct in (1, 0):
            for graphs in (1, 0):
                eng.set_option("direct_result", direct)
                eng.set_option("graphs", graphs)
                for rep in range(3):                      # capture, replay, replay
                    for k in (0, 1):
                        assert eng.msm_dev(d_p, d_s[k], n) == wants[k]
                o0, o1 = eng.msm2_bytes(pb, sbs[0], n, pb, sbs[1], n)
                assert (o0, o1) == (wants[0], wants[1])
                for rep in range(2):
                    eng.msm_dev_enqueue(0, d_p, d_s[0], n)
                    eng.msm_dev_enqueue(1, d_p, d_s[1], n)
                    assert eng.msm_finish(1) == wants[1] and eng.msm_finish(0) == wants[0]
                for tail in (1, 2):
                    eng.set_option("tail", tail)
                    eng.set_option("small_n", -1)
                    assert eng.msm_dev(d_p, d_s[0], n) == wants[0]
                    eng.set_option("small_n", 0)
                eng.set_option("tail", 0)
    finally:
        eng.set_option("graphs", 0)
        eng.set_option("direct_result", 1)
        eng.set_option("tail", 0)
        eng.set_option("small_n", 0)
        d_p.free()
        for d in d_s:
            d.free()


@pytest.mark.parametrize("shape", ["uniform", "one_scalar", "bits", "bits_and_blinding", "small_values", "identities_and_negatives", "near_q"])
@pytest.mark.parametrize("n", [1536, 2049, 4097, 8193, 8448])
def test_msm_mid_kernel_one_block_per_window(gp, shape, n):
    """k_msm_mid (round 4): a whole window's bucket method in one block -- digits, counting sort and lane shares in LDS, the
    segmented scan over the lanes, the weighted bucket sum on quads.  Every size class it takes, digit distributions that put
    everything in one bucket or leave almost all buckets empty, identity points, scalars around q / 2 and q; single MSMs and the
    pair in one launch; against the C oracle and against the pipeline (mid_min = -1) and the small kernel."""
    eng = gp.engine()
    pts, _ = gp.rand_points(n, 91)
    rnd = random.Random(n * 3 + len(shape))
    half = (Q - 1) // 2
    if shape == "uniform":
        es = [rnd.randrange(Q) for _ in range(n)]
    elif shape == "one_scalar":
        es = [rnd.randrange(Q)] * n
    elif shape == "bits":
        es = [rnd.randrange(2) for _ in range(n)]
    elif shape == "bits_and_blinding":
        es = [rnd.randrange(Q) if i % 1024 == 0 else (Q - 1 if rnd.random() < 0.5 else 0) for i in range(n)]
    elif shape == "small_values":
        es = [rnd.randrange(1 << rnd.choice((3, 7, 8, 14, 64))) for _ in range(n)]
    elif shape == "identities_and_negatives":
        es = [rnd.randrange(Q) for _ in range(n)]
        pts = [INF if i % 5 == 0 else (-pts[i - 1] if i % 5 == 1 else p) for i, p in enumerate(pts)]
    else:
        es = [[half, half + 1, half - 1, Q - 1, Q - 2, 1, (1 << 255) - 19, Q + 5][i % 8] for i in range(n)]
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars([e % (1 << 256) for e in es])
    want = cbind.msm_bytes(pb, sb, n, 4)
    try:
        eng.set_option("mid_min", 1)
        eng.set_option("mid_single_min", 1)
        assert eng.msm_bytes(pb, sb, n) == want
        want1 = cbind.msm_bytes(pb[:64 * (n - 7)], sb[:32 * (n - 7)], n - 7, 4)
        o0, o1 = eng.msm2_bytes(pb, sb, n, pb[:64 * (n - 7)], sb[:32 * (n - 7)], n - 7)      # the pair in ONE launch, unequal sizes
        assert o0 == want and o1 == want1
        for parts in (1, 2, 3, 4):                           # round 5: a window's pairs over 1 .. 4 blocks, their sums added by the host tail
            eng.set_option("mid_parts", parts)
            assert eng.msm_bytes(pb, sb, n) == want, parts
            assert tuple(eng.msm2_bytes(pb, sb, n, pb[:64 * (n - 7)], sb[:32 * (n - 7)], n - 7)) == (want, want1), parts
        eng.set_option("mid_parts", 0)
        eng.set_option("mid_min", -1)
        eng.set_option("mid_single_min", -1)
        assert eng.msm_bytes(pb, sb, n) == want
        o0, o1 = eng.msm2_bytes(pb, sb, n, pb[:64 * (n - 7)], sb[:32 * (n - 7)], n - 7)
        assert o0 == want
    finally:
        eng.set_option("mid_parts", 0)
        eng.set_option("mid_min", 0)
        eng.set_option("mid_single_min", 0)
