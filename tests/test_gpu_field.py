"""The DEVICE bodies of the field-multiplication family (csrc/field_gen.hpp: generated v_mad_u64_u32 chains) against
the HOST build of the same header (the C body fe_mac_c, itself checked against Python integers in
tests/test_csrc_host.py): identical limbs on adversarial limb patterns -- every variant at the largest
magnitudes it allows, the largest addends, loose and tight maxima, zero / one / p -- and on random lazy values.
This is the arithmetic under every `Point + Point` of the reference (/root/reference/src/pippenger/group.py:31-32)."""
import ctypes
import random

import pytest

from test_csrc_host import P, assert_loose, fe_raw, limbs_value, mul_family_cases, mul_family_expected, shim  # noqa: F401

pytestmark = pytest.mark.gpu


def test_device_multiplication_family_equals_host_limb_for_limb(shim):  # noqa: F811
    import gpu_common
    eng = gpu_common.engine()
    shim.t_fe_raw.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_uint32)] * 5
    rnd = random.Random(2026)
    cases = mul_family_cases(rnd, 60000)
    z = [0] * 9
    by_op = {}
    for case in cases:
        by_op.setdefault(case[0], []).append(case)
    total = 0
    for op, lst in sorted(by_op.items()):
        n = len(lst)
        flat = lambda idx: (ctypes.c_uint32 * (9 * n))(*[v for case in lst for v in (case[idx] if case[idx] is not None else z)])
        out = (ctypes.c_uint32 * (9 * n))()
        eng._ck(eng.lib.bpmi_debug_fe_op(eng.ctx, op, flat(1), flat(2), flat(3), flat(4), n, out))
        got = list(out)
        for i, (_, a, b, c, d) in enumerate(lst):
            dev = got[9 * i: 9 * i + 9]
            if i < 3000:                    # the host shim call is the slow part; every case is still checked against Python below
                assert dev == fe_raw(shim, op, a, b, c, d), (op, a, b, c, d)
            assert_loose(dev)
            assert limbs_value(dev) % P == mul_family_expected(op, a, b, c, d), (op, a, b, c, d)
        total += n
    assert total >= 60000
    # carry / canonical form of lazy and loose inputs
    pats = [[0xFFFFFFFF] * 9, [0] * 9, [0x1FFFFC2F, 0x1FFFFFF7] + [0x1FFFFFFF] * 6 + [0x00FFFFFF]] + \
           [[rnd.randrange(1 << 32) for _ in range(9)] for _ in range(500)]
    n = len(pats)
    arr = (ctypes.c_uint32 * (9 * n))(*[v for p in pats for v in p])
    for op in (5, 6):
        out = (ctypes.c_uint32 * (9 * n))()
        eng._ck(eng.lib.bpmi_debug_fe_op(eng.ctx, op, arr, arr, arr, arr, n, out))
        for i, p in enumerate(pats):
            assert list(out)[9 * i: 9 * i + 9] == fe_raw(shim, op, p)


def test_device_mod_q_limb_arithmetic_equals_host(shim):  # noqa: F811
    """sq_mul_dev (csrc/scalar_gen.hpp, generated multiply-add chains), sq_add / sq_sub / sq_neg, the canonical form and the
    binary-Euclid inverse ON THE DEVICE against the host build of the same header (itself pinned to Python integers in
    tests/test_csrc_host.py): identical limbs, also from the largest loose operands."""
    import gpu_common
    from test_csrc_host import secp256k1
    eng = gpu_common.engine()
    q = secp256k1.q
    rnd = random.Random(31)
    u32x9 = ctypes.c_uint32 * 9
    LOOSE = (1 << 29) + (1 << 10)
    top = [LOOSE - 1] * 9

    def host(op, a, b):
        out_l, out_v = u32x9(), ctypes.create_string_buffer(32)
        shim.t_sq_raw(op, u32x9(*a), u32x9(*b), out_l, out_v)
        return list(out_l), int.from_bytes(out_v.raw, "little")

    def value(limbs):
        return sum(v << (29 * i) for i, v in enumerate(limbs))

    edge = [top, [0] * 9, [1] + [0] * 8, [(1 << 29) - 1] * 9, [1 << 29] * 9, [0] * 8 + [LOOSE - 1], [LOOSE - 1] + [0] * 8]
    edge += [[(v >> (29 * i)) & 0x1FFFFFFF for i in range(9)] for v in (q, q - 1, q + 1, 2**256 - 1, 2**255, 1 << 232)]
    rand = [[rnd.randrange(LOOSE) for _ in range(9)] for _ in range(3000)] + [[rnd.randrange(1 << 29) for _ in range(8)] + [rnd.randrange(1 << 24)] for _ in range(3000)]
    pairs = [(a, b) for a in edge for b in edge] + [(rnd.choice(rand + edge), rnd.choice(rand + edge)) for _ in range(20000)]
    n = len(pairs)
    A = (ctypes.c_uint32 * (9 * n))(*[v for a, _ in pairs for v in a])
    B = (ctypes.c_uint32 * (9 * n))(*[v for _, b in pairs for v in b])
    for dev_op, host_op, fn in ((10, 0, lambda x, y: x * y), (11, 1, lambda x, y: x + y), (12, 2, lambda x, y: x - y), (13, 3, lambda x, y: -x)):
        out = (ctypes.c_uint32 * (9 * n))()
        eng._ck(eng.lib.bpmi_debug_fe_op(eng.ctx, dev_op, A, B, A, A, n, out))
        got = list(out)
        for i, (a, b) in enumerate(pairs):
            dev = got[9 * i: 9 * i + 9]
            assert all(v < LOOSE for v in dev)
            assert value(dev) % q == fn(value(a), value(b)) % q, (dev_op, a, b)
            if i < 2500:
                assert dev == host(host_op, a, b)[0], (dev_op, a, b)
    # canonical form and inverse
    out = (ctypes.c_uint32 * (9 * n))()
    eng._ck(eng.lib.bpmi_debug_fe_op(eng.ctx, 14, A, B, A, A, n, out))
    got = list(out)
    canon = []
    for i, (a, _) in enumerate(pairs):
        w = got[9 * i: 9 * i + 8]
        c = sum(v << (32 * k) for k, v in enumerate(w))
        assert c == value(a) % q
        canon.append(c)
    m = 4000
    out = (ctypes.c_uint32 * (9 * m))()
    eng._ck(eng.lib.bpmi_debug_fe_op(eng.ctx, 15, A, B, A, A, m, out))
    got = list(out)
    for i in range(m):
        inv = sum(v << (32 * k) for k, v in enumerate(got[9 * i: 9 * i + 8]))
        assert inv == (pow(canon[i], -1, q) if canon[i] else 0)


def test_four_lane_point_addition_equals_the_affine_group_law():
    """csrc/msm_kernels.hpp quad_add (a general XYZZ addition spread over four lanes, the step of the bucket reduction's latency-
    bound stages) against the oracle's affine group law on 6 000 pairs: random points under random projective scalings, the
    identity on either side and on both, P + P (the doubling path), P + (-P), equal points under DIFFERENT scalings, and
    neighbours of every special case inside the same wave."""
    import gpu_common
    from oracle.ec import INF, secp256k1
    from oracle import cbind
    eng = gpu_common.engine()
    rnd = random.Random(77)
    M29 = (1 << 29) - 1

    def limbs(v):
        return [(v >> (29 * k)) & M29 for k in range(9)]

    def record(pt, z):
        if pt == INF:
            return [0] * 36
        zz, zzz = z * z % P, z * z * z % P
        return limbs(pt.x * zz % P) + limbs(pt.y * zzz % P) + limbs(zz) + limbs(zzz)

    pool = cbind.ec_mul_batch([secp256k1.G] * 300, [rnd.randrange(1, secp256k1.q) for _ in range(300)])
    pairs = []
    for i in range(6000):
        a, b = rnd.choice(pool), rnd.choice(pool)
        kind = i % 12
        if kind == 1: a = INF
        elif kind == 2: b = INF
        elif kind == 3: a = b = INF
        elif kind == 4: b = a
        elif kind == 5: b = -a
        pairs.append((a, b))
    n = len(pairs)
    za = [rnd.choice((1, rnd.randrange(1, P))) for _ in range(n)]
    zb = [rnd.choice((1, rnd.randrange(1, P))) for _ in range(n)]
    ra = (ctypes.c_uint32 * (36 * n))(*[v for (a, _), z in zip(pairs, za) for v in record(a, z)])
    rb = (ctypes.c_uint32 * (36 * n))(*[v for (_, b), z in zip(pairs, zb) for v in record(b, z)])
    out = (ctypes.c_uint32 * (36 * n))()
    eng._ck(eng.lib.bpmi_debug_quad_add(eng.ctx, ra, rb, n, out))
    got = list(out)
    for i, (a, b) in enumerate(pairs):
        rec = got[36 * i: 36 * i + 36]
        X, Y, ZZ, ZZZ = (limbs_value(rec[9 * k: 9 * k + 9]) % P for k in range(4))
        want = a + b
        if want == INF:
            assert ZZ == 0, (i, "expected the identity")
        else:
            assert ZZ != 0 and ZZZ != 0, i
            assert (X * pow(ZZ, -1, P) % P, Y * pow(ZZZ, -1, P) % P) == (want.x, want.y), (i, i % 12)
            assert pow(ZZ, 3, P) == ZZZ * ZZZ % P                      # a consistent (Z^2, Z^3) pair
