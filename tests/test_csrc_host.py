"""Unit tests of the DEVICE arithmetic headers (python-bulletproofs_amd/csrc/field.hpp,
curve.hpp) compiled for the host: limb arithmetic, lazy-magnitude discipline and the
complete group law are checked against Python integers / oracle.ec on CPU."""
import ctypes
import os
import random
import subprocess

import pytest

from helpers import gens, seed
from oracle.ec import INF, point_from_le64, point_to_le64, secp256k1

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "csrc_host", "host_shim.cpp")
INC = os.path.join(REPO, "python-bulletproofs_amd", "csrc")
SO = os.path.join(REPO, "tests", "csrc_host", "libhost_shim.so")
P = secp256k1.p


@pytest.fixture(scope="module")
def shim():
    deps = [SRC] + [os.path.join(INC, f) for f in os.listdir(INC) if f.endswith(".hpp")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", INC, SRC, "-o", SO])
    L = ctypes.CDLL(SO)
    L.t_fe_op.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    L.t_fe_is_zero.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.t_madd_chain.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    L.t_xyzz_sum.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    L.t_xyzz_dbl_n.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    L.t_jac_mul.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    L.t_jac_madd_chain.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    L.t_sc_op.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    L.t_sq_raw.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p]
    L.t_sq_from_sc.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
    L.t_host_f_op.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    L.t_host_tail.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                              ctypes.c_char_p]
    return L


def sc_op(L, op, a, b=0):
    out = ctypes.create_string_buffer(32)
    L.t_sc_op(op, a.to_bytes(32, "little"), b.to_bytes(32, "little"), out)
    return int.from_bytes(out.raw, "little")


def test_scalar_ops_mod_q(shim):
    """csrc/scalar.hpp against Python integers: multiplication, addition, negation, halving and the binary-Euclid inverse
    (the batch-preparation kernel's one inversion per proof) on edge values and random ones."""
    q = secp256k1.q
    rnd = random.Random(7)
    edge = [0, 1, 2, 3, q - 1, q - 2, (q - 1) // 2, (q + 1) // 2, 2**128, 2**255 % q, 2**32 - 1, 2**64, (1 << 200) - 1,
            0x14551231950B75FC4402DA1732FC9BEBF, q - 0x14551231950B75FC4402DA1732FC9BEBF]
    vals = edge + [rnd.randrange(q) for _ in range(400)] + [rnd.randrange(2**40) for _ in range(20)] + [1 << k for k in range(0, 256, 7)]
    for a in vals:
        assert sc_op(shim, 2, a) == (-a) % q
        assert sc_op(shim, 4, a) == a * pow(2, -1, q) % q
        inv = sc_op(shim, 3, a)
        assert inv == (pow(a, -1, q) if a else 0), hex(a)
    for a, b in [(x, y) for x in edge for y in edge] + [(rnd.choice(vals), rnd.choice(vals)) for _ in range(2000)]:
        assert sc_op(shim, 0, a, b) == a * b % q
        assert sc_op(shim, 1, a, b) == (a + b) % q
    for a in [q, q + 1, 2**256 - 1, 2**256 - 2, q - 1, 5]:
        assert sc_op(shim, 5, a) == a % q


def test_scalar_29bit_limb_arithmetic(shim):
    """The sq routines of csrc/scalar.hpp (9 x 29-bit limbs mod q, what the batch-preparation kernel multiplies with) against
    Python integers: every result is correct mod q AND loose (limbs < 2^29 + 2^10), also from the largest loose inputs, and
    long random chains of mul / add / sub / neg stay loose."""
    q = secp256k1.q
    rnd = random.Random(29)
    u32x9 = ctypes.c_uint32 * 9
    LOOSE = (1 << 29) + (1 << 10)

    def value(limbs):
        return sum(v << (29 * i) for i, v in enumerate(limbs))

    def raw(op, a, b):
        out_l, out_v = u32x9(), ctypes.create_string_buffer(32)
        shim.t_sq_raw(op, u32x9(*a), u32x9(*b), out_l, out_v)
        limbs = list(out_l)
        assert all(v < LOOSE for v in limbs), (op, [hex(v) for v in limbs])
        canon = int.from_bytes(out_v.raw, "little")
        assert canon < q and canon == value(limbs) % q
        return limbs, canon

    def from_int(x):
        out_l = u32x9()
        shim.t_sq_from_sc(x.to_bytes(32, "little"), out_l)
        assert value(list(out_l)) == x
        return list(out_l)

    top = [LOOSE - 1] * 9                                       # the largest loose operand
    edge = [from_int(v) for v in (0, 1, 2, q - 1, q - 2, q, q + 1, 2**256 - 1, 2**255, (1 << 232) - 1, 1 << 232, 2**128)] + [top, [0] * 8 + [LOOSE - 1]]
    edge += [[(1 << 29) - 1] * 9, [1 << 29] * 9, [LOOSE - 1] + [0] * 8]
    rand = [from_int(rnd.randrange(2**256)) for _ in range(200)] + [[rnd.randrange(LOOSE) for _ in range(9)] for _ in range(200)]
    pairs = [(a, b) for a in edge for b in edge] + [(rnd.choice(rand + edge), rnd.choice(rand + edge)) for _ in range(4000)]
    for a, b in pairs:
        va, vb = value(a), value(b)
        assert raw(0, a, b)[1] == va * vb % q
        assert raw(1, a, b)[1] == (va + vb) % q
        assert raw(2, a, b)[1] == (va - vb) % q
    for a in edge + rand:
        assert raw(3, a, a)[1] == (-value(a)) % q
        assert raw(4, a, a)[1] == value(a) % q
    # chains: the loose invariant is closed under every operation
    x, vx = top, value(top)
    for step in range(3000):
        y = rnd.choice(rand + edge)
        op = rnd.randrange(4)
        x, got = raw(op, x, y)
        vx = [vx * value(y), vx + value(y), vx - value(y), -vx][op] % q
        assert got == vx


def fe_op(L, op, a, b=0):
    out = ctypes.create_string_buffer(32)
    L.t_fe_op(op, a.to_bytes(32, "little"), b.to_bytes(32, "little"), out)
    return int.from_bytes(out.raw, "little")


EDGE = [0, 1, 2, P - 1, P - 2, P, P + 1, 2**256 - 1, 2**256 - 2, 2**255, 2**32 + 977, 2**32 + 976,
        (1 << 29) - 1, 1 << 29, (1 << 232) - 1, 1 << 232, 0x1FFFFFFF1FFFFFFF1FFFFFFF1FFFFFFF]


def test_field_ops_random_and_edges(shim):
    rnd = random.Random(42)
    vals = EDGE + [rnd.randrange(2**256) for _ in range(300)] + [rnd.randrange(2**40) for _ in range(20)]
    pairs = [(a, b) for a in EDGE for b in EDGE] + [(rnd.choice(vals), rnd.choice(vals)) for _ in range(3000)]
    for a, b in pairs:
        assert fe_op(shim, 0, a, b) == a * b % P
        assert fe_op(shim, 2, a, b) == (a + b) % P
        assert fe_op(shim, 3, a, b) == (a - b) % P
        assert fe_op(shim, 6, a, b) == (a + b) ** 2 % P
        assert fe_op(shim, 7, a, b) == (a - b) * b % P
        assert fe_op(shim, 9, a, b) == (a - 3 * b) % P
        assert fe_op(shim, 10, a, b) == (-b) % P
        assert bool(shim.t_fe_is_zero(a.to_bytes(32, "little"), b.to_bytes(32, "little"))) == ((a - b) % P == 0)
    for a in vals:
        assert fe_op(shim, 1, a) == a * a % P
        assert fe_op(shim, 4, a) == (-a) % P
        assert fe_op(shim, 8, a) == a % P
    for a in EDGE + vals[:40]:
        if a % P:
            assert fe_op(shim, 5, a) == pow(a, -1, P)
    assert fe_op(shim, 5, 0) == 0
    for a in EDGE + vals[:60]:
        assert fe_op(shim, 11, a) == pow(a % P, (P + 1) // 4, P)


def madd_chain(L, start, pts, negs):
    out = ctypes.create_string_buffer(64)
    L.t_madd_chain(point_to_le64(start), b"".join(point_to_le64(p) for p in pts), bytes(negs), len(pts), out)
    return point_from_le64(out.raw)


def test_madd_complete_group_law(shim):
    pts = gens(24, seed(31))
    rnd = random.Random(3)
    for trial in range(30):
        k = rnd.randrange(1, 12)
        sel = [rnd.choice(pts) for _ in range(k)]
        negs = [rnd.randrange(2) for _ in range(k)]
        want = INF
        for p, s in zip(sel, negs):
            want = want + (-p if s else p)
        assert madd_chain(shim, INF, sel, negs) == want
        assert madd_chain(shim, pts[0], sel, negs) == pts[0] + want
    A, B = pts[0], pts[1]
    # doubling, cancellation, identity addends, re-growth after cancellation
    assert madd_chain(shim, INF, [A, A], [0, 0]) == 2 * A
    assert madd_chain(shim, INF, [A, A, A, A], [0, 0, 0, 0]) == 4 * A
    assert madd_chain(shim, INF, [A, A], [0, 1]) == INF
    assert madd_chain(shim, INF, [A, A, B], [0, 1, 0]) == B
    assert madd_chain(shim, A, [INF, INF], [0, 1]) == A
    assert madd_chain(shim, INF, [A, B, A + B], [0, 0, 1]) == INF          # acc == addend negated
    assert madd_chain(shim, INF, [A, B, A + B], [0, 0, 0]) == 2 * (A + B)  # acc == addend (Z != 1 doubling)
    assert madd_chain(shim, 2 * A, [A, A, A], [1, 1, 1]) == -A


def test_xyzz_add_and_double(shim):
    pts = gens(20, seed(32))
    for pre in (0, 1, 2):
        out = ctypes.create_string_buffer(64)
        shim.t_xyzz_sum(b"".join(point_to_le64(p) for p in pts), len(pts), pre, out)
        want = INF
        for p in pts:
            want = want + p
        assert point_from_le64(out.raw) == want
    # same point twice (general doubling path), opposite points, identity members
    for lst, want in (([pts[0], pts[0]], 2 * pts[0]), ([pts[0], -pts[0]], INF), ([INF, pts[1], INF], pts[1]),
                      ([pts[0], pts[1], -pts[0], -pts[1]], INF), ([pts[2]] * 5, 5 * pts[2])):
        for pre in (0, 2):
            out = ctypes.create_string_buffer(64)
            shim.t_xyzz_sum(b"".join(point_to_le64(p) for p in lst), len(lst), pre, out)
            assert point_from_le64(out.raw) == want
    for n in (0, 1, 5, 64):
        out = ctypes.create_string_buffer(64)
        shim.t_xyzz_dbl_n(point_to_le64(pts[3]), n, out)
        assert point_from_le64(out.raw) == (2**n) * pts[3]
    out = ctypes.create_string_buffer(64)
    shim.t_xyzz_dbl_n(point_to_le64(INF), 3, out)
    assert point_from_le64(out.raw) == INF


def test_jacobian_ladder_and_exceptions(shim):
    from helpers import Q
    pts = gens(10, seed(33))
    rnd = random.Random(8)
    ks = [0, 1, 2, 3, Q - 1, Q - 2, (Q - 1) // 2, 2**255, 2**256 - 1] + [rnd.randrange(2**256) for _ in range(20)]
    for k in ks:
        for neg in (0, 1):
            out = ctypes.create_string_buffer(64)
            shim.t_jac_mul(point_to_le64(pts[0]), k.to_bytes(32, "little"), neg, out)
            assert point_from_le64(out.raw) == (k % Q) * pts[0], (k, neg)
    out = ctypes.create_string_buffer(64)
    shim.t_jac_mul(point_to_le64(INF), (5).to_bytes(32, "little"), 0, out)
    assert point_from_le64(out.raw) == INF
    A, B = pts[1], pts[2]
    for lst, negs, want in (([A, A], [0, 0], 2 * A), ([A, A], [0, 1], INF), ([A, A, B], [0, 1, 0], B),
                            ([A, B, A + B], [0, 0, 1], INF), ([A, B, A + B], [0, 0, 0], 2 * (A + B)),
                            ([A, A, A, A, A], [0] * 5, 5 * A), ([A, INF, B], [0, 0, 1], A - B)):
        out = ctypes.create_string_buffer(64)
        shim.t_jac_madd_chain(b"".join(point_to_le64(p) for p in lst), bytes(negs), len(lst), out)
        assert point_from_le64(out.raw) == want


# ---- the multiplication family on raw limbs: adversarial magnitudes, loose output bounds --------------------
M29 = (1 << 29) - 1
LOOSE_MAX = [M29 + (1 << 22), M29 + (1 << 15)] + [M29] * 6 + [(1 << 24) - 1]
TIGHT_MAX = [M29] * 8 + [(1 << 24) + (1 << 20)]


def limbs_value(l):
    return sum(v << (29 * k) for k, v in enumerate(l))


def fe_raw(L, op, a, b=None, c=None, d=None):
    z = [0] * 9
    arr = lambda v: (ctypes.c_uint32 * 9)(*(v if v is not None else z))
    out = (ctypes.c_uint32 * 9)()
    L.t_fe_raw(op, arr(a), arr(b), arr(c), arr(d), out)
    return list(out)


def assert_loose(l):
    assert l[0] < (1 << 29) + (1 << 22) and l[1] < (1 << 29) + (1 << 15), l
    assert all(v < (1 << 29) for v in l[2:8]) and l[8] < (1 << 24), l


def scaled(pattern, m, limb8=None):
    """a magnitude-m operand: every limb at its maximum"""
    out = [min(v * m, 0xFFFFFFFF) for v in pattern]
    if limb8 is not None:
        out[8] = limb8
    return out


def mul_family_cases(rnd, count):
    """(op, a, b, c, d, expected value mod p) with the largest limbs each variant allows, and random lazy values."""
    cases = []
    bias2 = [0x3FFFF85E, 0x3FFFFFEE] + [0x3FFFFFFE] * 6 + [0x01FFFFFE]
    bias8 = [0xBFFFE178, 0xBFFFFFBA] + [0xBFFFFFFA] * 6 + [0x07FFFFFA]
    full = [0xFFFFFFFF] * 9
    for ma, mb in ((1, 1), (1, 7), (7, 1), (2, 3), (3, 2), (1, 3), (2, 2)):
        for pat in (LOOSE_MAX, TIGHT_MAX):
            # limb 8 of a lazy sum is small; the all-limbs-maximal pattern is the worst case for every column
            a, b = scaled(pat, ma, pat[8] * ma), scaled(pat, mb, pat[8] * mb)
            cases.append((0, a, b, None, None))
            cases.append((2, a, b, full, None))                        # the largest possible addend
            cases.append((2, a, b, bias8, None))
    for m in (1, 2):
        for pat in (LOOSE_MAX, TIGHT_MAX):
            a = scaled(pat, m, pat[8] * m)
            cases.append((1, a, None, None, None))
            cases.append((3, a, None, full, None))
    for (ma, mb, mc, md) in ((1, 3, 2, 1), (1, 1, 1, 1), (1, 4, 1, 3), (2, 2, 1, 3), (1, 7, 0, 0)):
        for pat in (LOOSE_MAX, TIGHT_MAX):
            cases.append((4, scaled(pat, ma, pat[8] * ma), scaled(pat, mb, pat[8] * mb), scaled(pat, mc, pat[8] * mc), scaled(pat, md, pat[8] * md)))
    zero, one = [0] * 9, [1] + [0] * 8
    plimbs = [0x1FFFFC2F, 0x1FFFFFF7] + [M29] * 6 + [0x00FFFFFF]
    for x in (zero, one, plimbs, bias2):
        cases += [(0, x, LOOSE_MAX, None, None), (0, x, x, None, None), (1, x, None, None, None), (4, x, one, plimbs, one), (2, zero, zero, x, None)]
    for _ in range(count):
        op = rnd.randrange(5)
        def lazy(m):
            return [rnd.randrange(m * (1 << 29)) for _ in range(8)] + [rnd.randrange(m * ((1 << 24) + (1 << 20)))]
        if op == 0:
            ma = rnd.randrange(1, 8)
            cases.append((0, lazy(ma), lazy(max(1, 7 // ma)), None, None))
        elif op == 1:
            cases.append((1, lazy(rnd.randrange(1, 3)), None, None, None))
        elif op == 2:
            ma = rnd.randrange(1, 7)
            cases.append((2, lazy(ma), lazy(max(1, 6 // ma)), [rnd.randrange(1 << 32) for _ in range(9)], None))
        elif op == 3:
            cases.append((3, lazy(rnd.randrange(1, 3)), None, [rnd.randrange(1 << 32) for _ in range(9)], None))
        else:
            cases.append((4, lazy(1), lazy(rnd.randrange(1, 4)), lazy(rnd.randrange(1, 3)), lazy(1)))
    return cases


def mul_family_expected(op, a, b, c, d):
    va, vb, vc, vd = (limbs_value(x) if x is not None else 0 for x in (a, b, c, d))
    return {0: va * vb, 1: va * va, 2: va * vb + vc, 3: va * va + vc, 4: va * vb + vc * vd}[op] % P


def test_multiplication_family_raw_limbs(shim):
    shim.t_fe_raw.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_uint32)] * 5
    rnd = random.Random(77)
    for op, a, b, c, d in mul_family_cases(rnd, 4000):
        out = fe_raw(shim, op, a, b, c, d)
        assert_loose(out)
        assert limbs_value(out) % P == mul_family_expected(op, a, b, c, d), (op, a, b, c, d)
        assert limbs_value(out) < (1 << 256) + (1 << 45)
        # a loose value that is 0 (mod p) has exactly the limbs of 0 or of p
        if limbs_value(out) % P == 0:
            assert limbs_value(out) in (0, P)
    # carry / canon accept loose and lazy inputs
    for l in (LOOSE_MAX, TIGHT_MAX, [0xFFFFFFFF] * 9, [0] * 9):
        t = fe_raw(shim, 5, l)
        assert limbs_value(t) % P == limbs_value(l) % P and all(v < (1 << 29) for v in t[:8]) and t[8] <= (1 << 24) + (1 << 20)
        cn = fe_raw(shim, 6, l)
        assert limbs_value(cn) == limbs_value(l) % P


def test_generated_device_bodies_are_current():
    """csrc/field_gen.hpp is what tools/gen_field_asm.py emits today (the asm is never edited by hand)."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_field_asm.py"), "--check"])
    assert r.returncode == 0, "run `python tools/gen_field_asm.py` and commit csrc/field_gen.hpp"


def test_glv_split_and_beta(shim):
    """csrc/scalar.hpp glv_split: k = k1 + k2 lambda (mod q) with |k1|, |k2| < 2^128 on random and adversarial scalars (every
    single-bit value, values around multiples of the lattice constants, 0, 1, q - 1); fe_mul_beta: lambda (x, y) = (beta x, y)."""
    q, p = secp256k1.q, secp256k1.p
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    beta = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE
    assert pow(lam, 3, q) == 1 and pow(beta, 3, p) == 1
    shim.t_glv_split.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    shim.t_fe_mul_beta.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    rnd = random.Random(31)
    a1, mb1 = 0x3086D221A7D46BCDE86C90E49284EB15, 0xE4437ED6010E88286F547FA90ABFE4C3
    cases = [0, 1, 2, q - 1, q - 2, q // 2, q // 2 + 1, lam, q - lam, (lam * lam) % q, a1, mb1, q - a1, q - mb1]
    cases += [1 << b for b in range(256)] + [(1 << b) - 1 for b in range(1, 257) if (1 << b) - 1 < q]
    cases += [(t * a1 + d) % q for t in (1, 7, 1 << 64, (1 << 127) + 3) for d in (-1, 0, 1)]
    cases += [rnd.randrange(q) for _ in range(20000)]
    worst = 0
    for k in cases:
        k %= q
        k1b, k2b, sg = ctypes.create_string_buffer(16), ctypes.create_string_buffer(16), (ctypes.c_int * 2)()
        shim.t_glv_split(k.to_bytes(32, "little"), k1b, k2b, sg)
        k1, k2 = int.from_bytes(k1b.raw, "little"), int.from_bytes(k2b.raw, "little")
        worst = max(worst, k1, k2)
        v1, v2 = (-k1 if sg[0] else k1), (-k2 if sg[1] else k2)
        assert (v1 + v2 * lam - k) % q == 0, hex(k)
    assert worst < 1 << 128
    G = secp256k1.G
    for pt in [G] + gens(5, seed(3)):
        out = ctypes.create_string_buffer(32)
        shim.t_fe_mul_beta(pt.x.to_bytes(32, "little"), out)
        bx = int.from_bytes(out.raw, "little")
        assert bx == beta * pt.x % p
        img = lam * pt
        assert (img.x, img.y) == (bx, pt.y)


def test_glv_fold_operation_list_spells_the_coefficients(shim):
    """csrc/fold_ops_host.hpp glv_fold_ops -- the host half of the inner-product prover's 16-way generator fold (round 4): the ladder
    k_ec_multifold_w4g runs "double n times, then add (-)(j-th odd multiple) of (lambda?) point row / 2" from this list, so the list,
    read as integers, must spell every coefficient: with c_r = sum over the operations of row r of (+-)(2 j + 1) 2^(doublings still to
    come), c_(2t) + c_(2t+1) lambda == coefficient t (mod q).  Also: every digit odd and <= 7, at most 129 doublings in all, the
    first operation doubles nothing, rows within one position in ascending order."""
    q = secp256k1.q
    lam = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
    shim.t_glv_fold_ops.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    shim.t_glv_fold_ops.restype = ctypes.c_int
    rnd = random.Random(41)
    edge = [0, 1, 2, q - 1, q - 2, q // 2, lam, q - lam, (1 << 128) - 1, 1 << 128, (1 << 255) % q, 7, 8, 15, 16]
    for K in (1, 2, 16, 16, 16):
        coefs = [rnd.choice(edge) if rnd.random() < 0.3 else rnd.randrange(q) for _ in range(K)]
        ops = (ctypes.c_uint32 * 2048)()
        tail = ctypes.c_uint32()
        nops = shim.t_glv_fold_ops(b"".join(c.to_bytes(32, "little") for c in coefs), K, ops, ctypes.byref(tail))
        assert nops >= 0
        ops = list(ops[:nops])
        assert not ops or ops[0] & 255 == 0
        total_dbl = sum(o & 255 for o in ops) + tail.value
        assert total_dbl <= 132
        rows = [0] * (2 * K)
        left = total_dbl
        prev_row = -1
        for o in ops:
            n_dbl, r, j, neg = o & 255, (o >> 8) & 31, (o >> 13) & 7, (o >> 16) & 1
            assert r < 2 * K and j < 4
            left -= n_dbl
            if n_dbl == 0 and prev_row >= 0:
                assert r > prev_row
            prev_row = r
            rows[r] += (-1 if neg else 1) * (2 * j + 1) << left
        assert left == tail.value
        for t in range(K):
            assert abs(rows[2 * t]) < 1 << 129 and abs(rows[2 * t + 1]) < 1 << 129
            assert (rows[2 * t] + rows[2 * t + 1] * lam - coefs[t]) % q == 0, (K, t)
    assert shim.t_glv_fold_ops(bytes(32 * 17), 17, (ctypes.c_uint32 * 2048)(), ctypes.byref(ctypes.c_uint32())) == -1



def test_sha_block_on_cpu_extensions_equals_the_portable_one(shim):
    """csrc/rp_batch_host.hpp: the SHA-256 compression function on the host CPU's SHA extensions (round 4; chosen at run time) against
    the portable one, from random states on random and structured blocks; and the dispatched function against hashlib."""
    import hashlib
    import struct
    shim.t_sha_block.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint32), ctypes.c_char_p]
    shim.t_sha_block.restype = ctypes.c_int
    rnd = random.Random(77)
    h0 = (0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19)
    have = None
    for k in range(3000):
        block = bytes([0] * 64) if k == 0 else bytes([255] * 64) if k == 1 else bytes(range(64)) if k == 2 else rnd.randbytes(64)
        start = h0 if k % 2 == 0 else tuple(rnd.getrandbits(32) for _ in range(8))
        outs = []
        for which in (0, 1, 2):
            st = (ctypes.c_uint32 * 8)(*start)
            ok = shim.t_sha_block(which, st, block)
            outs.append(tuple(st) if ok else None)
        have = outs[1] is not None
        assert outs[0] == outs[2] and (outs[1] is None or outs[1] == outs[0]), k
    # the dispatched function as a whole hash: one padded block
    msg = b"bpmi" * 9
    block = msg + b"\x80" + bytes(64 - len(msg) - 9) + struct.pack(">Q", 8 * len(msg))
    st = (ctypes.c_uint32 * 8)(*h0)
    shim.t_sha_block(2, st, block)
    assert struct.pack(">8I", *st) == hashlib.sha256(msg).digest()
    if not have:
        pytest.skip("this CPU has no SHA extensions: only the portable path and the dispatch were checked")



def test_host_tail_field_arithmetic(shim):
    """host_tail.hpp's 4 x 64-bit field routines (round 5: a squaring of its own, the inversion by an addition chain) against Python
    integers: random values and the carry edges of the 2^256 - p folding."""
    rnd = random.Random(99)
    edge = [0, 1, 2, P - 1, P - 2, (1 << 255), (1 << 256) - (1 << 32) - 978, (1 << 128) - 1, (1 << 192) + 5, 0x1000003D1, P - 0x1000003D1, (P + 1) // 2]
    vals = edge + [rnd.randrange(P) for _ in range(300)]

    def op(code, a, b=0):
        out = ctypes.create_string_buffer(32)
        shim.t_host_f_op(code, a.to_bytes(32, "little"), b.to_bytes(32, "little"), out)
        return int.from_bytes(out.raw, "little")

    for i, a in enumerate(vals):
        b = vals[(i * 7 + 3) % len(vals)]
        assert op(0, a, b) == a * b % P
        assert op(1, a) == a * a % P
        assert op(3, a, b) == (a + b) % P and op(4, a, b) == (a - b) % P
        if a and i < 80:
            assert op(2, a) == pow(a, -1, P)
    assert op(2, 0) == 0


@pytest.mark.parametrize("c,nv,off,top,top_off", [(16, 4, (0, 4, 8, 12), 0, (0, 0, 0, 0)), (15, 4, (0, 4, 7, 11), 1, (0, 4, 8, 12)), (13, 4, (0, 3, 6, 9), 0, (0, 0, 0, 0)),
                                                  (13, 4, (0, 3, 6, 9), 9, (0, 4, 7, 10)), (12, 4, (0, 3, 6, 9), 4, (0, 3, 6, 9)), (10, 4, (0, 3, 5, 7), 6, (0, 3, 5, 8)),
                                                  (8, 1, (0, 0, 0, 0), 0, (0, 0, 0, 0)), (7, 1, (0, 0, 0, 0), 0, (0, 0, 0, 0)),
                                                  (7, 3, (0, 0, 0, 0), 0, (0, 0, 0, 0)), (7, 4, (0, 0, 0, 0), 0, (0, 0, 0, 0))])      # k_msm_mid's parts: several sums per window at offset 0
def test_host_tail_combines_the_window_sums(shim, c, nv, off, top, top_off):
    """tail_combine (the MSM's Horner chain over bit positions, on the host): sum_w 2^(start of window w) sum_v 2^(off_v) E[w][v] for
    window sums in projective form with random scalings, identities among them, against the oracle's affine arithmetic.  top = the
    number of WIDE windows at the top (mixed widths: 256 // c windows, the last `top` of them c + 1 bits wide and split at their
    own offsets; 1 at c = 15 is the round's first form of it)."""
    W = (256 // c) if top else 255 // c + 1
    assert not top or (W - top) * c + top * (c + 1) == 256
    rnd = random.Random(c * 100 + nv)
    base = gens(7, seed(3))
    pts, zs, want = [], [], INF
    for w in range(W):
        for v in range(nv):
            k = rnd.randrange(1, 1 << 40)
            kind = rnd.randrange(10)
            pt = INF if kind == 0 else base[rnd.randrange(7)] * k
            z = 0 if kind == 1 else rnd.randrange(1, P)
            pts.append(pt)
            zs.append(z)
            if kind > 1:
                wide = top and w >= W - top
                o = (top_off if wide else off)[v]
                want = want + pt * (1 << (c * w + max(0, w - (W - top)) * (1 if top else 0) + o))
    out = ctypes.create_string_buffer(64)
    o4, t4 = (ctypes.c_uint32 * 4)(*off), (ctypes.c_uint32 * 4)(*top_off)
    shim.t_host_tail(b"".join(point_to_le64(p) for p in pts), b"".join(z.to_bytes(32, "little") for z in zs), W, c, nv, o4, top, t4, out)
    assert out.raw == point_to_le64(want)
