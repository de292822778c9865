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
    return L


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
