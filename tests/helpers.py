"""Shared input derivation for tests: the same seed -> input rules as
tests/golden/make_golden.py (which uses the reference's own elliptic_hash /
mod_hash; oracle.bp_ref's versions are pinned to them by hash_codec.json)."""
from oracle import bp_ref as R
from oracle.ec import Point, secp256k1, INF

Q = secp256k1.q


def seed(i):
    return bytes([i]) * 10


def gens(n, s):
    return [R.elliptic_hash(str(i).encode() + s) for i in range(n)]


def scal(n, s):
    return [R.mod_hash(str(i).encode() + s, Q) for i in range(n)]


def P(xy):
    x, y = int(xy[0], 16), int(xy[1], 16)
    if x == 0 and y == 0:
        return INF
    return Point(x, y, secp256k1)


def hx(v):
    return "%x" % int(v)
