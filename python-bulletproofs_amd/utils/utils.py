"""Scalar-field element, Fiat-Shamir hash, point codecs, inner product
(reference: src/utils/utils.py).  Byte-exact where bytes matter (mod_hash, codecs)."""
import base64
from hashlib import sha256
from typing import List

from .. import engine as _engine
from ..ec import Point, mod_sqrt, pack_scalars, secp256k1

CURVE = secp256k1
BYTE_LENGTH = CURVE.q.bit_length() // 8


class ModP:
    """Integer mod p with the reference's observable behaviour (src/utils/utils.py:24-81):
    `+` and `*` with a plain int do not reduce, `-` does; `-v` is p - v; `v * Point` is a
    scalar multiplication (on the GPU); `v % m` gives an int; `inv()` raises
    Exception("modular inverse does not exist")."""

    __slots__ = ("x", "p")

    def __init__(self, x, p):
        self.x, self.p = x, p

    def _rhs(self, y):
        assert self.p == y.p
        return y.x

    def __add__(self, y):
        if isinstance(y, int):
            return ModP(self.x + y, self.p)
        return ModP((self.x + self._rhs(y)) % self.p, self.p)

    def __radd__(self, y):
        return self + y

    def __mul__(self, y):
        if isinstance(y, int):
            return ModP(self.x * y, self.p)
        if isinstance(y, Point):
            return self.x * y
        return ModP((self.x * self._rhs(y)) % self.p, self.p)

    def __sub__(self, y):
        if isinstance(y, int):
            return ModP((self.x - y) % self.p, self.p)
        return ModP((self.x - self._rhs(y)) % self.p, self.p)

    def __rsub__(self, y):
        return -(self - y)

    def __pow__(self, n):
        return ModP(pow(self.x, n, self.p), self.p)

    def __mod__(self, other):
        return self.x % other

    def __neg__(self):
        return ModP(self.p - self.x, self.p)

    def inv(self):
        try:
            return ModP(pow(self.x, -1, self.p), self.p)
        except ValueError:
            raise Exception("modular inverse does not exist")

    def __eq__(self, y):
        return self.p == y.p and (self.x - y.x) % self.p == 0

    def __hash__(self):
        return hash((self.x % self.p, self.p))

    def __int__(self):
        return self.x

    def __repr__(self):
        return str(self.x)

    __str__ = __repr__


def mod_hash(msg: bytes, p: int, non_zero: bool = True) -> ModP:
    """Try-and-increment SHA-256 into Z_p (src/utils/utils.py:84-97)."""
    keep = (1 << p.bit_length()) - 1
    counter = 0
    while True:
        counter += 1
        x = int.from_bytes(sha256(str(counter).encode() + msg).digest(), "big") & keep
        if x >= p or (non_zero and x == 0):
            continue
        return ModP(x, p)


def point_to_bytes(g: Point) -> bytes:
    if g == Point.IDENTITY_ELEMENT:
        return b"\x00"
    return (b"\x03" if g.y & 1 else b"\x02") + g.x.to_bytes(BYTE_LENGTH, "big")


def point_to_b64(g: Point) -> bytes:
    return base64.b64encode(point_to_bytes(g))


def bytes_to_point(b: bytes) -> Point:
    if b == 0:                      # (dead branch in the reference as well, utils.py:121)
        return Point.IDENTITY_ELEMENT
    odd = 0 if b[0] == 2 else 1
    x = int.from_bytes(b[1:], "big")
    y = mod_sqrt((x**3 + CURVE.a * x + CURVE.b) % CURVE.p, CURVE.p)[0]
    return Point(x, y if y % 2 == odd else CURVE.p - y, CURVE)


def b64_to_point(s: bytes) -> Point:
    return bytes_to_point(base64.b64decode(s))


def inner_product(a: List[ModP], b: List[ModP]) -> ModP:
    """<a, b> in Z_p (src/utils/utils.py:134-137); for p = q the products and the sum run
    on the GPU (bpmi_sc_dot)."""
    assert len(a) == len(b)
    p = a[0].p
    if p != CURVE.q:
        return sum([ai * bi for ai, bi in zip(a, b)], ModP(0, p))
    eng = _engine.default_engine()
    out = eng.sc_dot_bytes(pack_scalars(a, p), pack_scalars(b, p), len(a))
    return ModP(int.from_bytes(out, "little"), p)
