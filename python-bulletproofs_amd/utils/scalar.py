"""Scalars of the proofs: the field element class the reference's callers use, the Fiat-Shamir
hash into that field, and the inner product (reference: src/utils/utils.py:24-97,134-137).

ModP reproduces the OBSERVABLE behaviour of the reference's class, quirks included, because
callers rely on it (e.g. Pippenger does `e % order` on whatever it is handed):
  v + int, v * int      not reduced             v - int, v - w, v + w, v * w   reduced
  -v                    p - v (so -0 is p)      v % m                           a plain int
  v * Point             scalar multiplication   v.inv()                         Exception("modular
  v == w                equal residues, same p                                  inverse does not exist")
"""
from hashlib import sha256

from .. import engine as _engine
from ..ec import Point, pack_scalars, secp256k1


class ModP:
    __slots__ = ("x", "p")

    def __init__(self, x, p):
        self.x = x
        self.p = p

    # ---- helpers ---------------------------------------------------------------------------
    def _peer(self, other):
        assert other.p == self.p
        return other.x

    def _make(self, value):
        return ModP(value, self.p)

    # ---- ring operations -------------------------------------------------------------------
    def __add__(self, other):
        if isinstance(other, int):
            return self._make(self.x + other)
        return self._make((self.x + self._peer(other)) % self.p)

    __radd__ = __add__

    def __sub__(self, other):
        rhs = other if isinstance(other, int) else self._peer(other)
        return self._make((self.x - rhs) % self.p)

    def __rsub__(self, other):
        return -(self - other)

    def __mul__(self, other):
        if isinstance(other, Point):
            return self.x * other                     # -> Point.__rmul__: one-term MSM on the GPU
        if isinstance(other, int):
            return self._make(self.x * other)
        return self._make(self.x * self._peer(other) % self.p)

    def __pow__(self, exponent):
        return self._make(pow(self.x, exponent, self.p))

    def __neg__(self):
        return self._make(self.p - self.x)

    def inv(self):
        try:
            return self._make(pow(self.x, -1, self.p))
        except ValueError:
            raise Exception("modular inverse does not exist") from None

    # ---- conversions / comparisons ---------------------------------------------------------
    def __mod__(self, modulus):
        return self.x % modulus

    def __int__(self):
        return self.x

    def __eq__(self, other):
        return self.p == other.p and (self.x - other.x) % self.p == 0

    def __hash__(self):
        return hash((self.x % self.p, self.p))

    def __str__(self):
        return "%d" % self.x

    __repr__ = __str__


def mod_hash(msg: bytes, p: int, non_zero: bool = True) -> ModP:
    """First i = 1, 2, ... with  SHA-256(str(i) || msg) mod 2^bitlen(p)  in [1, p) (or [0, p)
    when non_zero is False): the reference's challenge derivation (src/utils/utils.py:84-97),
    pinned by tests/golden/hash_codec.json."""
    width_mask = (1 << p.bit_length()) - 1
    i = 1
    while True:
        candidate = int.from_bytes(sha256(b"%d" % i + msg).digest(), "big") & width_mask
        if candidate < p and (candidate or not non_zero):
            return ModP(candidate, p)
        i += 1


def inner_product(a, b) -> ModP:
    """<a, b> mod p.  For p = q (the only case in the proofs) the products and the sum run on the
    GPU (bpmi_sc_dot); any other modulus is summed here."""
    assert len(a) == len(b)
    p = a[0].p
    if p == secp256k1.q:
        raw = _engine.default_engine().sc_dot_bytes(pack_scalars(a, p), pack_scalars(b, p), len(a))
        return ModP(int.from_bytes(raw, "little"), p)
    total = ModP(0, p)
    for u, v in zip(a, b):
        total = total + u * v
    return total
