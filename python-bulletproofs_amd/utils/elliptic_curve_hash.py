"""Deriving curve points from byte strings ("nothing up my sleeve" generators).

Try-and-increment exactly as the reference does it (/root/reference/src/utils/
elliptic_curve_hash.py:7-23), because the generators are INPUTS of every golden vector: the
counter i = 1, 2, ... is prefixed in decimal, SHA-256 gives the candidate x, the first candidate
below p whose x^3 + ax + b is a square wins, and the parity of MD5(prefix) picks which of the
two square roots is y (root r = rhs^((p+1)/4) when the parity is odd, p - r otherwise).
Host integers only; this is set-up work, not part of the accelerated path."""
import hashlib
from itertools import count

from ..ec import Point, mod_sqrt, secp256k1


def _candidates(msg):
    for i in count(1):
        tagged = b"%d" % i + msg
        yield tagged, int.from_bytes(hashlib.sha256(tagged).digest(), "big")


def elliptic_hash(msg: bytes, CURVE=secp256k1):
    p, a, b = CURVE.p, CURVE.a, CURVE.b
    for tagged, x in _candidates(msg):
        if x >= p:
            continue
        root = mod_sqrt((x * x * x + a * x + b) % p, p)[0]
        if not CURVE.is_point_on_curve((x, root)):
            continue                                   # the right-hand side was not a square
        keep_root = int(hashlib.md5(tagged).hexdigest(), 16) & 1
        return Point(x, root if keep_root else p - root, CURVE)
