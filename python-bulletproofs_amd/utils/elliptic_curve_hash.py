"""Hash-to-curve by try-and-increment (reference: src/utils/elliptic_curve_hash.py:7-23).
Generator derivation only (inputs); host integers."""
from hashlib import md5, sha256

from ..ec import Curve, Point, mod_sqrt, secp256k1


def elliptic_hash(msg: bytes, CURVE: Curve = secp256k1):
    p = CURVE.p
    i = 0
    while True:
        i += 1
        pre = str(i).encode() + msg
        x = int.from_bytes(sha256(pre).digest(), "big")
        if x >= p:
            continue
        y = mod_sqrt((x**3 + CURVE.a * x + CURVE.b) % p, p)[0]
        if CURVE.is_point_on_curve((x, y)):
            odd = int(md5(pre).hexdigest(), 16) % 2
            return Point(x, y, CURVE) if odd else Point(x, p - y, CURVE)
