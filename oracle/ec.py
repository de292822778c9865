"""secp256k1 affine group law -- restatement of the `fastecdsa` surface the
reference touches (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference delegates all EC arithmetic to the third-party C extension
`fastecdsa` (GMP), which is not in /root/reference and not installable here.
Call sites that define the surface restated below:
  Point(x, y, curve), .x .y .curve      src/utils/utils.py:100-131
  Point + Point                          src/pippenger/group.py:31-32
  int * Point (-> Point.__rmul__)        src/utils/utils.py:43-44
  Point == Point                         src/innerproduct/inner_product_verifier.py:145
  Point.IDENTITY_ELEMENT                 src/pippenger/group.py:29, src/utils/utils.py:102
  Curve.p .a .b .q .G .is_point_on_curve src/utils/elliptic_curve_hash.py:8-21
  mod_sqrt(a, p) -> (r, p - r)           src/utils/utils.py:127

Published algorithm restated: SEC 1 v2 section 2.2.1 (affine chord-and-tangent
law on y^2 = x^3 + 7 over F_p) with curve constants from SEC 2 v2 section 2.4.1.
"""


class Curve:
    def __init__(self, name, p, a, b, q, gx, gy):
        self.name = name
        self.p = p
        self.a = a
        self.b = b
        self.q = q
        self.gx = gx
        self.gy = gy

    @property
    def G(self):
        return Point(self.gx, self.gy, self)

    def is_point_on_curve(self, pt):
        x, y = pt
        return (y * y - (x * x * x + self.a * x + self.b)) % self.p == 0

    def __repr__(self):
        return self.name


secp256k1 = Curve(
    "secp256k1",
    p=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F,
    a=0,
    b=7,
    q=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    gx=0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    gy=0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8,
)

P_FIELD = secp256k1.p
Q_ORDER = secp256k1.q


class Point:
    """Affine point; the identity is (0, 0, curve=None).  (0, 0) is never on
    y^2 = x^3 + 7 because 7 is a quadratic non-residue mod p."""

    __slots__ = ("x", "y", "curve")
    IDENTITY_ELEMENT = None  # set below

    def __init__(self, x, y, curve=secp256k1):
        if curve is None:
            assert x == 0 and y == 0
        else:
            if not curve.is_point_on_curve((x, y)):
                raise ValueError("point not on curve")
        self.x = x
        self.y = y
        self.curve = curve

    def is_identity(self):
        return self.curve is None

    def __eq__(self, other):
        return (
            isinstance(other, Point)
            and self.x == other.x
            and self.y == other.y
            and (self.curve is None) == (other.curve is None)
        )

    def __hash__(self):
        return hash((self.x, self.y))

    def __neg__(self):
        if self.curve is None:
            return self
        return _mk(self.x, (-self.y) % self.curve.p, self.curve)

    def __add__(self, other):
        if self.curve is None:
            return other
        if other.curve is None:
            return self
        p = self.curve.p
        x1, y1, x2, y2 = self.x, self.y, other.x, other.y
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return Point.IDENTITY_ELEMENT
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        y3 = (lam * (x1 - x3) - y1) % p
        return _mk(x3, y3, self.curve)

    def __sub__(self, other):
        return self + (-other)

    def __mul__(self, k):
        k = int(k)
        if self.curve is None:
            return self
        k %= self.curve.q
        return _from_jac(_jac_mul(self.x, self.y, k, self.curve.p), self.curve)

    __rmul__ = __mul__

    def __repr__(self):
        if self.curve is None:
            return "Point(inf)"
        return "Point(%x, %x)" % (self.x, self.y)


def _mk(x, y, curve):
    pt = Point.__new__(Point)
    pt.x = x
    pt.y = y
    pt.curve = curve
    return pt


Point.IDENTITY_ELEMENT = _mk(0, 0, None)
INF = Point.IDENTITY_ELEMENT


# --- Jacobian helpers (internal speed-up for scalar multiplication; the result
# is converted back to the canonical affine form, so it is value-identical to
# repeated affine addition) -------------------------------------------------
def _jac_dbl(X, Y, Z, p):
    if Z == 0 or Y == 0:
        return (0, 1, 0)
    A = X * X % p
    B = Y * Y % p
    C = B * B % p
    D = 2 * ((X + B) * (X + B) - A - C) % p
    E = 3 * A % p
    X3 = (E * E - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def _jac_add_affine(X1, Y1, Z1, x2, y2, p):
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % p
    U2 = x2 * Z1Z1 % p
    S2 = y2 * Z1 * Z1Z1 % p
    H = (U2 - X1) % p
    R = (S2 - Y1) % p
    if H == 0:
        if R == 0:
            return _jac_dbl(X1, Y1, Z1, p)
        return (0, 1, 0)
    HH = H * H % p
    HHH = H * HH % p
    V = X1 * HH % p
    X3 = (R * R - HHH - 2 * V) % p
    Y3 = (R * (V - X3) - Y1 * HHH) % p
    Z3 = Z1 * H % p
    return (X3, Y3, Z3)


def _jac_mul(x, y, k, p):
    acc = (0, 1, 0)
    for bit in bin(k)[2:] if k else "":
        acc = _jac_dbl(acc[0], acc[1], acc[2], p)
        if bit == "1":
            acc = _jac_add_affine(acc[0], acc[1], acc[2], x, y, p)
    return acc


def _from_jac(J, curve):
    X, Y, Z = J
    if Z == 0:
        return Point.IDENTITY_ELEMENT
    p = curve.p
    zi = pow(Z, -1, p)
    zi2 = zi * zi % p
    return _mk(X * zi2 % p, Y * zi2 * zi % p, curve)


def mod_sqrt(a, p):
    """(r, p - r) with r = a^((p+1)/4) mod p (p = 3 mod 4).  Which root comes
    first only matters for generator derivation (inputs), src/utils/utils.py:127,
    src/utils/elliptic_curve_hash.py:19."""
    assert p % 4 == 3
    r = pow(a, (p + 1) // 4, p)
    return (r, p - r)


# --- byte layout shared with the C-ABI (include/bpmi.h) ----------------------
def point_to_le64(pt):
    """Affine point -> 64 B: x (32 B little-endian) || y (32 B LE); inf = zeros."""
    return pt.x.to_bytes(32, "little") + pt.y.to_bytes(32, "little")


def point_from_le64(b):
    x = int.from_bytes(b[:32], "little")
    y = int.from_bytes(b[32:64], "little")
    if x == 0 and y == 0:
        return Point.IDENTITY_ELEMENT
    return Point(x, y, secp256k1)
