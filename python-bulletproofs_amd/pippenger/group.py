"""Groups the multi-exponentiation is generic over.

The reference's Pippenger is written against a three-member operator interface -- a neutral
element `unit`, the group `order`, and a binary operation `mult` with `square(x) = mult(x, x)`
(/root/reference/src/pippenger/group.py:6-32).  The same interface is kept here so that
`Pippenger(some_group)` stays constructible for any group:

  * EC(curve)            points of secp256k1 under addition.  Pippenger recognises this group and
                         hands the whole sum to the GPU engine; `mult` on its own is one device
                         point addition.
  * MultIntModP(p, q)    the multiplicative group of integers mod p (the reference's test group,
                         elements are the multiplication-counting pippenger.modp.ModP).
  * any subclass of Group providing `mult` runs on the generic host path of Pippenger.
"""
from ..ec import secp256k1
from .modp import ModP as _CountingModP


class Group:
    """Base class: subclasses supply `mult`; `unit` and `order` are plain attributes."""

    def __init__(self, unit, order):
        self.unit, self.order = unit, order

    def mult(self, x, y):
        raise NotImplementedError("%s does not define its group operation" % type(self).__name__)

    def square(self, x):
        return self.mult(x, x)

    def power(self, x, k):
        """x combined with itself k times (k >= 0) by square-and-multiply; convenience for tests."""
        result, base = self.unit, x
        while k:
            if k & 1:
                result = self.mult(result, base)
            base = self.square(base)
            k >>= 1
        return result


class EC(Group):
    """Elliptic-curve points, written additively: mult(P, Q) = P + Q."""

    def __init__(self, curve=secp256k1):
        Group.__init__(self, curve.G.IDENTITY_ELEMENT, curve.q)
        self.curve = curve

    def mult(self, P, Q):
        return P + Q


class MultIntModP(Group):
    """Integers mod p under multiplication; `order` is the order of the subgroup in use."""

    def __init__(self, p, order):
        Group.__init__(self, _CountingModP(1, p), order)
        self.p = p

    def mult(self, a, b):
        return a * b
