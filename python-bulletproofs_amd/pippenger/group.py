"""The operator API Pippenger is written against (reference: src/pippenger/group.py:6-32)."""
from abc import ABC, abstractmethod

from ..ec import Curve, secp256k1
from .modp import ModP


class Group(ABC):
    def __init__(self, unit, order):
        self.unit = unit
        self.order = order

    @abstractmethod
    def mult(self, x, y):
        ...

    def square(self, x):
        return self.mult(x, x)


class MultIntModP(Group):
    def __init__(self, p, order):
        super().__init__(ModP(1, p), order)

    def mult(self, x, y):
        return x * y


class EC(Group):
    def __init__(self, curve: Curve = secp256k1):
        self.curve = curve
        super().__init__(curve.G.IDENTITY_ELEMENT, curve.q)

    def mult(self, x, y):
        return x + y          # one GPU point addition
