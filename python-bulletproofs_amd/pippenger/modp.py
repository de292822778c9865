"""Counting test-group element (reference: src/pippenger/modp.py:1-53): an integer mod p
whose class-level `num_of_mult` counts every multiplication."""


class ModP:
    num_of_mult = 0

    @classmethod
    def reset(cls):
        cls.num_of_mult = 0

    def __init__(self, x, p):
        self.x, self.p = x, p

    def _other(self, y):
        if isinstance(y, int):
            return y, False
        assert self.p == y.p
        return y.x, True

    def __add__(self, y):
        v, red = self._other(y)
        return ModP((self.x + v) % self.p if red else self.x + v, self.p)

    def __sub__(self, y):
        v, red = self._other(y)
        return ModP((self.x - v) % self.p if red else self.x - v, self.p)

    def __mul__(self, y):
        type(self).num_of_mult += 1
        v, red = self._other(y)
        return ModP((self.x * v) % self.p if red else self.x * v, self.p)

    def __pow__(self, n):
        acc = ModP(self.x, self.p)
        for bit in bin(n)[3:]:          # square-and-multiply so the counter sees each mult
            acc = acc * acc
            if bit == "1":
                acc = acc * self
        return acc

    def __neg__(self):
        return ModP(self.p - self.x, self.p)

    def __eq__(self, y):
        return self.x == y.x and self.p == y.p

    def __repr__(self):
        return str(self.x)

    __str__ = __repr__
