"""A group element for TESTING multi-exponentiation algorithms: an integer modulo p that counts,
in the class attribute `num_of_mult`, how many multiplications have been performed on such
elements since the last `reset()` (reference: src/pippenger/modp.py:1-53).  The reference's
benchmark notebook uses the count to compare Pippenger with the naive product; here the same
count pins the oracle's restatement of the reference algorithm (tests/golden/modp_group.json).

Semantics kept from the reference: mixing with a plain int is allowed and leaves the result
UNREDUCED (only ModP-ModP operations reduce), `**` is square-and-multiply built from counted
multiplications, equality compares value and modulus."""
import operator


class ModP:
    num_of_mult = 0

    def __init__(self, x, p):
        self.x = x
        self.p = p

    @classmethod
    def reset(cls):
        cls.num_of_mult = 0

    # one place for the int / ModP distinction
    def _combine(self, other, op):
        if isinstance(other, int):
            return ModP(op(self.x, other), self.p)
        assert other.p == self.p
        return ModP(op(self.x, other.x) % self.p, self.p)

    def __add__(self, other):
        return self._combine(other, operator.add)

    def __sub__(self, other):
        return self._combine(other, operator.sub)

    def __mul__(self, other):
        ModP.num_of_mult += 1
        return self._combine(other, operator.mul)

    def __pow__(self, exponent):
        result = ModP(self.x, self.p)
        for bit in format(exponent, "b")[1:]:
            result = result * result
            if bit == "1":
                result = result * self
        return result

    def __neg__(self):
        return ModP(self.p - self.x, self.p)

    def __eq__(self, other):
        return (self.x, self.p) == (other.x, other.p)

    def __str__(self):
        return "%d" % self.x

    __repr__ = __str__
