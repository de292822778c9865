"""The operator API of the reference (`Group`, `MultIntModP`, the counting `ModP`)
through the PRODUCT's Pippenger on a non-EC group: results against pow(), and the number of group
operations against the counts the reference's own code produced (tests/golden/modp_group.json).
Reference: /root/reference/src/pippenger/pippenger.py:22-94, group.py:19-24, modp.py:3-19.  CPU only."""
import random

import pytest

import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.pippenger.group import Group, MultIntModP
from bulletproofs_amd.pippenger.modp import ModP
from bulletproofs_amd.pippenger.pippenger import Pippenger

from conftest import load_golden


def product_mod(gs, es, p):
    want = 1
    for g, e in zip(gs, es):
        want = want * pow(g, e, p) % p
    return want


def test_generic_group_matches_reference_results_and_operation_counts():
    for c in load_golden("modp_group.json")["cases"]:
        p, n = c["p"], c["n"]
        pip = Pippenger(MultIntModP(p, p - 1))
        gs = [ModP(2 + 3 * i, p) for i in range(n)]                  # same inputs as tests/golden/make_golden.py
        es = [(12345 * (i + 1) ** 3) % (p - 1) for i in range(n)]
        ModP.reset()
        r = pip.multiexp(gs, es)
        assert r.x == c["result"] == product_mod([g.x for g in gs], es, p)
        assert ModP.num_of_mult == c["num_of_mult"]                  # the reference's schedule, multiplication for multiplication


@pytest.mark.parametrize("p", [101, 65537, 2 ** 61 - 1])
def test_generic_group_against_pow(p):
    rnd = random.Random(p)
    pip = Pippenger(MultIntModP(p, p - 1))
    for n in (1, 2, 3, 7, 33, 100):
        gs = [rnd.randrange(1, p) for _ in range(n)]
        es = [rnd.randrange(p - 1) for _ in range(n)]
        es[0] = 0
        if n > 2:
            es[1] = p - 1 + 5            # >= order: reduced like the reference (pippenger.py:26)
            es[2] = -3                   # negative: Python's % makes it order - 3
        r = pip.multiexp([ModP(g, p) for g in gs], es)
        assert r.x == product_mod(gs, [e % (p - 1) for e in es], p)


def test_generic_group_edge_cases():
    p = 1000003
    pip = Pippenger(MultIntModP(p, p - 1))
    assert pip.multiexp([], []) == ModP(1, p)                        # N = 0 -> unit (pippenger.py:28-29)
    assert pip.multiexp([ModP(5, p)] * 3, [0, 0, 0]).x == 1
    with pytest.raises(Exception, match="Different number of group elements and exponents"):
        pip.multiexp([ModP(5, p)], [1, 2])
    assert pip.lamb == (p - 1).bit_length() and pip.order == p - 1


def test_generic_path_is_the_operator_api():
    """Any Group subclass that supplies `mult` works: here the additive group of integers mod m."""
    class AddMod(Group):
        def __init__(self, m):
            Group.__init__(self, 0, m)
            self.m = m

        def mult(self, a, b):
            return (a + b) % self.m

    m = 10007
    pip = Pippenger(AddMod(m))
    rnd = random.Random(3)
    gs = [rnd.randrange(m) for _ in range(20)]
    es = [rnd.randrange(m) for _ in range(20)]
    assert pip.multiexp(gs, es) == sum(g * e for g, e in zip(gs, es)) % m
