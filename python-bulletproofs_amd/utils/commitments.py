"""Pedersen commitments (reference: src/utils/commitments.py:5-13)."""
from ..ec import Point, pack_points, pack_scalars
from ..pippenger import PipSECP256k1
from .. import engine as _engine


def commitment(g, h, x, r):
    """x*g + r*h -- one 2-term MSM on the GPU instead of two scalar mults and an add."""
    eng = _engine.default_engine()
    return Point.from_le64(eng.msm_bytes(pack_points([g, h]), pack_scalars([x, r]), 2))


def vector_commitment(g, h, a, b):
    assert len(g) == len(h) == len(a) == len(b)
    return PipSECP256k1.multiexp(g + h, a + b)
