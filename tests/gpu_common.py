"""Helpers for the `-m gpu` parity tests: everything goes through the product package
(reference-shaped API -> ctypes C-ABI -> HIP) and is compared with the oracle."""
import random

import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.ec import Point as GP
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.utils.utils import ModP as GModP

from helpers import Q
from oracle import cbind
from oracle.ec import INF, secp256k1

G = secp256k1.G


def to_gpu(p):
    """oracle Point -> product Point"""
    return GP._raw(p.x, p.y)


def to_gpu_list(ps):
    return [to_gpu(p) for p in ps]


def same_point(gp, op):
    return gp.x == op.x and gp.y == op.y


def gsc(v):
    """oracle Zq / int -> product ModP"""
    return GModP(int(v.x) if hasattr(v, "x") else int(v), Q)


def rand_points(n, seed):
    """n distinct valid points k_i * G, k_i random (built by the C oracle)."""
    rnd = random.Random(seed)
    ks = [rnd.randrange(1, Q) for _ in range(n)]
    return cbind.ec_mul_batch([G] * n, ks), ks


def engine():
    return default_engine()
