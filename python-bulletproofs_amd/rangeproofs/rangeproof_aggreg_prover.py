"""Aggregated range proof prover (reference: src/rangeproofs/rangeproof_aggreg_prover.py)."""
from typing import List

from ..ec import Point
from ..utils.utils import ModP
from . import common


class AggregNIRangeProver:
    def __init__(self, vs: List[ModP], n: int, g: Point, h: Point, gs: List[Point], hs: List[Point],
                 gammas: List[ModP], u: Point, group, seed: bytes = b""):
        self.vs, self.n, self.g, self.h, self.gs, self.hs = vs, n, g, h, gs, hs
        self.gammas, self.u, self.group, self.seed = gammas, u, group, seed
        self.m = len(vs)

    def prove(self):
        return common.prove(list(self.vs), self.n, self.g, self.h, self.gs, self.hs, list(self.gammas), self.u,
                            self.group, self.seed, aggregated=True)
