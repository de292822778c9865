"""Single range proof prover (reference: src/rangeproofs/rangeproof_prover.py)."""
from typing import List

from ..ec import Point
from ..utils.utils import ModP
from . import common


class NIRangeProver:
    def __init__(self, v: ModP, n: int, g: Point, h: Point, gs: List[Point], hs: List[Point],
                 gamma: ModP, u: Point, group, seed: bytes = b""):
        self.v, self.n, self.g, self.h, self.gs, self.hs = v, n, g, h, gs, hs
        self.gamma, self.u, self.group, self.seed = gamma, u, group, seed

    def prove(self):
        return common.prove([self.v], self.n, self.g, self.h, self.gs, self.hs, self.gamma, self.u,
                            self.group, self.seed, aggregated=False)
