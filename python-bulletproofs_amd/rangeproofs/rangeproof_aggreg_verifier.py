"""Aggregated range proof verifier (reference: src/rangeproofs/rangeproof_aggreg_verifier.py)."""
from ..ec import secp256k1
from ..innerproduct.inner_product_verifier import Verifier1
from ..pippenger import PipSECP256k1
from ..utils.utils import ModP
from .common import Proof, VerifierBase, scaled_generators, _powers

CURVE = secp256k1


class AggregRangeVerifier(VerifierBase):
    def __init__(self, Vs, g, h, gs, hs, u, proof: Proof):
        self.Vs, self.g, self.h, self.gs, self.hs, self.u, self.proof = Vs, g, h, gs, hs, u, proof

    def verify(self):
        self.verify_transcript()
        g, h, gs, hs, x, y, z, proof = self.g, self.h, self.gs, self.hs, self.x, self.y, self.z, self.proof
        nm = len(gs)
        m = len(self.Vs)
        n = nm // m
        ysum = ModP(sum(_powers(y.x, nm, CURVE.q)) % CURVE.q, CURVE.q)
        delta_yz = (z - z ** 2) * ysum - sum([(z ** (j + 2)) * ModP(2 ** n - 1, CURVE.q) for j in range(1, m + 1)])
        hsp = scaled_generators(hs, y)
        lhs = PipSECP256k1.multiexp([g, h], [proof.t_hat, proof.taux])
        rhs = PipSECP256k1.multiexp(
            list(self.Vs) + [g, proof.T1, proof.T2],
            [z ** (j + 2) for j in range(m)] + [delta_yz, x, x ** 2],
        )
        self.assertThat(lhs == rhs)
        P_inner = self._getP(x, y, z, proof.A, proof.S, gs, hsp, n, m, aggregated=True,
                             extra_pts=[h], extra_sc=[-proof.mu])
        return Verifier1(gs, hsp, self.u, P_inner, proof.t_hat, proof.innerProof).verify()
