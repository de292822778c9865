"""Aggregated range proof verifier (reference: src/rangeproofs/rangeproof_aggreg_verifier.py)."""
from ..ec import Point, secp256k1
from ..innerproduct.inner_product_verifier import Verifier1
from ..pippenger import PipSECP256k1
from ..utils.utils import ModP
from .common import Proof, VerifierBase, verifier_vectors

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
        hsc, yscale, ysum = verifier_vectors(y, z, n, m, True)          # the O(n m) scalars: native host code
        ysum = ModP(ysum, CURVE.q)
        q = CURVE.q
        zp = [1, z.x % q]                                               # z^k for k <= m + 2, one multiplication each (not m + 2 pow() calls)
        for _ in range(m + 1):
            zp.append(zp[-1] * zp[1] % q)
        delta_yz = (z - z ** 2) * ysum - ModP(sum(zp[j + 2] for j in range(1, m + 1)) % q * ((2 ** n - 1) % q), q)
        # hsp[i] = y^-i * hs[i] is never materialised: y^-i goes into the MSM scalars
        # t_hat*g + taux*h == sum z^(j+2) V_j + delta*g + x*T1 + x^2*T2  (reference :82-89), one MSM == identity
        # ... overlapped with the independent MSM for P (:91-106) on the engine's second lane
        p_pts, p_scs = self._getP(x, y, z, proof.A, proof.S, gs, hs, n, m, aggregated=True,
                                  extra_pts=[h], extra_sc=[-proof.mu], h_scale=yscale, terms_only=True, hsc=hsc)
        check, P_inner = PipSECP256k1.multiexp2(
            [g, h] + list(self.Vs) + [proof.T1, proof.T2],
            [proof.t_hat - delta_yz, proof.taux] + [ModP(-zp[j + 2] % q, q) for j in range(m)] + [-x, -(x ** 2)],
            p_pts, p_scs,
        )
        self.assertThat(check == Point.IDENTITY_ELEMENT)
        return Verifier1(gs, hs, self.u, P_inner, proof.t_hat, proof.innerProof, h_scale=yscale).verify()
