"""Shared machinery of the single and aggregated range proofs.  The two reference
provers (src/rangeproofs/rangeproof_prover.py, rangeproof_aggreg_prover.py) differ only
in the power of z attached to bit i (z^2 for every bit vs z^(2 + i // n)) and in how the
blinding factors enter taux; likewise the two verifiers.  Scalar algebra is O(n m)
host-side integer work (out of scope for the GPU); every
group operation goes to the engine, fused into as few MSMs as the algebra allows."""
from .. import engine as _engine
from ..ec import Point, pack_points, pack_scalars, unpack_points
from ..innerproduct.inner_product_prover import NIProver
from ..innerproduct.inner_product_verifier import Verifier1
from ..pippenger import PipSECP256k1
from ..utils.commitments import commitment
from ..utils.transcript import Transcript
from ..utils.utils import ModP, inner_product, mod_hash, point_to_b64


class Proof:
    """Range-proof container (reference: src/rangeproofs/rangeproof_verifier.py:10-22)."""

    def __init__(self, taux, mu, t_hat, T1, T2, A, S, innerProof, transcript):
        self.taux, self.mu, self.t_hat = taux, mu, t_hat
        self.T1, self.T2, self.A, self.S = T1, T2, A, S
        self.innerProof, self.transcript = innerProof, transcript


def scaled_generators(hs, y):
    """hsp[i] = y^-i * hs[i] (reference rangeproof_prover.py:77): one batched GPU launch."""
    q = y.p
    yinv = y.inv()
    powers, cur = [], ModP(1, q)
    for _ in hs:
        powers.append(cur)
        cur = cur * yinv
    eng = _engine.default_engine()
    out = eng.ec_mul_batch_bytes(pack_points(hs), pack_scalars(powers, q), len(hs))
    return unpack_points(out, len(hs))


def z_term(z, i, n, aggregated):
    return (z ** (2 + i // n)) * (2 ** (i % n)) if aggregated else (z ** 2) * (2 ** i)


def prove(vs, n, g, h, gs, hs, gammas, u, group, seed, aggregated):
    """NIRangeProver.prove (rangeproof_prover.py:35-112) / AggregNIRangeProver.prove
    (rangeproof_aggreg_prover.py:36-146)."""
    q = group.q
    nm = n * len(vs)
    tr = Transcript(seed)
    aL = []
    for v in vs:
        aL += list(map(int, reversed(bin(v.x)[2:].zfill(n))))[:n]
    aR = [(bit - 1) % q for bit in aL]
    alpha = mod_hash(b"alpha" + tr.digest, q)
    # A = <aL, gs> + <aR, hs> + alpha*h as one MSM
    A = PipSECP256k1.multiexp(gs + hs + [h], aL + aR + [alpha])
    sL = [mod_hash(str(i).encode() + tr.digest, q) for i in range(nm)]
    sR = [mod_hash(str(i).encode() + tr.digest, q) for i in range(nm, 2 * nm)]
    rho = mod_hash(str(2 * n).encode() + tr.digest, q)     # sic: 2*n also when aggregated (:61)
    S = PipSECP256k1.multiexp(gs + hs + [h], sL + sR + [rho])
    tr.add_list_points([A, S])
    y = tr.get_modp(q)
    tr.add_number(y)
    z = tr.get_modp(q)
    tr.add_number(z)
    ypow, cur = [], ModP(1, q)
    for _ in range(nm):
        ypow.append(cur)
        cur = cur * y
    zt = [z_term(z, i, n, aggregated) for i in range(nm)]
    ysr = [ypow[i] * sR[i] for i in range(nm)]
    t1 = inner_product(sL, [ypow[i] * (aR[i] + z) + zt[i] for i in range(nm)]) + inner_product(
        [aL[i] - z for i in range(nm)], ysr)
    t2 = inner_product(sL, ysr)
    tau1 = mod_hash(b"tau1" + tr.digest, q)
    tau2 = mod_hash(b"tau2" + tr.digest, q)
    T1 = commitment(g, h, t1, tau1)
    T2 = commitment(g, h, t2, tau2)
    tr.add_list_points([T1, T2])
    x = tr.get_modp(q)
    tr.add_number(x)
    ls = [aL[i] - z + sL[i] * x for i in range(nm)]
    rs = [ypow[i] * (aR[i] + z + sR[i] * x) + zt[i] for i in range(nm)]
    t_hat = inner_product(ls, rs)
    if aggregated:
        blind = sum([(z ** (2 + j)) * gammas[j] for j in range(len(vs))])
    else:
        blind = (z ** 2) * gammas
    taux = tau2 * (x ** 2) + tau1 * x + blind
    mu = alpha + rho * x
    hsp = scaled_generators(hs, y)
    # P - mu*h = A + x*S + sum(-z)*gs + sum(z*y^i + zt_i)*hsp - mu*h, one MSM
    P_inner = PipSECP256k1.multiexp(
        gs + hsp + [A, S, h],
        [-z for _ in range(nm)] + [(z * ypow[i]) + zt[i] for i in range(nm)] + [1, x, -mu],
    )
    inner = NIProver(gs, hsp, u, P_inner, t_hat, ls, rs, group).prove()
    return Proof(taux, mu, t_hat, T1, T2, A, S, inner, tr.digest)


class VerifierBase:
    def assertThat(self, expr: bool):
        if not expr:
            raise Exception("Proof invalid")

    def verify_transcript(self):
        """A, S, T1, T2 must match the transcript; y, z, x are read from it, not
        re-hashed (reference rangeproof_verifier.py:42-53)."""
        proof = self.proof
        p = proof.taux.p
        items = proof.transcript.split(b"&")
        self.assertThat(items[1] == point_to_b64(proof.A))
        self.assertThat(items[2] == point_to_b64(proof.S))
        self.y = ModP(int(items[3]), p)
        self.z = ModP(int(items[4]), p)
        self.assertThat(items[5] == point_to_b64(proof.T1))
        self.assertThat(items[6] == point_to_b64(proof.T2))
        self.x = ModP(int(items[7]), p)

    def _getP(self, x, y, z, A, S, gs, hsp, n, m=1, aggregated=False, extra_pts=(), extra_sc=()):
        nm = n * m
        ypow, cur = [], ModP(1, y.p)
        for _ in range(nm):
            ypow.append(cur)
            cur = cur * y
        return PipSECP256k1.multiexp(
            gs + hsp + [A, S] + list(extra_pts),
            [-z for _ in range(nm)]
            + [(z * ypow[i]) + z_term(z, i, n, aggregated) for i in range(nm)]
            + [1, x] + list(extra_sc),
        )
