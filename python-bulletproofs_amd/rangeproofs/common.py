"""Shared machinery of the single and aggregated range proofs.  The two reference
provers (src/rangeproofs/rangeproof_prover.py, rangeproof_aggreg_prover.py) differ only
in the power of z attached to bit i (z^2 for every bit vs z^(2 + i // n)) and in how the
blinding factors enter taux; likewise the two verifiers.  Scalar algebra is O(n m)
host-side integer work (out of scope for the GPU); every
group operation goes to the engine, fused into as few MSMs as the algebra allows."""
from hashlib import sha256

from .. import engine as _engine
from ..ec import PackedPoints, PackedScalars, pack_points, secp256k1, unpack_points
from ..innerproduct.inner_product_prover import NIProver
from ..pippenger import PipSECP256k1
from ..utils.transcript import Transcript
from ..utils.utils import ModP, mod_hash, point_to_b64


class Proof:
    """Range-proof container (reference: src/rangeproofs/rangeproof_verifier.py:10-22)."""

    def __init__(self, taux, mu, t_hat, T1, T2, A, S, innerProof, transcript):
        self.taux, self.mu, self.t_hat = taux, mu, t_hat
        self.T1, self.T2, self.A, self.S = T1, T2, A, S
        self.innerProof, self.transcript = innerProof, transcript


def scaled_generators(hs, y):
    """hsp[i] = y^-i * hs[i] (reference rangeproof_prover.py:77): one batched GPU launch."""
    q = y.p
    yinv = pow(y.x, -1, q)
    powers, cur = [], 1
    for _ in hs:
        powers.append(cur)
        cur = cur * yinv % q
    eng = _engine.default_engine()
    out = eng.ec_mul_batch_bytes(pack_points(hs), b"".join(v.to_bytes(32, "little") for v in powers), len(hs))
    return unpack_points(out, len(hs))


def z_term(z, i, n, aggregated):
    return (z ** (2 + i // n)) * (2 ** (i % n)) if aggregated else (z ** 2) * (2 ** i)


def _z_terms(z, n, m, q, aggregated):
    """[z^(2 + i//n) * 2^(i%n) mod q for i < n*m] (aggregated) or [z^2 * 2^i] (single, m = 1)."""
    out = []
    for j in range(m):
        cur = pow(z, 2 + j, q) if aggregated else z * z % q
        for _ in range(n):
            out.append(cur)
            cur = cur * 2 % q
    return out


def _powers(y, count, q):
    out, cur = [], 1
    for _ in range(count):
        out.append(cur)
        cur = cur * y % q
    return out


def _mod_hash_ints(lo, hi, digest, q):
    """[mod_hash(str(i) + digest, q).x for i in range(lo, hi)] as a PackedScalars: for the curve's order the hashes run in
    native code (bpmi_mod_hash_range) and arrive as the bytes the MSM wants; any other modulus takes the Python loop."""
    if q == secp256k1.q and hi > lo:
        import ctypes
        import os
        from .. import _native
        out = ctypes.create_string_buffer(32 * (hi - lo))
        threads = _threads(hi - lo)
        if _native.load().bpmi_mod_hash_range(digest, len(digest), lo, hi, threads, out) != 0:
            raise RuntimeError("bpmi_mod_hash_range failed")
        return PackedScalars.from_bytes(out.raw)
    mask = (1 << q.bit_length()) - 1
    out = []
    for i in range(lo, hi):
        c = int.from_bytes(sha256(b"1%d" % i + digest).digest(), "big") & mask
        out.append(c if 0 < c < q else mod_hash(b"%d" % i + digest, q).x)
    return out


_BIT_LE = ((0).to_bytes(32, "little"), (1).to_bytes(32, "little"))
_MINUS1_LE = (secp256k1.q - 1).to_bytes(32, "little")
_ASCII_BITS = bytes(range(256)).translate(bytes(c & 1 for c in range(256)))      # b"0" / b"1" -> 0 / 1


def _threads(count):
    """Host threads for the native O(n m) loops: one per 1 024 elements, at most BPMI_HOST_THREADS (default 8) and the CPUs of the process."""
    import os
    try:
        cap = int(os.environ.get("BPMI_HOST_THREADS", "8") or 8)
    except ValueError:                       # a malformed setting must not fail a proof
        cap = 8
    return max(1, min(cap, len(os.sched_getaffinity(0)), count // 1024))


def _le32(v, q):
    return (int(v) % q).to_bytes(32, "little")


def _native_prove(vs, n, g, h, gs, hs, gammas, u, group, seed, aggregated):
    """prove() for the curve's own order with the O(n m) algebra in native host code (csrc/rp_algebra_host.hpp:
    bpmi_rp_poly_coeffs / bpmi_rp_final_vectors) and every vector kept as the bytes the MSMs want -- no Python loop over the
    n m elements is left.  Same transcript, same proof (tests/test_rp_algebra_cpu.py, the golden tests)."""
    import ctypes
    from .. import _native
    lib = _native.load()
    q = group.q
    m = len(vs)
    nm = n * m
    thr = _threads(nm)
    tr = Transcript(seed)
    # aL: the n low bits of every value, least significant first (:40-47); one byte per bit for the native code
    bits = b"".join(bin(v.x)[2:].zfill(n)[::-1][:n].encode() for v in vs).translate(_ASCII_BITS)
    aL = PackedScalars.from_bytes(b"".join(map(_BIT_LE.__getitem__, bits)))
    aR = PackedScalars.from_bytes(b"".join(map((_MINUS1_LE, _BIT_LE[0]).__getitem__, bits)))
    alpha = mod_hash(b"alpha" + tr.digest, q).x
    sLR = _mod_hash_ints(0, 2 * nm, tr.digest, q)        # one native call (one set of threads) for both blinding vectors
    sL, sR = PackedScalars.from_bytes(sLR.packed[:32 * nm]), PackedScalars.from_bytes(sLR.packed[32 * nm:])
    rho = mod_hash(str(2 * n).encode() + tr.digest, q).x     # sic: 2*n also when aggregated (:61)
    gs, hs = PackedPoints(gs) if not isinstance(gs, PackedPoints) else gs, PackedPoints(hs) if not isinstance(hs, PackedPoints) else hs
    base = PackedPoints.join(gs, hs, [h])
    A, S = PipSECP256k1.multiexp2(base, PackedScalars.join(aL, aR, [alpha]), base, PackedScalars.join(sL, sR, [rho]))
    tr.add_list_points([A, S])
    yM = tr.get_modp(q)
    tr.add_number(yM)
    zM = tr.get_modp(q)
    tr.add_number(zM)
    y, z = yM.x, zM.x
    yb, zb = _le32(y, q), _le32(z, q)
    t1b, t2b = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    if lib.bpmi_rp_poly_coeffs(n, m, 1 if aggregated else 0, bits, sL.packed, sR.packed, yb, zb, thr, t1b, t2b) != 0:
        raise RuntimeError("bpmi_rp_poly_coeffs failed")
    t1, t2 = int.from_bytes(t1b.raw, "little"), int.from_bytes(t2b.raw, "little")
    tau1 = mod_hash(b"tau1" + tr.digest, q).x
    tau2 = mod_hash(b"tau2" + tr.digest, q).x
    T1, T2 = PipSECP256k1.multiexp2([g, h], [t1, tau1], [g, h], [t2, tau2])
    tr.add_list_points([T1, T2])
    xM = tr.get_modp(q)
    tr.add_number(xM)
    x = xM.x
    ls, rs = ctypes.create_string_buffer(32 * nm), ctypes.create_string_buffer(32 * nm)
    hsc, ysc, thb = ctypes.create_string_buffer(32 * nm), ctypes.create_string_buffer(32 * nm), ctypes.create_string_buffer(32)
    if lib.bpmi_rp_final_vectors(n, m, 1 if aggregated else 0, bits, sL.packed, sR.packed, yb, zb, _le32(x, q), thr, ls, rs, thb, hsc, ysc) != 0:
        raise RuntimeError("bpmi_rp_final_vectors failed")
    t_hat = int.from_bytes(thb.raw, "little")
    if aggregated:
        blind = sum(pow(z, 2 + j, q) * gammas[j].x for j in range(m))
    else:
        blind = z * z * gammas.x
    taux = (tau2 * x * x + tau1 * x + blind) % q
    mu = (alpha + rho * x) % q
    yscale = PackedScalars.from_bytes(ysc.raw)
    P_inner = PipSECP256k1.multiexp(
        PackedPoints.join(gs, hs, [A, S, h]),
        PackedScalars.join(PackedScalars.from_bytes(_le32(-z, q) * nm), PackedScalars.from_bytes(hsc.raw), [1, x, (-mu) % q]),
    )
    inner = NIProver(gs, hs, u, P_inner, ModP(t_hat, q), PackedScalars.from_bytes(ls.raw), PackedScalars.from_bytes(rs.raw), group,
                     h_scale=yscale).prove()
    return Proof(ModP(taux, q), ModP(mu, q), ModP(t_hat, q), T1, T2, A, S, inner, tr.digest)


def verifier_vectors(y, z, n, m, aggregated):
    """(hsc, yscale, ysum): hsc_i = (z y^i + zt_i) y^-i and yscale_i = y^-i as PackedScalars, ysum = sum_{i < n m} y^i -- the
    O(n m) part of RangeVerifier.verify / AggregRangeVerifier.verify; native for the curve's own order."""
    q = y.p
    nm = n * m
    if q == secp256k1.q and y.x % q:
        import ctypes
        from .. import _native
        hsc, ysc, ysum = ctypes.create_string_buffer(32 * nm), ctypes.create_string_buffer(32 * nm), ctypes.create_string_buffer(32)
        if _native.load().bpmi_rp_verifier_vectors(n, m, 1 if aggregated else 0, _le32(y.x, q), _le32(z.x, q), _threads(nm), hsc, ysc, ysum) != 0:
            raise RuntimeError("bpmi_rp_verifier_vectors failed")
        return PackedScalars.from_bytes(hsc.raw), PackedScalars.from_bytes(ysc.raw), int.from_bytes(ysum.raw, "little")
    ypow, zt = _powers(y.x, nm, q), _z_terms(z.x, n, m, q, aggregated)
    yscale = _powers(pow(y.x, -1, q), nm, q)
    return [(z.x * ypow[i] + zt[i]) * yscale[i] % q for i in range(nm)], yscale, sum(ypow) % q


def prove(vs, n, g, h, gs, hs, gammas, u, group, seed, aggregated):
    """NIRangeProver.prove (rangeproof_prover.py:35-112) / AggregNIRangeProver.prove
    (rangeproof_aggreg_prover.py:36-146).  The O(n m) scalar algebra runs on plain Python
    integers mod q (every value the reference exposes -- transcript numbers, taux, mu, t_hat,
    a, b -- is reduced there as well, so the results are identical); ModP objects appear
    only where the reference's surface shows them."""
    q = group.q
    if q == secp256k1.q:
        return _native_prove(vs, n, g, h, gs, hs, gammas, u, group, seed, aggregated)
    m = len(vs)
    nm = n * m
    tr = Transcript(seed)
    aL = []
    for v in vs:
        aL += list(map(int, reversed(bin(v.x)[2:].zfill(n))))[:n]
    aR = [(bit - 1) % q for bit in aL]
    if q == secp256k1.q:                                              # bits and bits - 1: two constant encodings each
        minus1 = (q - 1).to_bytes(32, "little")
        aL = PackedScalars(aL, b"".join([_BIT_LE[bit] for bit in aL]))
        aR = PackedScalars(aR, b"".join([minus1 if not bit else _BIT_LE[0] for bit in aL]))
    alpha = mod_hash(b"alpha" + tr.digest, q).x
    sL = _mod_hash_ints(0, nm, tr.digest, q)
    sR = _mod_hash_ints(nm, 2 * nm, tr.digest, q)
    rho = mod_hash(str(2 * n).encode() + tr.digest, q).x     # sic: 2*n also when aggregated (:61)
    # A = <aL, gs> + <aR, hs> + alpha*h and S = <sL, gs> + <sR, hs> + rho*h: two MSMs over the same
    # points, independent of each other, overlapped on the engine's two lanes
    gs, hs = PackedPoints(gs), PackedPoints(hs)                      # packed once for the four MSMs and the argument below
    base = PackedPoints.join(gs, hs, [h])
    A, S = PipSECP256k1.multiexp2(base, PackedScalars.join(aL, aR, [alpha]), base, PackedScalars.join(sL, sR, [rho]))
    tr.add_list_points([A, S])
    yM = tr.get_modp(q)
    tr.add_number(yM)
    zM = tr.get_modp(q)
    tr.add_number(zM)
    y, z = yM.x, zM.x
    ypow = _powers(y, nm, q)
    zt = _z_terms(z, n, m, q, aggregated)
    ysr = [ypow[i] * sR[i] % q for i in range(nm)]
    # _get_polynomial_coeffs (:93-101 / aggreg :117-130)
    t1 = (sum(sL[i] * (ypow[i] * (aR[i] + z) + zt[i]) for i in range(nm))
          + sum((aL[i] - z) * ysr[i] for i in range(nm))) % q
    t2 = sum(sL[i] * ysr[i] for i in range(nm)) % q
    tau1 = mod_hash(b"tau1" + tr.digest, q).x
    tau2 = mod_hash(b"tau2" + tr.digest, q).x
    T1, T2 = PipSECP256k1.multiexp2([g, h], [t1, tau1], [g, h], [t2, tau2])      # = commitment(g, h, t_i, tau_i), as a pair
    tr.add_list_points([T1, T2])
    xM = tr.get_modp(q)
    tr.add_number(xM)
    x = xM.x
    # _final_compute (:103-112 / aggreg :132-146)
    ls = [(aL[i] - z + sL[i] * x) % q for i in range(nm)]
    rs = [(ypow[i] * (aR[i] + z + sR[i] * x) + zt[i]) % q for i in range(nm)]
    t_hat = sum(ls[i] * rs[i] for i in range(nm)) % q
    if aggregated:
        blind = sum(pow(z, 2 + j, q) * gammas[j].x for j in range(m))
    else:
        blind = z * z * gammas.x
    taux = (tau2 * x * x + tau1 * x + blind) % q
    mu = (alpha + rho * x) % q
    # hsp[i] = y^-i * hs[i] (:77) is never materialised: y^-i goes into the scalars, both in
    # P - mu*h = A + x*S + sum(-z)*gs + sum(z*y^i + zt_i)*hsp - mu*h (one MSM) and in the IPA
    yscale = _powers(pow(y, -1, q), nm, q)
    P_inner = PipSECP256k1.multiexp(
        PackedPoints.join(gs, hs, [A, S, h]),
        [-z] * nm + [(z * ypow[i] + zt[i]) * yscale[i] % q for i in range(nm)] + [1, x, -mu],
    )
    inner = NIProver(gs, hs, u, P_inner, ModP(t_hat, q), ls, rs, group, h_scale=yscale).prove()
    return Proof(ModP(taux, q), ModP(mu, q), ModP(t_hat, q), T1, T2, A, S, inner, tr.digest)


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

    def _getP(self, x, y, z, A, S, gs, hsp, n, m=1, aggregated=False, extra_pts=(), extra_sc=(), h_scale=None, terms_only=False, hsc=None):
        """A + x*S + sum(-z)*gs_i + sum(z*y^i + zt_i)*hsp_i (+ extras) as one MSM.  With
        h_scale = [y^-i] the list `hsp` holds the UNSCALED hs and the factors ride in the
        scalars (hsc: those scalars already computed, see verifier_vectors)."""
        q = y.p
        nm = n * m
        zi = z.x
        if hsc is None:
            ypow, zt = _powers(y.x, nm, q), _z_terms(z.x, n, m, q, aggregated)
            hsc = [zi * ypow[i] + zt[i] for i in range(nm)]
            if h_scale is not None:
                hsc = [v * c % q for v, c in zip(hsc, h_scale)]
        pts = PackedPoints.join(gs, hsp, [A, S], list(extra_pts))
        scs = PackedScalars.join(PackedScalars.from_bytes(_le32(-zi, q) * nm) if q == secp256k1.q else [-zi] * nm, hsc,
                                 [1, x] + list(extra_sc))
        if terms_only:
            return pts, scs
        return PipSECP256k1.multiexp(pts, scs)
