"""CPU restatement of the reference's hot path and its callers, in plain Python
(TEST INFRASTRUCTURE -- see oracle/__init__.py; never imported by the product).

Every function cites the reference file:line it follows (paths relative to
/root/reference/).  Scalars are carried as `Zq` objects that reproduce the
reference `ModP`'s observable semantics, including its quirks (int operands are
not reduced on + and *, negation of 0 gives p), because proof fields and
transcript bytes depend on them.  The restatement is pinned against vectors
produced by the reference's own code in tests/golden/ (tests/test_oracle_golden.py).
"""

import base64
from hashlib import md5, sha256
from math import floor, isqrt, log2
from types import SimpleNamespace

from .ec import INF, Point, mod_sqrt, secp256k1

CURVE = secp256k1
Q = CURVE.q
BYTE_LENGTH = Q.bit_length() // 8  # src/utils/utils.py:12


# --------------------------------------------------------------------------
# scalar field element -- src/utils/utils.py:24-81
# --------------------------------------------------------------------------
class Zq:
    """Observable behaviour of the reference `ModP` (src/utils/utils.py:24-81).

    value-with-int:   +  and * leave the result unreduced (:30-31, :41-42);
                      -  reduces (:49-50)
    value-with-value: + - * reduce (:33, :46, :52)
    value * Point:    EC scalar multiplication by .x (:43-44)
    -v:               p - x, so -0 is p, not 0 (:63-64)
    v % m:            plain int (:60-61)   -- what Pippenger.multiexp relies on
    v.inv():          the inverse in [0, p); Exception text on failure (:66-72)
    """

    __slots__ = ("x", "p")

    def __init__(self, x, p=Q):
        self.x = x
        self.p = p

    def _same(self, o):
        assert self.p == o.p
        return o.x

    def __add__(self, o):
        if isinstance(o, int):
            return Zq(self.x + o, self.p)
        return Zq((self.x + self._same(o)) % self.p, self.p)

    __radd__ = __add__

    def __mul__(self, o):
        if isinstance(o, int):
            return Zq(self.x * o, self.p)
        if isinstance(o, Point):
            return self.x * o
        return Zq((self.x * self._same(o)) % self.p, self.p)

    def __sub__(self, o):
        if isinstance(o, int):
            return Zq((self.x - o) % self.p, self.p)
        return Zq((self.x - self._same(o)) % self.p, self.p)

    def __rsub__(self, o):
        return -(self - o)

    def __neg__(self):
        return Zq(self.p - self.x, self.p)

    def __pow__(self, n):
        return Zq(pow(self.x, n, self.p), self.p)

    def __mod__(self, m):
        return self.x % m

    def inv(self):
        try:
            return Zq(pow(self.x, -1, self.p), self.p)
        except ValueError:
            raise Exception("modular inverse does not exist")

    def __eq__(self, o):
        return self.p == o.p and self.x % self.p == o.x % o.p

    def __hash__(self):
        return hash((self.x % self.p, self.p))

    def __int__(self):
        return self.x

    def __str__(self):
        return str(self.x)

    __repr__ = __str__


ModP = Zq  # the reference's name


# --------------------------------------------------------------------------
# hashing, codecs, transcript -- src/utils/utils.py:84-137, src/utils/transcript.py
# --------------------------------------------------------------------------
def mod_hash(msg, p, non_zero=True):
    """src/utils/utils.py:84-97: first counter i >= 1 such that
    sha256(str(i) || msg) mod 2^bitlen(p) is < p (and non-zero)."""
    mask = (1 << p.bit_length()) - 1
    i = 1
    while True:
        x = int.from_bytes(sha256(str(i).encode() + msg).digest(), "big") & mask
        if x < p and not (non_zero and x == 0):
            return Zq(x, p)
        i += 1


def point_to_bytes(g):
    """src/utils/utils.py:100-106: SEC1 compressed form; identity is b'\\x00'."""
    if g == INF:
        return b"\x00"
    return (b"\x03" if g.y & 1 else b"\x02") + g.x.to_bytes(BYTE_LENGTH, "big")


def point_to_b64(g):
    """src/utils/utils.py:109-111."""
    return base64.b64encode(point_to_bytes(g))


def bytes_to_point(b):
    """src/utils/utils.py:119-131.  (The reference's `b == 0` identity test is a
    dead branch for bytes input; reproduced as such.)"""
    if b == 0:
        return INF
    want_odd = 0 if b[0] == 2 else 1
    x = int.from_bytes(b[1:], "big")
    y = mod_sqrt((x**3 + CURVE.a * x + CURVE.b) % CURVE.p, CURVE.p)[0]
    if y % 2 != want_odd:
        y = CURVE.p - y
    return Point(x, y, CURVE)


def b64_to_point(s):
    """src/utils/utils.py:114-116."""
    return bytes_to_point(base64.b64decode(s))


def inner_product(a, b):
    """src/utils/utils.py:134-137."""
    assert len(a) == len(b)
    acc = Zq(0, a[0].p)
    for ai, bi in zip(a, b):
        acc = acc + ai * bi
    return acc


class Transcript:
    """src/utils/transcript.py:6-33: digest = b64(seed)& then b64(point)& /
    decimal(number)& items; a challenge is mod_hash of the whole digest."""

    def __init__(self, seed=b""):
        self.digest = base64.b64encode(seed) + b"&"

    def add_point(self, g):
        self.digest += point_to_b64(g) + b"&"

    def add_list_points(self, gs):
        for g in gs:
            self.add_point(g)

    def add_number(self, x):
        self.digest += str(x).encode() + b"&"

    def get_modp(self, p):
        return mod_hash(self.digest, p)


def elliptic_hash(msg, curve=CURVE):
    """src/utils/elliptic_curve_hash.py:7-23 (try-and-increment; MD5 parity bit
    picks the sign).  Only used to derive generators (inputs)."""
    p = curve.p
    i = 1
    while True:
        pre = str(i).encode() + msg
        x = int.from_bytes(sha256(pre).digest(), "big")
        if x < p:
            y = mod_sqrt((x**3 + curve.a * x + curve.b) % p, p)[0]
            if curve.is_point_on_curve((x, y)):
                odd = int(md5(pre).hexdigest(), 16) % 2
                return Point(x, y, curve) if odd else Point(x, p - y, curve)
        i += 1


# --------------------------------------------------------------------------
# groups and the reference's multi-exponentiation -- src/pippenger/*
# --------------------------------------------------------------------------
class CountedModP:
    """src/pippenger/modp.py:1-53: int mod p with a class-level mult counter."""

    num_of_mult = 0

    @classmethod
    def reset(cls):
        cls.num_of_mult = 0

    def __init__(self, x, p):
        self.x = x
        self.p = p

    def __mul__(self, o):
        type(self).num_of_mult += 1
        if isinstance(o, int):
            return CountedModP(self.x * o, self.p)
        assert self.p == o.p
        return CountedModP(self.x * o.x % self.p, self.p)

    def __pow__(self, n):
        # square-and-multiply so the counter sees every mult (:31-40)
        out = CountedModP(self.x, self.p)
        for bit in bin(n)[3:]:
            out = out * out
            if bit == "1":
                out = out * self
        return out

    def __eq__(self, o):
        return self.x == o.x and self.p == o.p


class Group:
    """src/pippenger/group.py:6-16."""

    def __init__(self, unit, order):
        self.unit = unit
        self.order = order

    def mult(self, x, y):
        raise NotImplementedError

    def square(self, x):
        return self.mult(x, x)


class MultIntModP(Group):
    """src/pippenger/group.py:19-24."""

    def __init__(self, p, order):
        super().__init__(CountedModP(1, p), order)

    def mult(self, x, y):
        return x * y


class EC(Group):
    """src/pippenger/group.py:27-32, with a group-op counter (= the number of
    `Point.__add__` calls the reference would make)."""

    def __init__(self, curve=CURVE):
        super().__init__(INF, curve.q)
        self.ops = 0

    def mult(self, x, y):
        self.ops += 1
        return x + y


class Pippenger:
    """The reference's multi-exponentiation (src/pippenger/pippenger.py:8-94):
    Pippenger's bit-matrix / subset-table formulation, NOT the bucket method.
    Same s, t, b parametrisation and the same number of group operations, with
    subset tables indexed by bitmask instead of index tuples."""

    def __init__(self, group):
        self.G = group
        self.order = group.order
        self.lamb = group.order.bit_length()

    def _pow2powof2(self, g, j):  # :15-19
        for _ in range(j):
            g = self.G.square(g)
        return g

    def multiexp(self, gs, es):  # :22-61
        if len(gs) != len(es):
            raise Exception("Different number of group elements and exponents")
        es = [e % self.G.order for e in es]
        N = len(gs)
        if N == 0:
            return self.G.unit
        lamb = self.lamb
        s = isqrt(lamb // N) + 1  # :33
        t = isqrt(lamb * N) + 1  # :34
        bases = []  # g_i^(2^j), row-major (i, j)  :35-40, :52
        for g in gs:
            bases.append(g)
            for _ in range(1, s):
                g = self.G.square(g)
                bases.append(g)
        # bit (j + s*k) of e_i for base (i, j), position k  :41-49, :53
        rows = [
            sum(((e >> (j + s * k)) & 1) << k for k in range(t))
            for e in es
            for j in range(s)
        ]
        Gs = self._multiexp_bin(bases, rows, t)
        acc = Gs[-1]  # :56-59
        for k in range(t - 2, -1, -1):
            acc = self._pow2powof2(acc, s)
            acc = self.G.mult(acc, Gs[k])
        return acc

    def _multiexp_bin(self, gs, rows, t):  # :63-94
        M = len(gs)
        b = floor(log2(M) - log2(log2(M))) if M > 1 else 0  # :66
        # M == 1: log2(log2(1)) raises in the reference too only for M==1... it
        # cannot happen: M = N*s >= 2 for every N >= 1 with lamb = 256.
        b = b if b else 1
        out = [self.G.unit] * t
        first = [True] * t
        tables = []
        for lo in range(0, M, b):
            grp = gs[lo : lo + b]
            T = [None] * (1 << len(grp))
            for mask in range(1, len(T)):  # :71-81 one mult per non-singleton
                top = mask.bit_length() - 1
                rest = mask ^ (1 << top)
                T[mask] = grp[top] if rest == 0 else self.G.mult(T[rest], grp[top])
            tables.append((lo, len(grp), T))
        for k in range(t):  # :83-92
            tmp = self.G.unit
            for lo, m, T in tables:
                mask = 0
                for j in range(m):
                    mask |= ((rows[lo + j] >> k) & 1) << j
                if mask:
                    tmp = self.G.mult(tmp, T[mask])
            out[k] = tmp
        return out


PipSECP256k1 = Pippenger(EC(CURVE))  # src/pippenger/__init__.py:5


def multiexp_naive(gs, es):
    """Sum e_i * g_i by independent scalar multiplications: the definition the
    reference's multiexp must equal (used for sizes where the subset tables are
    too slow)."""
    acc = INF
    for g, e in zip(gs, es):
        acc = acc + (int(e % Q)) * g
    return acc


# --------------------------------------------------------------------------
# commitments -- src/utils/commitments.py:5-13
# --------------------------------------------------------------------------
def commitment(g, h, x, r):
    return x * g + r * h


def vector_commitment(g, h, a, b, multiexp=None):
    assert len(g) == len(h) == len(a) == len(b)
    return (multiexp or PipSECP256k1.multiexp)(g + h, a + b)


# --------------------------------------------------------------------------
# inner-product argument -- src/innerproduct/*
# --------------------------------------------------------------------------
def Proof1(u_new, P_new, proof2, transcript):  # inner_product_verifier.py:10-17
    return SimpleNamespace(u_new=u_new, P_new=P_new, proof2=proof2, transcript=transcript)


def Proof2(a, b, xs, Ls, Rs, transcript, start_transcript=0):  # :61-73
    return SimpleNamespace(
        a=a, b=b, xs=xs, Ls=Ls, Rs=Rs, transcript=transcript, start_transcript=start_transcript
    )


def ipa2_prove(g, h, u, P, a, b, q=Q, transcript=None, multiexp=None, ecops=None):
    """FastNIProver2 (src/innerproduct/inner_product_prover.py:48-110).
    ecops (optional, config-size test cases only): an object with lincomb2(p1, p2, k1, k2) -> [k1*p1[i] + k2*p2[i]]
    and mul_batch(points, scalars) -> [scalars[i]*points[i]] that computes the SAME element-wise expressions in
    bulk (oracle/cbind.py: the plain-C affine group law); without it every element is a Python `int * Point`."""
    assert len(g) == len(h) == len(a) == len(b)
    assert len(a) & (len(a) - 1) == 0
    tr = Transcript()
    if transcript:
        tr.digest += transcript
        init_len = len(transcript.split(b"&"))
    else:
        init_len = 1
    xs, Ls, Rs = [], [], []
    while len(a) > 1:
        half = len(a) // 2
        cl = inner_product(a[:half], b[half:])  # :96
        cr = inner_product(a[half:], b[:half])  # :97
        L = vector_commitment(g[half:], h[:half], a[:half], b[half:], multiexp) + cl * u
        R = vector_commitment(g[:half], h[half:], a[half:], b[:half], multiexp) + cr * u
        Ls.append(L)
        Rs.append(R)
        tr.add_list_points([L, R])
        x = tr.get_modp(q)
        xs.append(x)
        tr.add_number(x)
        xi = x.inv()
        if ecops is not None:
            g = ecops.lincomb2(g[:half], g[half:], xi, x)  # :107
            h = ecops.lincomb2(h[:half], h[half:], x, xi)  # :108
        else:
            g = [xi * g[i] + x * g[half + i] for i in range(half)]  # :107
            h = [x * h[i] + xi * h[half + i] for i in range(half)]  # :108
        a = [x * a[i] + xi * a[half + i] for i in range(half)]  # :109
        b = [xi * b[i] + x * b[half + i] for i in range(half)]  # :110
    return Proof2(a[0], b[0], xs, Ls, Rs, tr.digest, init_len)


def ipa1_prove(g, h, u, P, c, a, b, q=Q, seed=b"", multiexp=None, ecops=None):
    """NIProver (src/innerproduct/inner_product_prover.py:11-45)."""
    assert len(g) == len(h) == len(a) == len(b)
    tr = Transcript(seed)
    x = tr.get_modp(q)
    tr.add_number(x)
    P_new = P + (x * c) * u
    u_new = x * u
    p2 = ipa2_prove(g, h, u_new, P_new, a, b, q, tr.digest, multiexp, ecops)
    return Proof1(u_new, P_new, p2, tr.digest)


def _must(ok):
    if not ok:
        raise Exception("Proof invalid")


def get_ss(xs, n):
    """Verifier2.get_ss (src/innerproduct/inner_product_verifier.py:91-102):
    s_i = prod_j x_j^{+1 if bit j (MSB first) of i is set else -1}, i = 0..n-1.
    Built by doubling instead of n*log n products; values are identical."""
    log_n = n.bit_length() - 1
    ss = [Zq(1, Q)]
    for j in range(log_n - 1, -1, -1):  # LSB of i <-> last challenge
        x, xi = xs[j], xs[j].inv()
        # prepend a more significant index bit: new index = bit * len(ss) + old
        ss = [s * xi for s in ss] + [s * x for s in ss]
    return ss


def ipa2_verify(g, h, u, P, proof, multiexp=None, out=None):
    """Verifier2.verify (src/innerproduct/inner_product_verifier.py:104-147)."""
    mexp = multiexp or PipSECP256k1.multiexp
    n = len(g)
    log_n = n.bit_length() - 1
    parts = proof.transcript.split(b"&")
    k0 = proof.start_transcript
    for i in range(log_n):  # verify_transcript :104-125
        _must(parts[k0 + 3 * i] == point_to_b64(proof.Ls[i]))
        _must(parts[k0 + 3 * i + 1] == point_to_b64(proof.Rs[i]))
        want = str(mod_hash(b"&".join(parts[: k0 + 3 * i + 2]) + b"&", Q)).encode()
        _must(str(proof.xs[i]).encode() == parts[k0 + 3 * i + 2] == want)
    ss = get_ss(proof.xs, n)
    lhs = mexp(
        g + h + [u],
        [proof.a * s for s in ss] + [proof.b * s.inv() for s in ss] + [proof.a * proof.b],
    )
    rhs = P + mexp(
        proof.Ls + proof.Rs,
        [x**2 for x in proof.xs] + [x.inv() ** 2 for x in proof.xs],
    )
    if out is not None:
        out["lhs"], out["rhs"] = lhs, rhs
    _must(lhs == rhs)
    return True


def ipa1_verify(g, h, u, P, c, proof1, multiexp=None):
    """Verifier1.verify (src/innerproduct/inner_product_verifier.py:36-58)."""
    parts = proof1.transcript.split(b"&")
    _must(parts[1] == str(mod_hash(b"&".join(parts[:1]) + b"&", Q)).encode())
    x = Zq(int(parts[1]), Q)
    _must(proof1.P_new == P + (x * c) * u)
    _must(proof1.u_new == x * u)
    return ipa2_verify(g, h, proof1.u_new, proof1.P_new, proof1.proof2, multiexp)


# --------------------------------------------------------------------------
# range proofs (callers of the hot path) -- src/rangeproofs/*
# --------------------------------------------------------------------------
def RangeProof(taux, mu, t_hat, T1, T2, A, S, innerProof, transcript):
    """Proof container, src/rangeproofs/rangeproof_verifier.py:10-22."""
    return SimpleNamespace(
        taux=taux, mu=mu, t_hat=t_hat, T1=T1, T2=T2, A=A, S=S,
        innerProof=innerProof, transcript=transcript,
    )


def _zpow_term(z, i, n):
    """z^(2 + i//n) * 2^(i % n): aggregated form (rangeproof_aggreg_prover.py:99);
    with m = 1 it is the single-proof z^2 * 2^i (rangeproof_prover.py:84)."""
    return (z ** (2 + i // n)) * (2 ** (i % n))


def range_prove_generic(vs, n, g, h, gs, hs, gammas, u, q=Q, seed=b"", multiexp=None, ecops=None):
    """NIRangeProver.prove (src/rangeproofs/rangeproof_prover.py:35-112) for
    len(vs) == 1, AggregNIRangeProver.prove (rangeproof_aggreg_prover.py:36-146)
    otherwise.  The two differ only in the z-power term and in gamma handling;
    note rho hashes str(2*n), not 2*n*m, in both (aggreg :61)."""
    mexp = multiexp or PipSECP256k1.multiexp
    m = len(vs)
    nm = n * m
    tr = Transcript(seed)
    aL = []
    for v in vs:
        aL += [int(c) for c in reversed(bin(v.x)[2:].zfill(n))][:n]
    aR = [(bit - 1) % q for bit in aL]
    alpha = mod_hash(b"alpha" + tr.digest, q)
    A = vector_commitment(gs, hs, aL, aR, mexp) + alpha * h
    sL = [mod_hash(str(i).encode() + tr.digest, q) for i in range(nm)]
    sR = [mod_hash(str(i).encode() + tr.digest, q) for i in range(nm, 2 * nm)]
    rho = mod_hash(str(2 * n).encode() + tr.digest, q)
    S = vector_commitment(gs, hs, sL, sR, mexp) + rho * h
    tr.add_list_points([A, S])
    y = tr.get_modp(q)
    tr.add_number(y)
    z = tr.get_modp(q)
    tr.add_number(z)
    ypow = [y**i for i in range(nm)]
    # _get_polynomial_coeffs (:93-101 / aggreg :117-130)
    t1 = inner_product(
        sL, [ypow[i] * (aR[i] + z) + _zpow_term(z, i, n) for i in range(nm)]
    ) + inner_product([aL[i] - z for i in range(nm)], [ypow[i] * sR[i] for i in range(nm)])
    t2 = inner_product(sL, [ypow[i] * sR[i] for i in range(nm)])
    tau1 = mod_hash(b"tau1" + tr.digest, q)
    tau2 = mod_hash(b"tau2" + tr.digest, q)
    T1 = commitment(g, h, t1, tau1)
    T2 = commitment(g, h, t2, tau2)
    tr.add_list_points([T1, T2])
    x = tr.get_modp(q)
    tr.add_number(x)
    # _final_compute (:103-112 / aggreg :132-146)
    ls = [aL[i] - z + sL[i] * x for i in range(nm)]
    rs = [ypow[i] * (aR[i] + z + sR[i] * x) + _zpow_term(z, i, n) for i in range(nm)]
    t_hat = inner_product(ls, rs)
    if m == 1 and not isinstance(gammas, (list, tuple)):
        gsum = (z**2) * gammas
    else:
        gsum = sum([(z ** (2 + j)) * gammas[j] for j in range(m)])
    taux = tau2 * (x**2) + tau1 * x + gsum
    mu = alpha + rho * x
    yinv = y.inv()
    if ecops is not None:
        hsp = ecops.mul_batch(hs, [yinv**i for i in range(nm)])  # :77 / aggreg :82
    else:
        hsp = [(yinv**i) * hs[i] for i in range(nm)]  # :77 / aggreg :82
    P = (
        A
        + x * S
        + mexp(
            gs + hsp,
            [-z for _ in range(nm)] + [(z * ypow[i]) + _zpow_term(z, i, n) for i in range(nm)],
        )
    )
    inner = ipa1_prove(gs, hsp, u, P + (-mu) * h, t_hat, ls, rs, q, b"", mexp, ecops)
    return RangeProof(taux, mu, t_hat, T1, T2, A, S, inner, tr.digest)


def range_prove(v, n, g, h, gs, hs, gamma, u, q=Q, seed=b"", multiexp=None, ecops=None):
    return range_prove_generic([v], n, g, h, gs, hs, gamma, u, q, seed, multiexp, ecops)


def aggreg_range_prove(vs, n, g, h, gs, hs, gammas, u, q=Q, seed=b"", multiexp=None, ecops=None):
    return range_prove_generic(list(vs), n, g, h, gs, hs, list(gammas), u, q, seed, multiexp, ecops)


def _range_transcript(proof):
    """verify_transcript (rangeproof_verifier.py:42-53): A, S, T1, T2 must match;
    y, z, x are READ from the transcript, not re-hashed."""
    p = proof.taux.p
    parts = proof.transcript.split(b"&")
    _must(parts[1] == point_to_b64(proof.A))
    _must(parts[2] == point_to_b64(proof.S))
    y = Zq(int(parts[3]), p)
    z = Zq(int(parts[4]), p)
    _must(parts[5] == point_to_b64(proof.T1))
    _must(parts[6] == point_to_b64(proof.T2))
    x = Zq(int(parts[7]), p)
    return x, y, z


def range_verify_generic(Vs, g, h, gs, hs, u, proof, aggregated, multiexp=None):
    """RangeVerifier.verify (rangeproof_verifier.py:55-99) when not `aggregated`,
    AggregRangeVerifier.verify (rangeproof_aggreg_verifier.py:55-108) otherwise."""
    mexp = multiexp or PipSECP256k1.multiexp
    x, y, z = _range_transcript(proof)
    nm = len(gs)
    m = len(Vs) if aggregated else 1
    n = nm // m
    ysum = Zq(0, Q)
    ypow = []
    for i in range(nm):
        yi = y**i
        ypow.append(yi)
        ysum = ysum + yi
    if aggregated:
        delta = (z - z**2) * ysum - sum(
            [(z ** (j + 2)) * Zq(2**n - 1, Q) for j in range(1, m + 1)]
        )
    else:
        delta = (z - z**2) * ysum - (z**3) * Zq(2**n - 1, Q)
    yinv = y.inv()
    hsp = [(yinv**i) * hs[i] for i in range(nm)]
    lhs = proof.t_hat * g + proof.taux * h
    if aggregated:
        rhs = mexp(
            list(Vs) + [g, proof.T1, proof.T2],
            [z ** (j + 2) for j in range(m)] + [delta, x, x**2],
        )
    else:
        rhs = (z**2) * Vs + delta * g + x * proof.T1 + (x**2) * proof.T2
    _must(lhs == rhs)
    P = (
        proof.A
        + x * proof.S
        + mexp(
            gs + hsp,
            [-z for _ in range(nm)] + [(z * ypow[i]) + _zpow_term(z, i, n) for i in range(nm)],
        )
    )
    return ipa1_verify(gs, hsp, u, P + (-proof.mu) * h, proof.t_hat, proof.innerProof, mexp)


def range_verify(V, g, h, gs, hs, u, proof, multiexp=None):
    return range_verify_generic(V, g, h, gs, hs, u, proof, False, multiexp)


def aggreg_range_verify(Vs, g, h, gs, hs, u, proof, multiexp=None):
    return range_verify_generic(Vs, g, h, gs, hs, u, proof, True, multiexp)
