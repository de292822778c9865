"""Pippenger(group).multiexp(gs, es) -- same call surface as the reference
(src/pippenger/pippenger.py:8-61).

For EC(secp256k1) the whole product runs on the MI355X through libbpmi (bpmi_msm);
there is no CPU path for that group.  For any other Group (e.g. MultIntModP) the reference's
bit-matrix schedule runs on the host, written against Group.mult / Group.square -- that is
the operator API itself, not a fallback for the EC path.
"""
from .. import engine as _engine
from ..ec import Point, pack_points, pack_scalars, secp256k1
from .group import EC


class DevicePoints:
    """A point list uploaded once and reused across MSMs (the reference re-passes the
    same gs / hs lists to 3-4 multiexp calls per proof)."""

    def __init__(self, points, engine=None):
        self.engine = engine or _engine.default_engine()
        self.n = len(points)
        self.buf = self.engine.upload(pack_points(points))


class Pippenger:
    def __init__(self, group):
        self.G = group
        self.order = group.order
        self.lamb = group.order.bit_length()

    def multiexp(self, gs, es):
        if len(gs) != len(es):
            raise Exception("Different number of group elements and exponents")
        if isinstance(self.G, EC) and self.G.curve is secp256k1:
            return self._multiexp_native(gs, es)
        return self._multiexp_generic(gs, es)

    # -- secp256k1: HIP engine ---------------------------------------------------
    def _multiexp_native(self, gs, es):
        n = len(gs)
        if n == 0:
            return self.G.unit
        eng = _engine.default_engine()
        scalars = pack_scalars(es, self.order)       # es[i] % order, pippenger.py:26
        if isinstance(gs, DevicePoints):
            d_sc = eng.upload(scalars)
            out = eng.msm_dev(gs.buf, d_sc, n)
            d_sc.free()
        else:
            out = eng.msm_bytes(pack_points(gs), scalars, n)
        return Point.from_le64(out)

    def multiexp2(self, gs0, es0, gs1, es1):
        """(multiexp(gs0, es0), multiexp(gs1, es1)) for two independent sums; on secp256k1 they are
        overlapped on the engine's two lanes (bpmi_msm2).  Not in the reference: its callers compute
        such pairs (A and S, T1 and T2) one after the other."""
        if len(gs0) != len(es0) or len(gs1) != len(es1):
            raise Exception("Different number of group elements and exponents")
        native = isinstance(self.G, EC) and self.G.curve is secp256k1
        if not native or not gs0 or not gs1 or isinstance(gs0, DevicePoints) or isinstance(gs1, DevicePoints):
            return self.multiexp(gs0, es0), self.multiexp(gs1, es1)
        o0, o1 = _engine.default_engine().msm2_bytes(pack_points(gs0), pack_scalars(es0, self.order), len(gs0),
                                                     pack_points(gs1), pack_scalars(es1, self.order), len(gs1))
        return Point.from_le64(o0), Point.from_le64(o1)

    # -- any other group: the reference's own schedule on the host ------------------------------
    def _multiexp_generic(self, gs, es):
        """Pippenger's bit-matrix multi-exponentiation as the reference parametrises it
        (src/pippenger/pippenger.py:22-94), written against Group.mult / Group.square only, so that
        for a counting group such as MultIntModP the NUMBER of group operations equals the
        reference's as well as the result (tests/golden/modp_group.json pins both):

          * every base g_i is expanded into its s successive squarings g_i^(2^j), j < s, and the
            lamb-bit exponent is read as a t-column bit matrix per expanded base
            (s = isqrt(lamb // N) + 1, t = isqrt(lamb N) + 1; :33-53);
          * the M = N s expanded bases are cut into groups of b = floor(log2 M - log2 log2 M) and
            every group gets the table of all its subset products -- one multiplication per subset
            of two or more elements (:66-81);
          * column k of the matrix costs one table lookup + multiplication per group with a
            non-empty subset (:83-92), and the t column values are combined by Horner's rule with
            s squarings per step (:56-59).
        """
        from math import floor, isqrt, log2
        G = self.G
        order = G.order
        es = [e % order for e in es]
        N = len(gs)
        if N == 0:
            return G.unit
        s = isqrt(self.lamb // N) + 1
        t = isqrt(self.lamb * N) + 1
        # expanded bases and, per expanded base, its matrix row packed into one integer:
        # bit k of row (i, j) = bit (j + s k) of e_i
        bases, rows = [], []
        for g, e in zip(gs, es):
            for j in range(s):
                if j:
                    g = G.square(g)
                bases.append(g)
                row, k, bits = 0, 0, e >> j
                while bits:
                    row |= (bits & 1) << k
                    bits >>= s
                    k += 1
                rows.append(row)
        M = len(bases)
        b = (floor(log2(M) - log2(log2(M))) if M > 1 else 0) or 1
        # subset tables by the lowest-set-bit recurrence: T[mask] = T[mask without its lowest bit] * base(lowest bit)
        tables = []
        for lo in range(0, M, b):
            members = bases[lo: lo + b]
            T = [None] * (1 << len(members))
            for mask in range(1, len(T)):
                low = (mask & -mask).bit_length() - 1
                rest = mask & (mask - 1)
                T[mask] = members[low] if not rest else G.mult(T[rest], members[low])
            tables.append((lo, len(members), T))
        columns = []
        for k in range(t):
            value = G.unit
            for lo, width, T in tables:
                mask = 0
                for j in range(width):
                    mask |= ((rows[lo + j] >> k) & 1) << j
                if mask:
                    value = G.mult(value, T[mask])
            columns.append(value)
        acc = columns[-1]
        for k in range(t - 2, -1, -1):
            for _ in range(s):
                acc = G.square(acc)
            acc = G.mult(acc, columns[k])
        return acc
