"""Pippenger(group).multiexp(gs, es) -- same call surface as the reference
(src/pippenger/pippenger.py:8-61).

For EC(secp256k1) the whole product runs on the MI355X through libbpmi (bpmi_msm);
there is no CPU path for that group.  For any other Group (e.g. MultIntModP) a small
generic windowed product written against Group.mult / Group.square is used -- that is
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

    # -- any other group: generic 4-bit windowed product -------------------------
    def _multiexp_generic(self, gs, es):
        G = self.G
        es = [e % G.order for e in es]
        if not gs:
            return G.unit
        c = 4
        nwin = (self.lamb + c - 1) // c
        acc = G.unit
        for w in range(nwin - 1, -1, -1):
            for _ in range(c):
                acc = G.square(acc)
            buckets = [None] * (1 << c)
            for g, e in zip(gs, es):
                d = (e >> (c * w)) & ((1 << c) - 1)
                if d:
                    buckets[d] = g if buckets[d] is None else G.mult(buckets[d], g)
            run = total = None
            for d in range((1 << c) - 1, 0, -1):
                if buckets[d] is not None:
                    run = buckets[d] if run is None else G.mult(run, buckets[d])
                if run is not None:
                    total = run if total is None else G.mult(total, run)
            if total is not None:
                acc = G.mult(acc, total)
        return acc
