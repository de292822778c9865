"""Point / Curve: the surface of the third-party `fastecdsa` package that the reference
uses (Point(x, y, curve), +, int *, ==, .x .y .curve, IDENTITY_ELEMENT; Curve.p .a .b .q
.G .is_point_on_curve; mod_sqrt) -- call sites: /root/reference/src/pippenger/group.py:29-32,
src/utils/utils.py:43-44,100-131, src/innerproduct/inner_product_verifier.py:145.

Group operations are NOT computed here: `+` and `*` go to the HIP engine
(bpmi_ec_sum / bpmi_ec_mul_batch).  Only representation-level work (equality, negation
of y, the on-curve check of the constructor, byte packing) is done with Python ints.
"""
import threading

from . import engine as _engine


class Curve:
    def __init__(self, name, p, a, b, q, gx, gy):
        self.name, self.p, self.a, self.b, self.q, self.gx, self.gy = name, p, a, b, q, gx, gy

    @property
    def G(self):
        return Point(self.gx, self.gy, self)

    def is_point_on_curve(self, xy):
        x, y = xy
        return (y * y - x * x * x - self.a * x - self.b) % self.p == 0

    def __repr__(self):
        return self.name


secp256k1 = Curve(
    "secp256k1",
    p=2**256 - 2**32 - 977,
    a=0,
    b=7,
    q=0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    gx=0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    gy=0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8,
)


class Point:
    __slots__ = ("x", "y", "curve", "_le")        # points are immutable by convention; _le caches the wire form
    IDENTITY_ELEMENT = None

    def __init__(self, x, y, curve=secp256k1):
        if curve is None:
            if x or y:
                raise ValueError("identity must be (0, 0)")
        elif not curve.is_point_on_curve((x, y)):
            raise ValueError("coordinates are not on curve %s" % curve)
        self.x, self.y, self.curve = x, y, curve
        self._le = None

    @classmethod
    def _raw(cls, x, y):
        pt = cls.__new__(cls)
        pt.x, pt.y = x, y
        pt.curve = secp256k1 if (x or y) else None
        pt._le = None
        return pt

    # -- representation-level ------------------------------------------------
    def __eq__(self, other):
        return isinstance(other, Point) and self.x == other.x and self.y == other.y

    def __hash__(self):
        return hash((self.x, self.y))

    def __neg__(self):
        if self.curve is None:
            return self
        return Point._raw(self.x, (-self.y) % self.curve.p)

    def to_le64(self):
        b = self._le
        if b is None:
            b = self._le = self.x.to_bytes(32, "little") + self.y.to_bytes(32, "little")
        return b

    @classmethod
    def from_le64(cls, b):
        pt = cls._raw(int.from_bytes(b[:32], "little"), int.from_bytes(b[32:64], "little"))
        pt._le = bytes(b[:64])
        return pt

    # -- group operations: on the GPU ----------------------------------------------
    def __add__(self, other):
        if not isinstance(other, Point):
            return NotImplemented
        eng = _engine.default_engine()
        return Point.from_le64(eng.ec_sum_bytes(self.to_le64() + other.to_le64(), 2))

    def __sub__(self, other):
        return self + (-other)

    def __mul__(self, k):
        k = int(k) % secp256k1.q
        # a one-term MSM: ~0.4 ms of launch latency, against ~2 ms for one lane walking the
        # 257-step ladder of bpmi_ec_mul_batch (which pays off only for many points at once)
        eng = _engine.default_engine()
        return Point.from_le64(eng.msm_bytes(self.to_le64(), k.to_bytes(32, "little"), 1))

    __rmul__ = __mul__

    def __repr__(self):
        return "Point(inf)" if self.curve is None else "Point(0x%x, 0x%x)" % (self.x, self.y)


Point.IDENTITY_ELEMENT = Point._raw(0, 0)


def mod_sqrt(a, p):
    """(r, p - r) for p = 3 (mod 4); used only by codecs / generator derivation."""
    r = pow(a, (p + 1) // 4, p)
    return (r, p - r)


class PackedPoints(list):
    """A list of points with its wire form (64 bytes per point) attached, for lists that go into several MSMs -- the
    generators of a proof are packed once instead of once per call.  Immutable by convention: build a new one instead of
    changing it."""

    def __init__(self, pts, packed=None):
        super().__init__(pts)
        self.packed = packed if packed is not None else b"".join(map(Point.to_le64, self))
        self._dev = {}                  # id(engine) -> (engine, DeviceBuffer): the same bytes in that engine's device memory
        self._dev_lock = threading.Lock()

    def device(self, engine):
        """The list in `engine`'s device memory, uploaded once PER ENGINE (generator lists are deployment constants shared by
        every engine and thread of a process: the verifiers read them from there instead of uploading them per proof).  A
        buffer is never freed on behalf of another engine -- its queued MSMs may still read it; `release()` or the
        collection of the list frees them (a DeviceBuffer whose engine is already closed frees nothing)."""
        with self._dev_lock:
            hit = self._dev.get(id(engine))
            if hit is None or hit[0] is not engine or hit[1].ptr is None:
                hit = self._dev[id(engine)] = (engine, engine.upload(self.packed))
            return hit[1]

    def release(self, engine=None):
        """Free the device copy held for `engine` (all engines when None)."""
        with self._dev_lock:
            keys = [k for k, (e, _) in self._dev.items() if engine is None or e is engine]
            for k in keys:
                e, buf = self._dev.pop(k)
                if getattr(e, "ctx", None):
                    buf.free()
                else:
                    buf.ptr = None        # the ctx is gone, and with it the right to call into it

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @classmethod
    def join(cls, *parts):
        pts = []
        for part in parts:
            pts.extend(part)
        return cls(pts, b"".join(pack_points(part) for part in parts))


def pack_points(pts):
    if isinstance(pts, PackedPoints):
        return pts.packed
    return b"".join(map(Point.to_le64, pts))


class PackedScalars(list):
    """A list of scalars (ints in [0, q)) with its wire form (32 bytes little-endian each) attached -- for vectors that are both
    computed with on the host and handed to an MSM (the provers' blinding vectors arrive from native code as bytes already).
    Built `from_bytes`, the integers are only materialised when somebody looks at them (len() and the wire form need none)."""

    def __init__(self, es, packed=None):
        super().__init__(es)
        self._lazy = False
        self.packed = packed if packed is not None else b"".join([e.to_bytes(32, "little") for e in self])

    @classmethod
    def from_bytes(cls, raw):
        self = cls((), bytes(raw))
        self._lazy = len(self.packed) > 0
        return self

    def _fill(self):
        if self._lazy:
            self._lazy = False
            raw = self.packed
            list.extend(self, [int.from_bytes(raw[i: i + 32], "little") for i in range(0, len(raw), 32)])

    def __len__(self):
        return len(self.packed) // 32 if self._lazy else list.__len__(self)

    def __iter__(self):
        self._fill()
        return list.__iter__(self)

    def __getitem__(self, i):
        self._fill()
        return list.__getitem__(self, i)

    def __eq__(self, other):
        self._fill()
        if isinstance(other, PackedScalars):
            other._fill()
        return list.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __add__(self, other):
        self._fill()
        return list(self) + list(other)

    def __radd__(self, other):
        self._fill()
        return list(other) + list(self)

    def __contains__(self, v):
        self._fill()
        return list.__contains__(self, v)

    def __repr__(self):
        self._fill()
        return list.__repr__(self)

    # every other read of the list sees the materialised integers ...
    def copy(self):
        self._fill()
        return list.copy(self)

    def __reversed__(self):
        self._fill()
        return list.__reversed__(self)

    def index(self, *args):
        self._fill()
        return list.index(self, *args)

    def count(self, v):
        self._fill()
        return list.count(self, v)

    def __mul__(self, k):
        self._fill()
        return list.__mul__(self, k)

    __rmul__ = __mul__

    # ... and a write would leave `packed` behind: the class is read-only (build a new one)
    def _read_only(self, *args, **kwargs):
        raise TypeError("PackedScalars is read-only: its wire form is attached (build a new one)")

    __setitem__ = __delitem__ = __iadd__ = __imul__ = _read_only
    append = extend = insert = pop = remove = clear = sort = reverse = _read_only

    @classmethod
    def join(cls, *parts):
        if any(isinstance(part, PackedScalars) and part._lazy for part in parts):
            return cls.from_bytes(b"".join(pack_scalars(part) for part in parts))
        es = []
        for part in parts:
            es.extend(part)
        return cls(es, b"".join(pack_scalars(part) for part in parts))


def pack_scalars(es, q=secp256k1.q):
    if isinstance(es, PackedScalars) and q == secp256k1.q:
        return es.packed
    return b"".join((int(e % q)).to_bytes(32, "little") for e in es)


def unpack_points(buf, n):
    return [Point.from_le64(buf[64 * i: 64 * i + 64]) for i in range(n)]
