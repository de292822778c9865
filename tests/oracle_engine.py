"""Engine stand-in for the CPU (gloo) tests of the sharded host logic: the same byte-level
interface as bulletproofs_amd.engine.Engine for the operations the sharded prover uses, computed by
the oracle the way the reference does it (eager generator folding, affine group law).  Test
infrastructure only -- the product's engine is the HIP library."""
from oracle import cbind
from oracle.ec import INF, point_to_le64, secp256k1

Q = secp256k1.q


def _scalars(buf, n):
    return [int.from_bytes(buf[32 * i: 32 * i + 32], "little") for i in range(n)]


class OracleIpaState:
    def __init__(self, g, h, a, b, n, u, h_scale=None):
        self.g, self.h = cbind.unpack_points(g, n), cbind.unpack_points(h, n)
        if h_scale is not None:
            self.h = cbind.ec_mul_batch(self.h, _scalars(h_scale, n), 1)
        self.a, self.b = _scalars(a, n), _scalars(b, n)
        self.u = cbind.unpack_points(u, 1)[0]

    def __len__(self):
        return len(self.a)

    def round_LR(self):
        np_ = len(self.a) // 2
        g, h, a, b = self.g, self.h, self.a, self.b
        cl = sum(x * y for x, y in zip(a[:np_], b[np_:])) % Q
        cr = sum(x * y for x, y in zip(a[np_:], b[:np_])) % Q
        L = cbind.msm(g[np_:] + h[:np_] + [self.u], a[:np_] + b[np_:] + [cl], 1)
        R = cbind.msm(g[:np_] + h[np_:] + [self.u], a[np_:] + b[:np_] + [cr], 1)
        return point_to_le64(L), point_to_le64(R)

    def fold(self, x, xinv):
        np_ = len(self.a) // 2
        self.g = cbind.ec_lincomb2_batch(self.g[:np_], self.g[np_:], xinv, x, 1)
        self.h = cbind.ec_lincomb2_batch(self.h[:np_], self.h[np_:], x, xinv, 1)
        self.a = [(x * lo + xinv * hi) % Q for lo, hi in zip(self.a[:np_], self.a[np_:])]
        self.b = [(xinv * lo + x * hi) % Q for lo, hi in zip(self.b[:np_], self.b[np_:])]

    def export(self):
        return (cbind.pack_points(self.g), cbind.pack_points(self.h),
                cbind.pack_scalars(self.a), cbind.pack_scalars(self.b))

    def finish(self):
        assert len(self.a) == 1
        return self.a[0], self.b[0]

    def close(self):
        pass


class OracleEngine:
    def ipa_create(self, g, h, a, b, n, u, h_scale=None):
        return OracleIpaState(g, h, a, b, n, u, h_scale)

    def ec_sum_bytes(self, buf, k):
        acc = INF
        for p in cbind.unpack_points(buf, k):
            acc = acc + p
        return point_to_le64(acc)

    def msm_bytes(self, pts, scalars, n):
        return cbind.msm_bytes(pts, scalars, n, 1)
