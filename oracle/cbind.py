"""ctypes binding of oracle/c/libbp_oracle.so (TEST INFRASTRUCTURE).

Point lists are `oracle.ec.Point`; scalars are ints (or anything with `% int`).
"""
import ctypes
import os
import subprocess

from .ec import INF, Point, point_from_le64, point_to_le64, secp256k1

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_SO = os.path.join(_DIR, "libbp_oracle.so")
Q = secp256k1.q
_lib = None


def build(force=False):
    src = os.path.join(_DIR, "bp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p, u64, i32 = ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int
        L.orc_msm.argtypes = [u8p, u8p, u64, i32, u8p]
        L.orc_ec_mul_batch.argtypes = [u8p, u8p, u64, i32, u8p]
        L.orc_ec_lincomb2_batch.argtypes = [u8p, u8p, u8p, u8p, u64, i32, u8p]
        L.orc_ec_add.argtypes = [u8p, u8p, u8p]
        L.orc_ec_decompress_batch.argtypes = [u8p, u64, i32, u8p, u8p]
        L.orc_ec_decompress_batch.restype = None
        L.orc_sc_dot.argtypes = [u8p, u8p, u64, u8p]
        L.orc_sc_fold.argtypes = [u8p, u8p, u8p, u8p, u64, u8p]
        L.orc_sc_mul.argtypes = [u8p, u8p, u8p]
        L.orc_fe_mul.argtypes = [u8p, u8p, u8p]
        for f in (L.orc_msm, L.orc_ec_mul_batch, L.orc_ec_lincomb2_batch, L.orc_ec_add,
                  L.orc_sc_dot, L.orc_sc_fold, L.orc_sc_mul, L.orc_fe_mul):
            f.restype = None
        _lib = L
    return _lib


def pack_points(pts):
    return b"".join(point_to_le64(p) for p in pts)


def pack_scalars(es):
    return b"".join(int(e % Q).to_bytes(32, "little") for e in es)


def unpack_points(buf, n):
    return [point_from_le64(buf[64 * i: 64 * i + 64]) for i in range(n)]


def msm_bytes(pts_bytes, sc_bytes, n, threads=None):
    out = ctypes.create_string_buffer(64)
    lib().orc_msm(pts_bytes, sc_bytes, n, threads or os.cpu_count() or 1, out)
    return out.raw


def msm(gs, es, threads=None):
    """= Pippenger.multiexp over EC(secp256k1) (src/pippenger/pippenger.py:22-61)."""
    if len(gs) != len(es):
        raise Exception("Different number of group elements and exponents")
    return point_from_le64(msm_bytes(pack_points(gs), pack_scalars(es), len(gs), threads))


def ec_mul_batch(pts, es, threads=None):
    n = len(pts)
    out = ctypes.create_string_buffer(64 * n)
    lib().orc_ec_mul_batch(pack_points(pts), pack_scalars(es), n, threads or os.cpu_count() or 1, out)
    return unpack_points(out.raw, n)


def ec_lincomb2_batch(p1, p2, k1, k2, threads=None):
    n = len(p1)
    out = ctypes.create_string_buffer(64 * n)
    lib().orc_ec_lincomb2_batch(pack_points(p1), pack_points(p2), pack_scalars([k1]), pack_scalars([k2]),
                                n, threads or os.cpu_count() or 1, out)
    return unpack_points(out.raw, n)


def ec_decompress_batch_bytes(comp, n, threads=None):
    """n x 33-byte compressed points -> (n x 64-byte points, n validity bytes): bytes_to_point (src/utils/utils.py:119-131)
    in bulk; 33 zero bytes decode to the identity."""
    out, ok = ctypes.create_string_buffer(64 * n), ctypes.create_string_buffer(n)
    lib().orc_ec_decompress_batch(bytes(comp), n, threads or os.cpu_count() or 1, out, ok)
    return out.raw, ok.raw


def ec_add(a, b):
    out = ctypes.create_string_buffer(64)
    lib().orc_ec_add(point_to_le64(a), point_to_le64(b), out)
    return point_from_le64(out.raw)


def sc_dot(a, b):
    out = ctypes.create_string_buffer(32)
    lib().orc_sc_dot(pack_scalars(a), pack_scalars(b), len(a), out)
    return int.from_bytes(out.raw, "little")


def sc_fold(lo, hi, x, xinv):
    n = len(lo)
    out = ctypes.create_string_buffer(32 * n)
    lib().orc_sc_fold(pack_scalars(lo), pack_scalars(hi), pack_scalars([x]), pack_scalars([xinv]), n, out)
    return [int.from_bytes(out.raw[32 * i: 32 * i + 32], "little") for i in range(n)]


class BulkEC:
    """`ecops` for oracle.bp_ref at configuration sizes: the element-wise `int * Point` / fold expressions of
    the reference computed in bulk by the C restatement of the affine group law."""

    def __init__(self, threads=None):
        self.threads = threads

    def mul_batch(self, pts, scalars):
        return ec_mul_batch(pts, [int(s % Q) for s in scalars], self.threads)

    def lincomb2(self, p1, p2, k1, k2):
        return ec_lincomb2_batch(p1, p2, int(k1 % Q), int(k2 % Q), self.threads)


def sc_dot_bytes(a, b, n):
    out = ctypes.create_string_buffer(32)
    lib().orc_sc_dot(a, b, n, out)
    return out.raw


def sc_fold_bytes(lo, hi, x, xinv, n):
    out = ctypes.create_string_buffer(32 * n)
    lib().orc_sc_fold(lo, hi, int(x % Q).to_bytes(32, "little"), int(xinv % Q).to_bytes(32, "little"), n, out)
    return out.raw
