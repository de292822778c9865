"""The C-ABI checks the points a caller hands in (option validate_points, default 1): what fastecdsa's Point constructor does for
the reference (reached from /root/reference/src/utils/utils.py:119-131) and ec.py for Python callers, the library does for raw
bytes.  An off-curve point gives BPMI_E_ARG with its index in bpmi_last_error -- never a result."""
import ctypes
import random

import pytest

from helpers import Q
from oracle import cbind
from oracle.ec import secp256k1 as OC

pytestmark = pytest.mark.gpu
P_FIELD = OC.p


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


def _bad_points():
    """64-byte encodings that are not points: y off by one, x >= p, y >= p, (0, y), only one coordinate zero."""
    G = OC.G
    le = lambda x, y: x.to_bytes(32, "little") + y.to_bytes(32, "little")
    return [le(G.x, G.y + 1), le(G.x + P_FIELD, G.y) if G.x + P_FIELD < (1 << 256) else le(P_FIELD, 1), le(G.x, P_FIELD + 1), le(0, 5), le(G.x, 0),
            le((1 << 256) - 1, (1 << 256) - 1)]


@pytest.mark.parametrize("n", [3, 64, 65, 129, 5000, 70000])
def test_msm_rejects_an_off_curve_point(gp, n):
    eng = gp.engine()
    lib, ctx = eng.lib, eng.ctx
    pts, _ = gp.rand_points(min(n, 500), 7)
    pts = (pts * (n // len(pts) + 1))[:n]
    rnd = random.Random(n)
    sb = cbind.pack_scalars([rnd.randrange(Q) for _ in range(n)])
    good = cbind.pack_points(pts)
    out = ctypes.create_string_buffer(b"\x55" * 64, 64)
    assert lib.bpmi_msm(ctx, good, sb, n, out) == 0 and out.raw == cbind.msm_bytes(good, sb, n)
    for j, bad in enumerate(_bad_points()):
        pos = (j * 7919 + n // 2) % n
        blob = good[: 64 * pos] + bad + good[64 * pos + 64:]
        out = ctypes.create_string_buffer(b"\x55" * 64, 64)
        assert lib.bpmi_msm(ctx, blob, sb, n, out) == -3
        assert b"pts[%d] is not a point of the curve" % pos in lib.bpmi_last_error(ctx)
        assert out.raw == bytes(64)
        # the pair entry point names the array
        o0, o1 = ctypes.create_string_buffer(64), ctypes.create_string_buffer(64)
        assert lib.bpmi_msm2(ctx, good, sb, n, o0, blob, sb, n, o1) == -3
        assert b"pts1[%d]" % pos in lib.bpmi_last_error(ctx)
    # the identity (64 zero bytes) is a valid input
    blob = bytes(64) + good[64:]
    assert lib.bpmi_msm(ctx, blob, sb, n, out) == 0 and out.raw == cbind.msm_bytes(blob, sb, n)
    # switched off: a result comes back (garbage, as before round 5); switched on again
    try:
        eng.set_option("validate_points", 0)
        blob = good[:64 * (n - 1)] + _bad_points()[0]
        assert lib.bpmi_msm(ctx, blob, sb, n, out) == 0
    finally:
        eng.set_option("validate_points", 1)


def test_dev_entry_points_check_at_level_2(gp):
    eng = gp.engine()
    n = 3000
    pts, _ = gp.rand_points(300, 9)
    good = cbind.pack_points((pts * 10)[:n])
    blob = good[: 64 * 1234] + _bad_points()[0] + good[64 * 1235:]
    sb = cbind.pack_scalars(list(range(1, n + 1)))
    d_p, d_s = eng.upload(blob), eng.upload(sb)
    out = ctypes.create_string_buffer(64)
    try:
        assert eng.lib.bpmi_msm_dev(eng.ctx, d_p.ptr, d_s.ptr, n, out) == 0              # level 1: device pointers are the caller's responsibility
        eng.set_option("validate_points", 2)
        assert eng.lib.bpmi_msm_dev(eng.ctx, d_p.ptr, d_s.ptr, n, out) == -3
        assert b"d_pts[1234]" in eng.lib.bpmi_last_error(eng.ctx) and out.raw == bytes(64)
        d_p.upload(good)
        assert eng.lib.bpmi_msm_dev(eng.ctx, d_p.ptr, d_s.ptr, n, out) == 0 and out.raw == cbind.msm_bytes(good, sb, n)
    finally:
        eng.set_option("validate_points", 1)
        d_p.free(); d_s.free()


def test_other_host_pointer_entry_points(gp):
    eng = gp.engine()
    lib, ctx = eng.lib, eng.ctx
    pts, _ = gp.rand_points(200, 11)
    good = cbind.pack_points(pts)
    bad = _bad_points()[0]
    blob = good[: 64 * 150] + bad + good[64 * 151:]
    sb = cbind.pack_scalars(list(range(1, 201)))
    out = ctypes.create_string_buffer(64 * 200)
    assert lib.bpmi_ec_mul_batch(ctx, blob, sb, 200, out) == -3 and b"pts[150]" in lib.bpmi_last_error(ctx)
    assert out.raw == bytes(64 * 200)                                       # nothing was written
    k = (5).to_bytes(32, "little")
    assert lib.bpmi_ec_lincomb2_batch(ctx, good, blob, k, k, 200, out) == -3 and b"p2[150]" in lib.bpmi_last_error(ctx)
    o64 = ctypes.create_string_buffer(64)
    assert lib.bpmi_ec_sum(ctx, blob, 200, o64) == -3 and b"pts[150]" in lib.bpmi_last_error(ctx)
    assert lib.bpmi_ec_sum(ctx, bad + good[64:640], 10, o64) == -3 and b"pts[0]" in lib.bpmi_last_error(ctx)       # few points: the host-side check
    assert lib.bpmi_ec_sum(ctx, good, 200, o64) == 0
    # inner-product prover: g, h and u
    n = 128
    a = cbind.pack_scalars(list(range(2, n + 2)))
    st = ctypes.c_void_p()
    g, h, u = good[: 64 * n], good[64 * 60: 64 * (60 + n)], good[64 * 199: 64 * 200]
    assert lib.bpmi_ipa_create(ctx, g[: 64 * 17] + bad + g[64 * 18:], h, a, a, n, u, ctypes.byref(st)) == -3 and b"g[17]" in lib.bpmi_last_error(ctx)
    assert lib.bpmi_ipa_create(ctx, g, h[: 64 * 99] + bad + h[64 * 100:], a, a, n, u, ctypes.byref(st)) == -3 and b"h[99]" in lib.bpmi_last_error(ctx)
    assert lib.bpmi_ipa_create(ctx, g, h, a, a, n, bad, ctypes.byref(st)) == -3 and b"u[0]" in lib.bpmi_last_error(ctx)
    assert lib.bpmi_ipa_create_scaled(ctx, g, h[: 64 * 5] + bad + h[64 * 6:], a, a, n, u, a, ctypes.byref(st)) == -3
    assert lib.bpmi_ipa_create(ctx, g, h, a, a, n, u, ctypes.byref(st)) == 0
    lib.bpmi_ipa_destroy(st)
    # the batched prover's generators
    pv = ctypes.c_void_p()
    gs, hs = good[: 64 * 8], good[64 * 8: 64 * 16]
    assert lib.bpmi_rp_prover_create(ctx, 8, u, u, u, gs[: 64 * 3] + bad + gs[64 * 4:], hs, ctypes.byref(pv)) == -3 and b"gs[3]" in lib.bpmi_last_error(ctx)
    assert lib.bpmi_rp_prover_create(ctx, 8, u, bad, u, gs, hs, ctypes.byref(pv)) == -3 and b"h[0]" in lib.bpmi_last_error(ctx)


def test_batch_verifier_rejects_an_off_curve_commitment(gp):
    """bpmi_rp_batch_verify_dev: a commitment that is not a point is an ARGUMENT error with its index -- neither a verdict nor a value."""
    from bulletproofs_amd.engine import EngineError
    from bulletproofs_amd.rangeproofs import BatchRangeProver, BatchRangeVerifier
    from bulletproofs_amd.utils import ModP, commitment
    from test_gpu_prove_batch import _setup
    n, count = 8, 40
    g, h, gs, hs, u = _setup(gp, n, b"vp")
    rnd = random.Random(8)
    vs = [ModP(rnd.randrange(1 << n), Q) for _ in range(count)]
    gammas = [ModP(rnd.randrange(Q), Q) for _ in range(count)]
    bp = BatchRangeProver(n, g, h, gs, hs, u)
    try:
        blobs = bp.prove_wire(vs, gammas, [b"s%d" % i for i in range(count)])
    finally:
        bp.close()
    Vb = b"".join(commitment(g, h, v, x).to_le64() for v, x in zip(vs, gammas))
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    assert bv.partial_wire(Vb, blobs) == bytes(64)
    with pytest.raises(EngineError, match=r"v_points\[23\] is not a point of the curve"):
        bv.partial_wire(Vb[: 64 * 23] + _bad_points()[0] + Vb[64 * 24:], blobs)
    assert bv.partial_wire(Vb, blobs) == bytes(64)
