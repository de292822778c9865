"""Large MSMs as slices (round 6): an input of more than slice_min pairs runs as K equal slices of about slice_n pairs, two in
flight, the slices' results added on the host (csrc/msm_host.hpp msm_run_sliced).  The reference's multiexp takes any N
(/root/reference/src/pippenger/pippenger.py:22-61) and its verifier calls it with 2n + 1 pairs
(/root/reference/src/innerproduct/inner_product_verifier.py:134-139).  Checked here: the slice boundaries (n = slice_min - 1,
slice_min, K slices + 5, a last slice of ONE pair) against the C oracle with a small slice_n, segments that straddle a slice
boundary, and the default geometry at 2^21 + 1 pairs through a size-independent property."""
import ctypes
import random

import pytest

from helpers import Q
from oracle import cbind

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


@pytest.fixture()
def small_slices(gp):
    eng = gp.engine()
    eng.set_option("slice_n", 1 << 16)            # slice_min = 106 496, a slice may be 69 632 pairs
    yield eng
    eng.set_option("slice_n", 0)
    eng.set_option("slice_min", 0)


CAP = (1 << 16) + (1 << 12)


@pytest.mark.parametrize("n", [106495, 106496, 2 * CAP, 2 * CAP + 1, 2 * CAP + 5, 3 * CAP + 2, 200001])
def test_sliced_msm_boundaries_vs_oracle(gp, small_slices, n):
    """2 CAP: two full slices; 2 CAP + 1: three slices of ceil(n / 3) (the last one shorter); 200 001: three."""
    eng = small_slices
    D = 1 << 11
    pts, _ = gp.rand_points(D, 77)
    rnd = random.Random(n)
    es = [rnd.randrange(Q) for _ in range(n)]
    es[0], es[1], es[n - 1] = 0, 1, Q - 1
    tiled = cbind.pack_points(pts) * (n // D + 1)
    pb, sb = tiled[: 64 * n], cbind.pack_scalars(es)
    folded = [0] * D
    for i, e in enumerate(es):
        folded[i % D] = (folded[i % D] + e) % Q
    want = cbind.msm_bytes(cbind.pack_points(pts), cbind.pack_scalars(folded), D)
    assert eng.msm_bytes(pb, sb, n) == want
    d_p, d_s = eng.upload(pb), eng.upload(sb)
    try:
        assert eng.msm_dev(d_p, d_s, n) == want
        eng.set_option("slice_n", -1)                       # ONE MSM: the path the slices replace
        assert eng.msm_dev(d_p, d_s, n) == want
        eng.set_option("slice_n", 1 << 16)
        eng.set_option("slice_min", n + 1)                   # below slice_min: one MSM again
        assert eng.msm_dev(d_p, d_s, n) == want
        eng.set_option("slice_min", 0)
    finally:
        d_p.free()
        d_s.free()


def test_sliced_msm_last_slice_of_one_pair(gp, small_slices):
    """slice_min forced low: n = 2 per + 1 with per = ceil(n / 3) ... the geometry that leaves ONE pair to the last slice is
    n = K (per - 1) + 1 for K = 3: per = 27 308 -> n = 81 922 gives slices 27 308 / 27 308 / 27 306; the degenerate split is
    forced with slice_n = 2^16 and n = 2 CAP + 1 above.  Here: every scalar 0 or 1 (empty buckets everywhere) across slices."""
    eng = small_slices
    n = 2 * CAP + 3
    D = 1 << 10
    pts, ks = gp.rand_points(D, 5)
    rnd = random.Random(9)
    es = [rnd.choice((0, 1, Q - 1)) for _ in range(n)]
    folded = [0] * D
    for i, e in enumerate(es):
        folded[i % D] = (folded[i % D] + e) % Q
    tiled = cbind.pack_points(pts) * (n // D + 1)
    want = cbind.msm_bytes(cbind.pack_points(pts), cbind.pack_scalars(folded), D)
    assert eng.msm_bytes(tiled[: 64 * n], cbind.pack_scalars(es), n) == want


def test_sliced_msm_segments_straddle_slices(gp, small_slices):
    """bpmi_msm_segs_dev: three device arrays of 50 000 + 70 001 + 30 000 pairs = two slices of 75 001 / 75 000 pairs; the first
    slice ends inside the second array."""
    eng = small_slices
    lib, ctx = eng.lib, eng.ctx
    ns = [50000, 70001, 30000]          # 150 001 >= slice_min
    D = 1 << 10
    pts, _ = gp.rand_points(D, 31)
    small = cbind.pack_points(pts)
    rnd = random.Random(2)
    bufs, folded, off = [], [0] * D, 0
    allp, alls = b"", b""
    for m in ns:
        es = [rnd.randrange(Q) for _ in range(m)]
        # segment k starts at point (off mod D) of the tile
        pb = (small * (m // D + 2))[64 * (off % D): 64 * (off % D) + 64 * m]
        for i, e in enumerate(es):
            folded[(off + i) % D] = (folded[(off + i) % D] + e) % Q
        sb = cbind.pack_scalars(es)
        bufs.append((eng.upload(pb), eng.upload(sb)))
        allp += pb
        alls += sb
        off += m
    want = cbind.msm_bytes(small, cbind.pack_scalars(folded), D)
    P = (ctypes.c_void_p * 3)(*[b[0].ptr for b in bufs])
    S = (ctypes.c_void_p * 3)(*[b[1].ptr for b in bufs])
    N = (ctypes.c_uint64 * 3)(*ns)
    out = ctypes.create_string_buffer(64)
    try:
        assert lib.bpmi_msm_segs_dev(ctx, 3, P, S, N, out) == 0
        assert out.raw == want
        assert eng.msm_bytes(allp, alls, sum(ns)) == want
    finally:
        for b in bufs:
            b[0].free()
            b[1].free()


def test_sliced_msm_refuses_while_a_slot_is_pending(gp, small_slices):
    eng = small_slices
    n = 120000
    pts, _ = gp.rand_points(256, 3)
    pb = cbind.pack_points(pts) * (n // 256 + 1)
    d_p, d_s = eng.upload(pb[: 64 * n]), eng.upload(bytes(32 * n))
    try:
        eng.msm_dev_enqueue(1, d_p, d_s, 1000)
        out = ctypes.create_string_buffer(64)
        assert eng.lib.bpmi_msm_dev(eng.ctx, d_p.ptr, d_s.ptr, n, out) == -5          # BPMI_E_STATE: slot 1 is the caller's
        assert eng.msm_finish(1) == bytes(64)
        assert eng.msm_dev(d_p, d_s, n) == bytes(64)
    finally:
        d_p.free()
        d_s.free()


def test_sliced_msm_default_geometry_2e21_plus_1(gp):
    """The IPA verifier's size at config C3 (2n + 1 pairs, n = 2^20) with the DEFAULT options: two slices of 2^20 + 1 / 2^20 pairs.
    Size-independent check: points tiled from D distinct ones, MSM(tiled, e) == MSM(distinct, column sums of e mod q)."""
    import numpy as np
    eng = gp.engine()
    D = 1 << 12
    n = (1 << 21) + 1
    pts, _ = gp.rand_points(D, 99)
    small = cbind.pack_points(pts)
    rng = np.random.default_rng(21)
    reps = n // D + 1
    e = rng.integers(0, 1 << 32, size=(reps, D, 8), dtype=np.uint64).astype(np.uint32)
    e[:, :, 7] &= 0x7FFFFFFF
    e = e.reshape(reps * D, 8)
    e[n:] = 0
    col = e.reshape(reps, D, 8).astype(np.uint64).sum(axis=0)
    folded = []
    for j in range(D):
        v = 0
        for k in range(7, -1, -1):
            v = (v << 32) + int(col[j, k])
        folded.append(v % Q)
    d_pts = eng.alloc(64 * n)
    tile = small * 64
    for r in range(0, reps, 64):
        cnt = min(64 * D, n - r * D)
        if cnt > 0:
            d_pts.upload(tile[: 64 * cnt], 64 * D * r)
    d_e = eng.upload(e[:n].tobytes())
    try:
        want = cbind.msm_bytes(small, cbind.pack_scalars(folded), D)
        got = eng.msm_dev(d_pts, d_e, n)
        assert got == want
        eng.set_option("slice_n", -1)
        assert eng.msm_dev(d_pts, d_e, n) == want
    finally:
        eng.set_option("slice_n", 0)
        d_pts.free()
        d_e.free()
