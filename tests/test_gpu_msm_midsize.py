"""GPU parity for the round-5 changes to the bucket pipeline at mid sizes (Pippenger.multiexp,
/root/reference/src/pippenger/pippenger.py:22-61): the unsigned last window of c = 15 (17 windows, no carry window), level B of
the sort handling partitions of any size itself up to 2^17 pairs, and the two "last block done" fusions (the segmented scan's last
level, the scan of the sort's partition counts: measured, lost to the cost of their device-scope fences, off by default).  Every
case is compared with the C oracle, and with the other setting of the same option."""
import random

import pytest

from helpers import Q
from oracle import cbind

pytestmark = pytest.mark.gpu

R5_DEFAULTS = {"top_window_unsigned": 1, "sort_inblock": 1, "segscan_fused": 0, "hist_scan_fused": 0, "final_spread": 3, "reduce_fit": 1, "mixed_windows": 1}


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


def _scalars(shape, n, rnd):
    half = (Q - 1) // 2
    if shape == "uniform":
        return [rnd.randrange(Q) for _ in range(n)]
    if shape == "top_window_edges":
        # the last window of c = 15 holds bits 240 .. 254 of min(s, q - s): 0x7FFF there with a carry from below is the digit 2^15 = 2B
        # (only next to q / 2); 2^14 = B and B + 1 sit either side of what a signed window would have recoded
        vals = [half, half - 1, half + 1, half - (1 << 200), half + (1 << 200) + 5, (1 << 254), (1 << 254) + (1 << 253), (1 << 254) - 1,
                (1 << 240), (1 << 240) - 1, (0x4000 << 240) | 12345, (0x4001 << 240), (0x3FFF << 240) | ((1 << 240) - 1), Q - 1, Q - 2, 1, 0,
                Q - half, (0x7FFF << 240) % Q, ((0x7FFE << 240) | ((1 << 240) - 1)) % Q]
        return [vals[rnd.randrange(len(vals))] if i % 3 else rnd.randrange(Q) for i in range(n)]
    if shape == "all_same":
        return [rnd.randrange(Q)] * n
    if shape == "two_values":
        vals = [12345678901234567890123, Q - 5]
        return [vals[rnd.randrange(2)] for _ in range(n)]
    if shape == "bits01":
        return [rnd.randrange(2) for _ in range(n // 2)] + [(rnd.randrange(2) - 1) % Q for _ in range(n - n // 2)]
    if shape == "bits_and_blinding":
        es = [rnd.randrange(2) for _ in range(n // 2)] + [(rnd.randrange(2) - 1) % Q for _ in range(n - n // 2)]
        for i in range(0, n, 4099):
            es[i] = rnd.randrange(Q)
        return es
    if shape == "top_heavy":                  # every scalar in one bucket of the LAST window, spread below it
        return [(0x5A5A << 240) | rnd.randrange(1 << 240) for _ in range(n)]
    if shape == "small_range":
        return [rnd.randrange(1 << 20) if i % 3 else rnd.randrange(Q) for i in range(n)]
    raise ValueError(shape)


def _reset(eng):
    for o, v in R5_DEFAULTS.items():
        eng.set_option(o, v)
    eng.set_option("window_bits", 0)
    eng.set_option("chunk", 0)
    eng.set_option("tail", 0)
    eng.set_option("fused_scan", 1)
    eng.set_option("reduce_epl", 0)
    eng.set_option("hist_threads", 0)
    eng.set_option("hist_blocks", 0)


@pytest.mark.parametrize("n,c", [(10240, 0), (11000, 12), (40000, 16), (40000, 15), (70001, 13), (300000, 16), (1 << 20, 0)])
def test_msm_partition_count_scan_in_the_histogram_kernels_last_block(gp, n, c):
    """k_coarse_hist's last block scans the partition counts itself (hist_scan_fused; 1 .. 256 blocks of 256 / 512 / 1024 threads: 8, 4 or 2
    partitions per thread in the scan) against the separate k_coarse_scan launch and the oracle."""
    eng = gp.engine()
    pts, _ = gp.rand_points(1000, 43)
    pts = (pts * (n // 1000 + 1))[:n]
    es = _scalars("uniform" if n % 2 else "bits_and_blinding", n, random.Random(n + c))
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    want = cbind.msm_bytes(pb, sb, n)
    try:
        eng.set_option("window_bits", c)
        for threads, blocks in ((0, 0), (256, 0), (512, 7), (1024, 1)):
            eng.set_option("hist_threads", threads)
            eng.set_option("hist_blocks", blocks)
            for fused in (1, 0, 1):
                eng.set_option("hist_scan_fused", fused)
                assert eng.msm_bytes(pb, sb, n) == want, (threads, blocks, fused)
    finally:
        _reset(eng)


@pytest.mark.parametrize("c", [10, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("n", [11000, 40000, 70001])
def test_msm_bucket_reduction_group_sizes(gp, n, c):
    """Stage 1 of the bucket reduction with 1 .. 64 lanes per sum (reduce_epl) -- any number, not only powers of two: a wave holds
    64 / lanes sums and the rest of its lanes idle -- the unsigned last window's own job set included (c = 15), against the oracle."""
    eng = gp.engine()
    pts, _ = gp.rand_points(1000, 41)
    pts = (pts * (n // 1000 + 1))[:n]
    es = _scalars("uniform" if n != 40000 else "top_window_edges", n, random.Random(n + c))
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    want = cbind.msm_bytes(pb, sb, n)
    try:
        eng.set_option("window_bits", c)
        for epl in (0, 1, 3, 5, 7, 11, 13, 16, 20, 33, 64):     # 64, 64, 32, 21, 12, 10, 8, 7, 4, 2 lanes per sum of 128 elements
            eng.set_option("reduce_epl", epl)
            assert eng.msm_bytes(pb, sb, n) == want, epl
        eng.set_option("reduce_epl", 0)
        eng.set_option("reduce_fit", 0)                         # round 4's rule for the lanes per sum
        assert eng.msm_bytes(pb, sb, n) == want
    finally:
        _reset(eng)


@pytest.mark.parametrize("shape", ["uniform", "top_window_edges", "all_same", "bits_and_blinding", "top_heavy"])
@pytest.mark.parametrize("n", [12000, 40000, 70001])
def test_msm_c15_unsigned_last_window(gp, shape, n):
    """c = 15 with the unsigned last window (17 windows, G = 18 B) against the oracle and against the 18-window recoding of round 4,
    under both tails (the host tail and the device tail walk the last window's own split offsets)."""
    eng = gp.engine()
    pts, _ = gp.rand_points(n, 31 + n)
    es = _scalars(shape, n, random.Random(n + len(shape)))
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    want = cbind.msm_bytes(pb, sb, n)
    try:
        eng.set_option("window_bits", 15)
        for top2 in (1, 0):
            eng.set_option("top_window_unsigned", top2)
            for tail in (2, 1):
                eng.set_option("tail", tail)
                assert eng.msm_bytes(pb, sb, n) == want, (top2, tail)
        eng.set_option("top_window_unsigned", 1)
        eng.set_option("tail", 0)
        eng.set_option("fused_scan", 0)                      # two records per thread: the multi-level segmented scan
        assert eng.msm_bytes(pb, sb, n) == want
        eng.set_option("fused_scan", 1)
        eng.set_option("chunk", 3)                            # many chunks per bucket, many records
        assert eng.msm_bytes(pb, sb, n) == want
    finally:
        _reset(eng)


def _digit_edge_scalars(c, n, rnd):
    """Scalars assembled window by window (the mixed layout of c) from the digits at the edges of the signed recoding: 0, 1, B - 1, B,
    B + 1, all ones -- for the narrow and for the wide windows -- and random ones; every second one negated."""
    W = 256 // c
    Wb = 256 - W * c
    widths = [c] * (W - Wb) + [c + 1] * Wb
    out = []
    for i in range(n):
        s, pos = 0, 0
        for wd in widths:
            B = 1 << (wd - 1)
            s |= rnd.choice((0, 1, B - 1, B, B + 1, (1 << wd) - 1, rnd.randrange(1 << wd), rnd.randrange(1 << wd))) << pos
            pos += wd
        s %= Q
        out.append(s if i % 2 else (Q - s) % Q)
    return out


@pytest.mark.parametrize("c", [10, 11, 12, 13, 14, 15])
@pytest.mark.parametrize("shape", ["uniform", "digit_edges", "top_window_edges", "all_same", "bits_and_blinding", "top_heavy"])
@pytest.mark.parametrize("n", [12000, 40000, 140000])
def test_msm_mixed_window_widths(gp, shape, n, c):
    """Window bits c as 256 // c windows of which the last 256 - (256 // c) c are c + 1 bits wide with twice the buckets (no carry
    window, no short top window): against the oracle and against the uniform recoding with its carry window, under both tails,
    with the unfused scan and with short chunks; every stage-1 group size of the wide windows' own job set."""
    eng = gp.engine()
    pts, _ = gp.rand_points(1000, 53 + c)
    pts = (pts * (n // 1000 + 1))[:n]
    es = _digit_edge_scalars(c, n, random.Random(n * 5 + c)) if shape == "digit_edges" else _scalars(shape, n, random.Random(n * 5 + c))
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    want = cbind.msm_bytes(pb, sb, n)
    try:
        eng.set_option("window_bits", c)
        for mixed in (1, 0):
            eng.set_option("mixed_windows", mixed)
            for tail in (2, 1):
                eng.set_option("tail", tail)
                assert eng.msm_bytes(pb, sb, n) == want, (mixed, tail)
        eng.set_option("mixed_windows", 1)
        eng.set_option("tail", 0)
        eng.set_option("fused_scan", 0)
        assert eng.msm_bytes(pb, sb, n) == want
        eng.set_option("fused_scan", 1)
        eng.set_option("chunk", 3)
        assert eng.msm_bytes(pb, sb, n) == want
        eng.set_option("chunk", 0)
        for epl in (1, 5, 11, 33):
            eng.set_option("reduce_epl", epl)
            assert eng.msm_bytes(pb, sb, n) == want, epl
        eng.set_option("reduce_epl", 0)
        for fs in (0, 1, 2):
            eng.set_option("final_spread", fs)
            assert eng.msm_bytes(pb, sb, n) == want, fs
        eng.set_option("quad_final", 0)
        assert eng.msm_bytes(pb, sb, n) == want
    finally:
        eng.set_option("quad_final", 1)
        _reset(eng)


@pytest.mark.parametrize("c", [12, 13, 15, 16])
@pytest.mark.parametrize("shape", ["all_same", "two_values", "bits01", "bits_and_blinding", "small_range", "top_heavy"])
@pytest.mark.parametrize("n", [30000, 131072, 131073])
def test_msm_sort_level_b_any_partition_size(gp, shape, n, c):
    """Up to 2^17 pairs k_fine_sort_part sorts heavy partitions itself (no staging buffer, wave-aggregated LDS slots) and the two
    heavy-tile kernels are not launched; one pair more and the tile kernels run.  Both against the oracle and each other."""
    eng = gp.engine()
    pts, _ = gp.rand_points(1024, 77)
    pts = (pts * (n // 1024 + 1))[:n]                         # (repeated points: the oracle's cost is the scalars')
    es = _scalars(shape, n, random.Random(n * 7 + c))
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    want = cbind.msm_bytes(pb, sb, n)
    try:
        eng.set_option("window_bits", c)
        for inblock in (1, 0):
            eng.set_option("sort_inblock", inblock)
            assert eng.msm_bytes(pb, sb, n) == want, inblock
    finally:
        _reset(eng)


@pytest.mark.parametrize("n,chunk,c", [(3000, 1, 16), (20000, 1, 12), (40000, 2, 16), (40000, 1, 15), (65536, 0, 0), (100000, 5, 13), (300000, 3, 16)])
@pytest.mark.parametrize("shape", ["uniform", "all_same", "bits_and_blinding"])
def test_msm_segscan_last_level_in_the_last_block(gp, shape, n, chunk, c):
    """One to 128 blocks in the first level of k_segscan: the block that finishes last runs the final level (ticket in the MSM's
    zeroed words); more than 128 blocks: two launches as before.  Fused and unfused wave scan (2 records per wave / per thread)."""
    eng = gp.engine()
    pts, _ = gp.rand_points(1000, 5)
    pts = (pts * (n // 1000 + 1))[:n]
    es = _scalars(shape, n, random.Random(n * 31 + chunk))
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    want = cbind.msm_bytes(pb, sb, n)
    try:
        eng.set_option("chunk", chunk)
        eng.set_option("window_bits", c)
        for fused_scan in (1, 0):
            eng.set_option("fused_scan", fused_scan)
            for seg in (1, 0, 1):
                eng.set_option("segscan_fused", seg)
                assert eng.msm_bytes(pb, sb, n) == want, (fused_scan, seg)
    finally:
        _reset(eng)


@pytest.mark.parametrize("c", [10, 11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize("n,shape", [(11000, "uniform"), (40000, "top_window_edges"), (70001, "bits_and_blinding"), (300000, "uniform")])
def test_msm_bucket_reduction_finish_spread_over_waves(gp, n, shape, c):
    """The finish of the bucket reduction with its 16 second-level sums per (window, array) as waves of their own
    (k_digit_final_spread: two launches with blocks of four waves -- the default -- or of one, or one launch with a ticket per array)
    against the 16-wave block per array (k_digit_final_quad), the one-lane k_digit_final and the oracle; called again and again on
    one workspace (the tickets go back to zero), both tails, the unsigned last window of c = 15 included."""
    eng = gp.engine()
    pts, _ = gp.rand_points(1000, 47)
    pts = (pts * (n // 1000 + 1))[:n]
    es = _scalars(shape, n, random.Random(n * 3 + c))
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    want = cbind.msm_bytes(pb, sb, n)
    try:
        eng.set_option("window_bits", c)
        for spread in (3, 1, 1, 0, 2, 3, 2, 1):               # 1: tickets; 2, 3 (default): two launches, sums in blocks of one wave / of four
            eng.set_option("final_spread", spread)
            for tail in (2, 1):
                eng.set_option("tail", tail)
                assert eng.msm_bytes(pb, sb, n) == want, (spread, tail)
        eng.set_option("tail", 0)
        eng.set_option("quad_final", 0)
        assert eng.msm_bytes(pb, sb, n) == want
        eng.set_option("quad_final", 1)
        handles = [eng.msm_bytes(pb, sb, n) for _ in range(5)]
        assert handles == [want] * 5
    finally:
        eng.set_option("quad_final", 1)
        _reset(eng)


def test_msm_default_geometry_sizes_around_the_switches(gp):
    """Either side of every size at which the default geometry changes (window bits, the in-block sort bound), edge scalars mixed in."""
    eng = gp.engine()
    pts, _ = gp.rand_points(2048, 9)
    edge = [0, 1, Q - 1, (Q - 1) // 2, (Q + 1) // 2, 1 << 254, (1 << 255) % Q, (1 << 15) - 1, 1 << 14, (0x7FFF << 240) % Q]
    for n in (5631, 5632, 8448, 8449, 15359, 15360, 18999, 19000, 32767, 32768, 65535, 65536, 65537, 131071, 131072, 131073, 184999, 185000, 200000, 262144, 262145):
        rnd = random.Random(n)
        es = [edge[rnd.randrange(len(edge))] if i % 7 == 0 else rnd.randrange(Q) for i in range(n)]
        p = (pts * (n // 2048 + 1))[:n]
        pb, sb = cbind.pack_points(p), cbind.pack_scalars(es)
        assert eng.msm_bytes(pb, sb, n) == cbind.msm_bytes(pb, sb, n), n
