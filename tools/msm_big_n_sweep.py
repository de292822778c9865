#!/usr/bin/env python3
"""Rate of ONE synchronous bpmi_msm_dev call from 2^20 to 2^24 pairs, as one MSM (slice_n = -1: the path before round 6) and as
slices (round 6, csrc/msm_host.hpp msm_run_sliced), for several slice sizes:
  python tools/msm_big_n_sweep.py [--max-log 24] [--slices 0,524288,2097152] > profiles/r06_msm_big_n.txt
Inputs: DISTINCT points k_i G (bpmi_ec_mul_batch_dev), uniform 255-bit scalars.  Check: every geometry gives the same 64 bytes, and
up to 2^22 pairs those bytes equal (sum e_i k_i mod q) G computed by a different kernel."""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.ec import secp256k1  # noqa: E402
from bulletproofs_amd.engine import default_engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--max-log", type=int, default=24)
ap.add_argument("--slices", default="0", help="comma-separated slice_n values to measure beside -1 (0 = the default, 2^20)")
ap.add_argument("--reps", type=int, default=0)
args = ap.parse_args()
Q = secp256k1.q
eng = default_engine()
nmax = 1 << args.max_log
rng = np.random.default_rng(2026)


def rand255(n):
    a = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
    a[:, 7] &= 0x7FFFFFFF          # < 2^255 < q
    return a


def to_ints(a):
    b = a.tobytes()
    return [int.from_bytes(b[32 * i: 32 * i + 32], "little") for i in range(len(a))]


t0 = time.time()
ks, es = rand255(nmax), rand255(nmax)
d_k = eng.upload(ks.tobytes())
d_G = eng.upload(secp256k1.G.to_le64() * nmax)
d_p = eng.alloc(64 * nmax)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, nmax, d_p.ptr))
eng.sync()
d_G.free()
d_k.free()
d_s = eng.upload(es.tobytes())
print("# %d distinct points + scalars resident after %.1f s" % (nmax, time.time() - t0), flush=True)
known_max = min(nmax, 1 << 22)
kk, ee = to_ints(ks[:known_max]), to_ints(es[:known_max])
G64 = secp256k1.G.to_le64()

sizes = []
for lg in range(20, args.max_log + 1):
    sizes.append(1 << lg)
    if lg < args.max_log:
        if lg == 20:
            sizes += [(1 << 20) + (1 << 18) - 1, (1 << 20) + (1 << 18), (1 << 20) + (1 << 19), (1 << 20) + 3 * (1 << 18)]
        if lg == 21:
            sizes.append((1 << 21) + 1)
        if 21 <= lg <= 22:
            sizes.append(3 << (lg - 1))
slice_opts = ["-1"] + args.slices.split(",")        # "slice_n" or "slice_n:slice_min"
print("# one synchronous bpmi_msm_dev per measurement; ms = mean of the timed calls; slice_n = -1: one MSM (slices of 2^23 beyond 2^23)")
print("%10s  %s" % ("n", "  ".join("slice %-13s" % v for v in slice_opts)))
for n in sizes:
    want = None
    if n <= known_max:
        dl = sum(e * k for e, k in zip(ee[:n], kk[:n])) % Q
        want = eng.ec_mul_batch_bytes(G64, dl.to_bytes(32, "little"), 1)
    reps = args.reps or max(4, min(40, (1 << 26) // n))
    cells, ref = [], None
    best = {}
    for rnd in range(2):                       # every geometry twice, alternating; the better of the two means
        for sl in slice_opts:
            eng.set_option("slice_n", int(sl.split(":")[0]))
            eng.set_option("slice_min", int(sl.split(":")[1]) if ":" in sl else 0)
            r = eng.msm_dev(d_p, d_s, n)
            ref = ref or r
            ok = (r == ref) and (want is None or r == want)
            t_h = time.perf_counter()            # ~60 ms of this very call first: the clocks (tools/step_ramp.py)
            while time.perf_counter() - t_h < 0.06:
                eng.msm_dev(d_p, d_s, n)
            t = time.perf_counter()
            for _ in range(reps):
                eng.msm_dev(d_p, d_s, n)
            dt = (time.perf_counter() - t) / reps
            if sl not in best or dt < best[sl][0]:
                best[sl] = (dt, ok and best.get(sl, (0, True))[1])
    for sl in slice_opts:
        dt, ok = best[sl]
        cells.append("%7.3f ms %.3e%s" % (dt * 1e3, n / dt, "" if ok else " MISMATCH"))
    print("%10d  %s   (%s)" % (n, "  ".join(cells), "known answer" if want else "geometries agree"), flush=True)
eng.set_option("slice_n", 0)
eng.set_option("slice_min", 0)
