#!/usr/bin/env python3
"""Randomised differential test of the HIP MSM against the C oracle: random sizes, scalar
shapes (uniform, tiny, near q/2 and q, repeated, zero), point shapes (duplicates, negated
pairs, identities) and engine options (window bits, chunk length, tail placement, window-group
split, small-MSM threshold, GLV split on / off, fused / unfused wave scan, the round-5 and round-6 options); a few percent of the cases are large enough for the LDS sort path.
  python tools/fuzz_msm.py [seconds]"""
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.engine import default_engine  # noqa: E402
from oracle import cbind  # noqa: E402
from oracle.ec import INF, secp256k1  # noqa: E402

Q = secp256k1.q
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(os.environ.get("FUZZ_SEED", "12345"))
rnd = random.Random(seed)
eng = default_engine()
pool = cbind.ec_mul_batch([secp256k1.G] * 4096, [rnd.randrange(1, Q) for _ in range(4096)])
half = (Q - 1) // 2


def scalar(kind):
    if kind == 0:
        return rnd.randrange(Q)
    if kind == 1:
        return rnd.randrange(1 << rnd.choice((1, 8, 16, 20, 64, 128)))
    if kind == 2:
        return (half + rnd.randrange(-3, 4)) % Q
    if kind == 3:
        return (Q - 1 - rnd.randrange(0, 4)) % Q
    if kind == 4:
        return 0
    return rnd.choice((1, 2, Q - 1, 1 << 255, (1 << 255) - 1, 1 << 240))


t0 = time.time()
cases = fails = 0
while time.time() - t0 < budget:
    n = rnd.choice((1, 2, 3, rnd.randrange(1, 64), rnd.randrange(1, 1500), rnd.randrange(1, 6000)))
    r = rnd.random()
    if r < 0.04:
        n = rnd.randrange(30000, 90000)            # default options: c = 16, LDS sort, 4096-scalar level-A tiles
    elif r < 0.05:
        n = rnd.randrange(1 << 19, (1 << 19) + 70000)   # 16384-scalar level-A tiles, L = 64
    elif r < 0.06:
        n = rnd.randrange((1 << 17) - 3000, (1 << 17) + 3000)      # either side of the in-block sort's bound
    elif r < 0.12:
        n = rnd.randrange(5000, 30000)             # the one-block kernel's end, 12-bit and 13-bit mixed-width windows (round 5)
    elif r < 0.13:
        n = rnd.randrange(180000, 190000)          # 13 / 16 bits
    pts = [rnd.choice(pool) for _ in range(n)]
    shape = rnd.randrange(6)
    if shape == 1:
        pts = [pts[0]] * n
    elif shape == 2:
        pts = [p if i % 2 else -pts[i - 1] for i, p in enumerate(pts)]
    elif shape == 3:
        pts = [INF if rnd.random() < 0.3 else p for p in pts]
    kind = rnd.randrange(7)
    es = [scalar(kind if kind < 6 else rnd.randrange(6)) for _ in range(n)]
    if rnd.random() < 0.2:
        es = [es[0]] * n
    opts = {"window_bits": rnd.choice((0, 0, 0, 2, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 15, 16)),
            "chunk": rnd.choice((0, 0, 1, 2, 5, 16, 33, 64, 200)), "tail": rnd.choice((0, 1, 2)),
            "split": rnd.choice((0, 0, 0, 1)), "small_n": rnd.choice((0, 0, -1, 100, 65536)),
            "glv": rnd.choice((0, 0, 1, 1, -1)),          # 1: the GLV split at every size the bucket pipeline takes, -1: never
            "fused_scan": rnd.choice((1, 1, 1, 0)),       # 0: two partial records per thread and k_segscan's first level (round 3)
            "direct_result": rnd.choice((1, 1, 0)), "graphs": rnd.choice((0, 0, 1)),
            "mid_single_min": rnd.choice((0, 0, 1, 300, -1)),   # k_msm_mid for single MSMs from that many pairs (0: default 2560, -1: never)
            # round 5: the unsigned last window of c = 15, level B of the sort on partitions of any size, the segmented scan's fused last level
            "top_window_unsigned": rnd.choice((1, 1, 0)), "sort_inblock": rnd.choice((1, 1, 0)), "segscan_fused": rnd.choice((0, 0, 1)),
            "hist_scan_fused": rnd.choice((0, 0, 1)), "final_spread": rnd.choice((3, 3, 2, 1, 0)), "reduce_fit": rnd.choice((1, 1, 0)), "mixed_windows": rnd.choice((1, 1, 0)), "mid_parts": rnd.choice((0, 0, 1, 2, 3, 4)),
            "reduce_epl": rnd.choice((0, 0, 0, 1, 3, 7, 11, 13, 20)),
            # round 6: the stages' wave priority (mask), slices of a large input (65 536: inputs from 106 496 pairs run as slices), rounds of the accumulation
            "priority": rnd.choice((1, 1, 0, 17, 26, 28, 31)), "slice_n": rnd.choice((0, 0, 0, 65536, -1)), "rounds": rnd.choice((0, 0, 1, 2, 5))}
    if n > 20000:
        opts["window_bits"] = rnd.choice((0, 0, 0, 10, 11, 12, 13, 13, 14, 15, 16))       # (10 .. 15: mixed window widths unless mixed_windows drew 0)
        opts["chunk"] = rnd.choice((0, 0, 16, 64))
    for k, v in opts.items():
        eng.set_option(k, v)
    pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
    got = eng.msm_bytes(pb, sb, n)
    want = cbind.msm_bytes(pb, sb, n, 4)
    cases += 1
    if got != want:
        fails += 1
        print("MISMATCH n=%d shape=%d kind=%d opts=%s seed=%d case=%d" % (n, shape, kind, opts, seed, cases), flush=True)
for k in ("window_bits", "chunk", "tail", "split", "small_n", "glv", "mid_single_min", "slice_n", "rounds"):
    eng.set_option(k, 0)
eng.set_option("priority", 1)
for k, v in (("top_window_unsigned", 1), ("sort_inblock", 1), ("segscan_fused", 0), ("hist_scan_fused", 0), ("final_spread", 3), ("reduce_fit", 1), ("mixed_windows", 1), ("mid_parts", 0), ("reduce_epl", 0)):
    eng.set_option(k, v)
eng.set_option("fused_scan", 1)
eng.set_option("direct_result", 1)
eng.set_option("graphs", 0)
print("fuzz: %d cases, %d mismatches, %.0f s, seed %d" % (cases, fails, time.time() - t0, seed))
sys.exit(1 if fails else 0)
