#!/usr/bin/env python3
"""CPU-side baselines of BASELINE.md section 3, timed on this host:
  1. the oracle's Python restatement of the REFERENCE ALGORITHM (src/pippenger/pippenger.py:
     subset tables, same s/t/b and group-op counts) on 1 core, n in {2^7, 2^10, 2^12, 2^14};
  2. the plain-C bucket MSM (oracle/c) on the host cores, n = 2^16 (and 2^20 with --big);
  3. verification of one 64-bit range proof by the Python restatement of RangeVerifier.verify
     on 1 core: with the reference's own multiexp algorithm, and with the C bucket MSM.
Prints JSON lines.  (The reference itself cannot travel to the GPU box; its survey-time
numbers are in BASELINE.md.)"""
import json
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import bp_ref as R, cbind  # noqa: E402
from oracle.ec import secp256k1  # noqa: E402

Q = secp256k1.q
rnd = random.Random(1)
for logn in (7, 10, 12, 14):
    n = 1 << logn
    pts = cbind.ec_mul_batch([secp256k1.G] * n, [rnd.randrange(1, Q) for _ in range(n)])
    es = [rnd.randrange(Q) for _ in range(n)]
    grp = R.EC()
    t = time.perf_counter()
    got = R.Pippenger(grp).multiexp(pts, es)
    dt = time.perf_counter() - t
    assert got == cbind.msm(pts, es)
    print(json.dumps({"baseline": "reference algorithm, Python restatement, 1 core", "n": n, "seconds": dt,
                      "pairs_per_s": n / dt, "group_ops": grp.ops, "ops_per_pair": grp.ops / n}), flush=True)
for logn in ((16, 20) if "--big" in sys.argv else (16,)):
    n = 1 << logn
    ks = [rnd.randrange(1, Q) for _ in range(n)]
    pts = cbind.pack_points(cbind.ec_mul_batch([secp256k1.G] * n, ks))
    es = cbind.pack_scalars([rnd.randrange(Q) for _ in range(n)])
    for threads in (1, os.cpu_count() or 1):
        t = time.perf_counter()
        cbind.msm_bytes(pts, es, n, threads)
        dt = time.perf_counter() - t
        print(json.dumps({"baseline": "plain-C bucket MSM (oracle/c)", "n": n, "threads_requested": threads,
                          "seconds": dt, "pairs_per_s": n / dt}), flush=True)

# 3. one 64-bit range proof (C1 / C5 shape), verified on the CPU
nbits = 64
gs = [R.elliptic_hash(b"g%d" % i) for i in range(nbits)]
hs = [R.elliptic_hash(b"h%d" % i) for i in range(nbits)]
g, h, u = R.elliptic_hash(b"g"), R.elliptic_hash(b"h"), R.elliptic_hash(b"u")
v, gamma = R.Zq(rnd.randrange(1 << nbits), Q), R.Zq(rnd.randrange(Q), Q)
V = R.commitment(g, h, v, gamma)
proof = R.range_prove(v, nbits, g, h, gs, hs, gamma, u, Q, b"seed", multiexp=cbind.msm)
for name, mexp, reps in (("reference multiexp algorithm (Python restatement)", None, 2), ("C bucket MSM for the multiexps", cbind.msm, 5)):
    t = time.perf_counter()
    for _ in range(reps):
        assert R.range_verify(V, g, h, gs, hs, u, proof, multiexp=mexp)
    dt = (time.perf_counter() - t) / reps
    print(json.dumps({"baseline": "RangeVerifier.verify restatement, 64-bit proof, 1 core; " + name,
                      "seconds_per_verify": dt, "verifies_per_s": 1 / dt}), flush=True)
