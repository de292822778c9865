#!/usr/bin/env python3
"""Where the wall time of one batched proving call goes beyond the device's (round 6): BatchRangeProver.prove_wire_packed on 2^14
64-bit proofs with the inputs as a service holds them (packed bytes, seeds joined + offsets), the offsets as a Python list or as a
ctypes array, the result copied into a bytes object (default) or left in the prover's page-locked buffer (copy=False).
    python tools/r06_prover_host_breakdown.py [log2 batch]"""
import ctypes, hashlib, os, sys, time
from itertools import accumulate
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import Point, secp256k1
from bulletproofs_amd.rangeproofs import BatchRangeProver
Q = secp256k1.q
eng = default_engine()
n = 64
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 14
m = 1 << lg
ks = b"".join((int.from_bytes(hashlib.sha256(b"pb/%d" % i).digest(), "big") % Q).to_bytes(32, "little") for i in range(2 * n + 3))
out = eng.ec_mul_batch_bytes(secp256k1.G.to_le64() * (2 * n + 3), ks, 2 * n + 3)
pts = [Point.from_le64(out[64 * i: 64 * i + 64]) for i in range(2 * n + 3)]
g, h, u, gs, hs = pts[0], pts[1], pts[2], pts[3:3 + n], pts[3 + n:]
vb = b"".join((int.from_bytes(hashlib.sha256(b"v%d" % i).digest()[:8], "big")).to_bytes(32, "little") for i in range(m))
gb = b"".join((int.from_bytes(hashlib.sha256(b"g%d" % i).digest(), "big") % Q).to_bytes(32, "little") for i in range(m))
seeds = [b"seed-%d" % i for i in range(m)]
sb, offs = b"".join(seeds), [0, *accumulate(map(len, seeds))]
offs_c = (ctypes.c_uint64 * (m + 1))(*offs)
for fmt in (2, 3):
    bp = BatchRangeProver(n, g, h, gs, hs, u, wire_format=fmt)
    for name, o, copy in (("offsets list, bytes out ", offs, True), ("offsets ctypes, bytes out", offs_c, True), ("offsets ctypes, view out ", offs_c, False)):
        bp.prove_wire_packed(vb, gb, (sb, o), copy=copy)
        best, dev = None, None
        for _ in range(5):
            t = time.perf_counter()
            bp.prove_wire_packed(vb, gb, (sb, o), copy=copy)
            dt = time.perf_counter() - t
            if best is None or dt < best:
                best, dev = dt, bp.last_ms()["total"]
        print("format %d  %s  wall %.2f ms  device %.2f ms  -> %.3g proofs/s wall" % (fmt, name, best * 1e3, dev, m / best), flush=True)
    bp.close()
