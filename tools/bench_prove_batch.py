#!/usr/bin/env python3
"""Throughput of the batched range-proof prover (bpmi_rp_prove_batch): 64-bit proofs per second for a range of batch sizes, the
device milliseconds of every phase, and the one-time table build.   python tools/bench_prove_batch.py [log2 batch ...]"""
import hashlib, os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import Point, secp256k1
from bulletproofs_amd.rangeproofs import BatchRangeProver
from bulletproofs_amd.utils import ModP
Q = secp256k1.q
eng = default_engine()
n = int(os.environ.get("PB_BITS", "64"))


def points(count, seed):
    ks = b"".join((int.from_bytes(hashlib.sha256(b"pb/%d/%d" % (seed, i)).digest(), "big") % Q).to_bytes(32, "little") for i in range(count))
    out = eng.ec_mul_batch_bytes(secp256k1.G.to_le64() * count, ks, count)
    return [Point.from_le64(out[64 * i: 64 * i + 64]) for i in range(count)]


pts = points(2 * n + 3, 1)
g, h, u, gs, hs = pts[0], pts[1], pts[2], pts[3:3 + n], pts[3 + n:]
tw = int(os.environ.get("PB_TW", "0"))
eng.set_option("prover_table_bits", tw)
for kv in os.environ.get("PB_OPTS", "").split():          # engine options, e.g. PB_OPTS="prover_split=0"
    eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
t = time.perf_counter()
bp = BatchRangeProver(n, g, h, gs, hs, u)
eng.sync()
print(json.dumps({"prover_create_ms": round((time.perf_counter() - t) * 1e3, 2), "bits": n, "table_window_bits": tw or 16,
                  "table_MB": round((3 + 2 * n) * ((256 + (tw or 16) - 1) // (tw or 16)) * (1 << ((tw or 16) - 1)) * 64 / 1e6, 1)}), flush=True)
for lg in [int(a) for a in sys.argv[1:]] or [8, 10, 12, 14, 16]:
    m = 1 << lg
    vs = [int.from_bytes(hashlib.sha256(b"v%d" % i).digest()[:8], "big") % (1 << n) for i in range(m)]
    gammas = [int.from_bytes(hashlib.sha256(b"g%d" % i).digest(), "big") % Q for i in range(m)]
    seeds = [b"seed-%d" % i for i in range(m)]
    bp.prove_wire_packed(vs[:8], gammas[:8], seeds[:8])
    best = None
    for rep in range(3):
        t = time.perf_counter()
        packed, off = bp.prove_wire_packed(vs, gammas, seeds)
        dt = time.perf_counter() - t
        ms = bp.last_ms()
        if best is None or dt < best[0]:
            best = (dt, ms)
    dt, ms = best
    print(json.dumps({"batch": m, "wall_ms": round(dt * 1e3, 2), "proofs_per_s_wall": round(m / dt), "proofs_per_s_device": round(m / (ms["total"] * 1e-3)),
                      "device_ms": {k: round(v, 3) for k, v in ms.items()}, "bytes_per_proof": len(packed) // m}), flush=True)
bp.close()
