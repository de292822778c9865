#!/usr/bin/env python3
"""Where the milliseconds of one 2^14-proof batch verification go (config C5, one GPU): wall time of each step of
BatchRangeVerifier.add_wire_native + verify."""
import hashlib
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.ec import secp256k1  # noqa: E402
from bulletproofs_amd.engine import default_engine  # noqa: E402
from bulletproofs_amd.rangeproofs import BatchRangeVerifier, NIRangeProver  # noqa: E402
from bulletproofs_amd.rangeproofs.codec import proof_to_bytes  # noqa: E402
from bulletproofs_amd.utils import ModP, commitment, elliptic_hash, mod_hash  # noqa: E402

Q = secp256k1.q
eng = default_engine()
n, total, distinct = 64, 1 << 14, int(os.environ.get("C5_DISTINCT", "256"))
gs = [elliptic_hash(str(i).encode() + b"gs") for i in range(n)]
hs = [elliptic_hash(str(i).encode() + b"hs") for i in range(n)]
g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
Vd, wire = [], []
for j in range(distinct):
    v = ModP(int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big"), Q)
    gamma = mod_hash(b"gamma%d" % j, Q)
    Vd.append(commitment(g, h, v, gamma))
    wire.append(proof_to_bytes(NIRangeProver(v, n, g, h, gs, hs, gamma, u, secp256k1, b"seed%d" % j).prove(), version=int(os.environ.get("C5_WIRE", "1"))))
Vs = [Vd[k % distinct] for k in range(total)]
blobs = [wire[k % distinct] for k in range(total)]
threads = min(32, len(os.sched_getaffinity(0)))
from itertools import accumulate
wire_buf = b"".join(blobs)
wire_off = [0, *accumulate(map(len, blobs))]
import cProfile
import pstats
PREPARE = os.environ.get("C5_PREPARE", "auto")          # device | host | auto
if os.environ.get("C5_PINNED"):                         # page-locked receive buffer + commitments already packed
    hb = eng.host_alloc(len(wire_buf))
    hb.view[:] = wire_buf
    wire_buf = hb
    vb = b"".join(V.to_le64() for V in Vs)
    Vs = eng.host_alloc(len(vb)) if os.environ.get("C5_ONECALL") else vb      # (the one-call path takes the commitments page-locked too)
    if os.environ.get("C5_ONECALL"):
        Vs.view[:] = vb
    import ctypes
    wire_off = (ctypes.c_uint64 * len(wire_off))(*wire_off)
if os.environ.get("C5_LANES"):
    eng.set_option("rp_lanes", int(os.environ["C5_LANES"]))
if os.environ.get("C5_SERIAL"):                         # point decoding behind the preparation kernels: every stage's time is its own
    eng.set_option("rp_overlap", 0)
if os.environ.get("C5_ONLY_ROLE"):
    eng.set_option("rp_only_role", int(os.environ["C5_ONLY_ROLE"]))
if PREPARE != "host" and not os.environ.get("C5_NO_STAGE_TIMERS"):
    eng.profile(1)
bv = BatchRangeVerifier(g, h, gs, hs, u)
REPS = int(os.environ.get("C5_REPS", "3"))
for rep in range(REPS):
    t0 = time.perf_counter()
    if rep == REPS - 1:
        pr = cProfile.Profile()
        pr.enable()
    try:
        if os.environ.get("C5_ONECALL"):                # the whole batch in one native call (bpmi_rp_batch_verify_dev)
            ok = bv.verify_wire(Vs, wire_buf, offsets=wire_off)
            t1 = time.perf_counter()
        else:
            bv.add_wire_native(Vs, wire_buf, threads=threads, offsets=wire_off, prepare=PREPARE)
            t1 = time.perf_counter()
            ok = bv.verify()
    except Exception as exc:            # a one-role profiling run (rp_only_role) rejects every batch by design
        t1 = time.perf_counter()
        ok = repr(exc)
    t2 = time.perf_counter()
    if rep == REPS - 1:
        pr.disable()
    bv.reset()
    print("prepare=%s" % PREPARE, end=" ")
    print("rep %d: add_wire_native %.2f ms, verify %.2f ms, ok=%s, threads=%d" % (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ok, threads))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
if PREPARE != "host":
    for name, (ms, cnt) in eng.profile_read().items():
        if cnt:
            print("  stage %-12s %8.3f ms over %d launches" % (name, ms, cnt))
