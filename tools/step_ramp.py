#!/usr/bin/env python3
"""Per-step wall time of the two-deep MSM pipeline at n = 2^20 right after an idle period: how long the GPU takes to reach its
steady clocks (what bench.py's warm-up has to cover).   python tools/step_ramp.py [idle_seconds]"""
import hashlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()
idle = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
n = 1 << 20
ks = b"".join(hashlib.sha256(b"%d" % i).digest() for i in range(n))
d_k = eng.upload(ks); d_G = eng.upload(secp256k1.G.to_le64() * n); d_p = eng.alloc(64 * n)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync()
eng.set_option("async_lanes", 1)
for trial in range(2):
    time.sleep(idle)
    K = 300
    ts = []
    eng.msm_dev_enqueue(0, d_p, d_k, n)
    t0 = time.perf_counter()
    for j in range(K):
        if j + 1 < K: eng.msm_dev_enqueue((j + 1) & 1, d_p, d_k, n)
        eng.msm_finish(j & 1)
        ts.append(time.perf_counter())
    d = [(b - a) * 1e3 for a, b in zip([t0] + ts[:-1], ts)]
    for lo in (0, 5, 10, 20, 40, 80, 160, 240):
        hi = min(K, lo * 2 if lo else 5)
        print("after %.1f s idle, steps %3d..%3d: %.4f ms per step" % (idle, lo, hi - 1, sum(d[lo:hi]) / (hi - lo)), flush=True)
