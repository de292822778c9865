#!/usr/bin/env python3
"""MSM beyond the LDS-sort range (n > 2^23 uses the global-atomic sort path) up to BPMI_MAX_N:
  python tools/big_n_check.py 24 25 26
Points: 2^16 distinct points tiled; scalars: uniform 255-bit.  Check (size-independent):
  MSM(tiled points, e) == MSM(distinct points, column sums of e mod q)
with the column sums computed on the host with numpy limbs."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.ec import secp256k1  # noqa: E402
from bulletproofs_amd.engine import default_engine  # noqa: E402

Q = secp256k1.q
eng = default_engine()
D = 1 << 16
rng = np.random.default_rng(7)
ks = rng.integers(0, 1 << 32, size=(D, 8), dtype=np.uint64).astype(np.uint32)
ks[:, 7] &= 0x7FFFFFFF
d_k = eng.upload(ks.tobytes())
d_G = eng.upload(secp256k1.G.to_le64() * D)
d_small = eng.alloc(64 * D)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, D, d_small.ptr))
eng.sync()
small = d_small.download()

for logn in [int(a) for a in sys.argv[1:]] or [24]:
    n = 1 << logn
    reps = n // D
    e = rng.integers(0, 1 << 32, size=(reps, D, 8), dtype=np.uint64).astype(np.uint32)
    e[:, :, 7] &= 0x7FFFFFFF                      # < 2^255 < q
    col = e.astype(np.uint64).sum(axis=0)         # (D, 8) limb sums, each < 2^42
    folded = bytearray()
    for j in range(D):
        v = 0
        for k in range(7, -1, -1):
            v = (v << 32) + int(col[j, k])
        folded += (v % Q).to_bytes(32, "little")
    d_pts = eng.alloc(64 * n)
    for r in range(reps):
        d_pts.upload(small, 64 * D * r)
    d_e = eng.upload(e.tobytes())
    del e
    want = eng.msm_bytes(small, bytes(folded), D)
    t = time.perf_counter()
    got = eng.msm_dev(d_pts, d_e, n)
    dt = time.perf_counter() - t
    t = time.perf_counter()
    got2 = eng.msm_dev(d_pts, d_e, n)
    dt2 = time.perf_counter() - t
    print("n=2^%d  ok=%s  deterministic=%s  first %.1f ms, second %.1f ms (%.3g pairs/s)" %
          (logn, got == want, got == got2, dt * 1e3, dt2 * 1e3, n / dt2), flush=True)
    d_pts.free()
    d_e.free()
