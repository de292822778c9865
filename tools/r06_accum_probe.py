#!/usr/bin/env python3
"""Why is the accumulation of an inner-product round's MSM (2^20 + 1 pairs over three segments) slower than the bench's (2^20 pairs, one
array)?  Stage-timer milliseconds of k_accum_l0, one MSM at a time, for: (A) one array; (B) three segments in separate buffers;
(C) the same through half-block selection as the deferred rounds use it; (D) an inner-product state's own round_LR."""
import ctypes, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bulletproofs_amd  # noqa
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.engine import Engine
from bulletproofs_amd.utils import elliptic_hash
eng = Engine(device=0)
for kv in sys.argv[1:]:
    k, v = kv.split("="); eng.set_option(k, int(v))
rng = np.random.default_rng(1)
n = 1 << 20


def rand255(m):
    a = rng.integers(0, 1 << 32, size=(m, 8), dtype=np.uint64).astype(np.uint32)
    a[:, 7] &= 0x7FFFFFFF
    return a


def points(m):
    d_k = eng.upload(rand255(m).tobytes()); d_G = eng.upload(secp256k1.G.to_le64() * m); d_p = eng.alloc(64 * m)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, m, d_p.ptr)); eng.sync(); d_G.free(); d_k.free()
    return d_p


def stage(fn, reps=12):
    for _ in range(40):
        fn()
    eng.profile(1); eng.profile_reset()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    dt = (time.perf_counter() - t) / reps
    pr = eng.profile_read(); eng.profile(False)
    return dt * 1e3, {k: round(v[0] / max(v[1], 1), 4) for k, v in pr.items() if v[1]}


d_g, d_h = points(n), points(n)
d_a, d_b = eng.upload(rand255(n).tobytes()), eng.upload(rand255(n).tobytes())
print("A one array 2^20:", stage(lambda: eng.msm_dev(d_g, d_a, n)), flush=True)
out = ctypes.create_string_buffer(64)
half = n // 2
u = elliptic_hash(b"bench-u").to_le64()
d_u = eng.upload(u); d_c = eng.upload(rand255(1).tobytes())
P = (ctypes.c_void_p * 3)(d_g.ptr + 64 * half, d_h.ptr, d_u.ptr)
S = (ctypes.c_void_p * 3)(d_a.ptr, d_b.ptr + 32 * half, d_c.ptr)
N = (ctypes.c_uint64 * 3)(half, half, 1)
print("B three segments (g_hi, h_lo, u):", stage(lambda: eng._ck(eng.lib.bpmi_msm_segs_dev(eng.ctx, 3, P, S, N, out))), flush=True)
st = eng.ipa_create_dev(d_g, d_h, d_a, d_b, n, u)
print("D ipa round 1 (pair of 2^20 + 1):", stage(lambda: st.round_LR(), 6), flush=True)
x = (12345).to_bytes(32, "little"); xi = pow(12345, -1, secp256k1.q).to_bytes(32, "little")
st.fold(int.from_bytes(x, "little"), int.from_bytes(xi, "little"))
print("D ipa round 2 (deferred, half-block selection):", stage(lambda: st.round_LR(), 6), flush=True)
