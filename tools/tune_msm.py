"""Sweep MSM tuning knobs on the GPU (window bits, chunk length) at a given size."""
import hashlib, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()

def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))

def device_points(n, seed):
    d_k = eng.upload(sha_scalars(n, seed)); d_G = eng.upload(secp256k1.G.to_le64() * n); d_p = eng.alloc(64 * n)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync(); d_G.free(); d_k.free()
    return d_p

CS = [int(v) for v in os.environ.get("TUNE_C", "0,8,10,13,16").split(",")]
CHUNKS = [int(v) for v in os.environ.get("TUNE_CHUNK", "16,32,64").split(",")]
EPLS = [int(v) for v in os.environ.get("TUNE_EPL", "0").split(",")]
for logn in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << logn
    d_p, d_s = device_points(n, 1), eng.upload(sha_scalars(n, 2))
    ref = None
    for c, chunk, epl in [(c, ch, e) for c in CS for ch in CHUNKS for e in EPLS]:
        if c and not (4 <= c <= 16): continue
        if True:
            eng.set_option("window_bits", c); eng.set_option("chunk", chunk); eng.set_option("reduce_epl", epl)
            r = eng.msm_dev(d_p, d_s, n)
            ref = ref or r
            assert r == ref
            eng.profile(True); eng.profile_reset()
            t = time.perf_counter(); reps = 8
            for _ in range(reps): eng.msm_dev(d_p, d_s, n)
            dt = (time.perf_counter() - t) / reps
            pr = eng.profile_read(); eng.profile(False)
            st = {k.replace("msm_", ""): round(v[0] / v[1], 3) for k, v in pr.items() if v[1]}
            print("n=2^%d c=%2d chunk=%3d epl=%2d  %.3f ms  %.3e pairs/s  %s" % (logn, c, chunk, epl, dt * 1e3, n / dt, st), flush=True)
    d_p.free(); d_s.free()
