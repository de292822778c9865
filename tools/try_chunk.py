import sys, time, hashlib
sys.path.insert(0, "/root/repo")
import bulletproofs_amd
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.engine import default_engine
Q = secp256k1.q; eng = default_engine()
def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))
for logn in [int(a) for a in sys.argv[1:]] or [20]:
    n = 1 << logn
    d_k = eng.upload(sha_scalars(n, 1)); d_G = eng.upload(secp256k1.G.to_le64() * n); d_p = eng.alloc(64 * n)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync()
    d_s = eng.upload(sha_scalars(n, 2))
    ref = None
    for L in (32, 40, 43, 48, 56, 64, 72, 80, 84, 85, 86, 88, 96, 128, 64, 85):
        eng.set_option("chunk", L)
        for _ in range(3): out = eng.msm_dev(d_p, d_s, n)
        ref = ref or out; assert out == ref
        t = time.perf_counter()
        for _ in range(20): eng.msm_dev(d_p, d_s, n)
        dt = (time.perf_counter() - t) / 20
        eng.profile(2); eng.profile_reset()
        for _ in range(5): eng.msm_dev(d_p, d_s, n)
        pr = eng.profile_read(); eng.profile(False)
        print("n=2^%d L=%3d  %.4f ms   accumulate %.4f" % (logn, L, dt * 1e3, pr["msm_accumulate"][0] / pr["msm_accumulate"][1]), flush=True)
    eng.set_option("chunk", 0)
    for b in (d_k, d_G, d_p, d_s): b.free()
