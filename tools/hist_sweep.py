import hashlib, os, sys, time
sys.path.insert(0, "/root/repo")
import bulletproofs_amd
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
eng = default_engine()
for lg in (20, 16):
    n = 1 << lg
    ks = b"".join(hashlib.sha256(b"%d" % i).digest() for i in range(n))
    d_k = eng.upload(ks); d_G = eng.upload(secp256k1.G.to_le64() * n); d_p = eng.alloc(64 * n)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync()
    ref = None
    for ht, hb in ((256, 512), (1024, 512), (1024, 256), (512, 512), (1024, 1024), (256, 2048), (256, 512)):
        eng.set_option("hist_threads", ht); eng.set_option("hist_blocks", hb)
        t = time.perf_counter()
        while time.perf_counter() - t < 0.06: r = eng.msm_dev(d_p, d_k, n)
        ref = ref or r; assert r == ref
        eng.profile(True); eng.profile_reset()
        for _ in range(10): eng.msm_dev(d_p, d_k, n)
        pr = eng.profile_read(); eng.profile(False)
        print("n=2^%d threads %4d blocks %4d: digits_hist %.4f ms" % (lg, ht, hb, pr["msm_digits_hist"][0] / pr["msm_digits_hist"][1]), flush=True)
