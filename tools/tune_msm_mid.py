"""Sweep of the MSM's window bits x chunk length at MID sizes (2^13 .. 2^19 and the batch verifier's 311 427 pairs), wall time of
synchronous calls WITHOUT stage timers (every recorded event is a bubble) and with two calls in flight; the table behind
pick_window_bits / the chunk choice of csrc/msm_host.hpp.   python tools/tune_msm_mid.py [n ...]   (TUNE_C, TUNE_CHUNK: lists)"""
import hashlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()
eng.set_option("small_n", -1)


def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))


def device_points(n, seed):
    d_k = eng.upload(sha_scalars(n, seed)); d_G = eng.upload(secp256k1.G.to_le64() * n); d_p = eng.alloc(64 * n)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr)); eng.sync(); d_G.free(); d_k.free()
    return d_p


CS = [int(v) for v in os.environ.get("TUNE_C", "0,8,9,10,11,12,13,14,15,16").split(",")]
CHUNKS = [int(v) for v in os.environ.get("TUNE_CHUNK", "0,8,16,32,64").split(",")]
if os.environ.get("TUNE_GLV"):
    eng.set_option("glv", int(os.environ["TUNE_GLV"]))
sizes = [int(a) for a in sys.argv[1:]] or [1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 17, 1 << 18, 311427, 1 << 19]
nmax = max(sizes)
d_p, d_s = device_points(nmax, 1), eng.upload(sha_scalars(nmax, 2))
for n in sizes:
    ref, best = None, None
    for c in CS:
        for chunk in CHUNKS:
            eng.set_option("window_bits", c); eng.set_option("chunk", chunk)
            r = eng.msm_dev(d_p, d_s, n)
            ref = ref or r
            assert r == ref, (n, c, chunk)
            reps = int(os.environ.get("TUNE_REPS", "40"))
            t_h = time.perf_counter()
            while time.perf_counter() - t_h < float(os.environ.get("TUNE_PREHEAT_S", "0.06")):      # steady clocks (tools/step_ramp.py)
                eng.msm_dev(d_p, d_s, n)
            t = time.perf_counter()
            for _ in range(reps): eng.msm_dev(d_p, d_s, n)
            dt = (time.perf_counter() - t) / reps
            eng.set_option("async_lanes", 1)
            for j in range(10):
                eng.msm_dev_enqueue(j & 1, d_p, d_s, n)
                if j: eng.msm_finish((j - 1) & 1)
            eng.msm_finish(1)
            eng.msm_dev_enqueue(0, d_p, d_s, n)
            t = time.perf_counter()
            for j in range(reps):
                if j + 1 < reps: eng.msm_dev_enqueue((j + 1) & 1, d_p, d_s, n)
                eng.msm_finish(j & 1)
            dp = (time.perf_counter() - t) / reps
            eng.set_option("async_lanes", 0)
            print("n=%7d c=%2d chunk=%2d  sync %.3f ms  two-in-flight %.3f ms" % (n, c, chunk, dt * 1e3, dp * 1e3), flush=True)
            if best is None or dt < best[0]: best = (dt, c, chunk)
            if os.environ.get("TUNE_STAGES"):           # the stage table of EVERY combination, not only of the best one
                eng.profile(True); eng.profile_reset()
                for _ in range(8): eng.msm_dev(d_p, d_s, n)
                pr = eng.profile_read(); eng.profile(False)
                print("   stages:", {k.replace("msm_", ""): round(v[0] / v[1], 4) for k, v in pr.items() if v[1]}, flush=True)
    print("## n=%d best sync: c=%d chunk=%d %.3f ms" % (n, best[1], best[2], best[0] * 1e3), flush=True)
    eng.set_option("window_bits", best[1]); eng.set_option("chunk", best[2])
    eng.profile(True); eng.profile_reset()
    for _ in range(8): eng.msm_dev(d_p, d_s, n)
    pr = eng.profile_read(); eng.profile(False)
    print("## stages:", {k.replace("msm_", ""): round(v[0] / v[1], 4) for k, v in pr.items() if v[1]}, flush=True)
