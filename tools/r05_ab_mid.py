#!/usr/bin/env python3
"""Round 5: A/B of the mid-size changes on one box, one process: synchronous MSMs and two in flight at the given sizes, alternating
configurations (a configuration = option assignments; every option returns to its default afterwards).
   python tools/r05_ab_mid.py [n ...]      R5_CONFIGS="name:opt=v,opt=v;name2:..." overrides the built-in list"""
import hashlib, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.engine import default_engine
from bulletproofs_amd.ec import secp256k1
Q = secp256k1.q
eng = default_engine()
# (an option a configuration names must be listed here with its default: the list is what is set before every measurement)
DEFAULTS = {"mid_single_min": 0, "mid_min": 0, "mid_parts": 0, "top_window_unsigned": 1, "sort_inblock": 1, "segscan_fused": 0, "hist_scan_fused": 0, "final_spread": 3, "reduce_fit": 1, "mixed_windows": 1, "graphs": 0, "window_bits": 0, "chunk": 0, "reduce_epl": 0}
CONFIGS = [("r4", {"top_window_unsigned": 0, "sort_inblock": 0, "segscan_fused": 0, "hist_scan_fused": 0}),
           ("inblock", {"top_window_unsigned": 0, "sort_inblock": 1, "segscan_fused": 0}),
           ("inblock+segfuse", {"top_window_unsigned": 0, "segscan_fused": 1}),
           ("inblock+histscan", {"top_window_unsigned": 0, "hist_scan_fused": 1}),
           ("c15old", {"top_window_unsigned": 0, "window_bits": 15}),
           ("c15top2", {"window_bits": 15}),
           ("c14", {"window_bits": 14}),
           ("c13", {"window_bits": 13}),
           ("default", {})]
if os.environ.get("R5_CONFIGS"):
    CONFIGS = []
    for part in os.environ["R5_CONFIGS"].split(";"):
        name, _, rest = part.partition(":")
        CONFIGS.append((name, {kv.split("=")[0]: int(kv.split("=")[1]) for kv in rest.split(",") if kv}))


def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    return b"".join((int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q).to_bytes(32, "little") for i in range(n))


sizes = [int(a) for a in sys.argv[1:]] or [1 << 15, 1 << 16, 1 << 17, 311427]
nmax = max(sizes)
d_k = eng.upload(sha_scalars(nmax, 1)); d_G = eng.upload(secp256k1.G.to_le64() * nmax); d_p = eng.alloc(64 * nmax)
eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, nmax, d_p.ptr)); eng.sync()
d_s = eng.upload(sha_scalars(nmax, 2))
rounds = int(os.environ.get("R5_ROUNDS", "3"))
for n in sizes:
    reps = 200 if n <= (1 << 17) else 80
    ref = None
    best = {}
    for rnd in range(rounds):
        for name, cfg in CONFIGS:
            for k, v in DEFAULTS.items(): eng.set_option(k, cfg.get(k, v))
            r = eng.msm_dev(d_p, d_s, n)
            ref = ref or r
            assert r == ref, (n, name)
            for _ in range(5): eng.msm_dev(d_p, d_s, n)
            t = time.perf_counter()
            for _ in range(reps): eng.msm_dev(d_p, d_s, n)
            ds = (time.perf_counter() - t) / reps
            eng.set_option("async_lanes", 1)
            eng.msm_dev_enqueue(0, d_p, d_s, n)
            t = time.perf_counter()
            for j in range(reps):
                if j + 1 < reps: eng.msm_dev_enqueue((j + 1) & 1, d_p, d_s, n)
                eng.msm_finish(j & 1)
            dp = (time.perf_counter() - t) / reps
            eng.set_option("async_lanes", 0)
            b = best.setdefault(name, [1e9, 1e9])
            b[0], b[1] = min(b[0], ds), min(b[1], dp)
            print("n=%7d %-16s sync %.4f ms  two in flight %.4f ms" % (n, name, ds * 1e3, dp * 1e3), flush=True)
    for name, _ in CONFIGS:
        print("## n=%7d %-16s best sync %.4f  best two-in-flight %.4f" % (n, name, best[name][0] * 1e3, best[name][1] * 1e3), flush=True)
    if os.environ.get("R5_STAGES"):
        for name, cfg in CONFIGS:
            for k, v in DEFAULTS.items(): eng.set_option(k, cfg.get(k, v))
            eng.profile(True); eng.profile_reset()
            for _ in range(8): eng.msm_dev(d_p, d_s, n)
            pr = eng.profile_read(); eng.profile(False)
            print("## n=%7d %-16s stages: %s" % (n, name, {k.replace("msm_", ""): round(v[0] / v[1], 4) for k, v in pr.items() if v[1]}), flush=True)
for k, v in DEFAULTS.items(): eng.set_option(k, v)
