#!/usr/bin/env python3
"""A/B of engine option SETS on one box in one process: the MSM at n = 2^logn, synchronous and with D in flight (the bench's
pipeline: bpmi_msm_dev_enqueue / bpmi_msm_finish over rotating slots, async_lanes = 1), the configurations alternating `rounds` times.
  python tools/ab_configs.py --logn 20 --depth 2 --rounds 3 base: chunk43:chunk=43 acc:accum_stream=1,chunk=43
A configuration is name:opt=v,opt=v (name: alone = the defaults).  Every (round, configuration) is a PROCESS of its own (--child): the
streams of an engine map onto the hardware queues in creation order, and two engines in one process could put two lanes on one
queue.  Prints ms per MSM per round and the minimum; every result is checked against a known answer computed by another kernel."""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.ec import secp256k1  # noqa: E402
from bulletproofs_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*")
ap.add_argument("--child", default=None)
ap.add_argument("--logn", type=int, nargs="+", default=[20])
ap.add_argument("--depth", type=int, nargs="+", default=[2])
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--ms", type=float, default=150.0, help="timed milliseconds per measurement")
ap.add_argument("--no-sync", action="store_true", help="skip the one-at-a-time measurement")
args = ap.parse_args()

if not args.child:
    import subprocess
    table, order = {}, []
    for rnd in range(args.rounds):
        for c in args.configs:
            cmd = [sys.executable, os.path.abspath(__file__), "--child", c, "--rounds", "1", "--ms", str(args.ms), "--logn"] + [str(v) for v in args.logn] + \
                  ["--depth"] + [str(v) for v in args.depth] + (["--no-sync"] if args.no_sync else [])
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                print("# %s failed: %s" % (c, (r.stdout + r.stderr)[-600:]), flush=True)
                continue
            for ln in r.stdout.splitlines():
                if ln.startswith("RESULT "):
                    _, lg, name, mode, ms, ok = ln.split("|")
                    key = (lg, name, mode)
                    if key not in table:
                        table[key] = []
                        order.append(key)
                    table[key].append(float(ms) if ok == "ok" else float("nan"))
    for key in order:
        v = table[key]
        print("n=2^%s  %-26s %-12s min %.4f ms   rounds %s" % (key[0], key[1], key[2], min(v), " ".join("%.4f" % x for x in v)), flush=True)
    sys.exit(0)

name, _, rest = args.child.partition(":")
opts = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in rest.split(",") if kv]
eng = Engine(device=0)
for k, v in opts:
    eng.set_option(k, v)
cfgs = [(name, opts, eng)]

rng = np.random.default_rng(6)
nmax = 1 << max(args.logn)


def rand255(n):
    a = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
    a[:, 7] &= 0x7FFFFFFF
    return a


e0 = cfgs[0][2]
# every engine owns its context but they share the device: inputs are allocated once through the first engine
ks, es = rand255(nmax), rand255(nmax)
d_k = e0.upload(ks.tobytes())
d_G = e0.upload(secp256k1.G.to_le64() * nmax)
d_p = e0.alloc(64 * nmax)
e0._ck(e0.lib.bpmi_ec_mul_batch_dev(e0.ctx, d_G.ptr, d_k.ptr, nmax, d_p.ptr))
e0.sync()
d_G.free()
d_k.free()
d_s = e0.upload(es.tobytes())
e0.sync()


def to_ints(a):
    b = a.tobytes()
    return [int.from_bytes(b[32 * i: 32 * i + 32], "little") for i in range(len(a))]


kk, ee = to_ints(ks), to_ints(es)
del ks, es


def run_sync(eng, n, ms):
    t_h = time.perf_counter()
    while time.perf_counter() - t_h < 0.05:
        r = eng.msm_dev(d_p, d_s, n)
    cnt, t = 0, time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        r = eng.msm_dev(d_p, d_s, n)
        cnt += 1
    return (time.perf_counter() - t) / cnt, r


def run_pipe(eng, n, D, ms):
    eng.set_option("async_lanes", 1)
    r = None
    try:
        def burst(k):
            nonlocal r
            for j in range(min(k, D - 1)):
                eng.msm_dev_enqueue(j % D, d_p, d_s, n)
            for j in range(k):
                if j + D - 1 < k:
                    eng.msm_dev_enqueue((j + D - 1) % D, d_p, d_s, n)
                r = eng.msm_finish(j % D)
        t_h = time.perf_counter()
        while time.perf_counter() - t_h < 0.08:
            burst(12)
        t = time.perf_counter()
        burst(6)
        per = (time.perf_counter() - t) / 6
        k = max(12, int(ms * 1e-3 / per))
        t = time.perf_counter()
        burst(k)
        return (time.perf_counter() - t) / k, r
    finally:
        eng.set_option("async_lanes", 0)


Q = secp256k1.q
for lg in args.logn:
    n = 1 << lg
    modes = ([] if args.no_sync else [("sync", 0)]) + [("%d in flight" % D, D) for D in args.depth]
    ref = eng.ec_mul_batch_bytes(secp256k1.G.to_le64(), (sum(e * k for e, k in zip(ee[:n], kk[:n])) % Q).to_bytes(32, "little"), 1)
    for name, opts, eng in cfgs:
        for mname, D in modes:
            dt, r = run_sync(eng, n, args.ms) if D == 0 else run_pipe(eng, n, D, args.ms)
            print("RESULT |%d|%s|%s|%.5f|%s" % (lg, name, mname, dt * 1e3, "ok" if r == ref else "MISMATCH"), flush=True)
