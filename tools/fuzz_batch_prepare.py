#!/usr/bin/env python3
"""Randomised differential test of the GPU batch preparation (bpmi_rp_batch_prepare_dev, csrc/rp_batch_kernels.hpp) against its
host twin (bpmi_rp_batch_prepare): batches of 1..48 proofs drawn from a pool of valid single and aggregated proofs (n = 2, 8, 16,
64 bits; m = 1, 2, 4 values), a random subset mutated (bit flips, byte overwrites, truncation, extension, length fields,
transcript splices, text edits), explicit or seed-derived weights, random proofs-per-wave and launch sizes.  Every batch must
give the same first failing proof on both sides (point encodings judged by the decompression), and a valid batch the same
scalars, shared coefficients and decoded points, byte for byte.  Round 6: every batch is drawn in wire format 1, 2 or 3 (format 3's y
coordinates get mutations of their own: bit flips, the other root, another point's y, values not below p).
    python tools/fuzz_batch_prepare.py [seconds]"""
import ctypes
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd import _native  # noqa: E402
from bulletproofs_amd.engine import default_engine  # noqa: E402
from bulletproofs_amd.rangeproofs.codec import proof_to_bytes  # noqa: E402
from oracle import bp_ref as R, cbind  # noqa: E402
from helpers import Q, gens  # noqa: E402
from test_batch_verify_cpu import convert_proof  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(os.environ.get("FUZZ_SEED", "7"))
rnd = random.Random(seed)
eng = default_engine()
lib = _native.load()

# ---- pool: (n_gens, m) -> list of wire proofs (made by the oracle's prover, the reference's algorithm)
pool = {}
for n, m in ((2, 1), (8, 1), (16, 1), (64, 1), (8, 2), (16, 4)):
    bits = n // m
    gs, hs = gens(n, b"fgs%d" % n), gens(n, b"fhs%d" % n)
    g, h, u = (R.elliptic_hash(s) for s in (b"fg", b"fh", b"fu"))
    blobs = []
    for t in range(6):
        vs = [R.Zq(rnd.randrange(2 ** bits), Q) for _ in range(m)]
        gammas = [R.mod_hash(b"fz%d-%d-%d" % (n, t, j), Q) for j in range(m)]
        if m == 1:
            pr = R.range_prove(vs[0], n, g, h, gs, hs, gammas[0], u, seed=b"fs%d-%d" % (n, t), multiexp=cbind.msm)
        else:
            pr = R.aggreg_range_prove(vs, bits, g, h, gs, hs, gammas, u, seed=b"fs%d-%d" % (n, t), multiexp=cbind.msm)
        blobs.append(tuple(proof_to_bytes(convert_proof(pr), version=f) for f in (1, 2, 3)))
    pool[(n, m)] = blobs

P_FIELD = 2 ** 256 - 2 ** 32 - 977


def mutate(src, k, fmt):
    ints_end = 6 + 32 * (5 + k)
    pts_end = ints_end + 33 * (6 + 2 * k)
    bad = bytearray(src)
    kind = rnd.randrange(9)
    if fmt != 1 and kind in (4, 5):                 # (the transcripts' length fields and text: format 1 only)
        kind = rnd.choice((0, 6, 7, 9 if fmt == 3 else 1))
    if fmt == 3 and rnd.random() < 0.3:
        kind = 9
    if kind == 9:                                   # a y coordinate of format 3
        npts = 6 + 2 * k
        at, t = len(bad) - 32 * npts, rnd.randrange(npts)
        y = int.from_bytes(bad[at + 32 * t: at + 32 * t + 32], "big")
        how = rnd.randrange(5)
        new = (P_FIELD - y) % P_FIELD if how == 0 else (0 if how == 1 else (rnd.choice((P_FIELD, 2 ** 256 - 1, min(y + P_FIELD, 2 ** 256 - 1))) if how == 2 else
               (int.from_bytes(bad[at + 32 * ((t + 1) % npts): at + 32 * ((t + 1) % npts) + 32], "big") if how == 3 else y ^ (1 << rnd.randrange(256)))))
        bad[at + 32 * t: at + 32 * t + 32] = new.to_bytes(32, "big")
        return bytes(bad)
    if kind == 0:
        for _ in range(rnd.randrange(1, 4)):
            bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
    elif kind == 1:
        bad[rnd.randrange(len(bad))] = rnd.randrange(256)
    elif kind == 2:
        del bad[rnd.randrange(len(bad)):]
    elif kind == 3:
        bad += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 12)))
    elif kind == 4:
        t0 = pts_end + 2
        l0 = int.from_bytes(src[t0:t0 + 4], "big")
        t1 = t0 + 4 + l0
        l1 = int.from_bytes(src[t1:t1 + 4], "big")
        pos = rnd.choice((pts_end, pts_end + 1, t0, t1, t1 + 4 + l1)) + rnd.randrange(2)
        bad[pos] = (bad[pos] + rnd.choice((1, 255, 128))) & 0xFF
    elif kind == 5:
        pos = rnd.randrange(pts_end + 14, len(bad))
        bad[pos] = rnd.choice(b"&0123456789=+/AZaz")
    elif kind == 6:
        bad[rnd.randrange(ints_end, pts_end)] ^= 1 << rnd.randrange(8)
    elif kind == 7:
        bad[rnd.randrange(6, ints_end)] ^= 1 << rnd.randrange(8)
    else:
        bad = bytearray(rnd.randrange(256) for _ in range(rnd.randrange(0, 600)))
    return bytes(bad)


def host(n, m, joined, offs, weights, sd):
    count = len(offs) - 1
    k = n.bit_length() - 1
    npts = count * (6 + 2 * k)
    o = (ctypes.c_uint64 * (count + 1))(*offs)
    v_sc, p_sc = ctypes.create_string_buffer(max(32 * count * m, 1)), ctypes.create_string_buffer(32 * npts)
    shared, comp = ctypes.create_string_buffer(32 * (5 + 2 * n)), ctypes.create_string_buffer(33 * npts)
    bad = ctypes.c_int64(-1)
    rc = lib.bpmi_rp_batch_prepare(n, m, count, joined, len(joined), ctypes.cast(o, ctypes.c_void_p), weights, sd, 2, v_sc, p_sc, shared, comp,
                                   ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p))
    return rc, bad.value, v_sc.raw[:32 * count * m], p_sc.raw, shared.raw, comp.raw


def dev(n, m, joined, offs, weights, sd):
    count = len(offs) - 1
    k = n.bit_length() - 1
    npts = count * (6 + 2 * k)
    o = (ctypes.c_uint64 * (count + 1))(*offs)
    d_v, d_p, d_pts = eng.alloc(32 * count * m), eng.alloc(32 * npts), eng.alloc(64 * npts)
    shared = ctypes.create_string_buffer(32 * (5 + 2 * n))
    bad = ctypes.c_int64(-1)
    try:
        rc = eng.lib.bpmi_rp_batch_prepare_dev(eng.ctx, n, m, count, joined, len(joined), ctypes.cast(o, ctypes.c_void_p), weights, sd, d_v.ptr, d_p.ptr,
                                               d_pts.ptr, shared, ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p))
        return rc, bad.value, d_v.download(), d_p.download(), shared.raw, d_pts.download()
    finally:
        for d in (d_v, d_p, d_pts):
            d.free()


batches = proofs = valid_batches = fails = 0
by_format = {1: 0, 2: 0, 3: 0}
t0 = time.time()
while time.time() - t0 < budget:
    (n, m), blobs0 = rnd.choice(list(pool.items()))
    k = n.bit_length() - 1
    count = rnd.randrange(1, 49)
    p_bad = rnd.choice((0.0, 0.0, 0.02, 0.2, 1.0))
    fmt = rnd.choice((1, 1, 2, 3, 3))
    blobs = [mutate(b, k, fmt) if rnd.random() < p_bad else b for b in (rnd.choice(blobs0)[fmt - 1] for _ in range(count))]
    # a mutation that turns a proof's magic into ANOTHER format's: the device reads a call in the format of its first proof and reports
    # the mix as an argument error, the host takes the formats proof by proof (tests/test_gpu_batch_dev.py) -- not a case for this comparison
    if len({b[4:5] for b in blobs if b[:4] == b"BPRP" and b[4:5] in (b"1", b"2", b"3")} | {b"%d" % fmt}) > 1:
        continue
    offs = [0]
    for b in blobs:
        offs.append(offs[-1] + len(b))
    joined = b"".join(blobs) + b"\x00"              # never an empty buffer
    if rnd.random() < 0.5:
        weights, sd = b"".join(rnd.randrange(0, 2 ** 256).to_bytes(32, "little") for _ in range(4 * count)), None
    else:
        weights, sd = None, bytes(rnd.randrange(256) for _ in range(32))
    eng.set_option("rp_lanes", rnd.choice((0, 0, 1, 2, 8, 32, 64)))
    eng.set_option("rp_rows", rnd.choice((0, 0, 1, 7, 64)))
    h = host(n, m, joined, offs, weights, sd)
    d = dev(n, m, joined, offs, weights, sd)
    host_bad = h[1]
    if h[0] == 0:
        # the host twin leaves the point encodings to the decompression: judge those of the proofs it got through (all of them,
        # or the ones before its first failing proof) and take the earlier verdict
        upto = count if host_bad < 0 else host_bad
        if upto:
            pts, ok = eng.ec_decompress_batch_bytes(h[5][:33 * upto * (6 + 2 * k)], upto * (6 + 2 * k))
            if 0 in ok:
                host_bad = ok.index(0) // (6 + 2 * k)
    good = h[0] == 0 and d[0] == 0 and d[1] == host_bad
    if good and host_bad < 0:
        good = d[2] == h[2] and d[3] == h[3] and d[4] == h[4] and d[5] == pts
        valid_batches += 1
    batches += 1
    by_format[fmt] += 1
    proofs += count
    if not good:
        fails += 1
        print("MISMATCH seed", seed, "batch", batches, "format", fmt, "n", n, "m", m, "count", count, "host", h[:2], host_bad, "dev", d[:2], flush=True)
eng.set_option("rp_lanes", 0)
eng.set_option("rp_rows", 0)
print("fuzz_batch_prepare: %d batches (formats 1/2/3: %d/%d/%d; %d proofs, %d fully valid batches compared byte for byte) in %.0f s, %d mismatches, seed %d"
      % (batches, by_format[1], by_format[2], by_format[3], proofs, valid_batches, time.time() - t0, fails, seed))
sys.exit(1 if fails else 0)
