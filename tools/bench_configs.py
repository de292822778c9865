#!/usr/bin/env python3
"""Times the BASELINE.json configurations that are not the headline bench line, on one
MI355X, through the product package (python tools/bench_configs.py [--quick]):
  C2  Pippenger MSM n = 2^16
  C3  inner-product argument prover n = 2^20 (log-n folding + L/R MSMs)
  C4  aggregated range proof m = 128 x 64-bit (prove + verify)
  C5  verify throughput of 64-bit range proofs (sample of the 2^14 batch)
Prints one JSON object per configuration."""
import hashlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.ec import Point, secp256k1  # noqa: E402
from bulletproofs_amd.engine import default_engine  # noqa: E402
from bulletproofs_amd.utils import ModP, Transcript, commitment, elliptic_hash, mod_hash  # noqa: E402

Q = secp256k1.q
quick = "--quick" in sys.argv
eng = default_engine()


def sha_scalars(n, seed):
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    out = bytearray(32 * n)
    for i in range(n):
        v = int.from_bytes(hashlib.sha256(pre + i.to_bytes(8, "little")).digest(), "big") % Q
        out[32 * i: 32 * i + 32] = v.to_bytes(32, "little")
    return bytes(out)


def device_points(n, seed):
    d_k = eng.upload(sha_scalars(n, seed))
    d_G = eng.upload(secp256k1.G.to_le64() * n)
    d_p = eng.alloc(64 * n)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, n, d_p.ptr))
    eng.sync()
    d_G.free()
    d_k.free()
    return d_p


def emit(obj):
    print(json.dumps(obj), flush=True)


# ---- C2 -----------------------------------------------------------------------------
def c2():
    n = 1 << 16
    d_p, d_s = device_points(n, 1), eng.upload(sha_scalars(n, 2))
    for _ in range(3):
        eng.msm_dev(d_p, d_s, n)
    t = time.perf_counter()
    reps = 50
    for _ in range(reps):
        eng.msm_dev(d_p, d_s, n)
    dt = (time.perf_counter() - t) / reps
    emit({"config": "C2 MSM n=2^16", "ms": dt * 1e3, "pairs_per_s": n / dt})


# ---- C3 -----------------------------------------------------------------------------
def c3(logn):
    n = 1 << logn
    d_g, d_h = device_points(n, 3), device_points(n, 4)
    d_a, d_b = eng.upload(sha_scalars(n, 5)), eng.upload(sha_scalars(n, 6))
    u = elliptic_hash(b"bench-u")
    eng.profile(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    st = eng.ipa_create_dev(d_g, d_h, d_a, d_b, n, u.to_le64())
    tr = Transcript(b"bench")
    t_lr = t_fold = t_hash = 0.0
    rounds = 0
    while len(st) > 1:
        t = time.perf_counter()
        Lb, Rb = st.round_LR()
        t_lr += time.perf_counter() - t
        t = time.perf_counter()
        tr.add_list_points([Point.from_le64(Lb), Point.from_le64(Rb)])
        x = tr.get_modp(Q)
        tr.add_number(x)
        xi = x.inv()
        t_hash += time.perf_counter() - t
        t = time.perf_counter()
        st.fold(x.x, xi.x)
        eng.sync()
        t_fold += time.perf_counter() - t
        rounds += 1
    a, b = st.finish()
    total = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile(False)
    st.close()
    emit({"config": "C3 IPA prover n=2^%d" % logn, "seconds": total, "rounds": rounds,
          "round_LR_s": t_lr, "fold_s": t_fold, "host_hash_s": t_hash,
          "elements_per_s": n / total,
          "stage_ms_total": {k: round(v[0], 3) for k, v in prof.items() if v[1]}})
    for d in (d_g, d_h, d_a, d_b):
        d.free()


# ---- C4 / C5 ---------------------------------------------------------------------------
def gens(n, s):
    return [elliptic_hash(str(i).encode() + s) for i in range(n)]


def c4(m):
    from bulletproofs_amd.rangeproofs import AggregNIRangeProver, AggregRangeVerifier
    n = 64
    t = time.perf_counter()
    gs, hs = gens(n * m, b"gs"), gens(n * m, b"hs")
    g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
    t_gen = time.perf_counter() - t
    vs = [ModP(int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big"), Q) for j in range(m)]
    gammas = [mod_hash(b"gamma%d" % j, Q) for j in range(m)]
    Vs = [commitment(g, h, vs[j], gammas[j]) for j in range(m)]
    t = time.perf_counter()
    proof = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, b"seed").prove()
    t_prove = time.perf_counter() - t
    t = time.perf_counter()
    ok = AggregRangeVerifier(Vs, g, h, gs, hs, u, proof).verify()
    t_verify = time.perf_counter() - t
    emit({"config": "C4 aggregated range proof m=%d x 64-bit" % m, "prove_s": t_prove, "verify_s": t_verify,
          "verified": ok, "generator_derivation_s": t_gen})


def c5(count, batch_total):
    from bulletproofs_amd.rangeproofs import NIRangeProver, RangeVerifier, BatchRangeVerifier
    n = 64
    gs, hs = gens(n, b"gs"), gens(n, b"hs")
    g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
    proofs = []
    t = time.perf_counter()
    for j in range(count):
        v = ModP(int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big"), Q)
        gamma = mod_hash(b"gamma%d" % j, Q)
        V = commitment(g, h, v, gamma)
        proofs.append((V, NIRangeProver(v, n, g, h, gs, hs, gamma, u, secp256k1, b"seed%d" % j).prove()))
    t_prove = time.perf_counter() - t
    t = time.perf_counter()
    oks = [RangeVerifier(V, g, h, gs, hs, u, pr).verify() for V, pr in proofs]
    t_verify = time.perf_counter() - t
    emit({"config": "C5a verify 64-bit range proofs one at a time (sample of %d, one GPU)" % count,
          "verifies_per_s": count / t_verify, "proves_per_s": count / t_prove, "all_ok": all(oks)})
    # batched: every proof's equations in ONE MSM (rangeproofs/batch.py); the sample is
    # cycled to reach the batch size
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    t = time.perf_counter()
    for k in range(batch_total):
        V, pr = proofs[k % count]
        bv.add(V, pr)
    t_host = time.perf_counter() - t
    t = time.perf_counter()
    ok = bv.verify()
    t_msm = time.perf_counter() - t
    emit({"config": "C5b batch-verify %d x 64-bit range proofs, one GPU, random-linear-combination MSM" % batch_total,
          "verifies_per_s": batch_total / (t_host + t_msm), "host_scalar_prep_s": t_host, "msm_and_pack_s": t_msm,
          "msm_points": 3 + 2 * n + 19 * batch_total, "ok": ok,
          "note": "host prep is single-threaded Python integer algebra; with one process per GPU the proofs shard "
                  "across ranks and the partials fold with one all_gather (tests/test_batch_verify_cpu.py)"})


if __name__ == "__main__":
    c2()
    c3(14 if quick else 20)
    c4(4 if quick else 128)
    c5(8 if quick else 64, 64 if quick else 1 << 12)
