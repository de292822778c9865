"""The CPU legs of the bench line: the oracle (test infrastructure; imported ONLY here) timed on the host cores."""
import argparse
import hashlib
import json
import os
import sys
import time

from .common import (ALGO_BYTES_PER_PAIR, FE_MUL_PEAK_G, HBM_PEAK_GBS, IPA_ALGO_BYTES_PER_ELEMENT, MULS_PER_MADD, Q, RAW_MAD_TOPS, REPO, committed_traffic, cpu_quota, isa_counts,
                     synth_scalars, usable_cpus)


def c5_cpu_baseline(g, h, gs, hs, u, v_packed, wire_joined, wire_off_c, total, usable):
    """verifies/s of the SAME batch verification with no GPU: libbpmi's host preparation (bpmi_rp_batch_prepare, the parity twin
    of the device kernels; `usable` threads) + the C oracle's point decompression and bucket MSM on the same threads; and ONE
    64-bit proof verified by the Python restatement of RangeVerifier.verify with the reference's own multiexp algorithm
    (oracle.bp_ref, 1 core) -- what /root/reference/src/rangeproofs/rangeproof_verifier.py:55-99 costs per proof."""
    from oracle import bp_ref as R, cbind
    from oracle.ec import secp256k1 as osecp
    from bulletproofs_amd.rangeproofs import BatchRangeVerifier
    thr = max(1, usable)
    bv = BatchRangeVerifier(g, h, gs, hs, u, msm=lambda p, s_, n_: cbind.msm_bytes(p, s_, n_, min(thr, 17)))
    dec = lambda comp, n_: cbind.ec_decompress_batch_bytes(comp, n_, thr)
    t0 = time.perf_counter()
    reps = 0
    ok = True
    while True:
        bv.reset()
        bv.add_wire_native(v_packed, wire_joined, decompress=dec, threads=thr, offsets=wire_off_c, prepare="host")
        ok = ok and bv.partial() == bytes(64)
        reps += 1
        if time.perf_counter() - t0 > 8.0:
            break
    dt = (time.perf_counter() - t0) / reps
    # one proof, the reference's way
    Qo = osecp.q
    og = lambda P_: R.elliptic_hash(P_)
    ogs = [R.elliptic_hash(str(i).encode() + b"gs") for i in range(64)]
    ohs = [R.elliptic_hash(str(i).encode() + b"hs") for i in range(64)]
    o_g, o_h, o_u = og(b"g"), og(b"h"), og(b"u")
    v, gamma = R.Zq(0x1234567890ABCDEF, Qo), R.mod_hash(b"gamma0", Qo)
    V = R.commitment(o_g, o_h, v, gamma)
    proof = R.range_prove(v, 64, o_g, o_h, ogs, ohs, gamma, o_u, Qo, b"seed0", multiexp=cbind.msm)
    t1 = time.perf_counter()
    ok1 = bool(R.range_verify(V, o_g, o_h, ogs, ohs, o_u, proof))
    dt1 = time.perf_counter() - t1
    return {"value": total / dt, "unit": "verifies/s", "cores": thr, "kind": "port",
            "sample": "the whole 2^14-proof batch, %d reps: bpmi_rp_batch_prepare (libbpmi's host preparation) + oracle/c decompression of %d points "
                      "+ oracle/c bucket MSM, %d threads" % (reps, 19 * total, thr),
            "seconds_per_batch": dt, "accepted": bool(ok),
            "python_reference_verify": {"value": 1.0 / dt1, "unit": "verifies/s", "cores": 1, "seconds_per_verify": round(dt1, 4), "accepted": ok1,
                                        "what": "oracle.bp_ref.range_verify: RangeVerifier.verify restated, the reference's subset-table multiexp, Python, one 64-bit proof"}}


# ---- extra: config C2, Pippenger MSM n = 2^16 --------------------------------------------------------------


def cpu_baseline(logn, d_pts, d_sc, eng):
    """The plain-C oracle MSM (bucket method, pthreads; "port") on this host's cores over the first 2^logn pairs of the same
    synthetic workload (default: ALL 2^20 of them), its result compared with the GPU's; and `reference_algorithm`: the oracle's
    restatement of src/pippenger's own subset-table schedule on ONE core (the reference is single-threaded)."""
    from oracle import cbind
    m = 1 << logn
    pts = bytes(d_pts[: 64 * m].cpu().numpy().tobytes())
    scs = bytes(d_sc[: 32 * m].cpu().numpy().tobytes())
    # the C oracle parallelises over windows, so it cannot use more threads than windows; and never more threads than the
    # CPUs this process may actually use (cgroup quota / affinity)
    c = max(2, min(16, m.bit_length() - 1 - 2))
    cores = max(1, min(usable_cpus(), (256 + c - 1) // c + 1))
    cbind.msm_bytes(pts[: 64 * 256], scs[: 32 * 256], 256, cores)     # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        ref = cbind.msm_bytes(pts, scs, m, cores)
        reps += 1
        if time.perf_counter() - t0 > 10.0:          # ~10 s of CPU work (bounded sample)
            break
    dt = time.perf_counter() - t0
    gpu = eng.msm_dev(d_pts, d_sc, m)
    out = {"value": m * reps / dt, "unit": "pairs/s", "cores": cores, "host_cores": os.cpu_count(), "host_cpu_quota": cpu_quota(), "kind": "port",
           "sample": "oracle/c bucket MSM, %s 2^%d pairs of the same inputs, %d reps, %d threads" % ("all" if logn >= 20 else "first", logn, reps, cores),
           "sample_matches_gpu": bool(gpu == ref)}
    try:
        out["reference_algorithm"] = cpu_reference_algorithm()
    except Exception as e:
        out["reference_algorithm"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def cpu_reference_algorithm():
    """src/pippenger/pippenger.py:22-94 as restated in oracle/bp_ref.py (same s / t / b, same subset tables), 1 core, on the
    inputs of the reference-generated goldens at n = 2^10 and 2^12: the group-operation counts must EQUAL the ones the
    reference itself performed on those inputs (tests/golden/multiexp.json; BASELINE.md quotes 53 815 / 259 068 for the survey's
    own random draw of the same sizes), and the result must equal the golden point.  2^16 and 2^20 are op-count extrapolations."""
    from oracle import bp_ref as R
    from oracle.ec import secp256k1
    with open(os.path.join(REPO, "tests", "golden", "multiexp.json")) as f:
        g = json.load(f)
    sg, ss = bytes.fromhex(g["seed_points"]), bytes.fromhex(g["seed_scalars"])
    want = {c["n"]: c for c in g["cases"] if c["label"] == "random"}
    pts_all = [R.elliptic_hash(str(i).encode() + sg) for i in range(4096)]
    es_all = [R.mod_hash(str(i).encode() + ss, secp256k1.q) for i in range(4096)]
    rows, sec_per_op = [], None
    for n in (1024, 4096):
        grp = R.EC()
        t0 = time.perf_counter()
        got = R.Pippenger(grp).multiexp(pts_all[:n], es_all[:n])
        dt = time.perf_counter() - t0
        same = ["%x" % got.x, "%x" % got.y] == want[n]["result"]
        rows.append({"n": n, "seconds": round(dt, 3), "pairs_per_s": round(n / dt, 1), "group_ops": grp.ops,
                     "group_ops_of_the_reference_on_these_inputs": want[n]["ops"], "ops_equal": grp.ops == want[n]["ops"],
                     "result_equals_reference_golden": bool(same)})
        sec_per_op = dt / grp.ops
    extrap = []
    try:
        with open(os.path.join(REPO, "tests", "golden", "multiexp_big.json")) as f:
            big = json.load(f)
        ops16, src16 = big["ops"], "op count of ONE run of the reference itself at this size (tests/golden/multiexp_big.json: %.0f s there)" % big["reference_seconds"]
    except (OSError, KeyError, ValueError):
        ops16, src16 = 23703378, "measured op count of the reference at survey time (BASELINE.md)"
    for n, ops, src in ((1 << 16, ops16, src16),
                        (1 << 20, 2.31e9, "closed form of SURVEY.md 3.1; 2.29e9 resident table entries: not runnable on any host")):
        extrap.append({"n": n, "group_ops": ops, "seconds_extrapolated": round(ops * sec_per_op, 1), "pairs_per_s_extrapolated": round(n / (ops * sec_per_op), 2),
                       "extrapolated": True, "op_count_source": src})
    return {"what": "oracle.bp_ref.Pippenger(EC): the reference's subset-table schedule, Python, 1 core", "cores": 1, "kind": "port of the reference algorithm",
            "measured": rows, "extrapolated": extrap,
            "reference_at_survey_time": "BASELINE.md section 2: 761 / 483 / 225 / 84.1 pairs/s at n = 2^10 / 2^12 / 2^14 / 2^16 (reference code + pure-Python EC, 1 core)"}
