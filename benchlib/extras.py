"""BASELINE.json's configs C2 .. C5 as `extra` entries of the bench line."""
import argparse
import hashlib
import json
import os
import sys
import time

from .common import (PCIE_PEAK_GBS, ALGO_BYTES_PER_PAIR, FE_MUL_PEAK_G, HBM_PEAK_GBS, IPA_ALGO_BYTES_PER_ELEMENT, MULS_PER_MADD, Q, RAW_MAD_TOPS, REPO, committed_traffic, cpu_quota, isa_counts,
                     synth_scalars, usable_cpus)
from .cpu_baseline import c5_cpu_baseline
from .launch import c5_inflight


def extra_c5(eng, world, rank, dev, ready, log_batch=14, distinct=None, per_gpu=False):
    """verifies/s of the random-linear-combination batch verifier on wire-format proofs: bytes in a page-locked receive
    buffer -> one upload -> GPU preparation (parse, SHA-256 transcript re-hashes, weighted scalars; one lane per proof) ->
    GPU decoding of 19 points per proof -> ONE MSM over 3 + 2*64 + 19*batch points; sharded by proof over the ranks.
    Replaces a loop of RangeVerifier.verify (/root/reference/src/rangeproofs/rangeproof_verifier.py:55-99,
    src/innerproduct/inner_product_verifier.py:127-147).  per_gpu: the batch is 2^log_batch proofs PER RANK (weak scaling; the
    default is BASELINE's fixed 2^14 batch split over the ranks)."""
    import torch
    import torch.distributed as dist
    from bulletproofs_amd.distributed import ShardedMSM, shard_bounds
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.engine import set_default_engine
    from bulletproofs_amd.rangeproofs import BatchRangeVerifier, NIRangeProver
    from bulletproofs_amd.rangeproofs.codec import proof_to_bytes
    from bulletproofs_amd.utils import ModP, commitment, elliptic_hash, mod_hash
    set_default_engine(eng)
    nbits = 64
    gs = [elliptic_hash(str(i).encode() + b"gs") for i in range(nbits)]
    hs = [elliptic_hash(str(i).encode() + b"hs") for i in range(nbits)]
    g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
    # Round 5: EVERY proof of the batch is its own proof (2^14 distinct values, blinding factors and seeds), made by the batched prover
    # (rangeproofs/batch_prover.py: one device call for all of them; round 4 proved 1 024 one at a time and repeated them 16 times).
    # The single-proof prover still proves a sample: its rate stays on record and its bytes must equal the batch's.
    from bulletproofs_amd.rangeproofs import BatchRangeProver
    from bulletproofs_amd.rangeproofs.codec import wire_v2_to_v1
    if distinct is None:
        distinct = 1 << log_batch
    vals = [int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big") for j in range(distinct)]
    gams = [int.from_bytes(hashlib.sha256(b"gamma%d" % j).digest(), "big") % Q for j in range(distinct)]
    seeds = [b"seed%d" % j for j in range(distinct)]
    t0 = time.perf_counter()
    bp = BatchRangeProver(nbits, g, h, gs, hs, u, engine=eng)
    eng.sync()
    t_tables = time.perf_counter() - t0
    bp.prove_wire(vals[:64], gams[:64], seeds[:64])                   # warm (buffers, clocks)
    # a service's inputs arrive as bytes: values and blinding factors packed (32 bytes little-endian each), the seeds joined with offsets
    vals_b = b"".join(int(v).to_bytes(32, "little") for v in vals)
    gams_b = b"".join(int(x).to_bytes(32, "little") for x in gams)
    from itertools import accumulate as _acc
    seeds_b, seeds_off = b"".join(seeds), [0, *_acc(map(len, seeds))]
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        packed2, off2 = bp.prove_wire_packed(vals_b, gams_b, (seeds_b, seeds_off))      # ONE native call; the proofs come back as one buffer + offsets
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, bp.last_ms())
    t_batch, batch_ms = best
    # the same call with the offsets as a ctypes array and the proofs left in the prover's page-locked buffer (copy=False): what a service
    # that forwards the bytes does -- no 18 MB host copy, no list of 2^14 Python integers (tools/r06_prover_host_breakdown.py)
    import ctypes as _ct
    seeds_off_c = (_ct.c_uint64 * (distinct + 1))(*seeds_off)
    t_view = None
    for _ in range(3):
        t0 = time.perf_counter()
        view, _off = bp.prove_wire_packed(vals_b, gams_b, (seeds_b, seeds_off_c), copy=False)
        dt = time.perf_counter() - t0
        t_view = dt if t_view is None else min(t_view, dt)
    view_same = bytes(view) == packed2
    del view, _off
    wire2 = [packed2[off2[j]: off2[j + 1]] for j in range(distinct)]    # format 2: no transcripts, the device rebuilds them (1.09 KB instead of 2.56 KB per proof)
    assert bp.prove_wire(vals[:16], gams[:16], seeds[:16]) == wire2[:16]
    # round 6, wire format 3: the same proofs followed by their points' y coordinates (the prover holds them anyway); the verifier
    # checks each y instead of taking a square root per point (rangeproofs/codec.py, csrc/rp_wire_v2_host.hpp)
    from bulletproofs_amd.rangeproofs.codec import wire_v2_to_v3
    bp.wire_format = 3
    t0 = time.perf_counter()
    packed3, off3 = bp.prove_wire_packed(vals_b, gams_b, (seeds_b, seeds_off))
    t_batch3 = time.perf_counter() - t0
    bp.wire_format = 2
    wire3 = [packed3[off3[j]: off3[j + 1]] for j in range(distinct)]
    wire3_ok = wire3[:64] == wire_v2_to_v3(wire2[:64], eng) and all(w3[5:len(w2)] == w2[5:] for w2, w3 in zip(wire2, wire3))
    bp.close()
    sample = min(distinct, 48)
    t0 = time.perf_counter()
    single = [proof_to_bytes(NIRangeProver(ModP(vals[j], Q), nbits, g, h, gs, hs, ModP(gams[j], Q), u, secp256k1, seeds[j]).prove(), version=2) for j in range(sample)]
    t_prove = time.perf_counter() - t0
    prover_info = {"proofs": distinct, "proves_per_s": distinct / t_batch, "proves_per_s_device_time": distinct / (batch_ms["total"] * 1e-3),
                   "seconds_per_batch": t_batch,
                   "proves_per_s_view_out": distinct / t_view, "view_out_same_bytes": view_same,
                   "view_out_note": "prove_wire_packed(..., copy=False) with ctypes offsets: the proofs stay in the prover's page-locked buffer (valid until its next call)",
                   "device_ms_by_phase": {k_: round(v_, 3) for k_, v_ in batch_ms.items()}, "tables_build_s_once_per_prover": round(t_tables, 4),
                   "single_proof_prover_proves_per_s": sample / t_prove, "byte_identical_to_single_proof_prover_on_sample": single == wire2[:sample],
                   "replaces": "a loop of NIRangeProver.prove (/root/reference/src/rangeproofs/rangeproof_prover.py:35-91)",
                   "how": "bpmi_rp_prove_batch: every protocol step one launch over the batch, fixed-base tables of the generators, Fiat-Shamir hashes on the device"}
    # round 6: the same prover for AGGREGATED proofs -- a quarter as many proofs of 4 x 16 bits (the same 64 elements per proof), a sample
    # compared byte for byte with AggregNIRangeProver (/root/reference/src/rangeproofs/rangeproof_aggreg_prover.py:36-146 behind the product's surface)
    try:
        from bulletproofs_amd.rangeproofs import AggregNIRangeProver
        am, ab, ap = 4, 16, max(8, distinct // 4)
        avals = [[int.from_bytes(hashlib.sha256(b"av%d/%d" % (j, t)).digest()[:2], "big") for t in range(am)] for j in range(ap)]
        agams = [[int.from_bytes(hashlib.sha256(b"ag%d/%d" % (j, t)).digest(), "big") % Q for t in range(am)] for j in range(ap)]
        bpa = BatchRangeProver(ab, g, h, gs, hs, u, engine=eng, m=am)
        av_b = b"".join(int(v).to_bytes(32, "little") for row in avals for v in row)
        ag_b = b"".join(int(x).to_bytes(32, "little") for row in agams for x in row)
        a_off = [0, *_acc(map(len, seeds[:ap]))]
        bpa.prove_wire_packed(av_b, ag_b, (b"".join(seeds[:ap]), a_off))
        t0 = time.perf_counter()
        apacked, aoff = bpa.prove_wire_packed(av_b, ag_b, (b"".join(seeds[:ap]), a_off))
        adt = time.perf_counter() - t0
        a_ms = bpa.last_ms()
        bpa.close()
        asample = min(ap, 6)
        asingle = [proof_to_bytes(AggregNIRangeProver([ModP(v, Q) for v in avals[j]], ab, g, h, gs, hs, [ModP(x, Q) for x in agams[j]], u, secp256k1, seeds[j]).prove(), version=2)
                   for j in range(asample)]
        prover_info["aggregated"] = {"proofs": ap, "values_per_proof": am, "bits_per_value": ab, "proves_per_s": ap / adt, "values_per_s": ap * am / adt,
                                     "device_ms": round(a_ms["total"], 3), "byte_identical_to_AggregNIRangeProver_on_sample": asingle == [apacked[aoff[j]: aoff[j + 1]] for j in range(asample)],
                                     "replaces": "a loop of AggregNIRangeProver.prove (/root/reference/src/rangeproofs/rangeproof_aggreg_prover.py:36-146)"}
    except Exception as e:
        prover_info["aggregated"] = {"error": "%s: %s" % (type(e).__name__, e)}
    wire = [wire_v2_to_v1(b_) for b_ in wire2]
    # the commitments V_j = v_j g + gamma_j h in bulk: two batched multiplications and one batched addition
    le = lambda xs: b"".join(int(x).to_bytes(32, "little") for x in xs)
    one = (1).to_bytes(32, "little")
    vg = eng.ec_mul_batch_bytes(g.to_le64() * distinct, le(vals), distinct)
    rh = eng.ec_mul_batch_bytes(h.to_le64() * distinct, le(gams), distinct)
    vsum = eng.ec_lincomb2_batch_bytes(vg, rh, one, one, distinct)
    from bulletproofs_amd.ec import Point
    proofs = [(Point.from_le64(vsum[64 * j: 64 * j + 64]), None) for j in range(distinct)]
    assert proofs[0][0] == commitment(g, h, ModP(vals[0], Q), ModP(gams[0], Q))

    def run_format(wire, first):
        """Everything measured on one wire format: one batch at a time, several in flight, the checks."""
        total = (1 << log_batch) * (world if per_gpu else 1)
        lo, hi = shard_bounds(total, world, rank)
        Vs_in = [proofs[k % distinct][0] for k in range(lo, hi)]
        blobs_in = [wire[k % distinct] for k in range(lo, hi)]
        # the proofs arrive as ONE receive buffer with an offset table (what a socket reader produces), not as 2^14 Python objects
        from itertools import accumulate
        wire_off = [0, *accumulate(map(len, blobs_in))]
        wire_joined = b"".join(blobs_in)
        wire_buf = eng.host_alloc(len(wire_joined))       # page-locked, as a receive buffer registered with the GPU would be
        wire_buf.view[:] = wire_joined
        v_packed = b"".join(V.to_le64() for V in Vs_in)   # commitments in the library's 64-byte point format
        v_pinned = eng.host_alloc(len(v_packed))          # ... page-locked like the receive buffer: no staging copy on the way up
        v_pinned.view[:] = v_packed
        import ctypes
        wire_off_c = (ctypes.c_uint64 * len(wire_off))(*wire_off)
        usable = usable_cpus()
        threads = max(1, min(32, usable // world))
        from bulletproofs_amd.engine import Engine
        eng_x = Engine(device=eng.device)                 # the exchange folds the ranks' partials on an engine of its own: the batch slots' engines
        sharded = ShardedMSM(engine=eng_x)                # are busy on other threads while this thread combines (one thread per engine at a time)

        bv = BatchRangeVerifier(g, h, gs, hs, u, engine=eng)

        errors = []

        def finish(part):
            failed = part is None
            if dist.is_initialized():        # a rank whose batch failed still takes part in the exchange (with a point that cannot sum to the identity by accident): no rank is left waiting
                part = sharded.combine(secp256k1.G.to_le64() if failed else part)
            return (not failed) and part == bytes(64)

        def finish_pipelined(parts):
            """finish() over a sequence of partials with ONE exchange in flight: the exchange of batch i (copy up, all_gather, fold, copy down
            on the exchange stream) is collected while batch i + 1's partial is awaited -- every rank still issues its collectives in batch order."""
            verdicts, pend = [], None
            for part in parts:
                failed = part is None
                if not dist.is_initialized():
                    verdicts.append((not failed) and part == bytes(64))
                    continue
                if pend is not None:
                    verdicts.append((not pend[0]) and sharded.combine_wait(pend[1]) == bytes(64))
                pend = (failed, sharded.combine_begin(secp256k1.G.to_le64() if failed else part))
            if pend is not None:
                verdicts.append((not pend[0]) and sharded.combine_wait(pend[1]) == bytes(64))
            return verdicts

        def one_batch(corrupt=False):
            buf = wire_buf
            if corrupt:           # flip one bit inside one proof of this rank's shard: the batch must reject
                bad = bytearray(wire_joined)
                bad[(wire_off[len(blobs_in) // 2] + wire_off[len(blobs_in) // 2 + 1]) // 2] ^= 1
                buf = bytes(bad)
            try:
                part = bv.partial_wire(v_pinned, buf, offsets=wire_off_c)      # ONE native call: upload, preparation, decoding, MSM
            except Exception as e:
                # "Proof invalid" is a verdict (the batch holds a bad proof); anything else is a defect and is reported as such
                if str(e) != "Proof invalid":
                    errors.append("%s: %s" % (type(e).__name__, e))
                part = None
            if corrupt:           # verified locally: the verdict on this rank's own shard is what is being checked
                return part == bytes(64)
            return finish(part)

        if first:
            ready()                                        # inputs, buffers and verifiers exist on every rank: the collectives start here
        for _ in range(4):                                 # warm: workspaces, pinned buffers, and the clocks (a batch is ~2 ms of GPU work)
            one_batch()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)
        eng.profile(1)
        eng.profile_reset()
        reps = 3
        t0 = time.perf_counter()
        oks = [one_batch() for _ in range(reps)]
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = (time.perf_counter() - t0) / reps
        prof = eng.profile_read()
        # every stage's OWN duration: one more batch with the point decoding behind the preparation kernels instead of beside them
        # (option rp_overlap = 0): beside each other the two stretch (0.63 ms for a decoding that takes 0.32), and a roofline fraction
        # computed on a stretched duration understates the kernel
        eng.set_option("rp_overlap", 0)
        eng.profile_reset()
        ok_serial = one_batch()
        prof_serial = eng.profile_read()
        eng.set_option("rp_overlap", 1)
        eng.profile(False)
        oks.append(ok_serial)
        rejected = not one_batch(corrupt=True)
        # Throughput: several batches in flight.  Verifiers with an engine (stream, workspaces) and a receive buffer of their own
        # work from their own threads (the library calls release the GIL): the upload of one batch overlaps the kernels of the other.  The
        # exchange of the partials stays on this thread, in batch order, so every rank issues its collectives in the same order.
        from concurrent.futures import ThreadPoolExecutor
        # batches in flight: throughput keeps growing with the depth (one GPU: 2: 7.9-9.2e6 verifies/s, 3: 9.1-9.4e6, 4: 9.5-10.4e6,
        # 6: 10.4-11.4e6, 8: 10.9-11.0e6, 10: 11.8-12.1e6); every slot is a host thread, so the default follows the CPUs this rank may use
        inflight = c5_inflight(usable, world)
        slots, extra_engines = [(bv, wire_buf)], []
        for _ in range(inflight - 1):
            e2 = Engine(device=eng.device)
            b2 = e2.host_alloc(len(wire_joined))
            b2.view[:] = wire_joined
            extra_engines.append((e2, b2))
            slots.append((BatchRangeVerifier(g, h, gs, hs, u, engine=e2), b2))

        def local_partial(slot):
            bv, buf = slots[slot]
            try:
                return bv.partial_wire(v_pinned, buf, offsets=wire_off_c)
            except Exception as e:
                if str(e) != "Proof invalid":
                    errors.append("%s: %s" % (type(e).__name__, e))
                return None

        pipe_batches = 32 * inflight                                                     # ~0.3 s of batches: run to run the figure moves by +-5 % (tools/c5_inflight_sweep.sh)
        lanes = [ThreadPoolExecutor(1) for _ in range(inflight)]                         # one thread per slot: a slot never runs two batches at once
        try:
            for _ in range(4):                                                            # warm every slot, and ~40 ms of this very load for the clocks
                for f in [lanes[i].submit(local_partial, i) for i in range(inflight)]:
                    finish(f.result())
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            futs = [lanes[i % inflight].submit(local_partial, i % inflight) for i in range(pipe_batches)]
            oks += finish_pipelined(f.result() for f in futs)
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize(dev)
            elapsed_pipe = (time.perf_counter() - t0) / pipe_batches
        finally:
            for ex in lanes:
                ex.shutdown()
        for bv, _ in slots:
            bv.release()
        for e2, b2 in extra_engines:
            b2.free()
            e2.close()
        wire_buf.free()
        eng_x.close()
        if dist.is_initialized():
            tt = torch.tensor([elapsed, elapsed_pipe], dtype=torch.float64, device="cpu" if dist.get_backend() != "nccl" else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed, elapsed_pipe = float(tt[0].item()), float(tt[1].item())
        msm_pairs = 3 + 2 * nbits + 19 * (hi - lo)
        stage_ms = {k: v[0] / reps for k, v in prof.items() if v[1]}
        serial_ms = {k: v[0] for k, v in prof_serial.items() if v[1]}
        dom = max(serial_ms, key=serial_ms.get) if serial_ms else None
        dom_s = serial_ms.get(dom, 0.0) / 1e3 if dom else 0.0
        wire_bytes = len(wire_joined)
        # what a batch moves over the host link: the wire bytes and the commitments (64 B per proof), both from page-locked buffers
        link_bytes = wire_bytes + len(v_packed)
        link = {"bytes_per_batch": link_bytes, "GBps": link_bytes / elapsed_pipe / 1e9, "GBps_one_batch_at_a_time": link_bytes / elapsed / 1e9,
                "peak_GBps": PCIE_PEAK_GBS, "frac": link_bytes / elapsed_pipe / 1e9 / PCIE_PEAK_GBS,
                "note": "host -> device bytes of this rank's share of a batch / seconds per batch; peak: PCIe Gen5 x16, 63 GB/s (MI355X_MICROARCH.md) -- "
                        "the bound a faster preparation would meet next on this format"}
        gpu_ms = sum(stage_ms.values())
        # algorithmic bytes of the dominant stage per batch: the preparation and the point decoding read the wire bytes once
        # (and write 32 B per scalar / 64 B per point); the MSM stages read 96 B per pair
        stage_bytes = {"rp_prepare": wire_bytes + 32 * msm_pairs, "ec_decompress": 33 * 19 * (hi - lo) + 64 * 19 * (hi - lo)}
        dom_bytes = stage_bytes.get(dom, ALGO_BYTES_PER_PAIR * msm_pairs)
        cpu = None
        if first and rank == 0 and world == 1 and not per_gpu and os.environ.get("BENCH_NO_CPU_BASELINE") != "1":
            try:
                cpu = c5_cpu_baseline(g, h, gs, hs, u, v_packed, wire_joined, wire_off_c, total, usable)
            except Exception as e:
                cpu = {"error": "%s: %s" % (type(e).__name__, e)}
        out_extra = {"cpu_baseline": cpu} if cpu is not None else {}
        if errors:
            out_extra["errors"] = sorted(set(errors))[:4]
        return {**out_extra, **{"metric": "range-proof verifies/sec (batched, 64-bit proofs, wire bytes in)", "value": total / elapsed_pipe, "unit": "verifies/s",
                "batch": total, "scaling": "weak (2^%d proofs per GPU)" % log_batch if per_gpu else "strong (one batch of 2^%d split over the ranks)" % log_batch, "seconds_per_batch": elapsed_pipe, "batches_in_flight": inflight, "batch_latency_s": elapsed,
                "verifies_per_s_one_batch_at_a_time": total / elapsed, "preparation": "device, one native call per batch (bpmi_rp_batch_verify_dev)",
                "accepted": all(oks), "corrupted_batch_rejected": rejected,
                "host_threads_per_rank": threads, "host_cores_usable": usable, "msm_pairs_per_rank": msm_pairs,
                "proves_per_s_one_gpu": prover_info["proves_per_s"], "distinct_proofs": distinct, "batch_prover": prover_info, "wire_bytes_per_batch": wire_bytes, "wire_bytes_per_proof": round(wire_bytes / max(hi - lo, 1), 1), "link": link,
                "gpu_stage_ms_per_batch": {k: round(v, 4) for k, v in stage_ms.items()},
                "gpu_stage_ms_per_batch_serial": {k: round(v, 4) for k, v in serial_ms.items()},
                "roofline": {"bound": "hbm", "kernel": "stage %s (the dominant GPU stage of a batch; duration from a batch whose stages run one after the other: gpu_stage_ms_per_batch_serial)" % dom,
                             "kernel_ms": dom_s * 1e3,
                             "achieved": (dom_bytes / dom_s / 1e9) if dom_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": (dom_bytes / dom_s / 1e9 / HBM_PEAK_GBS) if dom_s > 0 else None,
                             "algorithmic_bytes": dom_bytes, "dominant_gpu_stage": dom, "traffic": None,
                             "note": "GPU stages %.2f ms per batch (preparation kernels, point decoding, one MSM); one batch at a time takes %.2f ms "
                                     "(+ the %.1f MB upload from the page-locked receive buffer and the syncs), %d in flight %.2f ms per batch; integer-ALU bound like the MSM"
                                     % (gpu_ms, elapsed * 1e3, wire_bytes / 1e6, inflight, elapsed_pipe * 1e3)}}}

    res = run_format(wire, True)
    res["wire_format"] = "1 (the reference Proof object's fields, transcripts included: rangeproofs/codec.py)"
    try:
        r2 = run_format(wire2, False)
        res["wire_format_2"] = {k_: r2[k_] for k_ in ("value", "seconds_per_batch", "batch_latency_s", "verifies_per_s_one_batch_at_a_time", "accepted", "corrupted_batch_rejected",
                                                      "wire_bytes_per_batch", "wire_bytes_per_proof", "link", "gpu_stage_ms_per_batch", "gpu_stage_ms_per_batch_serial") if k_ in r2}
        res["wire_format_2"]["note"] = ("the same proofs without their three transcripts (csrc/rp_wire_v2_host.hpp): the device rebuilds them (k_rp_expand_v2) and runs "
                                        "the format-1 checks on the expansion; same verdicts (tests/test_gpu_configs.py::test_c5_wire_format_2_same_verdicts_as_format_1)")
        if "errors" in r2:
            res["wire_format_2"]["errors"] = r2["errors"]
    except Exception as e:
        res["wire_format_2"] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:
        r3 = run_format(wire3, False)
        res["wire_format_3"] = {k_: r3[k_] for k_ in ("value", "seconds_per_batch", "batch_latency_s", "verifies_per_s_one_batch_at_a_time", "accepted", "corrupted_batch_rejected",
                                                      "wire_bytes_per_batch", "wire_bytes_per_proof", "link", "gpu_stage_ms_per_batch", "gpu_stage_ms_per_batch_serial") if k_ in r3}
        res["wire_format_3"]["same_proofs_as_format_2_plus_y"] = wire3_ok
        res["wire_format_3"]["prover_seconds_per_batch"] = t_batch3
        res["wire_format_3"]["note"] = ("format 2 followed by the y coordinate of each of the proof's 18 points (32 B each): the verifier checks y^2 = x^3 + 7 and the "
                                        "parity of the encoding's tag instead of taking a square root per point (k_ec_decompress_wire; 253 squarings + 13 multiplications "
                                        "each, a quarter of a batch's device time on formats 1 and 2); a wrong y is an invalid proof; same verdicts "
                                        "(tests/test_gpu_batch_dev.py, tests/test_wire_v3_cpu.py)")
        if "errors" in r3:
            res["wire_format_3"]["errors"] = r3["errors"]
    except Exception as e:
        res["wire_format_3"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def extra_c2(eng, world, rank, dev, d_pts, d_sc, n, dlog, G64):
    """BASELINE config 2: one MSM of n = 2^16 pairs (the first 2^16 of the headline inputs), /root/reference/src/pippenger/
    pippenger.py:22-61.  pairs/s one call at a time (a caller that needs the result before it goes on) and with two calls in
    flight; per-stage times; the known-answer check."""
    import torch
    expect = eng.ec_mul_batch_bytes(G64, dlog.to_bytes(32, "little"), 1)
    got = eng.msm_dev(d_pts, d_sc, n)
    def measure():
        for _ in range(150):             # warm: ~60 ms of this very load (clocks; tools/step_ramp.py)
            eng.msm_dev(d_pts, d_sc, n)
        torch.cuda.synchronize(dev)
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.msm_dev(d_pts, d_sc, n)
        sync = (time.perf_counter() - t0) / reps
        eng.msm_dev_enqueue(0, d_pts, d_sc, n)
        t0 = time.perf_counter()
        for j in range(reps):
            if j + 1 < reps:
                eng.msm_dev_enqueue((j + 1) & 1, d_pts, d_sc, n)
            eng.msm_finish(j & 1)
        pipe2 = (time.perf_counter() - t0) / reps
        # ... and with three (the engine's three asynchronous slots): at this size an MSM is a chain of a dozen short kernels, two chains do
        # not fill the chip (the arithmetic of one MSM is ~75 us of the whole chip's issue slots)
        eng.msm_dev_enqueue(0, d_pts, d_sc, n)
        eng.msm_dev_enqueue(1, d_pts, d_sc, n)
        t0 = time.perf_counter()
        r3 = None
        for j in range(reps):
            if j + 2 < reps:
                eng.msm_dev_enqueue((j + 2) % 3, d_pts, d_sc, n)
            r3 = eng.msm_finish(j % 3)
        return sync, pipe2, (time.perf_counter() - t0) / reps, r3

    sync_s, pipe_s, pipe3_s, got3 = measure()
    eng.profile(1)
    eng.profile_reset()
    for _ in range(10):
        eng.msm_dev(d_pts, d_sc, n)
    prof = eng.profile_read()
    eng.profile(False)
    stage_ms = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items() if v[1]}
    acc_s = stage_ms.get("msm_accumulate", 0.0) / 1e3
    # the integer-issue figures, as for the headline: multiply-adds of the bucket additions (W windows x n mixed additions; the bucket
    # reduction's two GENERAL additions per bucket counted too, at 14/10.5 of a mixed one) against the chip's raw v_mad_u64_u32 rate.
    # Geometry as csrc/msm_host.hpp pick_window_bits: mixed window widths -- 256 // c windows of which the last 256 - (256 // c) c are
    # c + 1 bits wide with twice the buckets -- with c = 12 up to 19 000 pairs, 13 up to 185 000; 16 windows of 16 bits beyond
    isa = isa_counts() or {}
    mads = isa.get("v_mad_u64_u32_per_madd") or 1055
    geom = eng.msm_geometry(n)                      # the engine's own answer under the current options (bpmi_msm_geometry)
    c_bits, W, buckets = geom["window_bits"], geom["windows"], geom["buckets"]
    mad_accum = W * n * mads
    mad_reduce = 2 * buckets * mads * 14.0 / MULS_PER_MADD
    red_s = stage_ms.get("msm_bucket_reduce", 0.0) / 1e3

    def frac(work, secs):
        return (work / secs / 1e12 / RAW_MAD_TOPS) if secs > 0 else None
    alu = {"unit": "T lane multiply-adds/s (v_mad_u64_u32)", "peak": RAW_MAD_TOPS, "mads_per_madd": mads,
           "window_bits": c_bits, "windows": W, "buckets": buckets,
           "work": "%d windows x n mixed additions (accumulation) + 2 x %d general additions (bucket reduction, 14 / 10.5 of a mixed one)" % (W, buckets),
           "frac_vs_raw_mad_accumulate_kernel": frac(mad_accum, acc_s), "frac_vs_raw_mad_reduction_stage": frac(mad_reduce, red_s),
           "frac_vs_raw_mad_one_at_a_time": frac(mad_accum + mad_reduce, sync_s), "frac_vs_raw_mad_two_in_flight": frac(mad_accum + mad_reduce, pipe_s),
           "frac_vs_raw_mad_three_in_flight": frac(mad_accum + mad_reduce, pipe3_s),
           "frac_vs_raw_mad_accumulation_only_one_at_a_time": frac(mad_accum, sync_s),
           "note": "at this size one MSM's arithmetic is ~75 us of the whole chip's issue slots and its dozen kernels are a 0.22 ms chain of mostly "
                   "latency-bound stages (sort, segmented scan, the reduction's finish on quads) plus the host tail: the more calls in flight, the "
                   "closer to the arithmetic (DESIGN.md section 5)"}
    return {"metric": "Pippenger MSM scalar-point pairs/sec at n=2^16 (config C2)", "alu_roofline": alu, "value": n / pipe_s, "unit": "pairs/s", "n": n,
            "ms_per_msm_two_in_flight": pipe_s * 1e3, "ms_per_msm_three_in_flight": pipe3_s * 1e3, "ms_per_msm_one_at_a_time": sync_s * 1e3,
            "pairs_per_s_one_at_a_time": n / sync_s, "pairs_per_s_three_in_flight": n / pipe3_s,
            "result_ok": bool(got == expect and got3 == expect), "stage_ms_per_msm": stage_ms,
            "roofline": {"bound": "hbm", "kernel": "k_accum_l0 (msm_accumulate)", "kernel_ms": acc_s * 1e3,
                         "achieved": ALGO_BYTES_PER_PAIR * n / acc_s / 1e9 if acc_s > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ALGO_BYTES_PER_PAIR * n / acc_s / 1e9 / HBM_PEAK_GBS if acc_s > 0 else None, "traffic": None,
                         "note": "at this size every stage is a short chain of dependent point additions: latency-bound, not throughput-bound"}}


# ---- extra: config C3, inner-product-argument prover n = 2^20 ----------------------------------------


def extra_c3(eng, world, rank, dev, ready, logn=20):
    """Seconds per FastNIProver2.prove at n = 2^20 (/root/reference/src/innerproduct/inner_product_prover.py:70-110):
    g, h, a, b resident in HBM, 20 rounds of (c_L, c_R, L, R) -> host Fiat-Shamir -> fold.  With N > 1 the
    vectors are sharded cyclically (ShardedFastNIProver2) and the proof is the same one."""
    import torch
    import torch.distributed as dist
    from bulletproofs_amd.distributed import ShardedFastNIProver2
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.innerproduct import FastNIProver2
    from bulletproofs_amd.utils import elliptic_hash
    n = 1 << logn
    nl = n // world
    G64 = secp256k1.G.to_le64()

    def dev_points(seed):
        kb, _ = synth_scalars(nl, seed)
        d_k = eng.upload(kb)
        d_G = eng.upload(G64 * nl)
        d_p = eng.alloc(64 * nl)
        eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.ptr, d_k.ptr, nl, d_p.ptr))
        eng.sync()
        d_G.free()
        d_k.free()
        return d_p

    # rank r holds the elements i = r (mod world) of the global vectors: seeds depend on the rank
    d_g, d_h = dev_points(3000 + rank), dev_points(4000 + rank)
    d_a, d_b = eng.upload(synth_scalars(nl, 5000 + rank)[0]), eng.upload(synth_scalars(nl, 6000 + rank)[0])
    u = elliptic_hash(b"bench-u")

    def prove(profile):
        st = eng.ipa_create_dev(d_g, d_h, d_a, d_b, nl, u.to_le64())
        if profile:
            eng.profile(1)
            eng.profile_reset()
        t0 = time.perf_counter()
        pr = ShardedFastNIProver2(None, None, u, None, None, None, secp256k1, transcript=b"bench", engine=eng, state=st).prove()
        dt = time.perf_counter() - t0
        return dt, pr

    ready()                              # the shards are resident on every rank: the sharded prover's collectives start here
    prove(False)                         # warm: workspaces
    prove(False)                         # ... and clocks
    if dist.is_initialized():
        dist.barrier()
    times = []
    for _ in range(3):
        dt, pr = prove(False)
        times.append(dt)
    dt_prof, pr2 = prove(True)           # one more with stage timers (slower: every event is a bubble)
    prof = eng.profile_read()
    eng.profile(False)
    secs = min(times)
    # the same proof with the generators declared deployment constants (option ipa_fixed_generators: the tables of their odd multiples,
    # which the 16-way fold builds, are kept between proofs that name the same arrays); `value` stays the general case
    fixed = None
    try:
        eng.set_option("ipa_fixed_generators", 1)
        prove(False)
        ft = []
        for _ in range(3):
            dtf, prf = prove(False)
            ft.append(dtf)
        fixed = {"seconds": min(ft), "runs_s": [round(t, 5) for t in ft], "same_proof": bool(prf.transcript == pr.transcript),
                 "what": "bpmi_set_option(ctx, 'ipa_fixed_generators', 1): the fold's tables of 3P, 5P, 7P and beta x of the 2 x 2^%d generators are built once" % logn}
    except Exception as e:
        fixed = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        eng.set_option("ipa_fixed_generators", 0)
    if dist.is_initialized():
        tt = torch.tensor([secs], dtype=torch.float64, device="cpu" if dist.get_backend() != "nccl" else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        secs = float(tt.item())
    stage_ms = {k: v[0] for k, v in prof.items() if v[1]}
    dom = max(stage_ms, key=stage_ms.get) if stage_ms else None
    dom_s = stage_ms.get(dom, 0.0) / 1e3 if dom else 0.0
    algo = IPA_ALGO_BYTES_PER_ELEMENT * n
    same = (pr.transcript == pr2.transcript)
    for d in (d_g, d_h, d_a, d_b):
        d.free()
    return {"metric": "inner-product-argument prover seconds at n=2^%d" % logn, "value": secs, "unit": "s", "higher_is_better": False,
            "runs_s": [round(t, 5) for t in times], "rounds": len(pr.xs), "deterministic": bool(same), "with_fixed_generators": fixed,
            "transcript_sha256": hashlib.sha256(pr.transcript).hexdigest()[:16],
            "gpu_stage_ms_per_proof_with_timers": {k: round(v, 3) for k, v in stage_ms.items()},
            "seconds_with_stage_timers": dt_prof,
            "roofline": {"bound": "hbm", "kernel": "stage %s (sum over the proof's launches)" % dom, "kernel_ms": dom_s * 1e3,
                         "achieved": algo / secs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": algo / secs / 1e9 / HBM_PEAK_GBS,
                         "traffic": None,
                         "note": "960 algorithmic B per element over the whole proof (SURVEY 8d) / wall seconds of the proof; integer-ALU bound"}}


# ---- extra: config C4, aggregated range proof m = 128 x 64-bit ------------------------------------------


def extra_c4(eng, world, rank, dev, ready, m=128, nbits=64):
    """Seconds to prove and to verify ONE aggregated range proof over m = 128 values of 64 bits
    (/root/reference/src/rangeproofs/rangeproof_aggreg_prover.py:36-115, rangeproof_aggreg_verifier.py:55-108): vectors of
    n m = 8192 generators, one large Pedersen MSM per commitment, a 13-round inner-product argument.  Not sharded: with N > 1
    every rank runs the same proof (replicas) and rank 0's times are reported."""
    import torch
    from bulletproofs_amd.ec import Point
    from bulletproofs_amd.engine import set_default_engine
    from bulletproofs_amd.rangeproofs import AggregNIRangeProver, AggregRangeVerifier
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.utils import ModP, commitment, elliptic_hash, mod_hash
    set_default_engine(eng)
    nm = nbits * m
    G64 = secp256k1.G.to_le64()

    def gen_points(seed):                       # nm generators k_i * G (the reference derives them by hashing to the curve; only their number matters here)
        kb, _ = synth_scalars(nm, seed)
        raw = eng.ec_mul_batch_bytes(G64 * nm, kb, nm)
        return [Point.from_le64(raw[64 * i: 64 * i + 64]) for i in range(nm)]

    from bulletproofs_amd.ec import PackedPoints
    # lists of Points that carry their wire form (64 bytes per point): the generators of a deployment are fixed, they are packed once
    gs, hs = PackedPoints(gen_points(7000)), PackedPoints(gen_points(7001))
    g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
    vs = [ModP(int.from_bytes(hashlib.sha256(b"v%d" % j).digest()[:8], "big") % (1 << nbits), Q) for j in range(m)]      # values of nbits bits
    gammas = [mod_hash(b"gamma%d" % j, Q) for j in range(m)]
    Vs = [commitment(g, h, vs[j], gammas[j]) for j in range(m)]
    prove_s, verify_s = [], []
    proof = None
    for rep in range(3):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        proof = AggregNIRangeProver(vs, nbits, g, h, gs, hs, gammas, u, secp256k1, b"seed").prove()
        t1 = time.perf_counter()
        ok = AggregRangeVerifier(Vs, g, h, gs, hs, u, proof).verify()
        t2 = time.perf_counter()
        prove_s.append(t1 - t0)
        verify_s.append(t2 - t1)
    try:                                        # one commitment swapped for another value's: must be rejected
        rejected = not AggregRangeVerifier([Vs[1]] + Vs[1:], g, h, gs, hs, u, proof).verify()
    except Exception:
        rejected = True
    return {"metric": "aggregated range proof m=%d x %d-bit: seconds to prove / to verify" % (m, nbits), "value": min(prove_s), "unit": "s",
            "higher_is_better": False, "prove_s": round(min(prove_s), 5), "verify_s": round(min(verify_s), 5),
            "runs_prove_s": [round(t, 5) for t in prove_s], "runs_verify_s": [round(t, 5) for t in verify_s],
            "verified": bool(ok), "wrong_commitment_rejected": bool(rejected), "generators": 2 * nm,
            "note": "the reference-shaped Python call surface; the O(n m) scalar algebra runs in libbpmi's native host code (csrc/rp_algebra_host.hpp), "
                    "MSMs, the inner-product argument and the verifier's s-vector on the GPU; gs / hs are PackedPoints (lists of Points with their wire "
                    "form attached, packed once outside the timed region)"}
