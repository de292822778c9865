"""main() of bench.py: the MSM steps, the timed region, the roofline objects, the one JSON line."""
import argparse
import hashlib
import json
import os
import sys
import time

from .common import (ALGO_BYTES_PER_PAIR, FE_MUL_PEAK_G, HBM_PEAK_GBS, IPA_ALGO_BYTES_PER_ELEMENT, MULS_PER_MADD, Q, RAW_MAD_TOPS, REPO, committed_traffic, cpu_quota, isa_counts,
                     synth_scalars, usable_cpus)
from .cpu_baseline import cpu_baseline
from .extras import extra_c2, extra_c3, extra_c4, extra_c5
from .launch import PeerFailure, Ready, c5_inflight, launch_ranks, pipelined_exchange_loop, run_extras, strong_depth


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _r(v, digits):
    return round(v, digits) if isinstance(v, (int, float)) and not isinstance(v, bool) else v


def second_metric(out):
    """The second half of BASELINE.json's metric (range-proof verifies/sec: extra C5) at the top level of the line."""
    v = _get(out, "extra", "C5_batch_verify", "value")
    if v is None:
        return {}
    return {"metric2": "range-proof verifies/sec (batch of 2^14 x 64-bit proofs from wire bytes, %s batches in flight)" % _get(out, "extra", "C5_batch_verify", "batches_in_flight"),
            "value2": v, "unit2": "verifies/s"}


def summary_of(out):
    """Every BASELINE config in one compact object (< 600 bytes), emitted as the LAST key of the JSON line: a record that keeps only the
    tail of the line still shows C2 .. C5.  None = that extra did not run (or failed: see `extra`)."""
    ex = out.get("extra") or {}
    c2, c3, c4, c5 = ex.get("C2_msm_2e16") or {}, ex.get("C3_ipa_prover") or {}, ex.get("C4_aggregated_range_proof") or {}, ex.get("C5_batch_verify") or {}
    oks = [out.get("result_ok"), c2.get("result_ok"), c3.get("deterministic"), c4.get("verified"), c4.get("wrong_commitment_rejected"), c5.get("accepted"),
           c5.get("corrupted_batch_rejected"), _get(c5, "wire_format_3", "accepted"), _get(c5, "wire_format_3", "corrupted_batch_rejected"),
           _get(c5, "batch_prover", "byte_identical_to_single_proof_prover_on_sample"),
           _get(c5, "batch_prover", "aggregated", "byte_identical_to_AggregNIRangeProver_on_sample"), _get(ex, "MSM_strong", "result_ok")]
    ran = [v for v in oks if v is not None]
    return {"ms_per_step": _r(out.get("ms_per_step"), 4), "mad_frac_step": _r(_get(out, "alu_roofline", "frac_vs_raw_mad_step"), 3),
            "C2_ms_one": _r(c2.get("ms_per_msm_one_at_a_time"), 4), "C2_ms_two": _r(c2.get("ms_per_msm_two_in_flight"), 4), "C2_ms_three": _r(c2.get("ms_per_msm_three_in_flight"), 4),
            "C3_s": _r(c3.get("value"), 5), "C3_fixed_gens_s": _r(_get(c3, "with_fixed_generators", "seconds"), 5),
            "C4_prove_s": _r(c4.get("prove_s"), 5), "C4_verify_s": _r(c4.get("verify_s"), 5),
            "C5_verifies_per_s": _r(c5.get("value"), 0), "C5_one_batch_ms": _r((c5.get("batch_latency_s") or 0) * 1e3, 3) if c5 else None,
            "C5_link_GBps": _r(_get(c5, "link", "GBps"), 2), "C5_link_peak_GBps": _get(c5, "link", "peak_GBps"),
            "C5_v2_verifies_per_s": _r(_get(c5, "wire_format_2", "value"), 0), "C5_v3_verifies_per_s": _r(_get(c5, "wire_format_3", "value"), 0),
            "prover_proofs_per_s": _r(_get(c5, "batch_prover", "proves_per_s"), 0),
            "result_ok_all": bool(ran) and all(ran), "checks": len(ran)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # 0.2 s of timed MSMs: the fill and the drain of the two-deep pipeline are 1 % of it
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--logn", type=int, default=20)
    ap.add_argument("--preheat-ms", type=float, default=120.0,
                    help="untimed MSM steps for this long BEFORE the warm-up steps (and before every extra's timed region): after the "
                         "host-side input setup the GPU sits at idle clocks and needs ~40 steps (45 ms) to reach its steady ones "
                         "(tools/step_ramp.py, profiles/r03_clock_ramp_after_idle.txt); 0 = none")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C5 / C3 extra measurements")
    ap.add_argument("--no-pipeline", action="store_true", help="synchronous bpmi_msm_dev per step instead of the two-deep pipeline")
    ap.add_argument("--async-lanes", type=int, default=1, help="1: the in-flight MSMs run on the engine's lanes (one stream + workspace per slot)")
    ap.add_argument("--depth", type=int, default=2, choices=(2, 3), help="MSMs in flight in the timed loop (slots rotate)")
    ap.add_argument("--soak-seconds", type=float, default=6.0,
                    help="untimed MSMs after the timed region, so that an external sampler (rocm-smi every few seconds) sees the GPU busy")
    ap.add_argument("--cpu-logn", type=int, default=0, help="CPU baseline on the first 2^k pairs; 0 = the bench size itself (--logn)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="bpmi_set_option passthrough (tuning experiments)")
    ap.add_argument("--extra-scale", choices=("full", "small"), default="full",
                    help="small: the extras at test sizes (C5 2^8 proofs, C3 n = 2^12, C4 4 x 16 bits, the strong MSM at --logn): the N-rank "
                         "control flow with every extra in a minute (tests/test_gpu_dist.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libbpmi has no CPU fallback")
    local_dev = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    backend = None
    # BENCH_FORCE_DIST=1: initialise the process group even for ONE rank, so that a single-GPU box exercises the RCCL branch
    # of every collective this file uses (tests/test_gpu_dist.py)
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL ("nccl") is the backend of record; BENCH_DIST_BACKEND=gloo exists only to dry-run the N > 1
        # control flow with several ranks sharing ONE GPU (RCCL refuses duplicate devices)
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        import datetime
        # a rank that dies must cost its peers minutes, not the runtime's default half hour
        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("BENCH_PG_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
        assert dist.get_world_size() == args.gpus, "rendezvous gave %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus)
    # control plane of the extras (who is ready, who failed, with what text): host objects over gloo, whatever the data-path
    # backend is -- a timeout there is a Python exception on the waiting ranks, not a watchdog abort
    ctl = None
    if use_dist:
        ctl = dist.new_group(backend="gloo", timeout=pg_timeout) if backend == "nccl" else dist.group.WORLD

    def gather_objs(obj):
        """[obj of rank 0, ..., obj of rank N-1] on every rank (control group)."""
        if not use_dist:
            return [obj]
        got = [None] * world
        dist.all_gather_object(got, obj, group=ctl)
        return got

    import bulletproofs_amd  # noqa: F401
    from bulletproofs_amd.engine import Engine

    # a dedicated (non-null) torch stream, shared with the engine, so that torch.cuda.Event
    # and the library's own HIP events time the stream the kernels are launched on
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    eng = Engine(device=local_dev, stream=stream.cuda_stream)
    if args.async_lanes:
        eng.set_option("async_lanes", 1)
    for kv in args.opt:
        name, value = kv.split("=")
        eng.set_option(name, int(value))

    n_total = 1 << args.logn
    n = n_total if args.scaling == "weak" else n_total // world      # pairs on this rank
    # ---- synthetic inputs, resident in HBM ------------------------------------------
    # points P_i = k_i * G generated ON THE GPU (bpmi_ec_mul_batch), k_i and e_i from SHA-256.
    # weak: rank r draws its own n pairs (seeds 1000 + r / r); strong: every rank derives the same
    # n_total-pair problem and keeps the contiguous shard [rank * n, (rank + 1) * n).
    G64 = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798).to_bytes(32, "little") + \
          (0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8).to_bytes(32, "little")
    t_in = time.time()
    if args.scaling == "weak":
        kb, kv = synth_scalars(n, 1000 + rank)
        eb, ev = synth_scalars(n, rank)
    else:
        kb, kv = synth_scalars(n_total, 1000)
        eb, ev = synth_scalars(n_total, 0)
        lo = rank * n
        kb, kv, eb, ev = kb[32 * lo: 32 * (lo + n)], kv[lo: lo + n], eb[32 * lo: 32 * (lo + n)], ev[lo: lo + n]
    d_k = torch.frombuffer(bytearray(kb), dtype=torch.uint8).to(dev)
    d_G = torch.frombuffer(bytearray(G64), dtype=torch.uint8).to(dev).repeat(n)
    d_pts = torch.empty(64 * n, dtype=torch.uint8, device=dev)
    eng._ck(eng.lib.bpmi_ec_mul_batch_dev(eng.ctx, d_G.data_ptr(), d_k.data_ptr(), n, d_pts.data_ptr()))
    eng.sync()
    del d_G, d_k
    d_sc = torch.frombuffer(bytearray(eb), dtype=torch.uint8).to(dev)
    # known answer of this rank's shard: sum e_i * P_i = (sum e_i k_i mod q) * G, one scalar multiplication
    # by a different kernel (k_ec_mul_batch: a double-and-add ladder, no buckets)
    local_dlog = sum(e * k for e, k in zip(ev, kv)) % Q
    c2_n = min(n, 1 << 16)
    c2_dlog = sum(e * k for e, k in zip(ev[:c2_n], kv[:c2_n])) % Q
    ns_strong = n // world                 # extra MSM_strong: this rank's share (its first n / N pairs) of ONE n-pair MSM
    strong_dlog = sum(e * k for e, k in zip(ev[:ns_strong], kv[:ns_strong])) % Q
    del kb, kv, eb, ev
    t_in = time.time() - t_in

    from bulletproofs_amd.distributed import ShardedMSM
    sharded = ShardedMSM(engine=eng)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    host_t = [0.0, 0.0, 0.0, 0.0, 0]      # host seconds in: enqueue, finish (wait + tail), combine_wait, combine_begin; steps

    def run_steps(k):
        """k MSM steps; every step's global result is complete when this returns."""
        res = None
        if args.no_pipeline:
            for _ in range(k):
                res = sharded.multiexp_local_dev(d_pts, d_sc, n)
            return res
        # two-deep pipeline: MSM j + 1 is queued before MSM j is finished (host tail) and combined
        # (N > 1: ONE all_gather of the 64-byte partials + bpmi_ec_sum fold on every rank)
        # the exchange of step j is started (queued on its own stream) and collected one iteration later, so the host
        # never waits for the fold kernel before it has fed the GPU its next MSM
        # (round 6: the slot msm_finish has just freed is refilled AT ONCE, before the exchange's ~0.08 ms of host work -- behind it the next
        # MSM started that much later and every step with a process group was that much longer; without one the order is the same as before)
        D = args.depth
        no_exchange = os.environ.get("BENCH_EXCHANGE") == "0"
        for j in range(min(k, D)):
            eng.msm_dev_enqueue(j % D, d_pts, d_sc, n)
        pending = None
        for j in range(k):
            t1 = time.perf_counter()
            part = eng.msm_finish(j % D)
            t2 = time.perf_counter()
            if j + D < k:
                eng.msm_dev_enqueue(j % D, d_pts, d_sc, n)
            t3 = time.perf_counter()
            if no_exchange:                                  # (BENCH_EXCHANGE=0, experiments: a process group without the per-step exchange)
                res, t4, t5 = part, t3, t3
                host_t[0] += t3 - t2; host_t[1] += t2 - t1; host_t[4] += 1
                continue
            if pending is not None:
                res = sharded.combine_wait(pending)
            t4 = time.perf_counter()
            pending = sharded.combine_begin(part)
            t5 = time.perf_counter()
            host_t[0] += t3 - t2; host_t[1] += t2 - t1; host_t[2] += t4 - t3; host_t[3] += t5 - t4; host_t[4] += 1
        if pending is not None:
            res = sharded.combine_wait(pending)
        return res

    def preheat():
        """Untimed: the same pipelined steps for --preheat-ms (the clocks ramp over ~45 ms of work after an idle second)."""
        if args.preheat_ms <= 0:
            return
        t_h = time.perf_counter()
        while (time.perf_counter() - t_h) * 1e3 < args.preheat_ms:
            if use_dist:                     # with a process group: the very loop that is timed, exchange included (the first dozens of
                run_steps(8)                 # collectives of a group are slower -- work objects, events, staging -- and belong here, not into 20 timed steps)
                continue
            for sl in range(args.depth):
                eng.msm_dev_enqueue(sl, d_pts, d_sc, n)
            for sl in range(args.depth):
                eng.msm_finish(sl)
    # Round 6: the FIRST collective of an RCCL group costs tens of milliseconds (lazy set-up inside the first barrier); as the barrier that
    # opens the timed region it left the GPU idle that long right after the warm-up, the clocks dropped, and a 20-step region ran on the
    # ramp: 1.01-1.02 ms per step against 0.95 without a group -- nothing to do with the exchange (a 4 000-step run reads 0.943 either way;
    # profiles/r06_process_group_cost.txt).  One barrier BEFORE the untimed steps takes the set-up; the contract's barrier is then quick.
    barrier()
    preheat()
    result = run_steps(args.warmup)
    sharded.exchange_us()
    host_t[:] = [0.0, 0.0, 0.0, 0.0, 0]
    if not os.environ.get("BENCH_NO_KERNEL_EVENTS"):       # (experiments only: what the two events per step cost)
        eng.profile(2)          # HIP events around the dominant kernel only: each recorded event is a ~10 us bubble
    eng.profile_reset()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    t0 = time.perf_counter()
    result = run_steps(args.steps)
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    exchange_us = sharded.exchange_us()
    host_ms = {"enqueue": host_t[0], "finish_wait_and_tail": host_t[1], "combine_wait": host_t[2], "combine_begin": host_t[3]}
    host_ms = {k_: round(v / max(host_t[4], 1) * 1e3, 4) for k_, v in host_ms.items()}
    ev_ms = ev0.elapsed_time(ev1)
    prof = eng.profile_read()
    # per-stage breakdown: a few extra, UNTIMED synchronous steps with events around every stage
    eng.profile(1)
    eng.profile_reset()
    for _ in range(min(5, args.steps)):
        sharded.multiexp_local_dev(d_pts, d_sc, n)
    prof_all = eng.profile_read()
    eng.profile(False)

    n_ranks_seen = 1
    ms_by_rank = [elapsed / args.steps * 1e3]
    if use_dist:
        cpu_side = dist.get_backend() != "nccl"
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if cpu_side else dev)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        ms_by_rank = [float(t.item()) / args.steps * 1e3 for t in every]
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ones = torch.ones(1, dtype=torch.int64, device="cpu" if cpu_side else dev)
        dist.all_reduce(ones)
        n_ranks_seen = int(ones.item())
        dl = [None] * world
        dist.all_gather_object(dl, local_dlog)
        global_dlog = sum(dl) % Q
    else:
        global_dlog = local_dlog
    expect = eng.ec_mul_batch_bytes(G64, global_dlog.to_bytes(32, "little"), 1)
    result_ok = bool(result == expect)

    pairs_per_s = world * n * args.steps / elapsed
    acc_ms, acc_calls = prof["msm_accumulate"]
    acc_avg_s = acc_ms / max(acc_calls, 1) / 1e3
    achieved_gbs = ALGO_BYTES_PER_PAIR * n / acc_avg_s / 1e9 if acc_avg_s > 0 else 0.0
    stages = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof_all.items() if v[1]}
    geom = eng.msm_geometry(n, pipelined=not args.no_pipeline)      # the engine's own answer (csrc/msm_host.hpp msm_pick_geometry), not a constant
    windows = geom["windows"] * geom["slices"] * geom["pairs_per_slice"] / n      # bucket additions per pair
    isa = isa_counts()
    traffic, traffic_corrected, traffic_src = committed_traffic(args.logn if args.scaling == "weak" or world == 1 else -1)

    madds_per_launch = n * windows
    out = {
        "metric": "Pippenger MSM scalar-point pairs/sec at n=2^20",
        "value": pairs_per_s,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_by_rank": [round(v, 4) for v in ms_by_rank],
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u32x9 (29-bit limbs, 256-bit modular integer)",
        "data": "synthetic",
        "config": {"workload": ("MSM n=2^%d per GPU" % args.logn if args.scaling == "weak" else "ONE MSM n=2^%d split over %d GPUs" % (args.logn, world)) +
                               " over secp256k1, uniform 256-bit scalars (SHA-256), points k_i*G, inputs resident in HBM, "
                               "64-byte result to host every step",
                   "pairs_per_gpu": n, "sharding": "pairs across ranks, one all_gather of 64 B partials per step",
                   "pipeline": "synchronous" if args.no_pipeline else "%d MSMs in flight (bpmi_msm_dev_enqueue / bpmi_msm_finish)" % args.depth},
        "n_ranks_seen": n_ranks_seen,
        "dist_backend": backend,
        "result_ok": result_ok,
        "result_check": "timed MSM result == (sum e_i k_i mod q) * G computed by k_ec_mul_batch (different kernel), outside the timed region",
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_guide_corrected": traffic_corrected,
                     "traffic_note": "traffic = FETCH_SIZE + WRITE_SIZE as counted (raw); the guide's x2 on FETCH_SIZE is calibrated on wide coalesced reads and "
                                     "over-corrects this kernel's 64-B gathers (requested bytes 1.14e9 per launch): the corrected figure is kept beside it",
                     "traffic_measured_in_run": False,
                     "traffic_source": traffic_src,
                     "kernel": "k_accum_l0 (msm_accumulate)", "kernel_avg_ms": acc_avg_s * 1e3,
                     "duration_used": "kernel_avg_ms = the kernel's average over the timed steps from HIP events on its launch streams, i.e. IN the "
                                      "pipeline, beside the other lane's sort / reduction (alone, in synchronous steps, it is stage_ms_per_msm.msm_accumulate)",
                     "frac_step": ALGO_BYTES_PER_PAIR * n / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                     "frac_kernel_alone": (ALGO_BYTES_PER_PAIR * n / (stages["msm_accumulate"] / 1e3) / 1e9 / HBM_PEAK_GBS) if stages.get("msm_accumulate") else None,
                     "note": "integer-ALU bound path: 96 algorithmic B/pair vs ~1.5e4 multiply-adds/pair; frac_step = the same bytes over the whole step (ms_per_step)"},
        "alu_roofline": {"kernel": "k_accum_l0", "unit": "T lane multiply-adds/s (v_mad_u64_u32)",
                         "frac_vs_own_fe_mul": (madds_per_launch * MULS_PER_MADD / acc_avg_s / 1e9 / FE_MUL_PEAK_G) if acc_avg_s > 0 else 0.0,
                         "own_fe_mul": {"achieved_G_mul_s": madds_per_launch * MULS_PER_MADD / acc_avg_s / 1e9 if acc_avg_s > 0 else 0.0, "peak_G_mul_s": FE_MUL_PEAK_G,
                                        "note": "against the product's own fe_mul in isolation (profiles/r03_fe_microbench.txt, V8): NOT a hardware peak, kept for "
                                                "continuity with rounds 1-3 where it was `frac`"},
                         "geometry": geom,
                         "work": "%d windows x n mixed additions x %.1f multiplication-equivalents (8M + 2S)" % (geom["windows"], MULS_PER_MADD)},
        "stage_ms_per_msm": stages,
        "hip_event_ms_per_step": ev_ms / args.steps,
        "preheat_ms": args.preheat_ms,
        "host_ms_per_step": host_ms,
        "exchange_us": None if exchange_us is None else round(exchange_us, 1),      # device time of one RCCL exchange (copy up, all_gather of 64 B, fold kernel, copy down); None without a process group
        "input_setup_s": round(t_in, 2),
        "result_x_lo": result[:8].hex(),
    }
    if isa and acc_avg_s > 0:
        mads = isa["v_mad_u64_u32_per_madd"]
        lane_mads = madds_per_launch * mads / acc_avg_s / 1e12
        step_mads = madds_per_launch * mads / (elapsed / args.steps) / 1e12
        alone_s = stages.get("msm_accumulate", 0.0) / 1e3
        out["alu_roofline"].update({
            "achieved": lane_mads, "peak": RAW_MAD_TOPS, "frac": lane_mads / RAW_MAD_TOPS,      # = frac_vs_raw_mad: the hardware figure IS the headline one (round 4)
            "frac_vs_raw_mad": lane_mads / RAW_MAD_TOPS,                   # on kernel_avg_ms (the kernel in the pipeline)
            "frac_vs_raw_mad_step": step_mads / RAW_MAD_TOPS,               # the same multiply-adds over the WHOLE step (ms_per_step): what the chip delivers per MSM
            "frac_vs_raw_mad_kernel_alone": (madds_per_launch * mads / alone_s / 1e12 / RAW_MAD_TOPS) if alone_s > 0 else None})
        out["alu_roofline"]["raw_mad"] = {"achieved_T_lane_ops": lane_mads, "peak_T_lane_ops": RAW_MAD_TOPS,
                                          "mads_per_madd": mads, "instructions_per_madd": isa["instructions_per_madd"],
                                          "source": "profiles/r04_isa_counts.json (ISA of k_accum_l0's main path), "
                                                    "profiles/r01_fe_microbench.txt (raw v_mad_u64_u32 rate)"}

    usable = usable_cpus()
    out["host_budget"] = {"usable_cpus": usable, "ranks_on_this_host": world, "cpus_per_rank": max(1, usable // world),
                          "c5_batches_in_flight": c5_inflight(usable, world), "c5_host_threads_per_rank": max(1, min(32, usable // world)),
                          "cpu_baseline_threads": usable if world == 1 else 0,
                          "note": "every batch slot of C5 is a host thread of its rank; the C oracle (cpu_baseline) runs on rank 0 at N = 1 only"}

    if not args.no_extra:
        out["extra"] = {}
        small = args.extra_scale == "small"

        def msm_strong(e_, w_, r_, d_, ready):
            """ONE MSM of n pairs split over the N ranks (n / N each, the exchange of 64-byte partials every step): the strong-scaling
            line of the MSM in the same run as the weak headline.  Below 185 000 pairs per rank three MSMs are kept in flight (the measured
            winner at those sizes: 0.14 against 0.18 ms at 2^16).  Beside the measured time: the same loop on this rank
            WITHOUT the exchange (`expected_ms_per_step_if_ideal`: what N perfect GPUs could deliver at this per-rank size -- fixed latencies
            dominate small shards, so a 3x strong result on 8 GPUs is the expectation, not a defect) and the exchange's own device time."""
            ready()
            steps = min(args.steps, 50)
            D = strong_depth(ns_strong)

            def loop(exchange):
                return pipelined_exchange_loop(lambda sl: e_.msm_dev_enqueue(sl, d_pts, d_sc, ns_strong), e_.msm_finish, sharded.combine_begin,
                                               sharded.combine_wait, steps, D, exchange)

            for _ in range(3):
                sharded.multiexp_local_dev(d_pts, d_sc, ns_strong)
            loop(False)
            torch.cuda.synchronize(dev)
            t_s = time.perf_counter()
            loop(False)                              # this rank alone: no collective, no barrier
            ideal = (time.perf_counter() - t_s) / steps
            sharded.exchange_us()
            barrier()
            t_s = time.perf_counter()
            res_s = loop(True)
            barrier()
            dt = time.perf_counter() - t_s
            xus = sharded.exchange_us()
            dts = gather_objs(dt)
            ideals = gather_objs(ideal)
            dl_ = gather_objs(strong_dlog)
            want = e_.ec_mul_batch_bytes(G64, (sum(dl_) % Q).to_bytes(32, "little"), 1)
            return {"metric": "Pippenger MSM scalar-point pairs/sec, ONE MSM of n = %d pairs split over %d GPUs" % (ns_strong * w_, w_),
                    "value": ns_strong * w_ * steps / max(dts), "unit": "pairs/s", "scaling": "strong", "steps": steps, "msms_in_flight": D,
                    "ms_per_step": max(dts) / steps * 1e3, "ms_per_step_by_rank": [round(v / steps * 1e3, 4) for v in dts],
                    "expected_ms_per_step_if_ideal": round(max(ideals) * 1e3, 4),
                    "expected_note": "the slowest rank's own pipelined MSM of %d pairs WITHOUT the exchange: the per-rank floor at this shard size "
                                     "(one 2^20-pair MSM on one GPU is ms_per_step of the headline)" % ns_strong,
                    "exchange_us": None if xus is None else round(xus, 1),
                    "pairs_per_gpu": ns_strong, "result_ok": bool(res_s == want)}

        extras = [("C2_msm_2e16", lambda e_, w_, r_, d_, ready: extra_c2(e_, w_, r_, d_, d_pts, d_sc, c2_n, c2_dlog, G64)),
                  ("C5_batch_verify", (lambda *a: extra_c5(*a, log_batch=8)) if small else extra_c5),
                  ("C3_ipa_prover", (lambda *a: extra_c3(*a, logn=12)) if small else extra_c3),
                  ("C4_aggregated_range_proof", (lambda *a: extra_c4(*a, m=4, nbits=16)) if small else extra_c4)]
        if world > 1:         # the same verifier with 2^14 proofs per GPU: a rank's 2048-proof share of the fixed batch is mostly fixed latencies
            extras.insert(1, ("C5_batch_verify_per_gpu_batches", (lambda *a: extra_c5(*a, log_batch=8, per_gpu=True)) if small
                              else (lambda *a: extra_c5(*a, per_gpu=True))))
            if args.scaling == "weak":
                extras.insert(0, ("MSM_strong", msm_strong))
        out["extra"] = run_extras(extras, (eng, world, rank, dev), gather_objs, rank, lambda: torch.cuda.synchronize(dev))

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(min(args.cpu_logn or args.logn, args.logn), d_pts, d_sc, eng)

    out.update(second_metric(out))
    out["summary"] = summary_of(out)                     # LAST key: the driver's record keeps the tail of the line
    # soak: keep the GPU visibly busy for an external sampler; not part of any reported number
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < args.soak_seconds:
        run_steps(8)
    barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()
