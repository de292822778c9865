#!/usr/bin/env python3
"""bench.py -- headline benchmark: Pippenger MSM scalar-point pairs/s at n = 2^20 over
secp256k1 on MI355X, plus the second half of BASELINE.json's metric (range-proof verifies/s,
config C5) and the inner-product-argument prover of config C3 as `extra`.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--logn 20] [--scaling weak|strong]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU; it
starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>`
as a CHILD process (one rank per GPU over RCCL), relays rank 0's JSON line and exits with the
child's code.  Under torch.distributed.run (the driver's own launch form) it is a rank.

One "step" = one MSM of n pairs per GPU with scalars and points already resident in HBM; the
64-byte result comes back to the host every step.  Steps are pipelined two deep through
bpmi_msm_dev_enqueue / bpmi_msm_finish: the host tail of step k (and, with N > 1, the
all_gather of the 64-byte partials + bpmi_ec_sum fold) overlaps the kernels of step k + 1;
all K results are complete inside the timed region.
  weak   (default): every rank owns its own n pairs; the global MSM has N*n pairs.
  strong: ONE MSM of n pairs split N ways (n/N pairs per rank), north_star's "large MSMs
          shard across the GPUs of one node".

Prints ONE JSON line on rank 0 (driver contract), with
  roofline     -- the dominant kernel (k_accum_l0) against the HBM roofline that north_star
                  prescribes: algorithmic bytes = 96 B/pair, duration from HIP events on the
                  launch stream inside the timed region;
  alu_roofline -- the honest utilisation figure for this integer path: multiply-add lane
                  operations per second against the raw v_mad_u64_u32 rate of the chip;
  cpu_baseline -- the plain-C oracle MSM ("port") on this host's cores, bounded sample;
  result_ok    -- the timed MSM's 64 bytes against the known answer (sum e_i k_i) * G;
  extra        -- C5 batch verification (verifies/s) and C3 IPA prover (seconds), each with its
                  own roofline object.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# The two MSM lanes must sit on DIFFERENT hardware queues to overlap.  HIP maps streams onto GPU_MAX_HW_QUEUES (default 4)
# queues round-robin in creation order; once RCCL and the framework have created theirs, both lanes can land on one queue and
# the pipeline degrades to the one-lane rate (measured: 1.27 ms per step against 1.13 with 8 queues, profiles/r02_hw_queues.txt).
# Must be set before the HIP runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

Q = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
ALGO_BYTES_PER_PAIR = 96       # 32-B scalar + 64-B affine point
IPA_ALGO_BYTES_PER_ELEMENT = 960   # whole proof, per element of the n-vector
# multiply-add content of one bucket update (xyzz_madd, csrc/curve.hpp): v_mad_u64_u32 per wave-lane, counted in
# the ISA of k_accum_l0 (profiles/r04_isa_counts.json); and the chip's raw rate for that instruction
MADS_PER_MADD = None           # filled from profiles/r04_isa_counts.json when present
RAW_MAD_TOPS = 28.85           # T lane-ops/s, tools/fe_microbench.hip (profiles/r01_fe_microbench.txt)
MULS_PER_MADD = 10.5           # 8M + 2S plus carries/subtractions in multiplication-equivalents
FE_MUL_PEAK_G = 221.4          # the product's own fe_mul in isolation, G multiplications/s (profiles/r03_fe_microbench.txt, V8; round 1's fe_mul: 196)


def synth_scalars(n, seed):
    """e_i = SHA-256("bpmi/scalar" || seed || LE64(i)) mod q -> (bytes, list of ints)."""
    pre = b"bpmi/scalar" + seed.to_bytes(8, "little")
    out = bytearray(32 * n)
    vals = [0] * n
    sha = hashlib.sha256
    for i in range(n):
        v = int.from_bytes(sha(pre + i.to_bytes(8, "little")).digest(), "big") % Q
        vals[i] = v
        out[32 * i: 32 * i + 32] = v.to_bytes(32, "little")
    return bytes(out), vals


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as a child process.  Nothing in this
    process has touched a GPU (no torch.cuda call, no libbpmi call), and it never execs."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return proc.returncode if (proc.returncode or line is not None) else 1


class PeerFailure(Exception):
    """Some rank failed in the local phase of an extra; .bad = {rank: text}."""

    def __init__(self, bad):
        super().__init__("; ".join("rank %d: %s" % kv for kv in sorted(bad.items())))
        self.bad = bad


class Ready:
    """Every extra calls ready() ONCE, after its local setup (inputs, allocations, warm-up: where a rank can fail on its own)
    and before its first collective: the ranks exchange `None` or an error text over the control group, and if any rank
    failed, ALL of them leave the extra with PeerFailure -- nobody waits in a collective for a rank that is gone."""

    def __init__(self, gather):
        self.gather, self.called = gather, False

    def __call__(self, err=None):
        self.called = True
        bad = {r: t for r, t in enumerate(self.gather(err)) if t}
        if bad:
            raise PeerFailure(bad)


def run_extras(extras, call_args, gather, rank, sync=lambda: None):
    """Run [(name, fn)] one after the other on every rank; fn(*call_args, ready) -> dict.  Returns {name: dict}.
    The contract that keeps one failing rank from costing the others (or the headline line):
      * every rank makes exactly TWO exchanges per extra over the control group (`gather`): ready() -- inside fn, after its local
        setup and before its first collective, or by this wrapper when fn has none or failed before it -- and the report at the end;
      * a rank that raises in its local phase tells the others through ready(text): they all leave the extra with PeerFailure
        before any collective, and every rank's entry carries `errors_by_rank`;
      * a failure of the control group itself (a rank that vanished: the exchange times out) marks the group broken: the
        remaining extras are skipped, not waited for.
    BENCH_INJECT_FAILURE="<extra>:<rank>" makes that rank raise at the start of that extra (tests)."""
    inject = os.environ.get("BENCH_INJECT_FAILURE", "")
    results, dist_broken = {}, False
    for name, fn in extras:
        if dist_broken:
            results[name] = {"error": "skipped: the control group failed in an earlier extra"}
            continue
        ready = Ready(gather)
        res = None
        try:
            if inject.split(":")[:2] == [name, str(rank)]:
                raise RuntimeError("injected failure in %s on rank %d" % (name, rank))
            res = fn(*call_args, ready)
            if not ready.called:     # an extra without collectives: the exchange still happens once per rank and extra,
                try:                 # and this rank keeps its own (complete) result beside the others' errors
                    ready()
                except PeerFailure as pf:
                    if isinstance(res, dict):
                        res["errors_by_rank"] = {str(k_): v for k_, v in pf.bad.items()}
        except PeerFailure as pf:        # another rank failed before the extra's collectives: every rank leaves it here
            res = {"error": "skipped: " + str(pf), "errors_by_rank": {str(k_): v for k_, v in pf.bad.items()}}
        except Exception as e:      # an extra must never cost the headline line
            res = {"error": "%s: %s" % (type(e).__name__, e)}
            if not ready.called:    # the others wait in ready(): tell them
                try:
                    ready(res["error"])
                except PeerFailure as pf:
                    res["errors_by_rank"] = {str(k_): v for k_, v in pf.bad.items()}
                except Exception as e2:
                    dist_broken = True
                    res["control_plane"] = "%s: %s" % (type(e2).__name__, e2)
        # end of the extra: every rank reports (this is also the barrier between two extras)
        if not dist_broken:
            try:
                sync()
                sts = gather(res.get("error") if isinstance(res, dict) else None)
                bad = {str(r_): t_ for r_, t_ in enumerate(sts) if t_}
                if bad and isinstance(res, dict):
                    res.setdefault("errors_by_rank", bad)
            except Exception as e2:
                dist_broken = True
                if isinstance(res, dict):
                    res["control_plane"] = "%s: %s" % (type(e2).__name__, e2)
        results[name] = res
    return results


def c5_inflight(usable, world):
    """Batches in flight of the C5 extra: every slot is a host thread of its rank (BENCH_C5_INFLIGHT overrides)."""
    return max(1, int(os.environ.get("BENCH_C5_INFLIGHT", "0")) or min(8, max(2, usable // world)))


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
        D = args.depth
        for j in range(min(k, D - 1)):
            eng.msm_dev_enqueue(j % D, d_pts, d_sc, n)
        pending = None
        for j in range(k):
            ta = time.perf_counter()
            if j + D - 1 < k:
                eng.msm_dev_enqueue((j + D - 1) % D, d_pts, d_sc, n)
            tb = time.perf_counter()
            part = eng.msm_finish(j % D)
            tc = time.perf_counter()
            if pending is not None:
                res = sharded.combine_wait(pending)
            td = time.perf_counter()
            pending = sharded.combine_begin(part)
            te = time.perf_counter()
            host_t[0] += tb - ta; host_t[1] += tc - tb; host_t[2] += td - tc; host_t[3] += te - td; host_t[4] += 1
        if pending is not None:
            res = sharded.combine_wait(pending)
        return res

    def preheat():
        """Untimed: the same pipelined steps for --preheat-ms (the clocks ramp over ~45 ms of work after an idle second)."""
        if args.preheat_ms <= 0:
            return
        t_h = time.perf_counter()
        while (time.perf_counter() - t_h) * 1e3 < args.preheat_ms:
            for sl in range(args.depth):
                eng.msm_dev_enqueue(sl, d_pts, d_sc, n)
            for sl in range(args.depth):
                eng.msm_finish(sl)
    preheat()
    result = run_steps(args.warmup)
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
    windows = 16 if n >= (1 << 15) else 32          # pick_window_bits (csrc/msm_host.hpp): c = 16 -> 16 windows
    isa = isa_counts()
    traffic, traffic_src = committed_traffic(args.logn if args.scaling == "weak" or world == 1 else -1)

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
                         "work": "%d windows x n mixed additions x %.1f multiplication-equivalents (8M + 2S)" % (windows, MULS_PER_MADD)},
        "stage_ms_per_msm": stages,
        "hip_event_ms_per_step": ev_ms / args.steps,
        "preheat_ms": args.preheat_ms,
        "host_ms_per_step": host_ms,
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
            line of the MSM in the same run as the weak headline."""
            ready()
            steps = min(args.steps, 50)
            for _ in range(3):
                sharded.multiexp_local_dev(d_pts, d_sc, ns_strong)
            barrier()
            t_s = time.perf_counter()
            e_.msm_dev_enqueue(0, d_pts, d_sc, ns_strong)
            res_s, pend = None, None
            for j in range(steps):
                if j + 1 < steps:
                    e_.msm_dev_enqueue((j + 1) & 1, d_pts, d_sc, ns_strong)
                part = e_.msm_finish(j & 1)
                if pend is not None:
                    res_s = sharded.combine_wait(pend)
                pend = sharded.combine_begin(part)
            res_s = sharded.combine_wait(pend)
            barrier()
            dt = time.perf_counter() - t_s
            dts = gather_objs(dt)
            dl_ = gather_objs(strong_dlog)
            want = e_.ec_mul_batch_bytes(G64, (sum(dl_) % Q).to_bytes(32, "little"), 1)
            return {"metric": "Pippenger MSM scalar-point pairs/sec, ONE MSM of n = %d pairs split over %d GPUs" % (ns_strong * w_, w_),
                    "value": ns_strong * w_ * steps / max(dts), "unit": "pairs/s", "scaling": "strong", "steps": steps,
                    "ms_per_step": max(dts) / steps * 1e3, "ms_per_step_by_rank": [round(v / steps * 1e3, 4) for v in dts],
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

    # soak: keep the GPU visibly busy for an external sampler; not part of any reported number
    t_s = time.perf_counter()
    while time.perf_counter() - t_s < args.soak_seconds:
        run_steps(8)
    barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def isa_counts():
    try:
        for name in ("r04_isa_counts.json", "r03_isa_counts.json", "r02_isa_counts.json"):
            path = os.path.join(REPO, "profiles", name)
            if os.path.exists(path):
                with open(path) as f:
                    return json.load(f)["k_accum_l0_madd_main_path"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def committed_traffic(logn):
    """Per-launch HBM bytes of the dominant kernel from the committed rocprofv3 --pmc passes (NOT measured in this run)."""
    for name in ("r04_pmc_traffic_msm_n2e20.json", "r03_pmc_traffic_msm_n2e20.json", "r02_pmc_traffic_msm_n2e20.json", "r01_pmc_traffic_msm_n2e20.json"):
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                for row in json.load(f)["kernels"]:
                    if row["kernel"] == "k_accum_l0" and logn == 20:
                        return row["hbm_bytes_per_launch_guide_corrected"], "profiles/%s (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes; committed file, not this run)" % name
        except (OSError, KeyError, ValueError):
            pass
    return None, None


# ---- extra: config C5, batch verification of 2^14 64-bit range proofs -----------------------------
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
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        wire2 = bp.prove_wire(vals, gams, seeds)                       # format 2: no transcripts, the device rebuilds them (1.09 KB instead of 2.56 KB per proof)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, bp.last_ms())
    t_batch, batch_ms = best
    bp.close()
    sample = min(distinct, 48)
    t0 = time.perf_counter()
    single = [proof_to_bytes(NIRangeProver(ModP(vals[j], Q), nbits, g, h, gs, hs, ModP(gams[j], Q), u, secp256k1, seeds[j]).prove(), version=2) for j in range(sample)]
    t_prove = time.perf_counter() - t0
    prover_info = {"proofs": distinct, "proves_per_s": distinct / t_batch, "proves_per_s_device_time": distinct / (batch_ms["total"] * 1e-3),
                   "seconds_per_batch": t_batch, "device_ms_by_phase": {k_: round(v_, 3) for k_, v_ in batch_ms.items()}, "tables_build_s_once_per_prover": round(t_tables, 4),
                   "single_proof_prover_proves_per_s": sample / t_prove, "byte_identical_to_single_proof_prover_on_sample": single == wire2[:sample],
                   "replaces": "a loop of NIRangeProver.prove (/root/reference/src/rangeproofs/rangeproof_prover.py:35-91)",
                   "how": "bpmi_rp_prove_batch: every protocol step one launch over the batch, fixed-base tables of the generators, Fiat-Shamir hashes on the device"}
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

        def one_batch(corrupt=False):
            buf = wire_buf
            if corrupt:           # flip one bit inside one proof of this rank's shard: the batch must reject
                bad = bytearray(wire_joined)
                bad[(wire_off[len(blobs_in) // 2] + wire_off[len(blobs_in) // 2 + 1]) // 2] ^= 1
                buf = bytes(bad)
            try:
                part = bv.partial_wire(v_packed, buf, offsets=wire_off_c)      # ONE native call: upload, preparation, decoding, MSM
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
                return bv.partial_wire(v_packed, buf, offsets=wire_off_c)
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
            oks += [finish(f.result()) for f in futs]
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
                "proves_per_s_one_gpu": prover_info["proves_per_s"], "distinct_proofs": distinct, "batch_prover": prover_info, "wire_bytes_per_batch": wire_bytes, "wire_bytes_per_proof": round(wire_bytes / max(hi - lo, 1), 1),
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
                                                      "wire_bytes_per_batch", "wire_bytes_per_proof", "gpu_stage_ms_per_batch", "gpu_stage_ms_per_batch_serial") if k_ in r2}
        res["wire_format_2"]["note"] = ("the same proofs without their three transcripts (csrc/rp_wire_v2_host.hpp): the device rebuilds them (k_rp_expand_v2) and runs "
                                        "the format-1 checks on the expansion; same verdicts (tests/test_gpu_configs.py::test_c5_wire_format_2_same_verdicts_as_format_1)")
        if "errors" in r2:
            res["wire_format_2"]["errors"] = r2["errors"]
    except Exception as e:
        res["wire_format_2"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


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
        return sync, (time.perf_counter() - t0) / reps

    sync_s, pipe_s = measure()
    eng.profile(1)
    eng.profile_reset()
    for _ in range(10):
        eng.msm_dev(d_pts, d_sc, n)
    prof = eng.profile_read()
    eng.profile(False)
    stage_ms = {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items() if v[1]}
    acc_s = stage_ms.get("msm_accumulate", 0.0) / 1e3
    return {"metric": "Pippenger MSM scalar-point pairs/sec at n=2^16 (config C2)", "value": n / pipe_s, "unit": "pairs/s", "n": n,
            "ms_per_msm_two_in_flight": pipe_s * 1e3, "ms_per_msm_one_at_a_time": sync_s * 1e3, "pairs_per_s_one_at_a_time": n / sync_s,
            "result_ok": bool(got == expect), "stage_ms_per_msm": stage_ms,
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
            "runs_s": [round(t, 5) for t in times], "rounds": len(pr.xs), "deterministic": bool(same),
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


def usable_cpus():
    usable = len(os.sched_getaffinity(0))
    q = cpu_quota()
    if q:
        usable = max(1, min(usable, int(q)))
    return usable


def cpu_quota():
    """CPUs this container may use at once (cgroup v2 cpu.max), or None when unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if quota == "max" else round(int(quota) / int(period), 2)
    except Exception:
        return None


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


if __name__ == "__main__":
    main()
