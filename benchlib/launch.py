"""Rank launch and the control-plane exchanges of bench.py."""
import os
import socket
import subprocess
import sys

BENCH_PY = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")


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
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), BENCH_PY] + argv
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
    """Batches in flight of the C5 extra: every slot is a host thread of its rank (BENCH_C5_INFLIGHT overrides).  Round 6: at most ten
    (eight until then) -- on sixteen hardware queues wire format 3 gains 8 % from the two more, formats 1 and 2 nothing
    (profiles/r06_C5_in_flight_at_16_queues.txt)."""
    return max(1, int(os.environ.get("BENCH_C5_INFLIGHT", "0")) or min(10, max(3, usable // world)))


def strong_depth(pairs_per_rank):
    """MSMs to keep in flight on a rank whose shard has this many pairs: three below 185 000 (the window table's last step; measured
    0.14 against 0.18 ms per MSM at 2^16 with two), two above (three lose 2 % at 2^20: profiles/r05_pipeline_depth_ab.txt)."""
    return 3 if pairs_per_rank < 185000 else 2


def pipelined_exchange_loop(enqueue, finish, combine_begin, combine_wait, steps, depth, exchange=True):
    """`steps` MSMs through `depth` rotating slots: slot j % depth holds MSM j; `depth` MSMs are in flight whenever the host waits, the
    exchange of MSM j's partial is started behind it and collected one iteration later (a rank never waits for the fold before it has fed
    its GPU the next MSM).  enqueue(slot), finish(slot) -> 64 bytes, combine_begin(bytes) -> handle, combine_wait(handle) -> 64 bytes.
    Round 6: the slot that finish() has just freed is refilled AT ONCE, before the exchange's host work (combine_wait + combine_begin are
    ~0.08 ms of Python and RCCL launches per step: queued behind them, the next MSM started that much later and every step was that much
    longer -- 1.02 against 0.95 ms per step with a process group of one rank, profiles/r06_exchange_host_order.txt).
    Returns the LAST step's global result (exchange) or local partial (no exchange: the per-rank floor).  bench.py's MSM_strong line;
    tests/test_distributed_cpu.py drives it over gloo with 2 and 8 ranks."""
    res, pend = None, None
    for j in range(min(steps, depth)):
        enqueue(j % depth)
    for j in range(steps):
        part = finish(j % depth)
        if j + depth < steps:
            enqueue(j % depth)
        if exchange:
            if pend is not None:
                res = combine_wait(pend)
            pend = combine_begin(part)
        else:
            res = part
    if exchange and pend is not None:
        res = combine_wait(pend)
    return res
