"""Worker for tests/test_distributed_cpu.py: world_size-2 gloo run of the sharding logic
with the ORACLE standing in for the HIP engine (tests may use the oracle; the product's
default is the HIP engine)."""
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import torch.distributed as dist  # noqa: E402

import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.distributed import ShardedMSM, shard_bounds, shard_verdicts  # noqa: E402
from oracle import cbind  # noqa: E402
from oracle.ec import secp256k1, point_from_le64, point_to_le64, INF  # noqa: E402


def oracle_fold(buf, k):
    acc = INF
    for p in cbind.unpack_points(buf, k):
        acc = acc + p
    return point_to_le64(acc)


def batch_mode():
    """Each rank batch-verifies its own half of the proofs; the partial values fold to the
    identity iff the whole batch is valid."""
    from test_batch_verify_cpu import make_batch, oracle_msm
    from bulletproofs_amd.rangeproofs.batch import BatchRangeVerifier
    from bulletproofs_amd.utils.utils import ModP
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    b = make_batch(5)
    sm = ShardedMSM(msm=oracle_msm, fold=oracle_fold)
    # the two lines bench.py reports for N > 1 (extra_c5): "strong" = ONE batch split over the ranks by proof (5 proofs: ranks
    # without a proof contribute the identity), "weak" = a batch per rank (3 proofs each, the distinct proofs repeated: total 3 N)
    for line, total in (("strong", 5), ("weak", 3 * world)):
        lo, hi = shard_bounds(total, world, rank)
        for corrupt in (False, True):
            bad_at = total - 1                        # (one proof of the last rank's shard)
            bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm)
            for k in range(lo, hi):
                pr = b["proofs"][k % 5]
                mu = pr.mu
                if corrupt and k == bad_at:
                    pr.mu = pr.mu + ModP(1, secp256k1.q)
                bv.add(b["Vs"][k % 5], pr)
                pr.mu = mu
            try:
                ok = bv.verify(sharded=sm)
            except Exception as e:
                ok = str(e)
            assert ok == ("Proof invalid" if corrupt else True), (rank, line, corrupt, ok)
    # a partial result is exactly one 64-byte point: anything else is refused before a collective is entered (on every rank alike)
    for wrong in (b"", bytes(63), bytes(128)):
        try:
            sm.combine(wrong)
            raise AssertionError("a %d-byte partial was accepted" % len(wrong))
        except ValueError:
            pass
    dist.barrier()
    if rank == 0:
        print("DIST_BATCH_OK world=%d" % world)
    dist.destroy_process_group()


def ipa_mode(use_gpu):
    """Sharded FastNIProver2 against the oracle's single-process restatement: cyclic shards,
    one 128-byte all_gather per round, bit-identical Proof2 on every rank."""
    from oracle import bp_ref
    from bulletproofs_amd.distributed import ShardedFastNIProver2, cyclic_shard
    from bulletproofs_amd.ec import Point as NPoint
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    Q = secp256k1.q
    if use_gpu:
        from bulletproofs_amd.engine import default_engine
        eng = default_engine()
        sizes = [(2, None), (64, None), (256, 7), (4096, None)]
    else:
        from oracle_engine import OracleEngine
        eng = OracleEngine()
        sizes = [(1, None), (2, None), (4, 5), (8, None), (32, None)]      # incl. vectors shorter than 4 / 8 ranks and one element per rank
    rnd = random.Random(4242)
    for n, hs_seed in sizes:
        ks = [rnd.randrange(1, Q) for _ in range(2 * n + 1)]
        pts = cbind.ec_mul_batch([secp256k1.G] * (2 * n + 1), ks)
        g, h, u = pts[:n], pts[n:2 * n], pts[2 * n]
        a = [rnd.randrange(Q) for _ in range(n)]
        b = [rnd.randrange(Q) for _ in range(n)]
        hsc = None if hs_seed is None else [pow(hs_seed, i, Q) for i in range(n)]
        start = b"c2VlZA==&12345&"
        conv = lambda ps: [NPoint.from_le64(point_to_le64(p)) for p in ps]
        grp = type("G", (), {"q": Q})()
        pr = ShardedFastNIProver2(cyclic_shard(conv(g), world, rank), cyclic_shard(conv(h), world, rank), conv([u])[0], None,
                                  cyclic_shard(a, world, rank), cyclic_shard(b, world, rank), grp, start,
                                  None if hsc is None else cyclic_shard(hsc, world, rank), engine=eng).prove()
        if n <= 256:
            hh = h if hsc is None else cbind.ec_mul_batch(h, hsc)
            want = bp_ref.ipa2_prove(g, hh, u, None, [bp_ref.ModP(x, Q) for x in a], [bp_ref.ModP(x, Q) for x in b], Q, start,
                                     multiexp=cbind.msm)
            want_L = [(p.x, p.y) for p in want.Ls]
            want_R = [(p.x, p.y) for p in want.Rs]
            want_t = (want.a.x, want.b.x, [x.x for x in want.xs], want.transcript, want.start_transcript)
        else:
            from bulletproofs_amd.innerproduct import FastNIProver2
            one = FastNIProver2(conv(g), conv(h), conv([u])[0], None, a, b, grp, start, hsc).prove()
            want_L = [(p.x, p.y) for p in one.Ls]
            want_R = [(p.x, p.y) for p in one.Rs]
            want_t = (one.a.x, one.b.x, [x.x for x in one.xs], one.transcript, one.start_transcript)
        assert [(p.x, p.y) for p in pr.Ls] == want_L, (rank, n)
        assert [(p.x, p.y) for p in pr.Rs] == want_R, (rank, n)
        assert (pr.a.x, pr.b.x, [x.x for x in pr.xs], pr.transcript, pr.start_transcript) == want_t, (rank, n)
    dist.barrier()
    if rank == 0:
        print("DIST_IPA_OK world=%d" % world)
    dist.destroy_process_group()


def strong_mode():
    """bench.py's MSM_strong line on gloo: ONE MSM split over the ranks, `depth` MSMs in flight through rotating slots
    (benchlib.launch.pipelined_exchange_loop), the exchange of step j collected one iteration later; with and without the exchange (the
    per-rank floor `expected_ms_per_step_if_ideal`).  A stand-in engine records the slot protocol: a slot is never queued twice without
    being finished, never finished empty, and the rotation is j % depth."""
    from benchlib.launch import pipelined_exchange_loop, strong_depth
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    Q = secp256k1.q
    rnd = random.Random(99)
    assert strong_depth(1 << 17) == 3 and strong_depth(184999) == 3 and strong_depth(185000) == 2 and strong_depth(1 << 20) == 2
    n = 40
    pts = cbind.ec_mul_batch([secp256k1.G] * n, [rnd.randrange(1, Q) for _ in range(n)])
    pb = cbind.pack_points(pts)
    for depth in (2, 3):
        for steps in (1, 2, 3, 7):
            # step j multiplies by scalars of its own: the result of the loop must be the LAST step's global MSM
            scal = [[rnd.randrange(Q) for _ in range(n)] for _ in range(steps)]
            lo, hi = shard_bounds(n, world, rank)
            slots, log, queued = {}, [], [0]

            def enqueue(slot):
                assert slot not in slots, ("slot queued twice", slot)
                j = queued[0]
                queued[0] += 1
                assert slot == j % depth
                slots[slot] = cbind.msm_bytes(pb[64 * lo: 64 * hi], cbind.pack_scalars(scal[j][lo:hi]), hi - lo, 1)
                log.append(("q", slot))

            def finish(slot):
                assert slot in slots, ("finish of an empty slot", slot)
                log.append(("f", slot))
                return slots.pop(slot)

            sm = ShardedMSM(msm=lambda p, s, k: cbind.msm_bytes(p, s, k, 1), fold=oracle_fold)
            got = pipelined_exchange_loop(enqueue, finish, sm.combine_begin, sm.combine_wait, steps, depth, True)
            assert got == cbind.msm_bytes(pb, cbind.pack_scalars(scal[-1]), n, 1), (rank, depth, steps)
            assert not slots and queued[0] == steps and max(sum(1 for k, _ in log[:i] if k == "q") - sum(1 for k, _ in log[:i] if k == "f")
                                                            for i in range(len(log) + 1)) == min(depth, steps)
            assert sm.exchange_us() is None                       # gloo: no device-side exchange to time
            queued[0] = 0
            local = pipelined_exchange_loop(enqueue, finish, sm.combine_begin, sm.combine_wait, steps, depth, False)
            assert local == cbind.msm_bytes(pb[64 * lo: 64 * hi], cbind.pack_scalars(scal[-1][lo:hi]), hi - lo, 1)
    dist.barrier()
    if rank == 0:
        print("DIST_STRONG_OK world=%d" % world)
    dist.destroy_process_group()


def main():
    mode = os.environ.get("BPMI_DIST_MODE")
    if mode == "batch":
        return batch_mode()
    if mode == "strong":
        return strong_mode()
    if mode in ("ipa", "ipa_gpu"):
        return ipa_mode(mode == "ipa_gpu")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    Q = secp256k1.q
    rnd = random.Random(77)           # same inputs on every rank
    for n in (1, 2, 5, 64, 301):
        pts = cbind.ec_mul_batch([secp256k1.G] * n, [rnd.randrange(1, Q) for _ in range(n)])
        es = [rnd.randrange(Q) for _ in range(n)]
        pb, sb = cbind.pack_points(pts), cbind.pack_scalars(es)
        sm = ShardedMSM(msm=lambda p, s, k: cbind.msm_bytes(p, s, k, 1), fold=oracle_fold)
        got = sm.multiexp_replicated(pb, sb, n)
        want = cbind.msm_bytes(pb, sb, n, 1)
        assert got == want, (rank, n)
        # weak-scaling form: each rank owns a different shard
        lo, hi = shard_bounds(n, world, rank)
        part = cbind.msm_bytes(pb[64 * lo: 64 * hi], sb[32 * lo: 32 * hi], hi - lo, 1)
        assert sm.combine(part) == want, (rank, n)
    verdicts = shard_verdicts(lambda k: k % 3 != 0, list(range(11)))
    assert verdicts == [k % 3 != 0 for k in range(11)], verdicts
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d" % world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
