"""N > 1 path on CPU: two gloo processes run the sharded-MSM / sharded-verdict host logic
(bulletproofs_amd.distributed) with the oracle standing in for the GPU engine."""
import os
import subprocess
import sys

import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.distributed import shard_bounds

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 8, 9, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_sharded_msm():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29511", os.path.join(REPO, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DIST_OK world=2" in r.stdout


def test_two_rank_gloo_sharded_ipa_prover():
    """ShardedFastNIProver2 on cyclic shards == the oracle's single-process FastNIProver2."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", BPMI_DIST_MODE="ipa")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29513", os.path.join(REPO, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DIST_IPA_OK world=2" in r.stdout


def _run_ranks(world, mode, port, marker):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    if mode:
        env["BPMI_DIST_MODE"] = mode
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "%s world=%d" % (marker, world) in r.stdout


def test_four_and_eight_rank_gloo_sharded_ipa_prover():
    """The target machine has 8 GPUs: the cyclic layout with 4 and 8 ranks (log2(world) hand-over rounds), vectors SHORTER
    than the number of ranks (every rank runs the whole tiny argument) and exactly one element per rank."""
    _run_ranks(4, "ipa", 29521, "DIST_IPA_OK")
    _run_ranks(8, "ipa", 29523, "DIST_IPA_OK")


def test_eight_rank_gloo_sharded_msm_and_batch_verifier():
    """8 ranks, fewer items than ranks in places (n = 1, 2, 5 pairs; 5 proofs): empty shards contribute the identity."""
    _run_ranks(8, "", 29525, "DIST_OK")
    _run_ranks(8, "batch", 29527, "DIST_BATCH_OK")


def test_two_rank_gloo_batch_verifier_strong_and_weak_lines():
    """Both C5 lines of `bench.py --gpus N` (one batch split over the ranks; one batch per rank) at world 2 -- world 8 runs in
    test_eight_rank_gloo_sharded_msm_and_batch_verifier -- and the 64-byte guard of ShardedMSM.combine."""
    _run_ranks(2, "batch", 29531, "DIST_BATCH_OK")


def test_strong_line_three_slot_rotation_on_two_and_eight_ranks():
    """The MSM_strong line of `bench.py --gpus N`: two and THREE MSMs in flight through rotating slots with the exchange one step behind
    (benchlib.launch.pipelined_exchange_loop), with and without the exchange, at world 2 and 8 (round 6: shards below 185 000 pairs
    keep three in flight)."""
    _run_ranks(2, "strong", 29533, "DIST_STRONG_OK")
    _run_ranks(8, "strong", 29535, "DIST_STRONG_OK")


def test_sharded_prover_rejects_a_non_cyclic_layout():
    """Unequal shard lengths that are not 'shorter than the ranks' must be a clear ValueError on every rank, not a torch
    buffer error from a collective."""
    code = (
        "import os, sys; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
        "import torch.distributed as dist, bulletproofs_amd\n"
        "from bulletproofs_amd.distributed import ShardedFastNIProver2\n"
        "from bulletproofs_amd.ec import Point, secp256k1\n"
        "from oracle_engine import OracleEngine\n"
        "dist.init_process_group('gloo'); r = dist.get_rank()\n"
        "G = Point(secp256k1.G.x, secp256k1.G.y, secp256k1)\n"
        "n = 2 if r == 0 else 0\n"
        "grp = type('G', (), {'q': secp256k1.q})()\n"
        "try:\n"
        "    ShardedFastNIProver2([G] * n, [G] * n, G, None, [1] * n, [1] * n, grp, b'x&', engine=OracleEngine()).prove()\n"
        "    print('NO_ERROR')\n"
        "except ValueError as e:\n"
        "    print('VALUE_ERROR_OK', r)\n"
        "dist.destroy_process_group()\n" % (REPO, REPO))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29529", "--no-python", sys.executable, "-c", code]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("VALUE_ERROR_OK") == 2 and "NO_ERROR" not in r.stdout
