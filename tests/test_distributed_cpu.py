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
