"""The RCCL ("nccl") code path of bulletproofs_amd.distributed on the GPU box's single GPU: a one-rank
process group still runs all_gather_into_tensor -> bpmi_ec_sum_dev, the device-side branch of
ShardedMSM.combine that the CPU (gloo) tests cannot reach; and bench.py's own launcher (`--gpus 2`
outside torch.distributed.run) with two gloo ranks time-sharing the GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world1_allgather_and_device_fold():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(REPO, "tools", "nccl_selftest.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "NCCL_SELFTEST_OK world=1" in r.stdout


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` (no torchrun, no WORLD_SIZE): the parent starts the ranks as a child process and
    relays ONE JSON line.  gloo backend so that two ranks may share this box's single GPU."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--logn", "16",
           "--no-extra", "--soak-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["result_ok"] is True
    assert out["value"] > 0 and out["scaling"] == "weak"
    # the extras with two ranks: the batch verifier's shards are verified on worker threads while the main thread exchanges the
    # partials (its own engine), the fixed 2^14 batch and the 2^14-per-GPU variant both accept and both reject a corrupted shard
    r = subprocess.run([a for a in cmd if a != "--no-extra"] + ["--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    for name, batch in (("C5_batch_verify", 1 << 14), ("C5_batch_verify_per_gpu_batches", 1 << 15)):
        c5 = out["extra"][name]
        assert c5.get("accepted") is True and c5["corrupted_batch_rejected"] is True and c5["batch"] == batch, c5
    assert out["extra"]["C3_ipa_prover"].get("rounds") == 20 and out["extra"]["C4_aggregated_range_proof"].get("verified") is True
    # strong scaling: one 2^16 MSM split over the two ranks, same known-answer check
    r = subprocess.run(cmd + ["--scaling", "strong"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["result_ok"] is True and out["config"]["pairs_per_gpu"] == 1 << 15


def test_bench_collectives_on_rccl_with_one_rank():
    """bench.py under torch.distributed.run with ONE rank and BENCH_FORCE_DIST=1: every collective of the N > 1 path (the
    all_gather of partials on the exchange stream + device fold, the MAX all_reduce of the elapsed time, the gather of the
    known-answer terms, the barriers, the sharded batch verifier's combine) runs on the real RCCL backend."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--logn", "18",
           "--soak-seconds", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["dist_backend"] == "nccl" and out["n_ranks_seen"] == 1 and out["result_ok"] is True
    assert out["extra"]["C5_batch_verify"].get("accepted") is True and out["extra"]["C5_batch_verify"]["corrupted_batch_rejected"] is True
    assert out["extra"]["C3_ipa_prover"].get("rounds") == 20


def test_bench_two_gpus_on_rccl_when_the_box_has_them():
    """On a box with >= 2 GPUs (the driver's scaling node; the builder's boxes have one): `python bench.py --gpus 2` on the
    real RCCL backend, weak and strong, with the extras -- the same command line SCALE_rNN.json is produced by, so a defect of the
    N > 1 path shows up here first.  torch.cuda.device_count() does not initialise the GPU in this process."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("BENCH_DIST_BACKEND", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--soak-seconds", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["dist_backend"] == "nccl" and out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["result_ok"] is True
    assert len(out["ms_per_step_by_rank"]) == 2
    assert out["extra"]["C5_batch_verify"].get("accepted") is True and out["extra"]["C5_batch_verify"]["corrupted_batch_rejected"] is True
    assert out["extra"]["C3_ipa_prover"].get("rounds") == 20
    r = subprocess.run(cmd + ["--no-extra", "--scaling", "strong"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "strong" and out["result_ok"] is True
