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


def _bench_line(cmd, env, timeout):
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_eight_ranks_time_sharing_this_gpu_with_every_extra():
    """What the driver's 8-GPU node will run, rehearsed on ONE GPU: `bench.py --gpus 8` with eight gloo ranks time-sharing the device,
    EVERY extra (at test sizes: --extra-scale small), weak and strong.  The line must say eight ranks were seen, carry eight per-rank
    step times, the host budget, the strong-scaling MSM beside the weak headline, and both C5 variants."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--logn", "14", "--preheat-ms", "0",
           "--soak-seconds", "0", "--no-cpu-baseline", "--extra-scale", "small"]
    out = _bench_line(cmd, env, 1500)
    assert out["n_gpus"] == 8 and out["n_ranks_seen"] == 8 and out["result_ok"] is True and out["scaling"] == "weak"
    assert len(out["ms_per_step_by_rank"]) == 8 and all(v > 0 for v in out["ms_per_step_by_rank"])
    hb = out["host_budget"]
    assert hb["ranks_on_this_host"] == 8 and hb["c5_batches_in_flight"] >= 1 and hb["cpus_per_rank"] >= 1
    ex = out["extra"]
    assert set(ex) >= {"MSM_strong", "C2_msm_2e16", "C5_batch_verify", "C5_batch_verify_per_gpu_batches", "C3_ipa_prover", "C4_aggregated_range_proof"}
    for name, v in ex.items():
        assert "error" not in v and "errors_by_rank" not in v, (name, v)
    st = ex["MSM_strong"]
    assert st["scaling"] == "strong" and st["result_ok"] is True and st["pairs_per_gpu"] == (1 << 14) // 8 and len(st["ms_per_step_by_rank"]) == 8
    assert st["msms_in_flight"] == 3 and st["expected_ms_per_step_if_ideal"] > 0 and st["exchange_us"] is None      # (gloo: no device-side exchange)
    assert ex["C5_batch_verify"]["accepted"] is True and ex["C5_batch_verify"]["corrupted_batch_rejected"] is True and ex["C5_batch_verify"]["batch"] == 256
    assert ex["C5_batch_verify"]["scaling"].startswith("strong") and ex["C5_batch_verify_per_gpu_batches"]["scaling"].startswith("weak")
    assert ex["C5_batch_verify_per_gpu_batches"]["batch"] == 8 * 256 and ex["C5_batch_verify_per_gpu_batches"]["accepted"] is True
    assert ex["C3_ipa_prover"]["rounds"] == 12 and ex["C3_ipa_prover"]["deterministic"] is True
    assert ex["C4_aggregated_range_proof"]["verified"] is True
    # the record's tail: `summary` is the LAST key of the line, compact, and shows every config; BASELINE's second metric at the top level
    sm = out["summary"]
    assert list(out)[-1] == "summary" and len(json.dumps(sm)) < 600, sm
    assert sm["result_ok_all"] is True and sm["checks"] >= 8
    for k in ("ms_per_step", "C2_ms_one", "C2_ms_two", "C2_ms_three", "C3_s", "C4_prove_s", "C4_verify_s", "C5_verifies_per_s", "C5_one_batch_ms", "prover_proofs_per_s"):
        assert isinstance(sm[k], (int, float)) and sm[k] > 0, (k, sm)
    assert isinstance(sm["C5_link_GBps"], float) and sm["C5_link_GBps"] >= 0 and ex["C5_batch_verify"]["link"]["bytes_per_batch"] > 0      # (a rank's share of 256 proofs: megabytes per second)
    assert sm["C5_link_peak_GBps"] == 63.0 and out["value2"] == ex["C5_batch_verify"]["value"] and out["unit2"] == "verifies/s"
    assert out["alu_roofline"]["geometry"]["windows"] >= 16
    # strong scaling as the headline of its own run
    out = _bench_line(cmd + ["--scaling", "strong", "--no-extra"], env, 900)
    assert out["scaling"] == "strong" and out["n_ranks_seen"] == 8 and out["result_ok"] is True and out["config"]["pairs_per_gpu"] == (1 << 14) // 8


def test_an_extra_failing_on_one_rank_costs_neither_the_line_nor_the_other_extras():
    """Rank 5 of eight raises in the C3 extra (BENCH_INJECT_FAILURE): the headline is intact, every rank left C3 before its
    collectives, the text of the exception is in rank 0's line, and the extras behind it ran."""
    env = dict(os.environ, BENCH_DIST_BACKEND="gloo", BENCH_INJECT_FAILURE="C3_ipa_prover:5")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1", "--logn", "14", "--preheat-ms", "0",
           "--soak-seconds", "0", "--no-cpu-baseline", "--extra-scale", "small"]
    out = _bench_line(cmd, env, 1500)
    assert out["n_ranks_seen"] == 8 and out["result_ok"] is True and out["value"] > 0
    c3 = out["extra"]["C3_ipa_prover"]
    assert c3["errors_by_rank"] == {"5": "RuntimeError: injected failure in C3_ipa_prover on rank 5"} and c3["error"].startswith("skipped: rank 5")
    assert out["extra"]["C4_aggregated_range_proof"].get("verified") is True and "error" not in out["extra"]["C5_batch_verify"]


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
    assert out["exchange_us"] > 0                           # the RCCL exchange's own device time: the reference for the first N = 8 line
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
