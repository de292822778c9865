"""CPU checks of the batch verifier's host logic (coefficient algebra, transcript checks)
and of its 2-rank sharding, with the ORACLE's MSM injected in place of the HIP engine and
proofs produced by the oracle's restatement of the reference prover."""
import os
import random
import subprocess
import sys
from types import SimpleNamespace

import pytest

import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.ec import Point as GP
from bulletproofs_amd.rangeproofs.batch import BatchRangeVerifier
from bulletproofs_amd.utils.utils import ModP as GModP

from helpers import Q, gens
from oracle import bp_ref as R
from oracle import cbind

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpt(p):
    return GP._raw(p.x, p.y)


def gsc(v):
    return GModP(v.x, Q)


def convert_proof(pr):
    ip, p2 = pr.innerProof, pr.innerProof.proof2
    proof2 = SimpleNamespace(a=gsc(p2.a), b=gsc(p2.b), xs=[gsc(x) for x in p2.xs], Ls=[gpt(p) for p in p2.Ls],
                             Rs=[gpt(p) for p in p2.Rs], transcript=p2.transcript, start_transcript=p2.start_transcript)
    inner = SimpleNamespace(u_new=gpt(ip.u_new), P_new=gpt(ip.P_new), proof2=proof2, transcript=ip.transcript)
    return SimpleNamespace(taux=gsc(pr.taux), mu=gsc(pr.mu), t_hat=gsc(pr.t_hat), T1=gpt(pr.T1), T2=gpt(pr.T2),
                           A=gpt(pr.A), S=gpt(pr.S), innerProof=inner, transcript=pr.transcript)


def make_batch(count, n=8):
    gs, hs = gens(n, b"cgs"), gens(n, b"chs")
    g, h, u = (R.elliptic_hash(s) for s in (b"cg", b"ch", b"cu"))
    rnd = random.Random(5)
    Vs, proofs = [], []
    for k in range(count):
        v = R.Zq(rnd.randrange(2 ** n), Q)
        gamma = R.mod_hash(b"g%d" % k, Q)
        Vs.append(R.commitment(g, h, v, gamma))
        proofs.append(R.range_prove(v, n, g, h, gs, hs, gamma, u, seed=b"s%d" % k, multiexp=cbind.msm))
    return dict(g=gpt(g), h=gpt(h), u=gpt(u), gs=[gpt(p) for p in gs], hs=[gpt(p) for p in hs],
                Vs=[gpt(V) for V in Vs], proofs=[convert_proof(p) for p in proofs])


def oracle_msm(pts, scs, n):
    return cbind.msm_bytes(pts, scs, n, 1)


def test_batch_algebra_with_oracle_msm():
    b = make_batch(4)
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm)
    for V, pr in zip(b["Vs"], b["proofs"]):
        bv.add(V, pr)
    assert bv.verify() is True
    # one wrong scalar anywhere -> rejected
    for field in ("taux", "mu", "t_hat"):
        bv.reset()
        for k, (V, pr) in enumerate(zip(b["Vs"], b["proofs"])):
            if k == 2:
                pr = SimpleNamespace(**vars(pr))
                setattr(pr, field, getattr(pr, field) + GModP(1, Q))
            bv.add(V, pr)
        with pytest.raises(Exception, match="Proof invalid"):
            bv.verify()
    # a transcript that does not match the proof's points fails on the host, before any MSM
    bad = SimpleNamespace(**vars(b["proofs"][0]))
    bad.transcript = bad.transcript.replace(b"&", b"&x", 1)
    bv.reset()
    with pytest.raises(Exception):
        bv.add(b["Vs"][0], bad)


def test_two_rank_gloo_sharded_batch_verify():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", BPMI_DIST_MODE="batch")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29512", os.path.join(REPO, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "DIST_BATCH_OK world=2" in r.stdout


def oracle_decompress(comp, n):
    """Stand-in for bpmi_ec_decompress_batch on CPU: the oracle's bytes_to_point."""
    from oracle.ec import INF, point_to_le64
    out, ok = [], bytearray(n)
    for i in range(n):
        c = comp[33 * i: 33 * i + 33]
        try:
            P = INF if c == bytes(33) else R.bytes_to_point(c)
            out.append(point_to_le64(P))
            ok[i] = 1
        except Exception:
            out.append(bytes(64))
    return b"".join(out), bytes(ok)


@pytest.mark.parametrize("workers", [0, 2])
def test_batch_wire_path_with_worker_processes(workers):
    """add_wire: proofs arrive as bytes, the per-proof host work runs in spawned worker
    processes, their states merge linearly; mixing add(), add_wire() and a corrupted proof."""
    from bulletproofs_amd.rangeproofs.codec import proof_to_bytes
    b = make_batch(7)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm)
    try:
        if workers:
            bv.start_workers(workers)
        bv.add(b["Vs"][0], b["proofs"][0])
        bv.add_wire(b["Vs"][1:6], blobs[1:6], decompress=oracle_decompress, chunk=2)
        bv.add(b["Vs"][6], b["proofs"][6])
        assert bv.count == 7
        assert bv.verify() is True
        # one flipped scalar byte inside a wire proof (t_hat): the batch is rejected
        bad = bytearray(blobs[3])
        bad[6 + 32 * 2 + 31] ^= 1
        bv.reset()
        bv.add_wire(b["Vs"][:6], blobs[:3] + [bytes(bad)] + blobs[4:6], decompress=oracle_decompress, chunk=4)
        with pytest.raises(Exception, match="Proof invalid"):
            bv.verify()
        # a malformed blob / a transcript byte that does not match the proof's points: rejected on
        # the host (in the worker), before any MSM
        bv.reset()
        with pytest.raises(Exception, match="Proof invalid"):
            bv.add_wire(b["Vs"][2:3], [b"XX" + blobs[2]], decompress=oracle_decompress)
        flipped = bytearray(blobs[2])
        flipped[-3] ^= 1                      # inside the Protocol-2 transcript
        with pytest.raises(Exception, match="Proof invalid"):
            bv.add_wire(b["Vs"][2:3], [bytes(flipped)], decompress=oracle_decompress)
    finally:
        bv.stop_workers()
