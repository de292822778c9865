"""Random-linear-combination batch verifier (bulletproofs_amd.rangeproofs.batch): it must
accept exactly when every proof is individually accepted.  The oracle for it is the
individual verifier (product and oracle restatement), since the reference has no batch
verifier; every cheating mutation of the reference's tests, applied to one proof of the
batch, must make the batch fail."""
import copy
import random

import pytest

from helpers import Q, gens
from oracle import bp_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


@pytest.fixture(scope="module")
def batch(gp):
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.rangeproofs import NIRangeProver
    from bulletproofs_amd.utils import ModP, mod_hash, commitment
    n = 16
    gs, hs = gp.to_gpu_list(gens(n, b"bgs")), gp.to_gpu_list(gens(n, b"bhs"))
    g, h, u = (gp.to_gpu(R.elliptic_hash(s)) for s in (b"bg", b"bh", b"bu"))
    rnd = random.Random(1)
    Vs, proofs = [], []
    for k in range(6):
        v = ModP(rnd.randrange(2 ** n), Q)
        gamma = mod_hash(b"gamma%d" % k, Q)
        Vs.append(commitment(g, h, v, gamma))
        proofs.append(NIRangeProver(v, n, g, h, gs, hs, gamma, u, secp256k1, b"seed%d" % k).prove())
    return dict(n=n, gs=gs, hs=hs, g=g, h=h, u=u, Vs=Vs, proofs=proofs)


def run(b, Vs, proofs, seed=7):
    from bulletproofs_amd.rangeproofs import batch_verify
    rnd = random.Random(seed)
    return batch_verify(Vs, proofs, b["g"], b["h"], b["gs"], b["hs"], b["u"], rng=lambda: rnd.getrandbits(320))


def test_batch_accepts_valid_proofs(gp, batch):
    from bulletproofs_amd.rangeproofs import RangeVerifier, BatchRangeVerifier
    b = batch
    assert run(b, b["Vs"], b["proofs"]) is True
    assert run(b, b["Vs"][:1], b["proofs"][:1]) is True
    for V, pr in zip(b["Vs"], b["proofs"]):
        assert RangeVerifier(V, b["g"], b["h"], b["gs"], b["hs"], b["u"], pr).verify() is True
    # default CSPRNG weights, incremental use and reset
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    for V, pr in zip(b["Vs"], b["proofs"]):
        bv.add(V, pr)
    assert bv.count == 6 and bv.verify() is True
    bv.reset()
    assert bv.verify() is True          # empty batch: all coefficients zero -> identity


MUTATIONS = ["taux", "mu", "t_hat", "T1", "T2", "A_point_only", "V", "inner_a", "inner_b", "L0", "R_last",
             "u_new", "P_new", "swap_V"]


@pytest.mark.parametrize("what", MUTATIONS)
def test_batch_rejects_one_bad_proof(gp, batch, what):
    from bulletproofs_amd.rangeproofs import RangeVerifier
    from bulletproofs_amd.utils import ModP
    b = batch
    Vs, proofs = list(b["Vs"]), [copy.deepcopy(p) for p in b["proofs"]]
    k = 3
    pr = proofs[k]
    one = ModP(1, Q)
    if what == "taux":
        pr.taux = pr.taux + one
    elif what == "mu":
        pr.mu = pr.mu + one
    elif what == "t_hat":
        pr.t_hat = pr.t_hat + one
    elif what == "T1":
        pr.T1 = pr.T1 + b["g"]
    elif what == "T2":
        pr.T2 = pr.T2 + b["g"]
    elif what == "A_point_only":
        pr.A = pr.A + b["g"]            # transcript no longer matches -> host check fails
    elif what == "V":
        Vs[k] = Vs[k] + b["g"]
    elif what == "inner_a":
        pr.innerProof.proof2.a = pr.innerProof.proof2.a + one
    elif what == "inner_b":
        pr.innerProof.proof2.b = pr.innerProof.proof2.b + one
    elif what == "L0":
        pr.innerProof.proof2.Ls[0] = pr.innerProof.proof2.Ls[0] + b["g"]
    elif what == "R_last":
        pr.innerProof.proof2.Rs[-1] = pr.innerProof.proof2.Rs[-1] + b["g"]
    elif what == "u_new":
        pr.innerProof.u_new = pr.innerProof.u_new + b["g"]
    elif what == "P_new":
        pr.innerProof.P_new = pr.innerProof.P_new + b["g"]
    elif what == "swap_V":
        Vs[k], Vs[k + 1] = Vs[k + 1], Vs[k]
    # the individual verifier rejects the mutated proof ...
    with pytest.raises(Exception, match="Proof invalid"):
        RangeVerifier(Vs[k], b["g"], b["h"], b["gs"], b["hs"], b["u"], proofs[k]).verify()
    # ... and so does the batch, under several independent weight draws
    for seed in (1, 2, 3):
        with pytest.raises(Exception, match="Proof invalid"):
            run(b, Vs, proofs, seed)


def test_batch_partials_add_up(gp, batch):
    """Sharding by proof: the partial values of two disjoint sub-batches fold to the
    identity exactly when the whole batch is valid."""
    from bulletproofs_amd.rangeproofs import BatchRangeVerifier
    b = batch
    eng = gp.engine()
    parts = []
    for lo, hi in ((0, 2), (2, 6)):
        bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
        for V, pr in zip(b["Vs"][lo:hi], b["proofs"][lo:hi]):
            bv.add(V, pr)
        parts.append(bv.partial())
    assert eng.ec_sum_bytes(b"".join(parts), 2) == bytes(64)


def test_batch_aggregated_proofs(gp):
    """Aggregated proofs (m values each) in the same batch machinery, mixed with nothing
    else; one bad commitment must fail the batch."""
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.rangeproofs import AggregNIRangeProver, AggregRangeVerifier, BatchRangeVerifier
    from bulletproofs_amd.utils import ModP, mod_hash, commitment
    n, m = 8, 4
    gs, hs = gp.to_gpu_list(gens(n * m, b"ags")), gp.to_gpu_list(gens(n * m, b"ahs"))
    g, h, u = (gp.to_gpu(R.elliptic_hash(s)) for s in (b"ag", b"ah", b"au"))
    rnd = random.Random(3)
    items = []
    for k in range(3):
        vs = [ModP(rnd.randrange(2 ** n), Q) for _ in range(m)]
        gammas = [mod_hash(b"ag%d_%d" % (k, j), Q) for j in range(m)]
        Vs = [commitment(g, h, vs[j], gammas[j]) for j in range(m)]
        pr = AggregNIRangeProver(vs, n, g, h, gs, hs, gammas, u, secp256k1, b"aseed%d" % k).prove()
        assert AggregRangeVerifier(Vs, g, h, gs, hs, u, pr).verify() is True
        items.append((Vs, pr))
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    for Vs, pr in items:
        bv.add(Vs, pr)
    assert bv.verify() is True
    bv.reset()
    for k, (Vs, pr) in enumerate(items):
        if k == 1:
            Vs = list(Vs)
            Vs[2] = Vs[2] + g
        bv.add(Vs, pr)
    with pytest.raises(Exception, match="Proof invalid"):
        bv.verify()
    # the same through the wire format and the native preparation (lists of m commitments per proof)
    from bulletproofs_amd.rangeproofs.codec import proof_to_bytes
    blobs = [proof_to_bytes(pr) for _, pr in items]
    bv.reset()
    bv.add_wire_native([Vs for Vs, _ in items], blobs, threads=2)
    assert bv.count == 3 and bv.verify() is True
    bv.reset()
    bad_Vs = [list(Vs) for Vs, _ in items]
    bad_Vs[1][2] = bad_Vs[1][2] + g
    bv.add_wire_native(bad_Vs, blobs, threads=2)
    with pytest.raises(Exception, match="Proof invalid"):
        bv.verify()


@pytest.mark.parametrize("workers", [0, 3])
def test_batch_wire_path(gp, batch, workers):
    """Proofs arrive serialised: one GPU launch decompresses every point, the host work runs in
    spawned worker processes (or in-process), the merged combination is one MSM.  A proof that
    was valid on its own but whose wire bytes were altered (scalar, point sign) must fail."""
    from bulletproofs_amd.rangeproofs import BatchRangeVerifier
    from bulletproofs_amd.rangeproofs.codec import proof_to_bytes
    b = batch
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    try:
        if workers:
            bv.start_workers(workers)
        bv.add_wire(b["Vs"], blobs, chunk=2)
        bv.add(b["Vs"][0], b["proofs"][0])
        assert bv.count == 7 and bv.verify() is True
        k = blobs[4][5]
        for off, what in ((6 + 32 * 1 + 31, "mu"), (6 + 32 * (5 + k), "sign of T1")):
            bad = bytearray(blobs[4])
            bad[off] ^= 1
            bv.reset()
            with pytest.raises(Exception, match="Proof invalid"):     # T1's transcript check or the MSM
                bv.add_wire(b["Vs"], blobs[:4] + [bytes(bad)] + blobs[5:])
                bv.verify()
        bad = bytearray(blobs[1])
        bad[6 + 32 * (5 + k) + 1: 6 + 32 * (5 + k) + 33] = (7).to_bytes(32, "big")      # x = 7: 7^3 + 7 = 350 is not a square mod p
        bv.reset()
        with pytest.raises(Exception, match="Proof invalid"):
            bv.add_wire(b["Vs"][1:2], [bytes(bad)])
            bv.verify()
    finally:
        bv.stop_workers()


def test_batch_wire_path_native(gp, batch):
    """add_wire_native: GPU decompression, per-proof host work in libbpmi (transcript checks and
    scalars), one MSM; mixed with add() in the same verifier; altered wire bytes are rejected."""
    from bulletproofs_amd.rangeproofs import BatchRangeVerifier
    from bulletproofs_amd.rangeproofs.codec import proof_to_bytes
    b = batch
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    bv.add_wire_native(b["Vs"], blobs, threads=2)
    bv.add(b["Vs"][0], b["proofs"][0])
    assert bv.count == 7 and bv.verify() is True
    k = blobs[4][5]
    for off in (6 + 32 * 1 + 31, 6 + 32 * (5 + k), 6 + 32 * (5 + k) + 33 * 3 + 20, len(blobs[4]) - 2):   # mu, sign of T1, x of S, last x
        bad = bytearray(blobs[4])
        bad[off] ^= 1
        bv.reset()
        with pytest.raises(Exception, match="Proof invalid"):
            bv.add_wire_native(b["Vs"], blobs[:4] + [bytes(bad)] + blobs[5:], threads=3)
            bv.verify()


def test_state_and_merge_carry_device_resident_chunks(gp, batch):
    """state() of a verifier whose proofs came in through add_wire_native(prepare="device") -- their points and scalars live only
    in device memory -- exports them, and a second verifier that merges the state accepts / rejects like the first; the offset
    table is checked against the REAL size of the proof buffer."""
    from bulletproofs_amd.rangeproofs import BatchRangeVerifier
    from bulletproofs_amd.rangeproofs.codec import proof_to_bytes
    b = batch
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    bv.add_wire_native(b["Vs"], blobs, prepare="device")
    assert bv._dev_chunks, "the device path must have been taken"
    st = bv.state()
    assert st[9] == sum(len(b["Vs"][:1]) + 6 + 2 * blobs[i][5] for i in range(len(blobs)))      # every V, T1, T2, A, S, P', u', L_j, R_j
    other = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    other.merge(st)
    other.add(b["Vs"][1], b["proofs"][1])
    assert other.count == len(blobs) + 1 and other.verify() is True
    # a wrong commitment inside the exported state is still noticed after the merge
    Vs_bad = list(b["Vs"])
    Vs_bad[2] = b["Vs"][3]
    bv2 = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    bv2.add_wire_native(Vs_bad, blobs, prepare="device")
    other = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    other.merge(bv2.state())
    with pytest.raises(Exception, match="Proof invalid"):
        other.verify()
    # offsets that end past the buffer: refused before anything is read
    joined = b"".join(blobs)
    from itertools import accumulate
    offs = [0, *accumulate(map(len, blobs))]
    offs[-1] += 64
    bv3 = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    with pytest.raises(ValueError):
        bv3.add_wire_native(b["Vs"], joined, offsets=offs, prepare="device")
    offs[-1] -= 64
    offs[3], offs[4] = offs[4], offs[3]                    # not monotone: the native walk of the table refuses it
    with pytest.raises(Exception):
        bv3.add_wire_native(b["Vs"], joined, offsets=offs, prepare="device")
