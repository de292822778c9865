"""GPU parity of the batched range-proof prover (bpmi_rp_prove_batch, rangeproofs/batch_prover.py): every proof of a batch is
byte-identical to the one the single-proof prover (NIRangeProver.prove, the reference's call surface:
/root/reference/src/rangeproofs/rangeproof_prover.py:35-91) makes for the same inputs, and to the reference's own golden proofs
(tests/golden/rangeproofs.json); the proofs verify one by one and in the batch verifier."""
import random

import pytest

from conftest import load_golden
from helpers import P, Q, gens, hx
from oracle import bp_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


def _setup(gp, n, tag=b"bp"):
    gs, hs = gp.to_gpu_list(gens(n, tag + b"g")), gp.to_gpu_list(gens(n, tag + b"h"))
    g, h, u = (gp.to_gpu(R.elliptic_hash(tag + s)) for s in (b"G", b"H", b"U"))
    return g, h, gs, hs, u


def _seeds(count, rnd):
    out = []
    for i in range(count):
        kind = i % 5
        if kind == 0:
            out.append(b"")
        elif kind == 1:
            out.append(bytes([rnd.randrange(256)]))
        elif kind == 2:
            out.append(rnd.randbytes(rnd.randrange(2, 40)))
        elif kind == 3:
            out.append(b"seed-%d" % i)
        else:
            out.append(rnd.randbytes(rnd.randrange(40, 200)))
    return out


@pytest.mark.parametrize("n,count", [(64, 1024), (32, 200), (8, 130), (2, 70), (16, 1), (128, 40)])
def test_batch_prover_equals_the_single_proof_prover(gp, n, count):
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.rangeproofs import BatchRangeProver, NIRangeProver, proof_to_bytes
    from bulletproofs_amd.utils import ModP
    g, h, gs, hs, u = _setup(gp, n)
    rnd = random.Random(1000 + n)
    vs = [ModP(rnd.randrange(1 << n), Q) for _ in range(count)]
    for j, edge in enumerate((0, 1, (1 << n) - 1, 1 << (n - 1))):
        if j < count:
            vs[j] = ModP(edge, Q)
    if count > 8:
        vs[5] = ModP((1 << n) + 5, Q)                    # out of range: the prover still proves the low n bits (and the proof fails)
    gammas = [ModP(rnd.randrange(Q), Q) for _ in range(count)]
    seeds = _seeds(count, rnd)
    bp = BatchRangeProver(n, g, h, gs, hs, u)
    try:
        blobs = bp.prove_wire(vs, gammas, seeds)
        again = bp.prove_wire(vs[: max(1, count // 3)], gammas[: max(1, count // 3)], seeds[: max(1, count // 3)])     # a second, smaller batch on the same prover
    finally:
        bp.close()
    assert again == blobs[: len(again)]
    # the same inputs as packed bytes (what a service receives) give the same proofs
    if n == 32:
        from itertools import accumulate
        bp2 = BatchRangeProver(n, g, h, gs, hs, u)
        try:
            packed, off = bp2.prove_wire_packed(b"".join(v.x.to_bytes(32, "little") for v in vs), b"".join(x.x.to_bytes(32, "little") for x in gammas),
                                                (b"".join(seeds), [0, *accumulate(map(len, seeds))]))
        finally:
            bp2.close()
        assert [packed[off[i]: off[i + 1]] for i in range(count)] == blobs
        # copy=False: a view of the prover's page-locked buffer and the ctypes offsets (valid until the prover's next call)
        import ctypes
        bp3 = BatchRangeProver(n, g, h, gs, hs, u)
        try:
            offs_c = (ctypes.c_uint64 * (count + 1))(*[0, *accumulate(map(len, seeds))])
            view, voff = bp3.prove_wire_packed(b"".join(v.x.to_bytes(32, "little") for v in vs), b"".join(x.x.to_bytes(32, "little") for x in gammas),
                                               (b"".join(seeds), offs_c), copy=False)
            assert isinstance(view, memoryview) and [bytes(view[voff[i]: voff[i + 1]]) for i in range(count)] == blobs
            del view
        finally:
            bp3.close()
    step = 1 if count <= 1024 else 4
    for i in range(0, count, step):
        want = proof_to_bytes(NIRangeProver(vs[i], n, g, h, gs, hs, gammas[i], u, secp256k1, seeds[i]).prove(), version=2)
        assert blobs[i] == want, (n, i)


@pytest.mark.parametrize("count", [4, 5, 333])
def test_batch_prover_two_halves_on_two_lanes(gp, count):
    """Round 6 experiment (option prover_split, off by default: measured neutral): a batch as two halves on the ctx's two lanes -- here
    forced from 4 proofs, with an odd count: the same bytes as one launch sequence."""
    from bulletproofs_amd.rangeproofs import BatchRangeProver
    from bulletproofs_amd.utils import ModP
    n = 32
    g, h, gs, hs, u = _setup(gp, n, b"sp")
    rnd = random.Random(count)
    vs = [ModP(rnd.randrange(1 << n), Q) for _ in range(count)]
    gammas = [ModP(rnd.randrange(Q), Q) for _ in range(count)]
    seeds = _seeds(count, rnd)
    eng = gp.engine()
    bp = BatchRangeProver(n, g, h, gs, hs, u)
    try:
        eng.set_option("prover_split", 0)
        one = bp.prove_wire(vs, gammas, seeds)
        eng.set_option("prover_split", 2)
        two = bp.prove_wire(vs, gammas, seeds)
        again = bp.prove_wire(vs, gammas, seeds)
    finally:
        eng.set_option("prover_split", 0)
        bp.close()
    assert two == one and again == one


@pytest.mark.parametrize("k", range(7))
def test_batch_prover_reproduces_the_reference_goldens(gp, k):
    """The single-value goldens of tests/golden/rangeproofs.json (made by the reference itself): the batch prover, given the same
    value, blinding factor, seed and generators, writes the proof whose fields are the golden's."""
    from bulletproofs_amd.rangeproofs import BatchRangeProver, proofs_from_bytes
    from bulletproofs_amd.utils import ModP, mod_hash
    from test_gpu_rangeproofs import check_range_proof, inputs
    c = load_golden("rangeproofs.json")["single"][k]
    s, n, gs, hs, g, h, u = inputs(gp, c, 1)
    v = ModP(int(c["v"], 16), Q)
    gamma = mod_hash(s[5], Q)
    bp = BatchRangeProver(n, g, h, gs, hs, u)
    try:
        pr = bp.prove([v, v], [gamma, gamma], [s[6], s[6]])
    finally:
        bp.close()
    check_range_proof(gp, pr[0], c["proof"])
    check_range_proof(gp, pr[1], c["proof"])


@pytest.mark.parametrize("n,m,count", [(16, 4, 150), (64, 2, 60), (8, 8, 70), (2, 2, 33), (32, 4, 25), (1, 2, 9), (4, 1, 17)])
def test_batch_prover_aggregated_equals_single(gp, n, m, count):
    """Round 6: batches of AGGREGATED proofs (m values of n bits per proof, n m <= 128): every proof byte-identical to the one
    AggregNIRangeProver.prove (/root/reference/src/rangeproofs/rangeproof_aggreg_prover.py:36-146 behind the product's call surface) makes
    for the same values, blinding factors, seed and generators -- including the reference's rho = mod_hash(str(2 n) ...) with n the bits
    PER VALUE, values out of range, and edge values per slot -- and accepted by AggregRangeVerifier."""
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.rangeproofs import AggregNIRangeProver, AggregRangeVerifier, BatchRangeProver, proof_to_bytes, proofs_from_bytes
    from bulletproofs_amd.utils import ModP, commitment
    g, h, gs, hs, u = _setup(gp, n * m, b"ag")
    rnd = random.Random(7000 + 16 * n + m)
    vss = [[ModP(rnd.randrange(1 << n), Q) for _ in range(m)] for _ in range(count)]
    vss[0] = [ModP(0, Q)] * m
    vss[1] = [ModP((1 << n) - 1, Q)] * m
    vss[2] = [ModP((1 << n) - 1 if j % 2 else 0, Q) for j in range(m)]
    if count > 8:
        vss[5][m - 1] = ModP((1 << n) + 1, Q)                # out of range in the last slot: proved on its low n bits, must fail to verify
    gss = [[ModP(rnd.randrange(Q), Q) for _ in range(m)] for _ in range(count)]
    gss[3] = [ModP(0, Q)] * m
    seeds = _seeds(count, rnd)
    bp = BatchRangeProver(n, g, h, gs, hs, u, m=m)
    try:
        blobs = bp.prove_wire(vss, gss, seeds)
        packed, off = bp.prove_wire_packed(b"".join(v.x.to_bytes(32, "little") for row in vss for v in row),
                                           b"".join(x.x.to_bytes(32, "little") for row in gss for x in row), seeds)
    finally:
        bp.close()
    assert [packed[off[i]: off[i + 1]] for i in range(count)] == blobs
    step = 1 if count <= 40 else 3
    for i in list(range(0, 6)) + list(range(6, count, step)):
        if i >= count:
            continue
        if m == 1:
            from bulletproofs_amd.rangeproofs import NIRangeProver
            want = NIRangeProver(vss[i][0], n, g, h, gs, hs, gss[i][0], u, secp256k1, seeds[i]).prove()
        else:
            want = AggregNIRangeProver(vss[i], n, g, h, gs, hs, gss[i], u, secp256k1, seeds[i]).prove()
        assert blobs[i] == proof_to_bytes(want, version=2), (n, m, i)
    if m > 1:
        proofs = proofs_from_bytes(blobs[:8])
        for i in (0, 1, 2, 3, 4, 6):
            Vs = [commitment(g, h, v, x) for v, x in zip(vss[i], gss[i])]
            assert AggregRangeVerifier(Vs, g, h, gs, hs, u, proofs[i]).verify() is True
        if count > 8:
            Vs = [commitment(g, h, v, x) for v, x in zip(vss[5], gss[5])]
            with pytest.raises(Exception, match="Proof invalid"):
                AggregRangeVerifier(Vs, g, h, gs, hs, u, proofs[5]).verify()


@pytest.mark.parametrize("k", [0, 1])
def test_batch_prover_reproduces_the_reference_aggregated_goldens(gp, k):
    """The aggregated goldens of tests/golden/rangeproofs.json that fit the batched prover (16 bits x 4 values, 64 x 2; the third, 16 x 32,
    is 512 elements: the single-proof prover's): same values, blinding factors, seed and generators -> the golden's fields, twice in one batch."""
    from bulletproofs_amd.rangeproofs import BatchRangeProver
    from bulletproofs_amd.utils import ModP, mod_hash
    from test_gpu_rangeproofs import check_range_proof, inputs
    c = load_golden("rangeproofs.json")["aggregated"][k]
    m = c["m"]
    s, n, gs, hs, g, h, u = inputs(gp, c, m)
    vs = [ModP(int(v, 16), Q) for v in c["vs"]]
    gammas = [mod_hash(str(j).encode() + s[5], Q) for j in range(m)]
    bp = BatchRangeProver(n, g, h, gs, hs, u, m=m)
    try:
        pr = bp.prove([vs, vs], [gammas, gammas], [s[6], s[6]])
    finally:
        bp.close()
    check_range_proof(gp, pr[0], c["proof"])
    check_range_proof(gp, pr[1], c["proof"])


@pytest.mark.parametrize("bits", [4, 5, 7, 9, 10, 11, 12, 13, 14, 15, 16])
def test_batch_prover_table_windows(gp, bits):
    """The fixed-base tables with windows of 4 .. 16 bits (ctx option prover_table_bits, read when the prover is created), built level by
    level from the window bases (round 6: k_pv_table_level): windows that straddle words, a top window of 1 .. 16 bits, digits of
    magnitude 2^(bits-1) -- the same proofs."""
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.rangeproofs import BatchRangeProver, NIRangeProver, proof_to_bytes
    from bulletproofs_amd.utils import ModP
    n, count = 16, 40
    g, h, gs, hs, u = _setup(gp, n, b"tw")
    rnd = random.Random(bits)
    vs = [ModP(rnd.randrange(1 << n), Q) for _ in range(count)]
    gammas = [ModP(rnd.randrange(Q), Q) for _ in range(count)]
    gammas[0], gammas[1], gammas[2] = ModP(0, Q), ModP(Q - 1, Q), ModP((Q - 1) // 2, Q)
    seeds = [b"tw%d" % i for i in range(count)]
    eng = gp.engine()
    try:
        eng.set_option("prover_table_bits", bits)
        bp = BatchRangeProver(n, g, h, gs, hs, u)
    finally:
        eng.set_option("prover_table_bits", 0)
    try:
        blobs = bp.prove_wire(vs, gammas, seeds)
    finally:
        bp.close()
    for i in range(0, count, 3):
        assert blobs[i] == proof_to_bytes(NIRangeProver(vs[i], n, g, h, gs, hs, gammas[i], u, secp256k1, seeds[i]).prove(), version=2), (bits, i)


def test_batch_prover_output_verifies(gp):
    """The wire bytes go straight into the verifiers: one by one (RangeVerifier) and as one batch (BatchRangeVerifier); a wrong
    commitment is rejected."""
    from bulletproofs_amd.rangeproofs import BatchRangeProver, BatchRangeVerifier, RangeVerifier, proofs_from_bytes
    from bulletproofs_amd.utils import ModP, commitment
    n, count = 64, 96
    g, h, gs, hs, u = _setup(gp, n, b"vf")
    rnd = random.Random(5)
    vs = [ModP(rnd.randrange(1 << n), Q) for _ in range(count)]
    gammas = [ModP(rnd.randrange(Q), Q) for _ in range(count)]
    seeds = [b"verify-%d" % i for i in range(count)]
    bp = BatchRangeProver(n, g, h, gs, hs, u)
    try:
        blobs = bp.prove_wire(vs, gammas, seeds)
        ms = bp.last_ms()
    finally:
        bp.close()
    assert ms["total"] > 0
    Vs = [commitment(g, h, v, x) for v, x in zip(vs, gammas)]
    proofs = proofs_from_bytes(blobs)
    for i in (0, 1, count - 1):
        assert RangeVerifier(Vs[i], g, h, gs, hs, u, proofs[i]).verify() is True
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    for V, pr in zip(Vs, proofs):
        bv.add(V, pr)
    assert bv.verify() is True
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    for i, (V, pr) in enumerate(zip(Vs, proofs)):
        bv.add(Vs[0] if i == 7 else V, pr)
    with pytest.raises(Exception, match="Proof invalid"):
        bv.verify()


def test_batch_prover_argument_errors(gp):
    from bulletproofs_amd.engine import EngineError
    from bulletproofs_amd.rangeproofs import BatchRangeProver
    from bulletproofs_amd.utils import ModP
    g, h, gs, hs, u = _setup(gp, 8)
    with pytest.raises(ValueError):
        BatchRangeProver(8, g, h, gs[:7], hs, u)
    with pytest.raises(EngineError, match="powers of two"):
        BatchRangeProver(3, g, h, gs[:3], hs[:3], u)
    with pytest.raises(EngineError, match="powers of two"):
        BatchRangeProver(2, g, h, gs[:6], hs[:6], u, m=3)
    bp = BatchRangeProver(4, g, h, gs, hs, u, m=2)
    try:
        with pytest.raises(ValueError, match="takes 2 values"):
            bp.prove_wire([[ModP(1, Q)]], [[ModP(1, Q)]], [b""])
    finally:
        bp.close()
    bp = BatchRangeProver(8, g, h, gs, hs, u)
    try:
        assert bp.prove_wire([], [], []) == []
        with pytest.raises(ValueError):
            bp.prove_wire([ModP(1, Q)], [], [b""])
    finally:
        bp.close()


def test_c_program_proves_and_verifies_a_batch_through_the_abi_only(gp, tmp_path):
    """examples/prove_batch_c_abi.c: a C99 program over include/bpmi.h and libbpmi.so alone builds a prover, proves a batch in one
    call and verifies the wire bytes as one batch (bpmi_rp_batch_verify_dev); with one commitment changed the batch is rejected."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(repo, "python-bulletproofs_amd")
    exe = str(tmp_path / "prove_batch_c_abi")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(repo, "include"),
                           os.path.join(repo, "examples", "prove_batch_c_abi.c"), "-o", exe, os.path.join(libdir, "libbpmi.so"),
                           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"])
    for args, code, word in ((["300", "64"], 0, "batch verification: VALID"), (["70", "8"], 0, "batch verification: VALID"),
                             (["300", "64", "1"], 1, "batch verification: INVALID"),
                             (["200", "16", "0", "4"], 0, "batch verification: VALID"),           # aggregated: 4 values of 16 bits per proof
                             (["90", "32", "1", "2"], 1, "batch verification: INVALID")):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert r.returncode == code and word in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("n,m,count", [(32, 1, 40), (64, 1, 9), (8, 4, 7), (2, 1, 3)])
def test_batch_prover_writes_wire_format_3(gp, n, m, count):
    """wire_format=3 (round 6): the same proofs followed by the y coordinates of their points -- byte for byte what the codec writes from
    the single-proof prover's Proof object (src/rangeproofs/rangeproof_prover.py:35-91 / rangeproof_aggreg_prover.py:36-146 behind it), the
    format-2 part unchanged; the batch verifier accepts them (ys checked, no square roots), and the next batch of the same prover can
    be format 2 again."""
    from bulletproofs_amd.rangeproofs import AggregNIRangeProver, BatchRangeProver, BatchRangeVerifier, NIRangeProver
    from bulletproofs_amd.rangeproofs.codec import proof_to_bytes, wire_v3_to_v2
    from bulletproofs_amd.ec import secp256k1
    from bulletproofs_amd.utils import ModP, commitment
    g, h, gs, hs, u = _setup(gp, n * m, b"w3")
    rnd = random.Random(31 * n + m)
    vs = [[ModP(rnd.randrange(1 << n), Q) for _ in range(m)] for _ in range(count)]
    gammas = [[ModP(rnd.randrange(Q), Q) for _ in range(m)] for _ in range(count)]
    seeds = _seeds(count, rnd)
    if m == 1:
        vs, gammas = [r[0] for r in vs], [r[0] for r in gammas]
    bp = BatchRangeProver(n, g, h, gs, hs, u, m=m, wire_format=3)
    try:
        v3 = bp.prove_wire(vs, gammas, seeds)
        bp.wire_format = 2
        v2 = bp.prove_wire(vs, gammas, seeds)
    finally:
        bp.close()
    k = (n * m).bit_length() - 1
    for i in range(count):
        assert v3[i][:5] == b"BPRP3" and len(v3[i]) == len(v2[i]) + 32 * (6 + 2 * k) and wire_v3_to_v2(v3[i]) == v2[i]
        if i % 3 == 0:
            if m == 1:
                pr = NIRangeProver(vs[i], n, g, h, gs, hs, gammas[i], u, secp256k1, seeds[i]).prove()
            else:
                pr = AggregNIRangeProver(vs[i], n, g, h, gs, hs, gammas[i], u, secp256k1, seeds[i]).prove()
            assert v3[i] == proof_to_bytes(pr, version=3), (n, m, i)
    Vs = [[commitment(g, h, v, ga) for v, ga in zip(vr, gr)] if m > 1 else commitment(g, h, vr, gr) for vr, gr in zip(vs, gammas)]
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    assert bv.verify_wire(Vs, v3) is True
    bad = bytearray(v3[count // 2])
    bad[-5] ^= 0x20                                   # a bit of the last point's y
    with pytest.raises(Exception, match="^Proof invalid$"):
        bv.verify_wire(Vs, v3[:count // 2] + [bytes(bad)] + v3[count // 2 + 1:])
