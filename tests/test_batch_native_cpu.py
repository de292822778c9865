"""bpmi_rp_batch_prepare (csrc/rp_batch_host.hpp, host code of libbpmi) against its Python twin
BatchRangeVerifier.add: the same weights in, the same numbers out -- every per-proof MSM scalar and
every shared-generator coefficient -- and the same verdict on corrupted proofs.  No GPU involved:
decompression is the oracle's, the MSM is the C oracle's."""
import random

import pytest

import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.rangeproofs.batch import BatchRangeVerifier
from bulletproofs_amd.rangeproofs.codec import proof_to_bytes

from helpers import Q
from test_batch_verify_cpu import make_batch, oracle_decompress, oracle_msm


def both(b, Vs, blobs, proofs, seed):
    r1, r2 = random.Random(seed), random.Random(seed)
    py = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm, rng=lambda: r1.getrandbits(320))
    for V, pr in zip(Vs, proofs):
        py.add(V, pr)
    nat = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm, rng=lambda: r2.getrandbits(320))
    nat.add_wire_native(Vs, blobs, decompress=oracle_decompress, threads=3)
    return py, nat


@pytest.mark.parametrize("n", [2, 8, 64])
def test_native_prepare_equals_python_add(n):
    b = make_batch(7, n=n)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    py, nat = both(b, b["Vs"], blobs, b["proofs"], seed=n)
    assert (py.c_g, py.c_h, py.c_u, py._gs_const, py._hs_const) == (nat.c_g, nat.c_h, nat.c_u, nat._gs_const, nat._hs_const)
    assert [v % Q for v in py.c_gs] == [v % Q for v in nat.c_gs]
    assert [v % Q for v in py.c_hs] == [v % Q for v in nat.c_hs]
    k = n.bit_length() - 1
    per = 7 + 2 * k
    want = [v % Q for v in py._scs]                   # per proof: V T1 T2 A S P_new u_new Ls Rs
    v_sc, p_sc = nat._raw_scs
    for j in range(7):
        got_v = int.from_bytes(v_sc[32 * j: 32 * j + 32], "little")
        got_p = [int.from_bytes(p_sc[32 * (j * (per - 1) + t): 32 * (j * (per - 1) + t) + 32], "little") for t in range(per - 1)]
        w = want[per * j: per * (j + 1)]
        # native order of the points: T1 T2 A S u_new P_new Ls Rs
        assert got_v == w[0]
        assert got_p[:4] == w[1:5] and got_p[4] == w[6] and got_p[5] == w[5] and got_p[6:] == w[7:]
    assert py.verify() is True and nat.verify() is True


def test_native_prepare_rejects_what_python_rejects():
    b = make_batch(4, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    k = 3
    ints_end = 6 + 32 * (5 + k)
    pts_end = ints_end + 33 * (6 + 2 * k)
    rnd = random.Random(1)
    cases = 0
    for trial in range(120):
        j = rnd.randrange(4)
        bad = bytearray(blobs[j])
        region = rnd.choice(("scalar", "transcript", "header", "tail"))
        if region == "scalar":
            pos = rnd.randrange(6, ints_end)
        elif region == "transcript":
            pos = rnd.randrange(pts_end, len(bad))
        elif region == "header":
            pos = rnd.randrange(0, 6)
        else:
            pos = len(bad) - 1 - rnd.randrange(0, 40)
        bad[pos] ^= 1 << rnd.randrange(8)
        mutated = blobs[:j] + [bytes(bad)] + blobs[j + 1:]
        nat = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm)
        try:
            nat.add_wire_native(b["Vs"], mutated, decompress=oracle_decompress, threads=2)
            nat_ok = nat.verify()
        except Exception as e:
            nat_ok = str(e)
        py = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm)
        try:
            py.add_wire(b["Vs"], mutated, decompress=oracle_decompress)
            py_ok = py.verify()
        except Exception:
            py_ok = "rejected"
        assert (nat_ok is True) == (py_ok is True), (trial, region, pos, nat_ok, py_ok)
        assert nat_ok is not True          # every single-bit flip here must be caught
        cases += 1
    assert cases == 120


def test_native_parser_under_sanitizers(tmp_path):
    """The parser of untrusted proof bytes, compiled for the host with ASan + UBSan, must survive
    20 000 corrupted inputs (bit flips, truncation, trailing bytes, length fields, garbage) without a
    report; unmodified proofs are the only ones it accepts."""
    import os
    import struct
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(repo, "tests", "csrc_host", "rp_fuzz.cpp")
    inc = os.path.join(repo, "python-bulletproofs_amd", "csrc")
    exe = str(tmp_path / "rp_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", inc, src, "-o", exe, "-lpthread"])
    b = make_batch(5, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    data = struct.pack("<I", len(blobs)) + b"".join(struct.pack("<I", len(x)) + x for x in blobs)
    path = tmp_path / "proofs.bin"
    path.write_bytes(data)
    r = subprocess.run([exe, str(path), "8", "20000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    accepted = int(r.stdout.split()[1])
    assert 1500 < accepted < 4500, r.stdout          # kind 0 (1 in 8) is the unmodified proof; almost nothing else gets through


@pytest.mark.parametrize("m,bits", [(2, 4), (4, 8), (1, 8)])
def test_native_prepare_aggregated_equals_python_add(m, bits):
    """Aggregated proofs (m values per proof): native preparation == BatchRangeVerifier.add on a list
    of commitments, scalar by scalar; m = 1 as a list takes the same route as a single proof."""
    from oracle import bp_ref as R
    from oracle import cbind
    from helpers import gens
    from test_batch_verify_cpu import convert_proof, gpt
    nm = m * bits
    gs, hs = gens(nm, b"ags"), gens(nm, b"ahs")
    g, h, u = (R.elliptic_hash(s) for s in (b"ag", b"ah", b"au"))
    rnd = random.Random(m * 100 + bits)
    Vs_all, proofs = [], []
    for k in range(3):
        vs = [R.Zq(rnd.randrange(2 ** bits), Q) for _ in range(m)]
        gammas = [R.mod_hash(b"ga%d-%d" % (k, j), Q) for j in range(m)]
        Vs_all.append([gpt(R.commitment(g, h, v, ga)) for v, ga in zip(vs, gammas)])
        proofs.append(convert_proof(R.aggreg_range_prove(vs, bits, g, h, gs, hs, gammas, u, seed=b"as%d" % k, multiexp=cbind.msm)))
    G = dict(g=gpt(g), h=gpt(h), u=gpt(u), gs=[gpt(p) for p in gs], hs=[gpt(p) for p in hs])
    blobs = [proof_to_bytes(pr) for pr in proofs]
    r1, r2 = random.Random(5), random.Random(5)
    py = BatchRangeVerifier(G["g"], G["h"], G["gs"], G["hs"], G["u"], msm=oracle_msm, rng=lambda: r1.getrandbits(320))
    for V, pr in zip(Vs_all, proofs):
        py.add(V, pr)
    nat = BatchRangeVerifier(G["g"], G["h"], G["gs"], G["hs"], G["u"], msm=oracle_msm, rng=lambda: r2.getrandbits(320))
    nat.add_wire_native(Vs_all, blobs, decompress=oracle_decompress, threads=2)
    assert (py.c_g, py.c_h, py.c_u, py._gs_const, py._hs_const) == (nat.c_g, nat.c_h, nat.c_u, nat._gs_const, nat._hs_const)
    assert [v % Q for v in py.c_gs] == [v % Q for v in nat.c_gs]
    assert [v % Q for v in py.c_hs] == [v % Q for v in nat.c_hs]
    k = nm.bit_length() - 1
    per = m + 6 + 2 * k
    want = [v % Q for v in py._scs]
    v_sc, p_sc = nat._raw_scs
    for j in range(3):
        w = want[per * j: per * (j + 1)]
        got_v = [int.from_bytes(v_sc[32 * (j * m + t): 32 * (j * m + t) + 32], "little") for t in range(m)]
        got_p = [int.from_bytes(p_sc[32 * (j * (6 + 2 * k) + t): 32 * (j * (6 + 2 * k) + t) + 32], "little") for t in range(6 + 2 * k)]
        assert got_v == w[:m]
        wp = w[m:]
        assert got_p[:4] == wp[:4] and got_p[4] == wp[5] and got_p[5] == wp[4] and got_p[6:] == wp[6:]
    assert py.verify() is True and nat.verify() is True


def test_native_prepare_checks_the_offset_table():
    """bpmi_rp_batch_prepare never reads outside blobs[0, blobs_len): an offset table that is not monotonic or runs
    past the buffer is BPMI_E_ARG (-3), not a host out-of-bounds read."""
    import ctypes
    from bulletproofs_amd import _native
    lib = _native.load()
    b = make_batch(2, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    joined = b"".join(blobs)
    k = 3
    npts = 2 * (6 + 2 * k)
    w = bytes([1] + [0] * 31) * 8

    def call(offsets, blobs_len):
        offs = (ctypes.c_uint64 * 3)(*offsets)
        v_sc, p_sc = ctypes.create_string_buffer(64), ctypes.create_string_buffer(32 * npts)
        shared, comp = ctypes.create_string_buffer(32 * (5 + 16)), ctypes.create_string_buffer(33 * npts)
        bad = ctypes.c_int64(-1)
        return lib.bpmi_rp_batch_prepare(8, 1, 2, joined, blobs_len, ctypes.cast(offs, ctypes.c_void_p), w, None, 1, v_sc, p_sc, shared, comp,
                                         ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p)), bad.value

    good = [0, len(blobs[0]), len(joined)]
    assert call(good, len(joined)) == (0, -1)
    assert call(good, len(joined) - 1)[0] == -3                       # last blob would end past the buffer
    assert call([0, len(joined) + 5, len(joined)], len(joined))[0] == -3
    assert call([len(blobs[0]), 0, len(joined)], len(joined))[0] == -3    # not monotonic
    rc, first_bad = call([0, len(blobs[0]) - 7, len(joined)], len(joined))   # in bounds but cut in the wrong place: a bad proof, not a crash
    assert rc == 0 and first_bad == 0


def test_native_seed_derived_weights_equal_explicit_weights():
    """weights = NULL + seed: the weights are SHA-256(seed || LE64(proof index) || t) cut to 248 bits; passing exactly those
    weights explicitly gives the same scalars, byte for byte."""
    import ctypes
    import hashlib
    from bulletproofs_amd import _native
    lib = _native.load()
    b = make_batch(5, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    joined = b"".join(blobs)
    count, k = 5, 3
    npts = count * (6 + 2 * k)
    offs = [0]
    for x in blobs:
        offs.append(offs[-1] + len(x))
    seed = bytes(range(32))
    derived = b""
    for g in range(count):
        for t in range(4):
            d = bytearray(hashlib.sha256(seed + g.to_bytes(8, "little") + bytes([t])).digest())
            d[31] = 0
            derived += bytes(d)

    def call(weights, sd):
        o = (ctypes.c_uint64 * (count + 1))(*offs)
        v_sc, p_sc = ctypes.create_string_buffer(32 * count), ctypes.create_string_buffer(32 * npts)
        shared, comp = ctypes.create_string_buffer(32 * (5 + 16)), ctypes.create_string_buffer(33 * npts)
        bad = ctypes.c_int64(-1)
        rc = lib.bpmi_rp_batch_prepare(8, 1, count, joined, len(joined), ctypes.cast(o, ctypes.c_void_p), weights, sd, 2, v_sc, p_sc, shared, comp,
                                       ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p))
        return rc, bad.value, v_sc.raw, p_sc.raw, shared.raw, comp.raw

    a1 = call(None, seed)
    a2 = call(derived, None)
    assert a1[0] == 0 and a1[1] == -1 and a1 == a2
    assert call(None, bytes(32))[2] != a1[2]                      # another seed, other weights
    assert call(None, None)[0] == -3                              # neither weights nor seed


def test_add_wire_native_accepts_one_buffer_with_offsets():
    b = make_batch(4, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    offs = [0]
    for x in blobs:
        offs.append(offs[-1] + len(x))
    r = random.Random(9)
    one = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm, rng=lambda: r.getrandbits(320))
    one.add_wire_native(b["Vs"], b"".join(blobs), decompress=oracle_decompress, threads=2, offsets=offs)
    r = random.Random(9)
    lst = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm, rng=lambda: r.getrandbits(320))
    lst.add_wire_native(b["Vs"], blobs, decompress=oracle_decompress, threads=2)
    assert one.state() == lst.state() and one.verify() is True
    # default weights (native, seed-derived): still a valid batch; a corrupted proof still fails
    nat = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm)
    nat.add_wire_native(b["Vs"], blobs, decompress=oracle_decompress, threads=2)
    assert nat.verify() is True
    bad = bytearray(blobs[2])
    bad[40] ^= 1
    nat = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], msm=oracle_msm)
    with pytest.raises(Exception, match="Proof invalid"):
        nat.add_wire_native(b["Vs"], blobs[:2] + [bytes(bad)] + blobs[3:], decompress=oracle_decompress, threads=2)
        nat.verify()
