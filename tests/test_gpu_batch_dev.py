"""bpmi_rp_batch_prepare_dev (csrc/rp_batch_kernels.hpp: parsing, SHA-256 transcript checks and the weighted scalars of the
batch verifier, one GPU lane per proof) against its host twin bpmi_rp_batch_prepare (csrc/rp_batch_host.hpp, itself pinned to
the Python / reference verifiers by tests/test_batch_native_cpu.py): same weights or seed in -> the same scalars, shared
coefficients, decoded points and verdicts out, byte for byte."""
import ctypes
import hashlib
import random

import pytest

import bulletproofs_amd  # noqa: F401
from bulletproofs_amd import _native
from bulletproofs_amd.rangeproofs.batch import BatchRangeVerifier
from bulletproofs_amd.rangeproofs.codec import proof_to_bytes

from helpers import Q
from test_batch_verify_cpu import make_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from bulletproofs_amd.engine import default_engine
    return default_engine()


@pytest.fixture(scope="module")
def gp():
    import gpu_common
    return gpu_common


def offsets_of(blobs):
    offs = [0]
    for x in blobs:
        offs.append(offs[-1] + len(x))
    return offs


def host_prepare(n, m, blobs, weights, seed, offs=None):
    lib = _native.load()
    count = len(blobs) if offs is None else len(offs) - 1
    joined = b"".join(blobs) if offs is None else blobs
    offs = offsets_of(blobs) if offs is None else offs
    k = n.bit_length() - 1
    npts = count * (6 + 2 * k)
    o = (ctypes.c_uint64 * (count + 1))(*offs)
    v_sc, p_sc = ctypes.create_string_buffer(32 * count * m), ctypes.create_string_buffer(32 * npts)
    shared, comp = ctypes.create_string_buffer(32 * (5 + 2 * n)), ctypes.create_string_buffer(33 * npts)
    bad = ctypes.c_int64(-1)
    rc = lib.bpmi_rp_batch_prepare(n, m, count, joined, len(joined), ctypes.cast(o, ctypes.c_void_p), weights, seed, 2, v_sc, p_sc, shared, comp,
                                   ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p))
    return rc, bad.value, v_sc.raw, p_sc.raw, shared.raw, comp.raw


def dev_prepare(eng, n, m, blobs, weights, seed, offs=None):
    count = len(blobs) if offs is None else len(offs) - 1
    joined = b"".join(blobs) if offs is None else blobs
    offs = offsets_of(blobs) if offs is None else offs
    k = n.bit_length() - 1
    npts = count * (6 + 2 * k)
    o = (ctypes.c_uint64 * (count + 1))(*offs)
    d_v, d_p, d_pts = eng.alloc(32 * count * m), eng.alloc(32 * npts), eng.alloc(64 * npts)
    shared = ctypes.create_string_buffer(32 * (5 + 2 * n))
    bad = ctypes.c_int64(-1)
    try:
        rc = eng.lib.bpmi_rp_batch_prepare_dev(eng.ctx, n, m, count, joined, len(joined), ctypes.cast(o, ctypes.c_void_p), weights, seed, d_v.ptr, d_p.ptr,
                                               d_pts.ptr, shared, ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p))
        return rc, bad.value, d_v.download(), d_p.download(), shared.raw, d_pts.download()
    finally:
        for d in (d_v, d_p, d_pts):
            d.free()


def assert_same(eng, n, m, blobs, weights, seed):
    h = host_prepare(n, m, blobs, weights, seed)
    d = dev_prepare(eng, n, m, blobs, weights, seed)
    assert h[0] == 0 and d[0] == 0 and h[1] == -1 and d[1] == -1
    assert d[2] == h[2], "V scalars"
    assert d[3] == h[3], "per-proof point scalars"
    assert d[4] == h[4], "shared coefficients"
    npts = len(h[5]) // 33
    pts, ok = eng.ec_decompress_batch_bytes(h[5], npts)
    assert 0 not in ok and d[5] == pts, "decoded points"


@pytest.mark.parametrize("n", [2, 8, 64])
def test_device_prepare_equals_host_prepare(eng, n):
    b = make_batch(7, n=n)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    rnd = random.Random(n)
    weights = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(4 * 7))
    assert_same(eng, n, 1, blobs, weights, None)
    assert_same(eng, n, 1, blobs, None, bytes(range(32)))
    # weights >= q are reduced on both sides
    big = b"".join((Q + rnd.randrange(1, 2 ** 100)).to_bytes(32, "little") for _ in range(4 * 7))
    assert_same(eng, n, 1, blobs, big, None)


def test_seed_weights_match_the_published_derivation(eng):
    b = make_batch(5, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    seed = hashlib.sha256(b"seed").digest()
    derived = b""
    for g in range(5):
        for t in range(4):
            dg = bytearray(hashlib.sha256(seed + g.to_bytes(8, "little") + bytes([t])).digest())
            dg[31] = 0
            derived += bytes(dg)
    a = dev_prepare(eng, 8, 1, blobs, None, seed)
    c = dev_prepare(eng, 8, 1, blobs, derived, None)
    assert a[0] == 0 and a[1] == -1 and a == c


@pytest.mark.parametrize("m,bits", [(2, 4), (4, 8), (1, 8)])
def test_device_prepare_aggregated(eng, m, bits):
    from oracle import bp_ref as R
    from oracle import cbind
    from helpers import gens
    from test_batch_verify_cpu import convert_proof
    nm = m * bits
    gs, hs = gens(nm, b"ags"), gens(nm, b"ahs")
    g, h, u = (R.elliptic_hash(s) for s in (b"ag", b"ah", b"au"))
    rnd = random.Random(m * 100 + bits)
    proofs = []
    for k in range(3):
        vs = [R.Zq(rnd.randrange(2 ** bits), Q) for _ in range(m)]
        gammas = [R.mod_hash(b"ga%d-%d" % (k, j), Q) for j in range(m)]
        proofs.append(convert_proof(R.aggreg_range_prove(vs, bits, g, h, gs, hs, gammas, u, seed=b"as%d" % k, multiexp=cbind.msm)))
    blobs = [proof_to_bytes(pr) for pr in proofs]
    weights = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(12))
    assert_same(eng, nm, m, blobs, weights, None)
    assert_same(eng, nm, m, blobs, None, b"\x07" * 32)


def test_many_proofs_any_lane_count_and_launch_size(eng):
    """300 proofs (7 distinct ones repeated; the seed weights differ per index): the result does not depend on how many proofs a
    wave takes or on how many go into one launch."""
    b = make_batch(7, n=8)
    base = [proof_to_bytes(pr) for pr in b["proofs"]]
    blobs = [base[i % 7] for i in range(300)]
    seed = b"\x21" * 32
    want = host_prepare(8, 1, blobs, None, seed)
    try:
        for lanes, rows in ((0, 0), (1, 0), (64, 0), (16, 37), (0, 128)):
            eng.set_option("rp_lanes", lanes)
            eng.set_option("rp_rows", rows)
            got = dev_prepare(eng, 8, 1, blobs, None, seed)
            assert got[:5] == want[:5], (lanes, rows)
    finally:
        eng.set_option("rp_lanes", 0)
        eng.set_option("rp_rows", 0)


def test_corrupted_proofs_same_verdict_as_host(eng):
    """Single-bit flips anywhere in a proof, truncation, trailing bytes, garbage: the device reports the same first failing
    proof as the host twin (point encodings are judged by the decompression on both sides)."""
    b = make_batch(4, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    k = 3
    ints_end = 6 + 32 * (5 + k)
    pts_end = ints_end + 33 * (6 + 2 * k)
    rnd = random.Random(11)
    w = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(16))
    rejected = 0
    for trial in range(160):
        j = rnd.randrange(4)
        bad = bytearray(blobs[j])
        kind = rnd.choice(("scalar", "point", "transcript", "header", "tail", "truncate", "extend", "garbage", "two"))
        if kind == "scalar":
            bad[rnd.randrange(6, ints_end)] ^= 1 << rnd.randrange(8)
        elif kind == "point":
            bad[rnd.randrange(ints_end, pts_end)] ^= 1 << rnd.randrange(8)
        elif kind == "transcript":
            bad[rnd.randrange(pts_end, len(bad))] ^= 1 << rnd.randrange(8)
        elif kind == "header":
            bad[rnd.randrange(0, 6)] ^= 1 << rnd.randrange(8)
        elif kind == "tail":
            bad[len(bad) - 1 - rnd.randrange(0, 40)] ^= 1 << rnd.randrange(8)
        elif kind == "truncate":
            del bad[rnd.randrange(0, len(bad)):]
        elif kind == "extend":
            bad += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9)))
        elif kind == "garbage":
            bad = bytearray(rnd.randrange(256) for _ in range(rnd.randrange(0, 700)))
        mutated = blobs[:j] + [bytes(bad)] + blobs[j + 1:]
        if kind == "two":                                  # two bad proofs: the smaller index is reported
            j2 = rnd.randrange(4)
            b2 = bytearray(mutated[j2])
            b2[rnd.randrange(6, ints_end)] ^= 0x10
            mutated[j2] = bytes(b2)
            b3 = bytearray(mutated[3])
            b3[-1] ^= 1
            mutated[3] = bytes(b3)
        h = host_prepare(8, 1, mutated, w, None)
        d = dev_prepare(eng, 8, 1, mutated, w, None)
        assert h[0] == 0 and d[0] == 0
        host_bad = h[1]
        upto = 4 if host_bad < 0 else host_bad             # the host twin leaves point encodings to the decompression: judge those of the proofs it got through
        if upto:
            _, ok = eng.ec_decompress_batch_bytes(h[5][:33 * upto * (6 + 2 * k)], upto * (6 + 2 * k))
            if 0 in ok:
                host_bad = ok.index(0) // (6 + 2 * k)
        assert d[1] == host_bad, (trial, kind, d[1], host_bad)
        rejected += d[1] >= 0
    assert rejected >= 140          # a flip inside a point's x can give another valid point: those fail in the MSM instead


def test_offset_table_is_checked(eng):
    b = make_batch(2, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    joined = b"".join(blobs)
    w = bytes([1] + [0] * 31) * 8
    good = [0, len(blobs[0]), len(joined)]
    assert dev_prepare(eng, 8, 1, joined, w, None, offs=good)[:2] == (0, -1)
    assert dev_prepare(eng, 8, 1, joined[:-1], w, None, offs=good)[0] == -3
    assert dev_prepare(eng, 8, 1, joined, w, None, offs=[0, len(joined) + 5, len(joined)])[0] == -3
    assert dev_prepare(eng, 8, 1, joined, w, None, offs=[len(blobs[0]), 0, len(joined)])[0] == -3
    rc, first_bad = dev_prepare(eng, 8, 1, joined, w, None, offs=[0, len(blobs[0]) - 7, len(joined)])[:2]
    assert rc == 0 and first_bad == 0
    assert dev_prepare(eng, 8, 1, joined, None, None, offs=good)[0] == -3      # neither weights nor seed
    assert dev_prepare(eng, 12, 1, joined, w, None, offs=good)[0] == -3        # n_gens not a power of two


def test_batch_verifier_end_to_end_on_the_device(eng):
    b = make_batch(9, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    for prepare in ("device", "host", "auto"):
        bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
        bv.add_wire_native(b["Vs"], blobs, prepare=prepare)
        bv.add_wire_native(b["Vs"][:4], blobs[:4], prepare=prepare)          # a second chunk in the same batch
        assert bv.verify() is True
        bv.reset()
    # receive-buffer form: one bytearray + offsets
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    bv.add_wire_native(b["Vs"], bytearray(b"".join(blobs)), offsets=offsets_of(blobs), prepare="device")
    assert bv.verify() is True
    # page-locked receive buffer + commitments already packed as 64-byte points
    joined = b"".join(blobs)
    hb = eng.host_alloc(len(joined) + 100)
    hb.view[:len(joined)] = joined
    for prepare in ("device", "host"):
        bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
        bv.add_wire_native(b"".join(V.to_le64() for V in b["Vs"]), hb, offsets=offsets_of(blobs), prepare=prepare)
        assert bv.verify() is True
        bv.reset()
    hb.free()
    # a wrong commitment passes the preparation and fails the MSM; a wrong transcript byte fails the preparation
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    bv.add_wire_native([b["Vs"][1]] + b["Vs"][1:], blobs, prepare="device")
    with pytest.raises(Exception, match="Proof invalid"):
        bv.verify()
    bad = bytearray(blobs[5])
    bad[-3] ^= 2
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    with pytest.raises(Exception, match="Proof invalid"):
        bv.add_wire_native(b["Vs"], blobs[:5] + [bytes(bad)] + blobs[6:], prepare="device")


def test_fuzz_single_proof_verdicts_equal_host(eng):
    """1500 mutated proofs, one per call, so that every verdict is compared (not only the first failing index of a batch):
    device preparation == host preparation + decompression flags.  Mutations: bit flips, byte overwrites, truncation,
    extension, length-field edits, splices of another proof's transcript."""
    b = make_batch(4, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    k = 3
    ints_end = 6 + 32 * (5 + k)
    pts_end = ints_end + 33 * (6 + 2 * k)
    rnd = random.Random(77)
    w = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(4))
    accepted = 0
    for trial in range(1500):
        src = blobs[rnd.randrange(4)]
        bad = bytearray(src)
        kind = rnd.randrange(8)
        if kind == 0:
            pass                                            # unmodified
        elif kind == 1:
            for _ in range(rnd.randrange(1, 4)):
                bad[rnd.randrange(len(bad))] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            bad[rnd.randrange(len(bad))] = rnd.randrange(256)
        elif kind == 3:
            del bad[rnd.randrange(len(bad)):]
        elif kind == 4:
            bad += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 12)))
        elif kind == 5:                                     # one of the three length fields, or the start index
            pos = pts_end + rnd.choice((0, 1))
            if rnd.random() < 0.7:
                t0 = pts_end + 2
                l0 = int.from_bytes(src[t0:t0 + 4], "big")
                t1 = t0 + 4 + l0
                l1 = int.from_bytes(src[t1:t1 + 4], "big")
                pos = rnd.choice((t0, t1, t1 + 4 + l1)) + rnd.randrange(4)
            bad[pos] = (bad[pos] + rnd.choice((1, 255, 128))) & 0xFF
        elif kind == 6:                                     # another proof's tail (its transcripts) behind this proof's head
            other = blobs[rnd.randrange(4)]
            cut = rnd.randrange(pts_end, len(bad))
            bad = bytearray(bytes(bad[:cut]) + other[cut:])
        else:                                               # an '&' or a digit changed inside a transcript
            pos = rnd.randrange(pts_end + 14, len(bad))
            bad[pos] = rnd.choice(b"&0123456789=")
        blob = bytes(bad)
        h = host_prepare(8, 1, [blob], w, None)
        d = dev_prepare(eng, 8, 1, [blob], w, None)
        assert h[0] == 0 and d[0] == 0
        host_bad = h[1]
        if host_bad < 0:
            _, ok = eng.ec_decompress_batch_bytes(h[5], len(h[5]) // 33)
            if 0 in ok:
                host_bad = 0
        assert d[1] == host_bad, (trial, kind, d[1], host_bad)
        if host_bad < 0:
            accepted += 1
            assert d[2] == h[2] and d[3] == h[3] and d[4] == h[4]
    assert 150 < accepted < 700          # the unmodified eighth, plus mutations that happen to leave the proof as it was


def test_one_role_profiling_run_never_reads_as_a_verification(eng):
    """The profiling option that runs only one role of the kernel makes the call report proof 0 as bad, whatever it saw."""
    b = make_batch(3, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    seed = b"\x05" * 32
    assert dev_prepare(eng, 8, 1, blobs, None, seed)[:2] == (0, -1)
    try:
        for role in (0, 1, 2, 3):
            eng.set_option("rp_only_role", role)
            assert dev_prepare(eng, 8, 1, blobs, None, seed)[:2] == (0, 0)
    finally:
        eng.set_option("rp_only_role", -1)
    assert dev_prepare(eng, 8, 1, blobs, None, seed)[:2] == (0, -1)


def test_long_proofs_up_to_the_wire_limit(eng):
    """Item 0 of the range-proof transcript is free text for the verifiers, so a valid proof can be long.  Up to the wire
    format's limit (32 KiB per proof) host and device agree byte for byte (the transposed array then has 4096 rows);
    beyond it both call the proof invalid."""
    b = make_batch(3, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    k = 3
    t0 = 6 + 32 * (5 + k) + 33 * (6 + 2 * k) + 2

    def padded(blob, extra):
        l0 = int.from_bytes(blob[t0:t0 + 4], "big")
        return blob[:t0] + (l0 + extra).to_bytes(4, "big") + b"x" * extra + blob[t0 + 4:]

    w = b"".join((7 + i).to_bytes(32, "little") for i in range(12))
    near = [blobs[0], padded(blobs[1], 32768 - len(blobs[1])), padded(blobs[2], 9001)]
    assert len(near[1]) == 32768
    assert_same(eng, 8, 1, near, w, None)
    over = [blobs[0], padded(blobs[1], 32769 - len(blobs[1])), blobs[2]]
    assert host_prepare(8, 1, over, w, None)[1] == 1
    assert dev_prepare(eng, 8, 1, over, w, None)[1] == 1


@pytest.mark.parametrize("m,bits", [(128, 64), (256, 64)])
def test_large_aggregated_proofs_device_equals_host(eng, m, bits):
    """Aggregated proofs over n m = 8192 and 16 384 generators (k = 13, 14: the s-vector walk's LDS tables cross 64 KB per block
    at k = 14): proofs made by the product's own prover, prepared by the device and by the host twin -- identical numbers --
    and the batch verifies."""
    import hashlib
    from bulletproofs_amd.ec import Point, secp256k1
    from bulletproofs_amd.rangeproofs import AggregNIRangeProver
    from bulletproofs_amd.utils import ModP, commitment, elliptic_hash, mod_hash
    nm = m * bits
    G64 = secp256k1.G.to_le64()

    def gens(seed):
        ks = b"".join(hashlib.sha256(b"g%d-%d" % (seed, i)).digest()[:31] + b"\x00" for i in range(nm))
        raw = eng.ec_mul_batch_bytes(G64 * nm, ks, nm)
        return [Point.from_le64(raw[64 * i: 64 * i + 64]) for i in range(nm)]

    gs, hs = gens(1), gens(2)
    g, h, u = elliptic_hash(b"g"), elliptic_hash(b"h"), elliptic_hash(b"u")
    blobs, Vs_all = [], []
    for t in range(2):
        vs = [ModP(int.from_bytes(hashlib.sha256(b"v%d-%d" % (t, j)).digest()[:8], "big"), Q) for j in range(m)]
        gammas = [mod_hash(b"gamma%d-%d" % (t, j), Q) for j in range(m)]
        Vs_all.append([commitment(g, h, vs[j], gammas[j]) for j in range(m)])
        blobs.append(proof_to_bytes(AggregNIRangeProver(vs, bits, g, h, gs, hs, gammas, u, secp256k1, b"seed%d" % t).prove()))
    seed = bytes(range(32))
    hst = host_prepare(nm, m, blobs, None, seed)
    dev = dev_prepare(eng, nm, m, blobs, None, seed)
    assert hst[:2] == (0, -1) and dev[:2] == (0, -1)
    assert dev[2] == hst[2] and dev[3] == hst[3] and dev[4] == hst[4]
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    bv.add_wire_native(Vs_all, blobs, prepare="device")
    assert bv.verify() is True
    bv.release()
    bad = BatchRangeVerifier(g, h, gs, hs, u)
    bad.add_wire_native([Vs_all[1], Vs_all[1]], blobs, prepare="device")          # proof 0 against proof 1's commitments
    with pytest.raises(Exception, match="Proof invalid"):
        bad.verify()
    bad.release()


def test_c_program_verifies_a_batch_through_the_abi_only(eng, tmp_path):
    """examples/batch_verify_c_abi.c: a C99 program over include/bpmi.h and libbpmi.so alone verifies a batch of wire proofs
    (page-locked receive buffer -> bpmi_rp_batch_prepare_dev -> shared coefficients -> bpmi_msm_segs_dev, and the same as ONE call,
    bpmi_rp_batch_verify_dev, with the same verdict): a valid batch, one
    with a flipped transcript bit (rejected before the MSM, with the proof's index) and one with exchanged commitments (the
    MSM is not the identity)."""
    import os
    import struct
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(repo, "python-bulletproofs_amd")
    exe = str(tmp_path / "batch_verify_c_abi")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-Wall", "-Wextra", "-I", os.path.join(repo, "include"),
                           os.path.join(repo, "examples", "batch_verify_c_abi.c"), "-o", exe, os.path.join(libdir, "libbpmi.so"),
                           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"])
    b = make_batch(6, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]

    def write(path, blobs, Vs):
        offs = offsets_of(blobs)
        with open(path, "wb") as f:
            f.write(struct.pack("<IIIQ", 8, 1, len(blobs), offs[-1]))
            for pt in [b["g"], b["h"], b["u"]] + b["gs"] + b["hs"] + Vs:
                f.write(pt.to_le64())
            f.write(struct.pack("<%dQ" % len(offs), *offs))
            f.write(b"".join(blobs))

    good = str(tmp_path / "good.bin")
    write(good, blobs, b["Vs"])
    r = subprocess.run([exe, good, "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("VALID in") == 6 and r.stdout.count("one call: VALID") == 3 and "INVALID" not in r.stdout, r.stdout + r.stderr
    flipped = bytearray(blobs[4])
    flipped[-9] ^= 4
    bad = str(tmp_path / "bad.bin")
    write(bad, blobs[:4] + [bytes(flipped)] + blobs[5:], b["Vs"])
    r = subprocess.run([exe, bad], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "proof 4 is invalid" in r.stdout, r.stdout + r.stderr
    swapped = str(tmp_path / "swapped.bin")
    write(swapped, blobs, [b["Vs"][1], b["Vs"][0]] + b["Vs"][2:])
    r = subprocess.run([exe, swapped], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "INVALID" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("n", [8, 64])
def test_device_scalars_equal_the_oracles_verifier_quantities(eng, n):
    """The DIRECT link (no host twin in between): for fixed weights w1..w4 per proof, every per-proof scalar and every shared
    coefficient the GPU preparation produces equals the combination of the quantities the reference's verifiers compute, taken
    from the oracle's restatement of them -- delta(y, z) and the hs-scalars of P (rangeproof_verifier.py:55-99), the s-vector of
    Verifier2.get_ss (inner_product_verifier.py:91-102), x_j^2 and x_j^-2 (:140-143), x_ip (:44-58):
        V: -w1 z^2 | T1: -w1 x | T2: -w1 x^2 | A: -w2 | S: -w2 x | u': w3 + w4 a b | P': w2 - w4 | L_j: -w4 x_j^2 | R_j: -w4 x_j^-2
        g: w1 (t_hat - delta) | h: w1 taux + w2 mu | u: -(w2 t_hat + w3) x_ip | gs_i: w2 z + w4 a s_i
        hs_i: w4 b s_i^-1 y^-i - w2 (z y^i + z^2 2^i) y^-i"""
    from oracle import bp_ref as R
    from oracle import cbind
    from helpers import gens
    gs, hs = gens(n, b"dgs"), gens(n, b"dhs")
    g, h, u = (R.elliptic_hash(s) for s in (b"dg", b"dh", b"du"))
    rnd = random.Random(99 + n)
    count = 5
    proofs, blobs = [], []
    from test_batch_verify_cpu import convert_proof
    for k in range(count):
        v = R.Zq(rnd.randrange(2 ** n), Q)
        gamma = R.mod_hash(b"dg%d" % k, Q)
        pr = R.range_prove(v, n, g, h, gs, hs, gamma, u, seed=b"ds%d" % k, multiexp=cbind.msm)
        proofs.append(pr)
        blobs.append(proof_to_bytes(convert_proof(pr)))
    ws = [[rnd.randrange(1, Q) for _ in range(4)] for _ in range(count)]
    weights = b"".join(w.to_bytes(32, "little") for row in ws for w in row)
    rc, bad, v_sc, p_sc, shared, _ = dev_prepare(eng, n, 1, blobs, weights, None)
    assert rc == 0 and bad == -1
    k = n.bit_length() - 1
    per = 6 + 2 * k
    le = lambda raw, i: int.from_bytes(raw[32 * i: 32 * i + 32], "little")
    want_shared = [0] * (5 + 2 * n)
    for j, (pr, (w1, w2, w3, w4)) in enumerate(zip(proofs, ws)):
        x, y, z = (c.x for c in R._range_transcript(pr))
        ip = pr.innerProof
        p2 = ip.proof2
        x_ip = int(ip.transcript.split(b"&")[1])
        a, b = p2.a.x, p2.b.x
        xs = [c.x for c in p2.xs]
        ss = [s.x for s in R.get_ss(p2.xs, n)]                                   # Verifier2.get_ss
        ypow = [pow(y, i, Q) for i in range(n)]
        delta = ((z - z * z) * sum(ypow) - pow(z, 3, Q) * (2 ** n - 1)) % Q      # as range_verify_generic computes it
        yinv = pow(y, -1, Q)
        want_p = [-w1 * x, -w1 * x * x, -w2, -w2 * x, w3 + w4 * a * b, w2 - w4] + [-w4 * c * c for c in xs] + \
                 [-w4 * pow(c, -2, Q) for c in xs]
        assert le(v_sc, j) == (-w1 * z * z) % Q, "V scalar"
        assert [le(p_sc, j * per + t) for t in range(per)] == [v % Q for v in want_p], "point scalars of proof %d" % j
        want_shared[0] += w1 * (pr.t_hat.x - delta)
        want_shared[1] += w1 * pr.taux.x + w2 * pr.mu.x
        want_shared[2] -= (w2 * pr.t_hat.x + w3) * x_ip
        want_shared[3] += w2 * z
        want_shared[4] -= w2 * z
        for i in range(n):
            want_shared[5 + i] += w4 * a * ss[i]
            # the hs-scalar of P over hsp_i = y^-i hs_i is z y^i + z^2 2^i (R._zpow_term); E2 carries it with weight -w2 (the +w2 z
            # part sits in the hs_const cell), E4 carries b s_i^-1 with weight w4, both over the unscaled hs_i
            want_shared[5 + n + i] += (w4 * b * pow(ss[i], -1, Q) - w2 * R._zpow_term(R.Zq(z, Q), i, n).x) * pow(yinv, i, Q)
    assert [le(shared, i) for i in range(5 + 2 * n)] == [v % Q for v in want_shared], "shared coefficients"


def test_one_call_batch_verification_equals_the_two_step_path(eng):
    """bpmi_rp_batch_verify_dev (upload in slices, preparation, shared coefficients folded on the device, one MSM): accepts what
    add_wire_native + verify accepts, gives the identity for valid batches (also aggregated ones, also below and above the
    slicing threshold), a non-identity value for a wrong commitment, "Proof invalid" for corrupted bytes; with fixed weights
    its 64-byte value equals the two-step path's partial()."""
    b = make_batch(6, n=8)
    blobs = [proof_to_bytes(pr) for pr in b["proofs"]]
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    assert bv.partial_wire(b["Vs"], blobs) == bytes(64)
    assert bv.verify_wire(b["Vs"], blobs) is True
    many = [blobs[i % 6] for i in range(5000)]                      # above the slicing threshold (4096 proofs)
    manyV = [b["Vs"][i % 6] for i in range(5000)]
    assert bv.partial_wire(manyV, many) == bytes(64)
    # the commitments packed: as bytes, and as a page-locked buffer (uploaded without a staging copy)
    packed = b"".join(V.to_le64() for V in manyV)
    assert bv.partial_wire(packed, many) == bytes(64)
    pinned = eng.host_alloc(len(packed))
    pinned.view[:] = packed
    assert bv.partial_wire(pinned, many) == bytes(64)
    pinned.view[64 * 4321: 64 * 4322] = packed[64 * 4322: 64 * 4323]
    assert bv.partial_wire(pinned, many) != bytes(64)
    with pytest.raises(Exception, match="Proof invalid"):
        bv.partial_wire(eng.host_alloc(64 * 4999), many)              # a buffer of the wrong size
    wrong = list(manyV)
    wrong[4321] = manyV[4322]
    assert bv.partial_wire(wrong, many) != bytes(64)
    with pytest.raises(Exception, match="Proof invalid"):
        bv.verify_wire(wrong, many)
    bad = bytearray(many[77])
    bad[len(bad) - 3] ^= 4
    with pytest.raises(Exception, match="Proof invalid"):
        bv.partial_wire(manyV, many[:77] + [bytes(bad)] + many[78:])
    # the same weights through both paths: the same 64 bytes (a non-zero value: one commitment is wrong)
    rnd = random.Random(3)
    ws = [rnd.randrange(1, Q) for _ in range(4 * 6)]
    Vs_bad = [b["Vs"][1]] + b["Vs"][1:]
    it1, it2 = iter(ws), iter(ws)
    one = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], rng=lambda: next(it1))
    two = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"], rng=lambda: next(it2))
    two.add_wire_native(Vs_bad, blobs, prepare="device")
    assert one.partial_wire(Vs_bad, blobs) == two.partial() != bytes(64)
    # aggregated proofs (m = 2 values each)
    from oracle import bp_ref as R
    from oracle import cbind
    from helpers import gens
    from test_batch_verify_cpu import convert_proof, gpt
    m, bits = 2, 4
    gs, hs = gens(m * bits, b"ags"), gens(m * bits, b"ahs")
    g, h, u = (R.elliptic_hash(s) for s in (b"ag", b"ah", b"au"))
    Vs_all, ab = [], []
    for j in range(3):
        vs = [R.Zq(rnd.randrange(2 ** bits), Q) for _ in range(m)]
        gammas = [R.mod_hash(b"gb%d-%d" % (j, t), Q) for t in range(m)]
        Vs_all.append([gpt(R.commitment(g, h, v, ga)) for v, ga in zip(vs, gammas)])
        ab.append(proof_to_bytes(convert_proof(R.aggreg_range_prove(vs, bits, g, h, gs, hs, gammas, u, seed=b"bs%d" % j, multiexp=cbind.msm))))
    av = BatchRangeVerifier(gpt(g), gpt(h), [gpt(p) for p in gs], [gpt(p) for p in hs], gpt(u))
    assert av.verify_wire(Vs_all, ab) is True
    Vs_all[1] = [Vs_all[1][1], Vs_all[1][0]]                        # two commitments of one proof exchanged
    with pytest.raises(Exception, match="Proof invalid"):
        av.verify_wire(Vs_all, ab)


# ---- wire format 2 (round 4): format 1 without the transcripts; the device expands it (k_rp_expand_v2) -----------------------------
def _v2(blobs_v1_proofs):
    return [proof_to_bytes(pr, version=2) for pr in blobs_v1_proofs]


@pytest.mark.parametrize("n", [2, 8, 64])
def test_format_2_gives_the_same_scalars_points_and_coefficients_as_format_1(eng, n):
    """The same proofs in both wire formats through the device preparation: identical V scalars, per-proof point scalars, shared
    coefficients and decoded points, byte for byte -- and identical to the host twin (which expands format 2 with
    rp_wire_v2_host.hpp before it parses)."""
    b = make_batch(7, n=n)
    v1 = [proof_to_bytes(pr) for pr in b["proofs"]]
    v2 = _v2(b["proofs"])
    assert all(len(y) < 0.62 * len(x) for x, y in zip(v1, v2))
    rnd = random.Random(n + 5)
    weights = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(4 * 7))
    for w, seed in ((weights, None), (None, bytes(range(32)))):
        d1 = dev_prepare(eng, n, 1, v1, w, seed)
        d2 = dev_prepare(eng, n, 1, v2, w, seed)
        assert d1[0] == 0 and d1[1] == -1 and d2 == d1
        assert_same(eng, n, 1, v2, w, seed)


def test_format_2_aggregated_and_many_proofs(eng):
    from oracle import bp_ref as R
    from oracle import cbind
    from helpers import gens
    from test_batch_verify_cpu import convert_proof
    m, bits = 4, 8
    nm = m * bits
    gs, hs = gens(nm, b"ags"), gens(nm, b"ahs")
    g, h, u = (R.elliptic_hash(s) for s in (b"ag", b"ah", b"au"))
    rnd = random.Random(77)
    proofs = []
    for k in range(3):
        vs = [R.Zq(rnd.randrange(2 ** bits), Q) for _ in range(m)]
        gammas = [R.mod_hash(b"ga%d-%d" % (k, j), Q) for j in range(m)]
        proofs.append(convert_proof(R.aggreg_range_prove(vs, bits, g, h, gs, hs, gammas, u, seed=b"a seed of some length %d" % k, multiexp=cbind.msm)))
    v1, v2 = [proof_to_bytes(pr) for pr in proofs], _v2(proofs)
    assert dev_prepare(eng, nm, m, v2, None, b"\x07" * 32) == dev_prepare(eng, nm, m, v1, None, b"\x07" * 32)
    # 300 proofs, several launch shapes
    b = make_batch(7, n=8)
    base1, base2 = [proof_to_bytes(pr) for pr in b["proofs"]], _v2(b["proofs"])
    want = dev_prepare(eng, 8, 1, [base1[i % 7] for i in range(300)], None, b"\x21" * 32)
    try:
        for lanes, rows in ((0, 0), (16, 37), (0, 128)):
            eng.set_option("rp_lanes", lanes)
            eng.set_option("rp_rows", rows)
            assert dev_prepare(eng, 8, 1, [base2[i % 7] for i in range(300)], None, b"\x21" * 32) == want
    finally:
        eng.set_option("rp_lanes", 0)
        eng.set_option("rp_rows", 0)


def test_format_2_corrupted_proofs_same_verdict_as_their_expansion(eng):
    """160 corruptions of format-2 proofs (bit flips in every section, truncation, trailing bytes, a seed length that lies, another
    proof's magic): the device reports the same first failing proof as the host twin, i.e. as the format-1 checks on the host
    expansion -- and a mix of the two formats in one call is refused at the first proof of the other format."""
    b = make_batch(4, n=8)
    v1s, blobs = [proof_to_bytes(pr) for pr in b["proofs"]], _v2(b["proofs"])
    k = 3
    body = 6 + 32 * (5 + k) + 33 * (6 + 2 * k)
    rnd = random.Random(12)
    w = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(16))
    rejected = 0
    for trial in range(160):
        j = rnd.randrange(4)
        bad = bytearray(blobs[j])
        kind = rnd.choice(("scalar", "point", "challenge", "seed", "header", "truncate", "extend", "seedlen", "two"))
        if kind == "scalar":
            bad[rnd.randrange(6, 6 + 32 * (5 + k))] ^= 1 << rnd.randrange(8)
        elif kind == "point":
            bad[rnd.randrange(6 + 32 * (5 + k), body)] ^= 1 << rnd.randrange(8)
        elif kind == "challenge":
            bad[rnd.randrange(body, body + 128)] ^= 1 << rnd.randrange(8)
        elif kind == "seed":
            bad[rnd.randrange(body + 130, len(bad))] ^= 1 << rnd.randrange(8)
        elif kind == "header":
            bad[rnd.randrange(0, 6)] ^= 1 << rnd.randrange(8)
        elif kind == "truncate":
            del bad[rnd.randrange(0, len(bad)):]
        elif kind == "extend":
            bad += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9)))
        elif kind == "seedlen":
            bad[body + 128 + rnd.randrange(2)] ^= 1 << rnd.randrange(8)
        mutated = blobs[:j] + [bytes(bad)] + blobs[j + 1:]
        if kind == "two":
            j2 = rnd.randrange(4)
            b2 = bytearray(mutated[j2])
            b2[rnd.randrange(6, 6 + 32 * (5 + k))] ^= 0x10
            mutated[j2] = bytes(b2)
        if mutated[0][:5] != b"BPRP2":           # the first proof tells the format of the call: keep it format 2 (the mix is tested below)
            continue
        h = host_prepare(8, 1, mutated, w, None)
        d = dev_prepare(eng, 8, 1, mutated, w, None)
        assert h[0] == 0 and d[0] == 0
        host_bad = h[1]
        upto = 4 if host_bad < 0 else host_bad
        if upto:
            _, ok = eng.ec_decompress_batch_bytes(h[5][:33 * upto * (6 + 2 * k)], upto * (6 + 2 * k))
            if 0 in ok:
                host_bad = ok.index(0) // (6 + 2 * k)
        assert d[1] == host_bad, (trial, kind, d[1], host_bad)
        rejected += d[1] >= 0
    assert rejected >= 90           # (flips in taux, mu, t_hat, a, b pass the byte-level checks: those proofs fail in the MSM)
    assert dev_prepare(eng, 8, 1, [blobs[0], v1s[1], blobs[2]], w, None)[1] == 1
    assert dev_prepare(eng, 8, 1, [v1s[0], blobs[1], v1s[2]], w, None)[1] == 1


def test_mixed_wire_formats_are_an_argument_error_not_a_verdict(eng):
    """A batch is read in the format of its first proof: a well-formed proof of the OTHER format inside it is reported by
    the device paths as an argument error that names it -- not as "Proof invalid", which is what a forged proof gets; the host
    preparation takes the formats proof by proof and accepts the mix."""
    from bulletproofs_amd.engine import EngineError
    b = make_batch(6, n=8)
    v1 = [proof_to_bytes(pr) for pr in b["proofs"]]
    v2 = _v2(b["proofs"])
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    assert bv.partial_wire(b["Vs"], v1) == bytes(64) and bv.partial_wire(b["Vs"], v2) == bytes(64)
    with pytest.raises(EngineError, match="mixed wire formats: proof 3 is format 1 in a format-2 batch"):
        bv.partial_wire(b["Vs"], v2[:3] + [v1[3]] + v2[4:])
    with pytest.raises(EngineError, match="mixed wire formats: proof 2 is format 2 in a format-1 batch"):
        bv.partial_wire(b["Vs"], v1[:2] + [v2[2]] + v1[3:])
    # a CORRUPTED proof of the batch's own format is still a verdict: a flipped scalar bit fails the MSM, a broken header the parser
    bad = bytearray(v2[4])
    bad[40] ^= 1
    with pytest.raises(Exception, match="^Proof invalid$"):
        bv.verify_wire(b["Vs"], v2[:4] + [bytes(bad)] + v2[5:])
    bad = bytearray(v2[4])
    bad[2] = ord("X")
    with pytest.raises(Exception, match="^Proof invalid$"):
        bv.partial_wire(b["Vs"], v2[:4] + [bytes(bad)] + v2[5:])
    seed = bytes(range(32))
    for mixed in (v2[:3] + [v1[3]] + v2[4:], v1[:2] + [v2[2]] + v1[3:]):
        h = host_prepare(8, 1, mixed, None, seed)
        want = host_prepare(8, 1, v1, None, seed)
        assert h[0] == 0 and h[1] == -1 and h[2:5] == want[2:5]
        d = dev_prepare(eng, 8, 1, mixed, None, seed)
        assert d[0] == -3


# ---- wire format 3 (round 6): format 2 + the points' y coordinates; the device CHECKS each y instead of taking a square root -------
def _v3(proofs):
    return [proof_to_bytes(pr, version=3) for pr in proofs]


@pytest.mark.parametrize("n", [2, 8, 64])
def test_format_3_gives_the_same_scalars_points_and_coefficients_as_format_1(eng, n):
    """The same proofs in formats 1, 2 and 3 through the device preparation: identical V scalars, per-proof point scalars, shared
    coefficients and DECODED POINTS byte for byte (format 3's come from the hinted branch of k_ec_decompress_wire: no square root),
    identical to the host twin; and codec.wire_v2_to_v3 (one batched decompression) writes the very bytes the codec writes from Proof objects."""
    from bulletproofs_amd.rangeproofs.codec import wire_v2_to_v3
    b = make_batch(7, n=n)
    v1, v2, v3 = [proof_to_bytes(pr) for pr in b["proofs"]], _v2(b["proofs"]), _v3(b["proofs"])
    assert wire_v2_to_v3(v2, eng) == v3
    rnd = random.Random(n + 6)
    weights = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(4 * 7))
    for w, seed in ((weights, None), (None, bytes(range(32)))):
        d1 = dev_prepare(eng, n, 1, v1, w, seed)
        d3 = dev_prepare(eng, n, 1, v3, w, seed)
        assert d1[0] == 0 and d1[1] == -1 and d3 == d1
        assert_same(eng, n, 1, v3, w, seed)
    # many proofs, several launch shapes, and the sliced upload (>= 4096 proofs: the decoding of a slice runs beside the next upload)
    want = dev_prepare(eng, n, 1, [v1[i % 7] for i in range(300)], None, b"\x21" * 32)
    try:
        for lanes, rows in ((0, 0), (16, 37)):
            eng.set_option("rp_lanes", lanes)
            eng.set_option("rp_rows", rows)
            assert dev_prepare(eng, n, 1, [v3[i % 7] for i in range(300)], None, b"\x21" * 32) == want
    finally:
        eng.set_option("rp_lanes", 0)
        eng.set_option("rp_rows", 0)
    if n == 8:
        assert dev_prepare(eng, n, 1, [v3[i % 7] for i in range(4100)], None, b"\x22" * 32) == dev_prepare(eng, n, 1, [v1[i % 7] for i in range(4100)], None, b"\x22" * 32)


def test_format_3_a_wrong_y_is_an_invalid_proof_on_the_device_as_on_the_host(eng):
    """200 corruptions of format-3 proofs -- every section of format 2, and the y coordinates: flipped bits, the other root, another
    point's y, values not below p, a zero, a flipped tag: the device names the same first failing proof as the host twin (which checks
    the ys with its own field arithmetic, rp_wire_v2_host.hpp hint_ok).  No corruption of a y passes."""
    from bulletproofs_amd.ec import secp256k1
    P = secp256k1.p
    b = make_batch(4, n=8)
    blobs = _v3(b["proofs"])
    v1s, v2s = [proof_to_bytes(pr) for pr in b["proofs"]], _v2(b["proofs"])
    k, npts = 3, 12
    body = 6 + 32 * (5 + k) + 33 * npts
    rnd = random.Random(13)
    w = b"".join(rnd.randrange(1, Q).to_bytes(32, "little") for _ in range(16))
    rejected = y_trials = 0
    for trial in range(200):
        j = rnd.randrange(4)
        bad = bytearray(blobs[j])
        at = len(bad) - 32 * npts
        kind = rnd.choice(("scalar", "point", "challenge", "seed", "header", "truncate", "extend", "seedlen", "ybit", "ybit", "yneg", "yswap", "ybig", "yzero", "tag"))
        t = rnd.randrange(npts)
        y = int.from_bytes(bad[at + 32 * t: at + 32 * t + 32], "big")
        if kind == "scalar":
            bad[rnd.randrange(6, 6 + 32 * (5 + k))] ^= 1 << rnd.randrange(8)
        elif kind == "point":
            bad[rnd.randrange(6 + 32 * (5 + k), body)] ^= 1 << rnd.randrange(8)
        elif kind == "challenge":
            bad[rnd.randrange(body, body + 128)] ^= 1 << rnd.randrange(8)
        elif kind == "seed":
            bad[rnd.randrange(body + 130, at)] ^= 1 << rnd.randrange(8)
        elif kind == "header":
            bad[rnd.randrange(0, 6)] ^= 1 << rnd.randrange(8)
        elif kind == "truncate":
            del bad[rnd.randrange(0, len(bad)):]
        elif kind == "extend":
            bad += bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 40)))
        elif kind == "seedlen":
            bad[body + 128 + rnd.randrange(2)] ^= 1 << rnd.randrange(8)
        elif kind == "ybit":
            bad[rnd.randrange(at, len(bad))] ^= 1 << rnd.randrange(8)
        elif kind == "yneg":
            bad[at + 32 * t: at + 32 * t + 32] = (P - y).to_bytes(32, "big")
        elif kind == "yswap":
            t2 = (t + 1 + rnd.randrange(npts - 1)) % npts
            bad[at + 32 * t: at + 32 * t + 32] = blobs[j][at + 32 * t2: at + 32 * t2 + 32]
        elif kind == "ybig":
            bad[at + 32 * t: at + 32 * t + 32] = rnd.choice((P, P + 1, (1 << 256) - 1, y + P if y + P < (1 << 256) else P)).to_bytes(32, "big")
        elif kind == "yzero":
            bad[at + 32 * t: at + 32 * t + 32] = bytes(32)
        elif kind == "tag":
            bad[6 + 32 * (5 + k) + 33 * t] ^= 1
        mutated = blobs[:j] + [bytes(bad)] + blobs[j + 1:]
        if mutated[0][:5] != b"BPRP3":           # the first proof tells the format of the call
            continue
        h = host_prepare(8, 1, mutated, w, None)
        d = dev_prepare(eng, 8, 1, mutated, w, None)
        assert h[0] == 0 and d[0] == 0
        host_bad = h[1]
        upto = 4 if host_bad < 0 else host_bad
        if upto:
            _, ok = eng.ec_decompress_batch_bytes(h[5][:33 * upto * npts], upto * npts)
            if 0 in ok:
                host_bad = ok.index(0) // npts
        assert d[1] == host_bad, (trial, kind, d[1], host_bad)
        rejected += d[1] >= 0
        if kind[0] == "y" or kind == "tag":
            y_trials += 1
            assert d[1] == j, (trial, kind)
    assert rejected >= 130 and y_trials >= 60
    for other in (v1s, v2s):
        assert dev_prepare(eng, 8, 1, [blobs[0], other[1], blobs[2]], w, None)[1] == 1
        assert dev_prepare(eng, 8, 1, [other[0], blobs[1], other[2]], w, None)[1] == 1


def test_format_3_verdicts_and_mixed_formats(eng):
    from bulletproofs_amd.engine import EngineError
    b = make_batch(6, n=8)
    v1, v2, v3 = [proof_to_bytes(pr) for pr in b["proofs"]], _v2(b["proofs"]), _v3(b["proofs"])
    bv = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    assert bv.partial_wire(b["Vs"], v3) == bytes(64) and bv.verify_wire(b["Vs"], v3) is True
    with pytest.raises(EngineError, match="mixed wire formats: proof 3 is format 2 in a format-3 batch"):
        bv.partial_wire(b["Vs"], v3[:3] + [v2[3]] + v3[4:])
    with pytest.raises(EngineError, match="mixed wire formats: proof 2 is format 3 in a format-1 batch"):
        bv.partial_wire(b["Vs"], v1[:2] + [v3[2]] + v1[3:])
    with pytest.raises(EngineError, match="mixed wire formats: proof 5 is format 3 in a format-2 batch"):
        bv.partial_wire(b["Vs"], v2[:5] + [v3[5]])
    # the other root of ONE point of ONE proof: a verdict, not an error
    from bulletproofs_amd.ec import secp256k1
    bad = bytearray(v3[4])
    y = int.from_bytes(bad[-32:], "big")
    bad[-32:] = (secp256k1.p - y).to_bytes(32, "big")
    with pytest.raises(Exception, match="^Proof invalid$"):
        bv.verify_wire(b["Vs"], v3[:4] + [bytes(bad)] + v3[5:])
    # a valid proof with the commitments of another: rejected by the MSM, as in the other formats
    with pytest.raises(Exception, match="^Proof invalid$"):
        bv.verify_wire(b["Vs"][1:] + b["Vs"][:1], v3)
    # the Python path (add_wire: points decoded from the encodings) takes format 3 as well
    bv2 = BatchRangeVerifier(b["g"], b["h"], b["gs"], b["hs"], b["u"])
    bv2.add_wire(b["Vs"], v3)
    assert bv2.verify() is True


@pytest.mark.parametrize("family,k", [("single", i) for i in range(7)] + [("aggregated", i) for i in range(3)])
def test_golden_proofs_in_the_three_formats(eng, gp, family, k):
    """The REFERENCE-MADE proofs of tests/golden/rangeproofs.json in the three wire formats (bytes pinned by tests/golden/wire_formats.json):
    the device preparation gives identical scalars, coefficients and points for each, the batch verifier accepts each with the golden's
    commitments and generators, and rejects each with a commitment exchanged for another point."""
    import hashlib
    import os
    import sys
    from conftest import load_golden
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_wire_golden
    from helpers import P
    from test_gpu_rangeproofs import inputs
    c = load_golden("rangeproofs.json")[family][k]
    want = [e for e in load_golden("wire_formats.json")["proofs"] if e["family"] == family and e["index"] == k][0]
    m = c.get("m", 1)
    s, n, gs, hs, g, h, u = inputs(gp, c, m)
    pr = make_wire_golden.proof_of(c["proof"])
    blobs = [proof_to_bytes(pr, version=v) for v in (1, 2, 3)]
    for v, b in zip((1, 2, 3), blobs):
        assert hashlib.sha256(b).hexdigest() == want["format_%d" % v]["sha256"]
    seed = bytes(range(32))
    d = [dev_prepare(eng, n * m, m, [b], None, seed) for b in blobs]
    assert d[0][0] == 0 and d[0][1] == -1 and d[1] == d[0] and d[2] == d[0]
    Vs = [gp.to_gpu(P(x)) for x in c["Vs"]] if m > 1 else gp.to_gpu(P(c["V"]))
    bv = BatchRangeVerifier(g, h, gs, hs, u)
    for b in blobs:
        assert bv.verify_wire([Vs], [b]) is True
        wrong = [[Vs[1], Vs[0]] + list(Vs[2:])] if m > 1 else [g]
        if m == 1 or Vs[0] != Vs[1]:
            with pytest.raises(Exception, match="^Proof invalid$"):
                bv.verify_wire(wrong, [b])


def test_preparation_options_never_change_a_result(eng):
    """Round 6's scheduling options of the batch preparation -- rp_priority (issue priority of the chain kernels: 0 / 1 / 2) and rp_slices
    (uploads per batch: 1 .. 4, from 4 096 proofs) -- are about WHEN things run: the scalars, coefficients, points and verdicts of a batch
    of 4 100 proofs are the same bytes under every setting, in every wire format."""
    b = make_batch(7, n=8)
    v1, v2, v3 = [proof_to_bytes(pr) for pr in b["proofs"]], _v2(b["proofs"]), _v3(b["proofs"])
    seed = b"\x31" * 32
    want = dev_prepare(eng, 8, 1, [v1[i % 7] for i in range(4100)], None, seed)
    assert want[0] == 0 and want[1] == -1
    bad = bytearray(v3[5])
    bad[-3] ^= 4                                                       # a y of proof 3 001
    try:
        for prio, slices in ((0, 0), (1, 1), (2, 2), (2, 3), (1, 4)):
            eng.set_option("rp_priority", prio)
            eng.set_option("rp_slices", slices)
            for blobs in (v1, v2, v3):
                assert dev_prepare(eng, 8, 1, [blobs[i % 7] for i in range(4100)], None, seed) == want, (prio, slices, blobs[0][:5])
            mutated = [v3[i % 7] for i in range(4100)]
            mutated[3001] = bytes(bad)
            assert dev_prepare(eng, 8, 1, mutated, None, seed)[1] == 3001
    finally:
        eng.set_option("rp_priority", 1)
        eng.set_option("rp_slices", 0)
