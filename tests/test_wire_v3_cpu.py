"""Wire format 3 of a range proof (round 6; rangeproofs/codec.py, csrc/rp_wire_v2_host.hpp): a format-2 proof followed by the y
coordinates of its points, which a verifier checks instead of computing.  CPU only: the Python codec, the native host expander
(bpmi_rp_wire_v2_to_v1 takes formats 2 and 3) and the host preparation must agree on oracle-made proofs; a y that is not THE y of its
point -- the other root, a flipped bit, another point's y, a value not below p -- makes the proof invalid everywhere.  The device
path (k_ec_decompress_wire's hinted branch, k_rp_expand_v2) and the verdicts are in tests/test_gpu_batch_dev.py."""
import random

import pytest

import bulletproofs_amd  # noqa: F401
from bulletproofs_amd.ec import secp256k1
from bulletproofs_amd.rangeproofs import codec
from bulletproofs_amd.rangeproofs.codec import proof_to_bytes, wire_v2_to_v1, wire_v3_to_v2

from test_batch_verify_cpu import make_batch
from test_wire_v2_cpu import _host_prepare, native_expand

P = secp256k1.p


def _ys_at(v3):
    k = v3[5]
    return len(v3) - 32 * (6 + 2 * k), 6 + 2 * k


@pytest.mark.parametrize("n", [2, 8, 64])
def test_format_3_is_format_2_plus_the_y_coordinates(n):
    b = make_batch(5, n=n)
    v1s = [proof_to_bytes(pr) for pr in b["proofs"]]
    v2s = [proof_to_bytes(pr, version=2) for pr in b["proofs"]]
    v3s = [proof_to_bytes(pr, version=3) for pr in b["proofs"]]
    for pr, v1, v2, v3 in zip(b["proofs"], v1s, v2s, v3s):
        k = v2[5]
        assert v3[:5] == b"BPRP3" and len(v3) == len(v2) + 32 * (6 + 2 * k) and v3[5:len(v2)] == v2[5:]
        assert wire_v3_to_v2(v3) == v2 and wire_v2_to_v1(v3) == v1
        assert codec.parse_blob(v3) == codec.parse_blob(v1) and codec.compressed_points(v3) == codec.compressed_points(v1)
        at, npts = _ys_at(v3)
        ip = pr.innerProof
        pts = [pr.T1, pr.T2, pr.A, pr.S, ip.u_new, ip.P_new] + list(ip.proof2.Ls) + list(ip.proof2.Rs)
        assert [int.from_bytes(v3[at + 32 * j: at + 32 * j + 32], "big") for j in range(npts)] == [pt.y for pt in pts]
    rc, bad, got = native_expand(v3s)
    assert (rc, bad) == (0, -1) and got == v1s
    rc, bad, got = native_expand([v3s[0], v2s[1], v3s[2]])                  # the host expander takes the formats proof by proof
    assert (rc, bad) == (0, -1) and got == v1s[:3]
    if n == 64:
        assert 1650 < len(v3s[0]) < 1750


def test_a_wrong_y_makes_the_proof_invalid():
    b = make_batch(3, n=8)
    v3 = proof_to_bytes(b["proofs"][0], version=3)
    good = proof_to_bytes(b["proofs"][1], version=3)
    at, npts = _ys_at(v3)

    def with_y(j, y):
        return v3[:at + 32 * j] + y.to_bytes(32, "big") + v3[at + 32 * j + 32:]

    def y_of(j):
        return int.from_bytes(v3[at + 32 * j: at + 32 * j + 32], "big")

    bads = []
    for j in (0, 3, 5, npts - 1):
        bads += [with_y(j, P - y_of(j)), with_y(j, 0), with_y(j, y_of(j) ^ 2), with_y(j, y_of((j + 1) % npts)), with_y(j, (1 << 256) - 1), with_y(j, P)]
    bads += [v3[:-1], v3 + b"\x00", v3[:at], v3[:at] + v3[at + 32:], b"BPRP2" + v3[5:]]
    # the tag flipped: the encoding now names the OTHER root, the y of the proof no longer belongs to it
    c0 = 6 + 32 * (5 + v3[5])
    bads.append(v3[:c0] + bytes([v3[c0] ^ 1]) + v3[c0 + 1:])
    for bad in bads:
        with pytest.raises(Exception, match="Proof invalid"):
            wire_v2_to_v1(bad)
        rc, first, _ = native_expand([good, bad, good])
        assert rc == 0 and first == 1
        assert _host_prepare(8, [good, bad, good])[:2] == (0, 1)


def test_every_bit_of_a_format_3_proof_counts():
    b = make_batch(1, n=8)
    v3 = proof_to_bytes(b["proofs"][0], version=3)
    ref = wire_v2_to_v1(v3)
    at, _ = _ys_at(v3)
    rnd = random.Random(3)
    for i in range(400):
        pos = rnd.randrange(at, len(v3)) if i % 2 else rnd.randrange(len(v3))
        flip = bytearray(v3)
        flip[pos] ^= 1 << rnd.randrange(8)
        try:
            out = wire_v2_to_v1(bytes(flip))
        except Exception:
            out = None
        rc, first, got = native_expand([bytes(flip)])
        assert rc == 0 and (out is None) == (first == 0), pos
        if pos >= at:
            assert out is None                                   # a y has ONE right value
        if out is not None:
            assert out != ref and got == [out]


def test_the_identity_has_the_zero_y():
    b = make_batch(1, n=8)
    v3 = proof_to_bytes(b["proofs"][0], version=3)
    at, npts = _ys_at(v3)
    c0 = 6 + 32 * (5 + v3[5])
    j = 7
    ident = v3[:c0 + 33 * j] + bytes(33) + v3[c0 + 33 * (j + 1): at + 32 * j] + bytes(32) + v3[at + 32 * (j + 1):]
    out = wire_v2_to_v1(ident)                                   # (not a valid PROOF any more, but a well-formed one)
    rc, first, got = native_expand([ident])
    assert (rc, first) == (0, -1) and got == [out]
    nonzero = ident[:at + 32 * j] + (1).to_bytes(32, "big") + ident[at + 32 * (j + 1):]
    with pytest.raises(Exception, match="Proof invalid"):
        wire_v2_to_v1(nonzero)
    assert native_expand([nonzero])[:2] == (0, 0)


def test_host_preparation_gives_the_same_numbers_for_the_three_formats():
    b = make_batch(6, n=8)
    v1 = [proof_to_bytes(pr) for pr in b["proofs"]]
    v2 = [proof_to_bytes(pr, version=2) for pr in b["proofs"]]
    v3 = [proof_to_bytes(pr, version=3) for pr in b["proofs"]]
    want = _host_prepare(8, v1)
    assert want[:2] == (0, -1)
    for batch in (v3, v3[:2] + [v1[2], v2[3]] + v3[4:], [v1[0]] + v3[1:]):
        assert _host_prepare(8, batch) == want
