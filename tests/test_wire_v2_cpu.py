"""Wire format 2 of a range proof (round 4; rangeproofs/codec.py, csrc/rp_wire_v2_host.hpp): format 1 without the transcripts.  CPU
only: the Python expander, the native host expander (bpmi_rp_wire_v2_to_v1) and the original format-1 bytes must agree on
oracle-made proofs of several shapes; malformed format-2 proofs are refused; the parsers take either format.  The device
expander and the verdicts are in tests/test_gpu_batch_dev.py."""
import ctypes
import random

import pytest

import bulletproofs_amd  # noqa: F401
from bulletproofs_amd import _native
from bulletproofs_amd.rangeproofs import codec
from bulletproofs_amd.rangeproofs.codec import proof_to_bytes, wire_v2_to_v1

from helpers import Q
from test_batch_verify_cpu import make_batch


def native_expand(v2s):
    lib = _native.load()
    joined = b"".join(v2s)
    off = (ctypes.c_uint64 * (len(v2s) + 1))(*([0] + [sum(map(len, v2s[:i + 1])) for i in range(len(v2s))]))
    cap = 4 * len(joined) + 4096
    out = ctypes.create_string_buffer(cap)
    out_off = (ctypes.c_uint64 * (len(v2s) + 1))()
    bad = ctypes.c_int64(-7)
    rc = lib.bpmi_rp_wire_v2_to_v1(joined, len(joined), off, len(v2s), out, cap, out_off, ctypes.byref(bad))
    return rc, bad.value, [out.raw[out_off[i]: out_off[i + 1]] for i in range(len(v2s))] if rc == 0 and bad.value < 0 else None


@pytest.mark.parametrize("n", [2, 8, 64])
def test_format_2_expands_to_the_format_1_bytes(n):
    b = make_batch(6, n=n)
    v1s = [proof_to_bytes(pr) for pr in b["proofs"]]
    v2s = [proof_to_bytes(pr, version=2) for pr in b["proofs"]]
    for v1, v2 in zip(v1s, v2s):
        assert v2[:5] == b"BPRP2" and len(v2) < 0.62 * len(v1)
        assert wire_v2_to_v1(v2) == v1
        assert codec.parse_blob(v2) == codec.parse_blob(v1) and codec.compressed_points(v2) == codec.compressed_points(v1)
    rc, bad, got = native_expand(v2s)
    assert (rc, bad) == (0, -1) and got == v1s
    if n == 64:
        assert len(v2s[0]) < 1150 and len(v1s[0]) > 2400         # what the upload of a batch shrinks by


def test_malformed_format_2_proofs_are_refused_by_both_expanders():
    b = make_batch(3, n=8)
    v2 = proof_to_bytes(b["proofs"][0], version=2)
    k = v2[5]
    body = 6 + 32 * (5 + k) + 33 * (6 + 2 * k)
    rnd = random.Random(2)
    bads = [v2[:-1], v2 + b"\x00", v2[:body + 100], b"BPRP3" + v2[5:], v2[:5] + bytes([17]) + v2[6:], v2[:10]]
    bads.append(v2[:body] + Q.to_bytes(32, "big") + v2[body + 32:])                    # y >= q
    bads.append(v2[:body + 96] + (Q + 5).to_bytes(32, "big") + v2[body + 128:])         # x_ip >= q
    bads.append(v2[:6 + 32 * 5] + ((1 << 256) - 1).to_bytes(32, "big") + v2[6 + 32 * 6:])   # a round challenge >= q
    bads.append(v2[:body + 128] + b"\xff\xff" + v2[body + 130:])                        # a seed length that leaves the proof
    for bad in bads:
        with pytest.raises(Exception, match="Proof invalid"):
            wire_v2_to_v1(bad)
        rc, first, _ = native_expand([v2, bad, v2])
        assert rc == 0 and first == 1
    # every single-bit flip either breaks the format or changes the expansion (nothing in format 2 is ignored)
    ref = wire_v2_to_v1(v2)
    for _ in range(300):
        pos = rnd.randrange(len(v2))
        flip = bytearray(v2)
        flip[pos] ^= 1 << rnd.randrange(8)
        try:
            out = wire_v2_to_v1(bytes(flip))
        except Exception:
            out = None
        rc, first, got = native_expand([bytes(flip)])
        assert (out is None) == (first == 0)
        if out is not None:
            assert out != ref and got == [out]
    rc, first, _ = native_expand([])
    assert (rc, first) == (0, -1)


def test_a_proof_with_foreign_transcripts_has_no_format_2_form():
    b = make_batch(1, n=8)
    pr = b["proofs"][0]
    pr.transcript = pr.transcript + b"extra&"
    with pytest.raises(ValueError, match="not canonical"):
        proof_to_bytes(pr, version=2)


def _host_prepare(n, blobs, seed=bytes(range(32))):
    lib = _native.load()
    count = len(blobs)
    k = n.bit_length() - 1
    npts = count * (6 + 2 * k)
    joined = b"".join(blobs) + b"\x00"
    offs = [0]
    for x in blobs:
        offs.append(offs[-1] + len(x))
    o = (ctypes.c_uint64 * (count + 1))(*offs)
    v_sc, p_sc = ctypes.create_string_buffer(32 * count), ctypes.create_string_buffer(32 * npts)
    shared, comp = ctypes.create_string_buffer(32 * (5 + 2 * n)), ctypes.create_string_buffer(33 * npts)
    bad = ctypes.c_int64(-1)
    rc = lib.bpmi_rp_batch_prepare(n, 1, count, joined, len(joined), ctypes.cast(o, ctypes.c_void_p), None, seed, 2, v_sc, p_sc, shared, comp,
                                   ctypes.cast(ctypes.pointer(bad), ctypes.c_void_p))
    return rc, bad.value, v_sc.raw, p_sc.raw, shared.raw


def test_host_preparation_takes_the_formats_proof_by_proof_and_names_the_first_bad_proof():
    """bpmi_rp_batch_prepare (round 5): a batch may mix the two wire formats; a blob that CLAIMS format 2 and does not expand is
    rejected at its own index, behind any earlier bad proof (returning at once from the expansion made the host name a later proof
    than the device: tools/fuzz_batch_prepare.py); an empty blob is a bad proof, not a crash (the sanitizer harness's find)."""
    b = make_batch(6, n=8)
    v1 = [proof_to_bytes(pr) for pr in b["proofs"]]
    v2 = [proof_to_bytes(pr, version=2) for pr in b["proofs"]]
    want = _host_prepare(8, v1)
    assert want[:2] == (0, -1)
    for mixed in (v2, v2[:3] + [v1[3]] + v2[4:], v1[:2] + [v2[2]] + v1[3:], [v1[0]] + v2[1:]):
        assert _host_prepare(8, mixed) == want
    garbage2 = b"BPRP2" + bytes(200)                       # claims format 2, is nothing
    broken1 = bytearray(v1[1])
    broken1[len(broken1) - 9] ^= 4                          # a flipped transcript bit
    assert _host_prepare(8, v1[:1] + [bytes(broken1)] + v1[2:4] + [garbage2] + v1[5:])[:2] == (0, 1)
    assert _host_prepare(8, v1[:4] + [garbage2] + v1[5:])[:2] == (0, 4)
    assert _host_prepare(8, v2[:2] + [b""] + v2[3:])[:2] == (0, 2)
    assert _host_prepare(8, [b""] + v1[1:])[:2] == (0, 0)
    rc, first, _ = native_expand([v2[0], b"", v2[1]])
    assert (rc, first) == (0, 1)
