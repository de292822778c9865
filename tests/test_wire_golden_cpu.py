"""tests/golden/wire_formats.json (made by tests/golden/make_wire_golden.py from the REFERENCE-MADE proofs of rangeproofs.json): the byte
layout of the three wire formats is pinned -- the codec must write exactly these bytes, and the native host expander
(bpmi_rp_wire_v2_to_v1) must turn formats 2 and 3 of every golden proof into its format 1, whose transcripts are the reference's own.
CPU only; the device expander on the same proofs: tests/test_gpu_batch_dev.py::test_golden_proofs_in_the_three_formats."""
import hashlib
import os
import sys

from conftest import load_golden

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_wire_golden  # noqa: E402

import bulletproofs_amd  # noqa: E402,F401
from bulletproofs_amd.rangeproofs import codec  # noqa: E402
from bulletproofs_amd.rangeproofs.codec import proof_to_bytes, wire_v2_to_v1, wire_v3_to_v2  # noqa: E402

from test_wire_v2_cpu import native_expand  # noqa: E402


def test_codec_writes_the_committed_bytes():
    want = load_golden("wire_formats.json")["proofs"]
    assert make_wire_golden.entries() == want
    assert len(want) == 10 and all(e["format_2"]["bytes"] < e["format_3"]["bytes"] < e["format_1"]["bytes"] for e in want)


def test_golden_proofs_expand_to_the_reference_transcripts():
    gold = load_golden("rangeproofs.json")
    want = {(e["family"], e["index"]): e for e in load_golden("wire_formats.json")["proofs"]}
    for family in ("single", "aggregated"):
        for i, c in enumerate(gold[family]):
            pr = make_wire_golden.proof_of(c["proof"])
            v1, v2, v3 = (proof_to_bytes(pr, version=v) for v in (1, 2, 3))
            e = want[(family, i)]
            for v, b in ((1, v1), (2, v2), (3, v3)):
                assert len(b) == e["format_%d" % v]["bytes"] and hashlib.sha256(b).hexdigest() == e["format_%d" % v]["sha256"]
                if "hex" in e["format_%d" % v]:
                    assert bytes.fromhex(e["format_%d" % v]["hex"]) == b
            # format 1 carries the reference's transcripts verbatim; formats 2 and 3 must rebuild exactly them
            assert c["proof"]["transcript"].encode() in v1 and c["proof"]["inner"]["proof2"]["transcript"].encode() in v1
            assert wire_v2_to_v1(v2) == v1 and wire_v2_to_v1(v3) == v1 and wire_v3_to_v2(v3) == v2
            assert native_expand([v2, v3]) == (0, -1, [v1, v1])
            assert codec.parse_blob(v3) == codec.parse_blob(v1)
