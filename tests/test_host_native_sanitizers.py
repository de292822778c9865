"""The native HOST code of libbpmi other than the wire parser -- the range proofs' O(n m) scalar algebra (threaded), the transcript
builder of bpmi_ipa_prove_rounds with its bounded export, bpmi_mod_hash_range (threaded), the host tail of an MSM -- compiled for
the host with ASan + UBSan and with TSan and driven by tests/csrc_host/host_native_fuzz.cpp (VERDICT r03 "next" #5).
Reference code these restate: src/rangeproofs/rangeproof_aggreg_prover.py:117-146, rangeproof_aggreg_verifier.py:96-108,
src/utils/transcript.py:13-33, src/utils/utils.py:84-97."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "csrc_host", "host_native_fuzz.cpp")
INC = os.path.join(REPO, "python-bulletproofs_amd", "csrc")


@pytest.mark.parametrize("name,flags,iters", [("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], 150),
                                              ("tsan", ["-fsanitize=thread"], 60)])
def test_native_host_code_under_sanitizers(tmp_path, name, flags, iters):
    exe = str(tmp_path / ("host_native_fuzz_" + name))
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17"] + flags + ["-I", INC, SRC, "-o", exe, "-lpthread"])
    for seed in (1, 2):
        r = subprocess.run([exe, str(iters), str(seed)], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
        assert "0 failed checks" in r.stdout
        assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr


def test_transcript_export_refuses_a_short_buffer_through_the_abi():
    """bpmi_ipa_prove_rounds' `cap` contract is enforced by rpt::export_digest (transcript_host.hpp): the same function, called
    through a host build, must refuse cap = len - 1 without writing and accept cap = len (the harness checks it under ASan on
    exact-size heap blocks; here the contract is stated once more as a plain test of the header's text)."""
    text = open(os.path.join(INC, "transcript_host.hpp")).read()
    assert "if (dg.size() > cap) return false;" in text
    bpmi = open(os.path.join(INC, "bpmi.hip")).read()
    assert 'if (!rpt::export_digest(dg, digest_out, cap, out_len)) return fail(ctx, BPMI_E_ARG, "digest_out too small");' in bpmi
