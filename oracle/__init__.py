"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the hot path of wborgeaud/python-bulletproofs
(src/pippenger + src/innerproduct and their callers), used as the checker for
the HIP engine in `python-bulletproofs_amd/`.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import anything from this package.  The product package never imports it; the
product path raises if the HIP library is missing.

Parity pinning (see DESIGN.md "Oracle"):
  * Everything ABOVE the `fastecdsa` boundary (Pippenger, commitments, IPA,
    range proofs, transcript, mod_hash, codecs) is pinned by golden vectors
    produced by running the reference's own, unmodified Python from
    /root/reference (tests/golden/make_golden.py), and the restatement in this
    package is checked against those vectors.
  * The `fastecdsa` boundary itself (third-party C extension, no pinned version:
    /root/reference/.travis.yml installs GitHub master; not vendored, not
    installable here) is a restatement of the published secp256k1 affine group
    law (oracle/ec.py), pinned by the public SEC2/secp256k1 known-answer vectors
    (G, 2G, 3G, (q-1)G = -G, qG = inf) in tests/test_oracle_ec.py.  The reference
    has no golden vector of its own at that boundary -> "parity unpinned by the
    reference's own tests" there; the group law is canonical, so any correct
    implementation is bit-identical.
"""
