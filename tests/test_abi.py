"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, exports
every symbol include/bpmi.h declares, and fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import bulletproofs_amd  # noqa: F401
    from bulletproofs_amd import build, _native
    build.build()
    return _native


def header_functions():
    text = open(os.path.join(REPO, "include", "bpmi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bpmi_[A-Za-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(native):
    declared = header_functions()
    assert len(declared) >= 30
    assert sorted(native.SIGNATURES) == declared
    lib = native.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_every_entry_cites_the_reference():
    text = open(os.path.join(REPO, "include", "bpmi.h")).read()
    for needle in ("src/pippenger/pippenger.py:22-61", "src/innerproduct/inner_product_prover.py:107-108",
                   "src/utils/utils.py:134-137", "src/utils/commitments.py:13"):
        assert needle in text


def test_no_cpu_fallback(native):
    import torch
    lib = native.load()
    assert lib.bpmi_version() >= 100
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.bpmi_device_count() == 0
    from bulletproofs_amd.engine import Engine, EngineError
    with pytest.raises(EngineError, match="no HIP device"):
        Engine()
    # the reference-shaped call surface must not silently compute on the CPU either
    from bulletproofs_amd.pippenger import PipSECP256k1
    from bulletproofs_amd.ec import secp256k1
    with pytest.raises(EngineError):
        PipSECP256k1.multiexp([secp256k1.G], [3])


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "python-bulletproofs_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "bp_oracle" not in src, f


def test_header_is_plain_c99(tmp_path):
    """include/bpmi.h is the drop-in boundary: it must compile as C (no C++ or torch types) and a C
    program must link against libbpmi.so through it (no GPU needed for either)."""
    import subprocess
    src = tmp_path / "use_abi.c"
    src.write_text('#include "bpmi.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%d %d\\n", bpmi_version(), BPMI_NSTAGES); return bpmi_last_error(0) ? 0 : 1; }\n')
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(repo, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", str(src)])
    lib = os.path.join(repo, "python-bulletproofs_amd", "libbpmi.so")
    if os.path.exists(lib):
        exe = tmp_path / "use_abi"
        subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), lib, "-Wl,-rpath," + os.path.dirname(lib),
                               "-Wl,--allow-shlib-undefined"])


def test_packed_points_and_bulk_hashes_equal_the_plain_forms():
    """Host-layer shortcuts of the range-proof prover: PackedPoints carries exactly pack_points' bytes (also through join), and
    the inlined bulk mod_hash gives mod_hash's values."""
    import hashlib
    from bulletproofs_amd.ec import PackedPoints, Point, pack_points, secp256k1
    from bulletproofs_amd.rangeproofs.common import _mod_hash_ints
    from bulletproofs_amd.utils.utils import mod_hash
    pts = [Point._raw(secp256k1.gx + i, secp256k1.gy + 7 * i) for i in range(9)] + [Point.IDENTITY_ELEMENT]      # wire form only: any coordinates
    plain = b"".join(p.to_le64() for p in pts)
    assert pack_points(pts) == plain and pack_points(PackedPoints(pts)) == plain
    joined = PackedPoints.join(PackedPoints(pts[:4]), pts[4:7], pts[7:])
    assert list(joined) == pts and joined.packed == plain and pack_points(joined) == plain
    q = secp256k1.q
    digest = hashlib.sha256(b"transcript").digest() + b"&"
    assert _mod_hash_ints(3, 40, digest, q) == [mod_hash(str(i).encode() + digest, q).x for i in range(3, 40)]
