"""The native O(n m) scalar algebra of the range-proof provers / verifiers (csrc/rp_algebra_host.hpp, host code of libbpmi)
against the formulas of the reference written on Python integers (src/rangeproofs/rangeproof_aggreg_prover.py:117-146,
rangeproof_prover.py:93-112, rangeproof_aggreg_verifier.py:96-108).  No GPU involved."""
import ctypes
import random

import pytest

import bulletproofs_amd  # noqa: F401
from bulletproofs_amd import _native
from bulletproofs_amd.ec import PackedScalars, pack_scalars, secp256k1

Q = secp256k1.q


def le(v):
    return (v % Q).to_bytes(32, "little")


def unle(raw):
    return [int.from_bytes(raw[i: i + 32], "little") for i in range(0, len(raw), 32)]


@pytest.mark.parametrize("n,m,agg,threads", [(2, 1, False, 1), (8, 1, False, 3), (64, 1, False, 1), (16, 4, True, 2), (4, 32, True, 5),
                                             (64, 128, True, 8), (64, 128, True, 1), (1, 8, True, 2)])
def test_native_algebra_equals_the_reference_formulas(n, m, agg, threads):
    lib = _native.load()
    rnd = random.Random(1000 * n + m)
    nm = n * m
    aL = [rnd.randrange(2) for _ in range(nm)]
    aR = [(b - 1) % Q for b in aL]
    sL = [rnd.randrange(Q) for _ in range(nm)]
    sR = [rnd.randrange(Q) for _ in range(nm)]
    y, z, x = (rnd.randrange(1, Q) for _ in range(3))
    ypow = [pow(y, i, Q) for i in range(nm)]
    zt = [pow(z, 2 + i // n, Q) * pow(2, i % n, Q) % Q if agg else z * z * pow(2, i, Q) % Q for i in range(nm)]
    ysr = [ypow[i] * sR[i] % Q for i in range(nm)]
    t1 = (sum(sL[i] * (ypow[i] * (aR[i] + z) + zt[i]) for i in range(nm)) + sum((aL[i] - z) * ysr[i] for i in range(nm))) % Q
    t2 = sum(sL[i] * ysr[i] for i in range(nm)) % Q
    ls = [(aL[i] - z + sL[i] * x) % Q for i in range(nm)]
    rs = [(ypow[i] * (aR[i] + z + sR[i] * x) + zt[i]) % Q for i in range(nm)]
    t_hat = sum(a * b for a, b in zip(ls, rs)) % Q
    yinv = pow(y, -1, Q)
    yscale = [pow(yinv, i, Q) for i in range(nm)]
    hsc = [(z * ypow[i] + zt[i]) * yscale[i] % Q for i in range(nm)]
    bits, sLb, sRb = bytes(aL), b"".join(map(le, sL)), b"".join(map(le, sR))
    o1, o2 = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
    assert lib.bpmi_rp_poly_coeffs(n, m, int(agg), bits, sLb, sRb, le(y), le(z), threads, o1, o2) == 0
    assert (int.from_bytes(o1.raw, "little"), int.from_bytes(o2.raw, "little")) == (t1, t2)
    bl, br, bh, by = (ctypes.create_string_buffer(32 * nm) for _ in range(4))
    th = ctypes.create_string_buffer(32)
    assert lib.bpmi_rp_final_vectors(n, m, int(agg), bits, sLb, sRb, le(y), le(z), le(x), threads, bl, br, th, bh, by) == 0
    assert unle(bl.raw) == ls and unle(br.raw) == rs and int.from_bytes(th.raw, "little") == t_hat
    assert unle(bh.raw) == hsc and unle(by.raw) == yscale
    bh2, by2, ys = ctypes.create_string_buffer(32 * nm), ctypes.create_string_buffer(32 * nm), ctypes.create_string_buffer(32)
    assert lib.bpmi_rp_verifier_vectors(n, m, int(agg), le(y), le(z), threads, bh2, by2, ys) == 0
    assert bh2.raw == bh.raw and by2.raw == by.raw and int.from_bytes(ys.raw, "little") == sum(ypow) % Q
    # y = 0 has no inverse: an argument error, not a crash
    assert lib.bpmi_rp_verifier_vectors(n, m, int(agg), le(0), le(z), threads, bh2, by2, ys) != 0


def test_packed_scalars_are_lazy_and_still_a_list():
    vals = [5, Q - 1, 0, 12345678901234567890]
    raw = b"".join(map(le, vals))
    ps = PackedScalars.from_bytes(raw)
    assert len(ps) == 4 and list.__len__(ps) == 0            # nothing materialised yet
    assert pack_scalars(ps) == raw
    j = PackedScalars.join(ps, [7], ps)
    assert len(j) == 9 and pack_scalars(j) == raw + le(7) + raw
    assert ps[1] == Q - 1 and list(ps) == vals and [v for v in j] == vals + [7] + vals
    assert len(PackedScalars.from_bytes(b"")) == 0
