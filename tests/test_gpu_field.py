"""The DEVICE bodies of the field-multiplication family (csrc/field_gen.hpp: generated v_mad_u64_u32 chains) against
the HOST build of the same header (the C body fe_mac_c, itself checked against Python integers in
tests/test_csrc_host.py): identical limbs on adversarial limb patterns -- every variant at the largest
magnitudes it allows, the largest addends, loose and tight maxima, zero / one / p -- and on random lazy values.
This is the arithmetic under every `Point + Point` of the reference (/root/reference/src/pippenger/group.py:31-32)."""
import ctypes
import random

import pytest

from test_csrc_host import P, assert_loose, fe_raw, limbs_value, mul_family_cases, mul_family_expected, shim  # noqa: F401

pytestmark = pytest.mark.gpu


def test_device_multiplication_family_equals_host_limb_for_limb(shim):  # noqa: F811
    import gpu_common
    eng = gpu_common.engine()
    shim.t_fe_raw.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_uint32)] * 5
    rnd = random.Random(2026)
    cases = mul_family_cases(rnd, 60000)
    z = [0] * 9
    by_op = {}
    for case in cases:
        by_op.setdefault(case[0], []).append(case)
    total = 0
    for op, lst in sorted(by_op.items()):
        n = len(lst)
        flat = lambda idx: (ctypes.c_uint32 * (9 * n))(*[v for case in lst for v in (case[idx] if case[idx] is not None else z)])
        out = (ctypes.c_uint32 * (9 * n))()
        eng._ck(eng.lib.bpmi_debug_fe_op(eng.ctx, op, flat(1), flat(2), flat(3), flat(4), n, out))
        got = list(out)
        for i, (_, a, b, c, d) in enumerate(lst):
            dev = got[9 * i: 9 * i + 9]
            if i < 3000:                    # the host shim call is the slow part; every case is still checked against Python below
                assert dev == fe_raw(shim, op, a, b, c, d), (op, a, b, c, d)
            assert_loose(dev)
            assert limbs_value(dev) % P == mul_family_expected(op, a, b, c, d), (op, a, b, c, d)
        total += n
    assert total >= 60000
    # carry / canonical form of lazy and loose inputs
    pats = [[0xFFFFFFFF] * 9, [0] * 9, [0x1FFFFC2F, 0x1FFFFFF7] + [0x1FFFFFFF] * 6 + [0x00FFFFFF]] + \
           [[rnd.randrange(1 << 32) for _ in range(9)] for _ in range(500)]
    n = len(pats)
    arr = (ctypes.c_uint32 * (9 * n))(*[v for p in pats for v in p])
    for op in (5, 6):
        out = (ctypes.c_uint32 * (9 * n))()
        eng._ck(eng.lib.bpmi_debug_fe_op(eng.ctx, op, arr, arr, arr, arr, n, out))
        for i, p in enumerate(pats):
            assert list(out)[9 * i: 9 * i + 9] == fe_raw(shim, op, p)
