#!/usr/bin/env python3
"""Generates python-bulletproofs_amd/csrc/field_gen.hpp: the DEVICE bodies of the field-multiplication family
of csrc/field.hpp (fe_mul, fe_sqr, fe_mul_add, fe_sqr_add, fe_mul2) for gfx950 -- and csrc/scalar_gen.hpp: the device
body of sq_mul (csrc/scalar.hpp, arithmetic mod the group order q on 9 x 29-bit limbs; see sq_mul_body below).

    python tools/gen_field_asm.py            # rewrites csrc/field_gen.hpp
    python tools/gen_field_asm.py --check    # exit 1 if the committed file differs (tests/test_csrc_host.py runs this)

Why generated inline asm and not C: every column of a 9 x 9 limb product is `carry + sum of products`.  Written
in C the compiler builds the product sums as independent chains and adds the carry with a separate 64-bit
addition, and it spends an `and` + 64-bit shift + zero-extending moves per column on the bookkeeping.  Here
each column is ONE asm statement that chains v_mad_u64_u32 through the running 64-bit accumulator (there is
no builtin for that instruction); the carry-out operand goes to a dummy SGPR pair, so VCC is never touched.

The arithmetic is exactly fe_mac_c (field.hpp) -- same columns, same fold, same loose output -- so host unit
tests of the C body and a GPU test that compares both bit for bit pin the asm:

  columns 9..16 (high half)   s = 8 * hi32(previous s) + products;  limb th = lo32(s) is kept DIRTY (32 bits):
                              it only feeds the fold multiply-adds, and hi32(s) is a register, so the high
                              half needs no `and` and no shift at all (one extra multiply-add by 8 instead)
  column 8                    raw sum + fold terms; everything above bit 24 (w, < 2^40) is folded into columns
                              0..2 as w * (2^32 + 977) BEFORE the low chain runs, so no second carry pass
  columns 0..7                s = carry + products + addend + fold terms;  limb = s & M29, carry = s >> 29
  end                         column 8 keeps 24 bits; the < 2^12 that the last carry pushes above them goes
                              to limbs 0 and 1 uncarried ("loose" output, see field.hpp)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "python-bulletproofs_amd", "csrc", "field_gen.hpp")
MAX_PAIRS = 14          # an asm statement takes at most 30 operands: accumulator, carry sink, 14 operand pairs


def products(k, kind, A, B):
    """operand pairs (x, y) of column k of one limb product.  kind 'mul': A[i] * B[k-i];  'sqr': A squared, with the
    doubled limbs d<A>[i] = 2 A[i] so that a symmetric pair costs one multiply-add;  'sqr3': 3 A^2 (d = 6 A, diagonal 3 A)."""
    out = []
    lo, hi = max(0, k - 8), min(k, 8)
    if kind == "mul":
        return [("%s.v[%d]" % (A, i), "%s.v[%d]" % (B, k - i)) for i in range(lo, hi + 1)]
    for i in range(lo, hi + 1):
        j = k - i
        if i < j:
            out.append(("d%s[%d]" % (A, i), "%s.v[%d]" % (A, j)))
        elif i == j:
            out.append((("%s.v[%d]" if kind == "sqr" else "t%s[%d]") % (A, i), "%s.v[%d]" % (A, i)))      # sqr3: (3 a_i) a_i
    return out


class Stmt:
    """One column: a list of multiply-add terms (x, y, kind) with kind in 'vv' (two VGPRs), 'vs' (VGPR x SGPR constant),
    'v8' / 'v1' (VGPR x inline constant 8 / 1)."""

    def __init__(self, fresh):
        self.fresh = fresh          # True: the accumulator starts at 0 (first multiply-add takes the literal 0)
        self.terms = []

    def add(self, x, y, kind="vv"):
        self.terms.append((x, y, kind))

    def emit(self, indent="  "):
        chunks = [self.terms[i:i + MAX_PAIRS] for i in range(0, len(self.terms), MAX_PAIRS)]
        text = ""
        for ci, chunk in enumerate(chunks):
            fresh = self.fresh and ci == 0
            ops, ins, n = [], [], 2
            for ti, (x, y, kind) in enumerate(chunk):
                acc = "0" if (fresh and ti == 0) else "%0"
                if kind in ("v8", "v1"):
                    ops.append("v_mad_u64_u32 %%0, %%1, %%%d, %s, %s" % (n, kind[1], acc))
                    ins.append('"v"(%s)' % x)
                    n += 1
                else:
                    ops.append("v_mad_u64_u32 %%0, %%1, %%%d, %%%d, %s" % (n, n + 1, acc))
                    ins.append('"v"(%s)' % x)
                    ins.append(('"s"(%s)' if kind == "vs" else '"v"(%s)') % y)
                    n += 2
            out = '"=&v"(s)' if fresh else '"+v"(s)'
            text += '%sasm("%s"\n%s    : %s, "=&s"(sink_) : %s);\n' % (indent, "\\n\\t".join(ops), indent, out, ", ".join(ins))
        return text


def body(prods, addend, add_kind="v1"):
    """prods: list of (kind, A, B).  addend: name of an fe whose limbs (times 1, or 8 with add_kind "v8") are added to
    columns 0..8, or None."""
    L = []
    L.append("  const u32 k31264 = 31264u, k256 = 256u, k65536 = 65536u, kf = 31264u * 256u, k977 = 977u;\n")
    L.append("  u64 s, sink_;\n  u32 th[8], t[8], hp;\n")
    for kind, A, B in prods:
        if kind == "sqr":
            L.append("  u32 d%s[9];\n#pragma unroll\n  for (int i = 0; i < 9; i++) d%s[i] = %s.v[i] << 1;\n" % (A, A, A))
        if kind == "sqr3":
            L.append("  u32 d%s[9], t%s[9];\n#pragma unroll\n  for (int i = 0; i < 9; i++) { t%s[i] = %s.v[i] * 3u; d%s[i] = t%s[i] << 1; }\n" % (A, A, A, A, A, A))
    # ---- high half: columns 9..16, dirty 32-bit limbs
    for k in range(9, 17):
        st = Stmt(True)
        if k > 9:
            st.add("hp", None, "v8")
        for kind, A, B in prods:
            for x, y in products(k, kind, A, B):
                st.add(x, y)
        L.append(st.emit())
        L.append("  th[%d] = (u32)s; hp = (u32)(s >> 32);\n" % (k - 9) if k < 16 else "  th[7] = (u32)s; const u32 t17 = (u32)(s >> 32) << 3;\n")
    # ---- column 8, raw
    st = Stmt(True)
    st.add("t17", "k31264", "vs")
    st.add("th[7]", "k256", "vs")
    for kind, A, B in prods:
        for x, y in products(8, kind, A, B):
            st.add(x, y)
    if addend:
        st.add("%s.v[8]" % addend, None, add_kind)
    L.append(st.emit())
    L.append("  const u32 s8m = (u32)s & M24;\n  const u64 w = s >> 24;\n  const u32 wl = (u32)w & M29, wh = (u32)(w >> 29);\n")
    # ---- low chain: columns 0..7
    for k in range(8):
        st = Stmt(k == 0)
        if k == 0:
            st.add("t17", "kf", "vs")
            st.add("wl", "k977", "vs")
        if k == 1:
            st.add("t17", "k65536", "vs")
            st.add("wl", None, "v8")
            st.add("wh", "k977", "vs")
        if k == 2:
            st.add("wh", None, "v8")
        st.add("th[%d]" % k, "k31264", "vs")
        if k >= 1:
            st.add("th[%d]" % (k - 1), "k256", "vs")
        for kind, A, B in prods:
            for x, y in products(k, kind, A, B):
                st.add(x, y)
        if addend:
            st.add("%s.v[%d]" % (addend, k), None, add_kind)
        L.append(st.emit())
        L.append("  t[%d] = (u32)s & M29; s >>= 29;\n" % k)
    # ---- end: column 8 = its 24 kept bits + the last carry; the overflow goes to limbs 0 and 1, uncarried
    st = Stmt(False)
    st.add("s8m", None, "v1")
    L.append(st.emit())
    L.append("  const u32 v2 = (u32)(s >> 24);\n")
    L.append("  r.v[0] = t[0] + v2 * 977u; r.v[1] = t[1] + (v2 << 3);\n")
    L.append("#pragma unroll\n  for (int k = 2; k < 8; k++) r.v[k] = t[k];\n")
    L.append("  r.v[8] = (u32)s & M24;\n  (void)sink_;\n")
    return "".join(L)


FUNCS = [
    ("fe_mul_dev", "fe &r, const fe &a, const fe &b", [("mul", "a", "b")], None, "v1"),
    ("fe_sqr_dev", "fe &r, const fe &a", [("sqr", "a", None)], None, "v1"),
    ("fe_sqr3_dev", "fe &r, const fe &a", [("sqr3", "a", None)], None, "v1"),
    ("fe_mul_add_dev", "fe &r, const fe &a, const fe &b, const fe &add", [("mul", "a", "b")], "add", "v1"),
    ("fe_mul_add8_dev", "fe &r, const fe &a, const fe &b, const fe &add", [("mul", "a", "b")], "add", "v8"),
    ("fe_sqr_add_dev", "fe &r, const fe &a, const fe &add", [("sqr", "a", None)], "add", "v1"),
    ("fe_mul2_dev", "fe &r, const fe &a, const fe &b, const fe &c, const fe &d", [("mul", "a", "b"), ("mul", "c", "d")], None, "v1"),
]


# ---- arithmetic mod q on 9 x 29-bit limbs (csrc/scalar.hpp "sq"): the device body of sq_mul, statement for statement sq_mul_c ----
OUT_SC = os.path.join(HERE, "..", "python-bulletproofs_amd", "csrc", "scalar_gen.hpp")


def sq_mul_body():
    L = []
    L.append("  const u32 kd[5] = BPMI_SQ_D;\n  u64 s, sink_;\n  u32 h[9], lo[9], g[5], u[9];\n")
    # columns 9..16 of a * b, carried: h[0..9)
    for k in range(9, 17):
        st = Stmt(k == 9)
        for i in range(k - 8, 9):
            st.add("a.v[%d]" % i, "b.v[%d]" % (k - i))
        L.append(st.emit())
        L.append("  h[%d] = (u32)s & M29; s >>= 29;\n" % (k - 9))
    L.append("  h[8] = (u32)s;\n")
    # columns 0..12 of lo(a * b) + h * D, carried: lo[0..9), g[0..5)
    for k in range(13):
        st = Stmt(k == 0)
        if k <= 8:
            for i in range(k + 1):
                st.add("a.v[%d]" % i, "b.v[%d]" % (k - i))
        for j in range(5):
            i = k - j
            if 0 <= i <= 8:
                st.add("h[%d]" % i, "kd[%d]" % j, "vs")
        L.append(st.emit())
        L.append("  %s = (u32)s & M29; s >>= 29;\n" % ("lo[%d]" % k if k <= 8 else "g[%d]" % (k - 9)))
    L.append("  g[4] = (u32)s;\n")
    # columns 0..8 of lo + g * D, carried: u[0..9), e
    for k in range(9):
        st = Stmt(k == 0)
        st.add("lo[%d]" % k, None, "v1")
        for j in range(5):
            i = k - j
            if 0 <= i <= 4:
                st.add("g[%d]" % i, "kd[%d]" % j, "vs")
        L.append(st.emit())
        L.append("  u[%d] = (u32)s & M29; s >>= 29;\n" % k)
    L.append("  const u32 e = (u32)s;\n")
    # e * D back into limbs 0..4; the second pass stops at limb 5 (loose output)
    for k in range(5):
        st = Stmt(k == 0)
        st.add("u[%d]" % k, None, "v1")
        st.add("e", "kd[%d]" % k, "vs")
        L.append(st.emit())
        L.append("  r.v[%d] = (u32)s & M29; s >>= 29;\n" % k)
    L.append("  r.v[5] = u[5] + (u32)s;\n#pragma unroll\n  for (int k = 6; k < 9; k++) r.v[k] = u[k];\n  (void)sink_;\n")
    return "".join(L)


def generate_sc():
    out = ["// scalar_gen.hpp -- GENERATED by tools/gen_field_asm.py; do not edit (tests/test_csrc_host.py checks it is current).\n",
           "// Device body of sq_mul (arithmetic mod q on 9 x 29-bit limbs); the arithmetic is sq_mul_c of scalar.hpp, column for column.\n",
           "#pragma once\n", "#if defined(__HIP_DEVICE_COMPILE__)\n",
           "__device__ __forceinline__ void sq_mul_dev(sq &r, const sq &a, const sq &b) {\n%s}\n" % sq_mul_body(), "#endif\n"]
    return "".join(out)


def generate():
    out = ["// field_gen.hpp -- GENERATED by tools/gen_field_asm.py; do not edit (tests/test_csrc_host.py checks it is current).\n",
           "// Device bodies of the field-multiplication family; the arithmetic is fe_mac_c of field.hpp, column for column.\n",
           "#pragma once\n", "#if defined(__HIP_DEVICE_COMPILE__)\n"]
    for name, sig, prods, addend, add_kind in FUNCS:
        out.append("__device__ __forceinline__ void %s(%s) {\n%s}\n" % (name, sig, body(prods, addend, add_kind)))
    out.append("#endif\n")
    return "".join(out)


if __name__ == "__main__":
    files = [(OUT, generate()), (OUT_SC, generate_sc())]
    if "--check" in sys.argv:
        sys.exit(0 if all(os.path.exists(path) and open(path).read() == text for path, text in files) else 1)
    for path, text in files:
        with open(path, "w") as f:
            f.write(text)
        print(os.path.normpath(path))
