// curve.hpp -- secp256k1 group law (y^2 = x^3 + 7) for the MSM / IPA kernels.
// Replaces fastecdsa's `Point.__add__` / `Point.__rmul__` as used by the reference
// (/root/reference/src/pippenger/group.py:31-32, src/utils/utils.py:43-44).
//
// Working form is XYZZ ("extended Jacobian": x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2),
// which makes the bucket update acc += affine cost 8M + 2S with no inversion.
// The identity is ZZ == 0.  All formulas are COMPLETE: acc = identity, P + P and
// P + (-P) are detected and handled, because real Bulletproofs inputs hit them
// (duplicate generators, scalars in {0, 1, q-1}; SURVEY.md section 7 "hard parts").
// Results are only ever compared / exported after to_affine + fe_canon, so the
// (non-unique) projective representative never leaks.
#pragma once
#include "field.hpp"

namespace bpmi {

struct affine { fe x, y; };              // tight limbs; identity = (0, 0)
struct xyzz { fe X, Y, ZZ, ZZZ; };       // tight limbs; identity = ZZ == 0

BPMI_HD void xyzz_set_inf(xyzz &r) {
  fe_set_zero(r.X); fe_set_zero(r.Y); fe_set_zero(r.ZZ); fe_set_zero(r.ZZZ);
}
// a necessary condition for a loose / tight value to be 0 (mod p): its limb 4 is 0 or 2^29 - 1 (the limbs of 0 and p).
// One compare pair instead of the 27-operation exact test; the exact test runs only behind it (probability 2^-28).
BPMI_HD bool fe_maybe_zero(const fe &a) { return (a.v[4] == 0u) | (a.v[4] == M29); }
BPMI_HD bool xyzz_is_inf(const xyzz &a) { return fe_is_zero_tight(a.ZZ); }
// the same for the accumulation loops, where the answer is almost always "no": ZZ is a multiplication result
// (never 0 mod p for a finite point) or the exact zeros of xyzz_set_inf
BPMI_HD bool xyzz_is_inf_fast(const xyzz &a) { return fe_maybe_zero(a.ZZ) && fe_is_zero_tight(a.ZZ); }
BPMI_HD bool affine_is_inf(const affine &a) {
  u32 z = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) z |= a.x.v[k] | a.y.v[k];
  return z == 0;
}
BPMI_HD void xyzz_from_affine(xyzz &r, const affine &a) {
  if (affine_is_inf(a)) { xyzz_set_inf(r); return; }
  r.X = a.x; r.Y = a.y; fe_set_one(r.ZZ); fe_set_one(r.ZZZ);
}

// 64-byte wire form (x || y, 32-byte little-endian each = 16 u32 words) <-> affine
// beta: the cube root of unity mod p with lambda (x, y) = (beta x, y) (scalar.hpp, GLV); limbs of
// 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE
#define BPMI_FE_BETA {0x119501EEu, 0x09CB6143u, 0x1D626570u, 0x0092EA25u, 0x034E99CFu, 0x03CF561Au, 0x1C41B991u, 0x056CAF80u, 0x007AE96Au}
BPMI_HD void fe_mul_beta(fe &r, const fe &a) {
  const fe beta = {BPMI_FE_BETA};
  fe_mul(r, a, beta);
}
BPMI_HD void affine_from_words(affine &r, const u32 w[16]) {
  fe_from_words(r.x, w);
  fe_from_words(r.y, w + 8);
}
BPMI_HD void affine_to_words(u32 w[16], const affine &a) {
  fe cx, cy;
  fe_canon(cx, a.x); fe_canon(cy, a.y);
  fe_to_words(w, cx); fe_to_words(w + 8, cy);
}
BPMI_HD void affine_neg(affine &r, const affine &a) {
  r.x = a.x;
  if (affine_is_inf(a)) { r.y = a.y; return; }
  fe t; fe_neg(t, a.y); fe_carry(r.y, t);
}

// r = 2 * (x, y) for an affine point (mdbl-2008-s-1): 2M + 3S + the Y3 pair, six reductions; y tight or loose
BPMI_HD void xyzz_dbl_affine(xyzz &r, const fe &x, const fe &y) {
  fe U, V, W, S, M, t, nW;
  fe_add(U, y, y);                       // magnitude 2
  fe_sqr(V, U);                          // V = 4y^2
  fe_mul(W, U, V);                       // W = 8y^3
  fe_mul(S, x, V);
  fe_sqr3(M, x);                         // M = 3x^2
  fe_bias4_sub_2a(t, S);
  fe_sqr_add(r.X, M, t);                 // X3 = M^2 - 2S
  fe_sub(t, S, r.X);                     // S - X3 + 2p, magnitude 3
  fe_neg(nW, W);
  fe_mul2(r.Y, M, t, nW, y);             // Y3 = M (S - X3) - W y
  r.ZZ = V; r.ZZZ = W;
}

// r = 2 * a (dbl-2008-s-1, a = 0): 6M + 3S in eight reductions.  2-torsion does not exist on
// secp256k1 (prime order), so Y == 0 only for the identity.
BPMI_HD void xyzz_dbl(xyzz &r, const xyzz &a) {
  if (xyzz_is_inf(a)) { r = a; return; }      // (every identity is the all-zero record of xyzz_set_inf; a copy, not 36 stores: with r == a nothing at all)
  fe U, V, W, S, M, t, nW, X3, Y3;
  fe_add(U, a.Y, a.Y);
  fe_sqr(V, U);
  fe_mul(W, U, V);
  fe_mul(S, a.X, V);
  fe_sqr3(M, a.X);
  fe_bias4_sub_2a(t, S);
  fe_sqr_add(X3, M, t);
  fe_sub(t, S, X3);
  fe_neg(nW, W);
  fe_mul2(Y3, M, t, nW, a.Y);
  fe_mul(r.ZZ, V, a.ZZ);
  fe_mul(r.ZZZ, W, a.ZZZ);
  r.X = X3; r.Y = Y3;
}

// acc += (x2, y2), affine addend that is NOT the identity (madd-2008-s): 8M + 2S in NINE reductions.
// y2 may be LAZY (magnitude <= 2, e.g. 2p - y): it only feeds a multiplication on the main path; the two rare
// branches that store or double it carry it first.  The subtractions ride in the reductions:
//   P  = x2 ZZ  + (2p - X)                 fe_mul_add        R  = y2 ZZZ + (2p - Y)         fe_mul_add
//   X3 = R^2 + (8p - PPP - 2Q)             fe_sqr_add        Y3 = R (Q - X3 + 2p) + (2p - Y) PPP   fe_mul2
// the main path in two pieces, so that the exceptional cases can be tested between them (and so that
// tools/isa_counts.py can compile exactly this code on its own)
BPMI_HD void xyzz_madd_pr(fe &P, fe &R, fe &nY, const xyzz &acc, const fe &x2, const fe &y2) {
  fe nX;
  fe_neg(nX, acc.X);                                      // 2p - X: a column addend
  fe_neg(nY, acc.Y);                                      // 2p - Y: addend now, magnitude-2 factor of Y3 later
  fe_mul_add(P, x2, acc.ZZ, nX);                          // U2 - X1
  fe_mul_add(R, y2, acc.ZZZ, nY);                         // S2 - Y1
}
BPMI_HD void xyzz_madd_finish(xyzz &acc, const fe &P, const fe &R, const fe &nY) {
  fe PP, PPP, Q, t;
  fe_sqr(PP, P);
  fe_mul(PPP, P, PP);
  fe_mul(Q, acc.X, PP);
  fe_bias8_sub_a_2b(t, PPP, Q);                           // 8p - PPP - 2Q
  fe_sqr_add(acc.X, R, t);                                // X3
  fe_sub(t, Q, acc.X);                                    // Q - X3 + 2p, magnitude 3
  fe_mul2(acc.Y, R, t, nY, PPP);                          // Y3 = R (Q - X3) - Y1 PPP      (3 + 2 <= 7)
  fe_mul(acc.ZZ, acc.ZZ, PP);
  fe_mul(acc.ZZZ, acc.ZZZ, PPP);
}
BPMI_HD void xyzz_madd(xyzz &acc, const fe &x2, const fe &y2) {
  if (xyzz_is_inf_fast(acc)) { acc.X = x2; fe_carry(acc.Y, y2); fe_set_one(acc.ZZ); fe_set_one(acc.ZZZ); return; }
  fe P, R, nY;
  xyzz_madd_pr(P, R, nY, acc, x2, y2);
  if (fe_maybe_zero(P)) {
    if (fe_is_zero_tight(P)) {
      fe t;
      if (fe_is_zero_tight(R)) { fe_carry(t, y2); xyzz_dbl_affine(acc, x2, t); return; }   // acc == addend
      xyzz_set_inf(acc); return;                                                             // acc == -addend
    }
  }
  xyzz_madd_finish(acc, P, R, nY);
}
// acc += P for an affine point that may be the identity, optionally negated
// The sign is data (random per entry), so it must be a SELECT, not a branch: a branch
// makes every wave execute two copies of the 8M+2S add with half the lanes masked.
BPMI_HD void xyzz_madd_signed(xyzz &acc, const affine &P, bool negate) {
  if (affine_is_inf(P)) return;
  fe ny;
  fe_neg(ny, P.y);                        // 2p - y, lazy (magnitude 2)
#pragma unroll
  for (int k = 0; k < 9; k++) ny.v[k] = negate ? ny.v[k] : P.y.v[k];
  xyzz_madd(acc, P.x, ny);
}

// r = a + b, both XYZZ (add-2008-s): 12M + 2S in thirteen reductions (the subtractions ride in them)
BPMI_HD void xyzz_add(xyzz &r, const xyzz &a, const xyzz &b) {
  if (xyzz_is_inf(a)) { r = b; return; }
  if (xyzz_is_inf(b)) { r = a; return; }
  fe U1, S1, nU1, nS1, P, R, PP, PPP, Q, t, X3, Y3, zz;
  fe_mul(U1, a.X, b.ZZ);
  fe_mul(S1, a.Y, b.ZZZ);
  fe_neg(nU1, U1);
  fe_neg(nS1, S1);
  fe_mul_add(P, b.X, a.ZZ, nU1);         // U2 - U1
  fe_mul_add(R, b.Y, a.ZZZ, nS1);        // S2 - S1
  if (fe_is_zero_tight(P)) {
    if (fe_is_zero_tight(R)) { xyzz_dbl(r, a); return; }
    xyzz_set_inf(r); return;
  }
  fe_sqr(PP, P);
  fe_mul(PPP, P, PP);
  fe_mul(Q, U1, PP);
  fe_bias8_sub_a_2b(t, PPP, Q);
  fe_sqr_add(X3, R, t);                  // R^2 - PPP - 2Q
  fe_sub(t, Q, X3);
  fe_mul2(Y3, R, t, nS1, PPP);           // R (Q - X3) - S1 PPP
  fe_mul(zz, a.ZZ, b.ZZ); fe_mul(r.ZZ, zz, PP);
  fe_mul(zz, a.ZZZ, b.ZZZ); fe_mul(r.ZZZ, zz, PPP);
  r.X = X3; r.Y = Y3;
}

BPMI_HD void xyzz_neg(xyzz &r, const xyzz &a) {
  r = a;
  if (xyzz_is_inf(a)) return;
  fe t; fe_neg(t, a.Y); fe_carry(r.Y, t);
}

// canonical affine form (one field inversion); identity -> (0, 0)
BPMI_HD void xyzz_to_affine(affine &r, const xyzz &a) {
  if (xyzz_is_inf(a)) { fe_set_zero(r.x); fe_set_zero(r.y); return; }
  fe zz_zzz, inv, izz, izzz;
  fe_mul(zz_zzz, a.ZZ, a.ZZZ);
  fe_inv(inv, zz_zzz);
  fe_mul(izz, inv, a.ZZZ);               // 1/ZZ
  fe_mul(izzz, inv, a.ZZ);               // 1/ZZZ
  fe_mul(r.x, a.X, izz);
  fe_mul(r.y, a.Y, izzz);
  fe_canon(r.x, r.x); fe_canon(r.y, r.y);
}

// ---- Jacobian coordinates (x = X/Z^2, y = Y/Z^3; identity = Z == 0) ------------------
// Used by the doubling-heavy ladders (generator fold, batch scalar multiplication):
// doubling is 2M + 5S here against 6M + 3S in XYZZ.
struct jac { fe X, Y, Z; };              // tight limbs

BPMI_HD void jac_set_inf(jac &r) { fe_set_zero(r.X); fe_set_one(r.Y); fe_set_zero(r.Z); }
BPMI_HD bool jac_is_inf(const jac &a) { return fe_is_zero_tight(a.Z); }

// r = 2a (dbl-2009-l, curve a = 0): 2M + 5S in seven reductions and one carry.  Z = 0 maps to Z = 0, so the
// identity needs no branch; Y = 0 cannot occur (no 2-torsion on a prime-order curve).
BPMI_HD void jac_dbl(jac &r, const jac &a) {
  fe A, B, C, Dh, E, t, X3, Y3, Z3;
  fe_sqr(A, a.X);
  fe_sqr(B, a.Y);
  fe_add(t, a.Y, a.Y);
  fe_mul(Z3, t, a.Z);                    // Z3 = 2YZ
  fe_sqr(C, B);
  fe_add(t, a.X, B);                     // magnitude 2
  fe_bias4_sub_a_b(E, A, C);
  fe_sqr_add(Dh, t, E);                  // Dh = (X + B)^2 - A - C   (= D / 2)
  fe_mul_small(E, A, 3); fe_carry(E, E); // E = 3A, tight
  fe_bias8_sub_4a(t, Dh);
  fe_sqr_add(X3, E, t);                  // X3 = E^2 - 2D
  fe_dbl_sub(t, Dh, X3);                 // D - X3 + 2p, magnitude 4
  fe_neg(C, C);                          // 2p - C
  fe_mul_add8(Y3, E, t, C);              // Y3 = E (D - X3) - 8C
  r.X = X3; r.Y = Y3; r.Z = Z3;
}
// acc += (x2, y2), affine addend that is NOT the identity: 8M + 3S in ten reductions, complete.
// y2 may be lazy (magnitude <= 2).
BPMI_HD void jac_madd(jac &acc, const fe &x2, const fe &y2) {
  if (jac_is_inf(acc)) { acc.X = x2; fe_carry(acc.Y, y2); fe_set_one(acc.Z); return; }
  fe ZZ, nX, nY, H, R, HH, HHH, V, t;
  fe_sqr(ZZ, acc.Z);                     // Z1Z1
  fe_neg(nX, acc.X);
  fe_neg(nY, acc.Y);
  fe_mul_add(H, x2, ZZ, nX);             // H = U2 - X1
  fe_mul(t, y2, acc.Z);
  fe_mul_add(R, t, ZZ, nY);              // R = S2 - Y1
  if (fe_is_zero_tight(H)) {
    if (fe_is_zero_tight(R)) {           // acc == addend: double the affine point
      jac d; d.X = x2; fe_carry(d.Y, y2); fe_set_one(d.Z);
      jac_dbl(acc, d);
      return;
    }
    jac_set_inf(acc); return;            // acc == -addend
  }
  fe_sqr(HH, H);
  fe_mul(acc.Z, acc.Z, H);               // Z3
  fe_mul(HHH, H, HH);
  fe_mul(V, acc.X, HH);
  fe_bias8_sub_a_2b(t, HHH, V);
  fe_sqr_add(acc.X, R, t);               // X3 = R^2 - HHH - 2V
  fe_sub(t, V, acc.X);
  fe_mul2(acc.Y, R, t, nY, HHH);         // Y3 = R (V - X3) - Y1 HHH
}
// y-sign as a select (see xyzz_madd_signed); the negated y stays lazy
BPMI_HD void jac_madd_signed(jac &acc, const fe &x2, const fe &y2, bool negate) {
  fe ny;
  fe_neg(ny, y2);
#pragma unroll
  for (int k = 0; k < 9; k++) ny.v[k] = negate ? ny.v[k] : y2.v[k];
  jac_madd(acc, x2, ny);
}
BPMI_HD void jac_to_affine(affine &r, const jac &a) {
  if (jac_is_inf(a)) { fe_set_zero(r.x); fe_set_zero(r.y); return; }
  fe zi, zi2, zi3;
  fe_inv(zi, a.Z);
  fe_sqr(zi2, zi);
  fe_mul(zi3, zi2, zi);
  fe_mul(r.x, a.X, zi2);
  fe_mul(r.y, a.Y, zi3);
  fe_canon(r.x, r.x); fe_canon(r.y, r.y);
}

// XYZZ record in memory: 4 x 9 u32 (144 B), limb form as is
BPMI_HD void xyzz_store(u32 *dst, const xyzz &a) {
#pragma unroll
  for (int k = 0; k < 9; k++) { dst[k] = a.X.v[k]; dst[9 + k] = a.Y.v[k]; dst[18 + k] = a.ZZ.v[k]; dst[27 + k] = a.ZZZ.v[k]; }
}
BPMI_HD void xyzz_load(xyzz &a, const u32 *src) {
#pragma unroll
  for (int k = 0; k < 9; k++) { a.X.v[k] = src[k]; a.Y.v[k] = src[9 + k]; a.ZZ.v[k] = src[18 + k]; a.ZZZ.v[k] = src[27 + k]; }
}

// bit offsets of the (<= 4) partial sums a bucket reduction leaves per MSM window:
// window value = sum_v 2^(off[v]) E[v], off ascending, off[0] = 0
// Is the 64-byte wire point (x, y: 8 little-endian words each) the identity (all zero) or a point of y^2 = x^3 + 7 with x, y < p?
// What fastecdsa's Point constructor checks for the reference (reached from /root/reference/src/utils/utils.py:119-131); the C-ABI
// checks it for callers that hand in raw bytes (bpmi.hip validate_*).  Host and device.
BPMI_HD bool wire_point_valid(const u32 w[16]) {
  u32 any = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) any |= w[k];
  if (!any) return true;
  bool ok = true;
#pragma unroll
  for (int h = 0; h < 2; h++) {               // coordinate < p:  v + (2^32 + 977) must not carry out of 256 bits
    const u32 *v = w + 8 * h;
    u64 c = (u64)v[0] + 977u; c >>= 32;
    c += (u64)v[1] + 1u; c >>= 32;
#pragma unroll
    for (int k = 2; k < 8; k++) { c += v[k]; c >>= 32; }
    ok = ok && (c == 0);
  }
  fe x, y, a, t, seven;
  fe_from_words(x, w);
  fe_from_words(y, w + 8);
  fe_sqr(t, x); fe_mul(a, t, x);
  fe_set_zero(seven); seven.v[0] = 7;
  fe_add(a, a, seven); fe_carry(a, a);          // x^3 + 7
  fe_sqr(t, y);
  fe ca, ct;
  fe_canon(ca, a); fe_canon(ct, t);
  return ok && fe_equal(ca, ct);
}

// bit offsets of the (<= 4) partial sums the bucket reduction leaves per window; top = 1: the LAST window was split at top_off instead
// (MsmGeom.top2).  The round-5 fields default to "none", so code that fills nv / off only stays right.
struct TailOffs { u32 nv; u32 off[4]; u32 top = 0; u32 top_off[4] = {0, 0, 0, 0}; };      // top: number of WIDE (c + 1 bit) windows at the top, split at top_off

}  // namespace bpmi
