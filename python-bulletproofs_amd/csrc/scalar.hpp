// scalar.hpp -- arithmetic mod q (the secp256k1 group order) on 8 x u32 words,
// canonical values in [0, q).  Replaces the reference's `ModP` arithmetic where
// it runs in bulk inside the IPA loop (/root/reference/src/utils/utils.py:24-81,
// inner_product :134-137, the a/b fold src/innerproduct/inner_product_prover.py:109-110).
// Plain C++ (host + device); this is O(n) work next to the O(n) EC work, so it is
// written for clarity, not for the last cycle.
#pragma once
#include "field.hpp"

namespace bpmi {

struct sc { u32 v[8]; };

#define BPMI_SC_Q  {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}
// 2^256 - q (129 bits)
#define BPMI_SC_C  {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 0x00000001u}
// (q - 1) / 2
#define BPMI_SC_HALF {0x681B20A0u, 0xDFE92F46u, 0x57A4501Du, 0x5D576E73u, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x7FFFFFFFu}

// word-wise add / subtract with carry: on the device the clang builtins become one v_addc / v_subb each (the 64-bit
// formulation below costs ~5 instructions per word there)
BPMI_HD u32 sc_adc(u32 a, u32 b, u32 &c) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 co;
  const u32 r = __builtin_addc(a, b, c, &co);
  c = co;
  return r;
#else
  const u64 t = (u64)a + b + c;
  c = (u32)(t >> 32);
  return (u32)t;
#endif
}
BPMI_HD u32 sc_sbb(u32 a, u32 b, u32 &br) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 bo;
  const u32 r = __builtin_subc(a, b, br, &bo);
  br = bo;
  return r;
#else
  const u64 t = (u64)a - b - br;
  br = (u32)(t >> 32) & 1u;
  return (u32)t;
#endif
}
BPMI_HD bool sc_is_zero(const sc &a) {
  u32 z = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) z |= a.v[i];
  return z == 0;
}
// a > b ?
BPMI_HD bool words_gt(const u32 a[8], const u32 b[8]) {
  bool gt = false, decided = false;
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    if (!decided && a[i] != b[i]) { gt = a[i] > b[i]; decided = true; }
  }
  return gt;
}
BPMI_HD bool sc_is_high(const sc &a) {   // a > (q-1)/2
  const u32 half[8] = BPMI_SC_HALF;
  return words_gt(a.v, half);
}
// r = a - b over 2^256, returns borrow
BPMI_HD u32 words_sub(u32 r[8], const u32 a[8], const u32 b[8]) {
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r[i] = sc_sbb(a[i], b[i], br);
  return br;
}
// s in [0, 2^256) -> s mod q (one conditional subtraction: 2^256 < 2q).  The MSM / ladder kernels apply it
// to every scalar they load, so a C caller that hands in an unreduced scalar gets the reference's
// `e % order` (pippenger.py:26) instead of a wrapped borrow in the sign recoding.
BPMI_HD void sc_reduce_once(sc &s) {
  const u32 q[8] = BPMI_SC_Q;
  u32 t[8];
  const u32 br = words_sub(t, s.v, q);
#pragma unroll
  for (int i = 0; i < 8; i++) s.v[i] = br ? s.v[i] : t[i];
}
BPMI_HD void sc_neg(sc &r, const sc &a) {   // q - a (a != 0)
  const u32 q[8] = BPMI_SC_Q;
  if (sc_is_zero(a)) { r = a; return; }
  words_sub(r.v, q, a.v);
}
BPMI_HD void sc_cond_sub_q(u32 t[9]) {
  const u32 q[8] = BPMI_SC_Q;
  u32 s[8];
  const u32 br = words_sub(s, t, q);
  // t >= q  <=>  t[8] != 0 or no borrow
  const bool ge = (t[8] != 0) | (br == 0);
  if (ge) {
    t[8] -= br;
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = s[i];
  }
}
BPMI_HD void sc_add(sc &r, const sc &a, const sc &b) {
  u32 t[9], c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) t[i] = sc_adc(a.v[i], b.v[i], c);
  t[8] = c;
  sc_cond_sub_q(t);
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
}
// out[0..NO) = lo[0..8) + hi[0..NH) * C   (NO >= max(8, NH + 5) + 1)
template <int NH, int NO>
BPMI_HD void sc_fold_once(u32 out[NO], const u32 lo[8], const u32 hi[NH]) {
  const u32 C[5] = BPMI_SC_C;
#pragma unroll
  for (int i = 0; i < NO; i++) out[i] = i < 8 ? lo[i] : 0;
#pragma unroll
  for (int i = 0; i < NH; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      c += (u64)hi[i] * C[j] + out[i + j];
      out[i + j] = (u32)c;
      c >>= 32;
    }
#pragma unroll
    for (int k = i + 5; k < NO; k++) { c += out[k]; out[k] = (u32)c; c >>= 32; }
  }
}
BPMI_HD void sc_mul(sc &r, const sc &a, const sc &b) {
  u32 t[16];
#pragma unroll
  for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (u64)a.v[i] * b.v[j] + t[i + j]; t[i + j] = (u32)c; c >>= 32; }
    t[i + 8] = (u32)c;
  }
  u32 f1[14], f2[11], f3[9];
  sc_fold_once<8, 14>(f1, t, t + 8);        // < 2^386
  sc_fold_once<6, 11>(f2, f1, f1 + 8);      // hi <= 130 bits -> < 2^260
  sc_fold_once<1, 9>(f3, f2, f2 + 8);       // hi <= 4 bits   -> < 2^256 + 2^134
  sc_cond_sub_q(f3);
  sc_cond_sub_q(f3);
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = f3[i];
}

// x / 2 mod q
BPMI_HD void sc_half(sc &x) {
  const u32 q[8] = BPMI_SC_Q;
  const u32 odd = x.v[0] & 1u;
  u32 t[9];
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { c += (u64)x.v[i] + (odd ? q[i] : 0u); t[i] = (u32)c; c >>= 32; }
  t[8] = (u32)c;
#pragma unroll
  for (int i = 0; i < 8; i++) x.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
}
// r = a^-1 mod q (a in [1, q); a = 0 gives 0) by the binary extended Euclid, with every run of trailing zero bits removed
// in ONE step: ~195 steps of ~165 instructions for a wave of 64 random inputs, against ~450 multiplications for a^(q-2)
// and ~385 steps for the one-bit-per-step form.  Invariants: a x1 == u, a x2 == v (mod q); v stays odd.  A step: if u is
// odd, order the pair so that u >= v and take u -= v, x1 -= x2 (u is even now, or 0: then v = gcd = 1 and x2 is the
// inverse); then u >>= tz with tz = its trailing zeros (at most 31 per step) and x1 = x1 / 2^tz mod q, which is
// (x1 + m q) >> tz for the m < 2^tz that makes the low tz bits vanish, m = x1 * (-q^-1) mod 2^tz.  Written with selects,
// not branches: the lanes of a wave disagree on every condition.  Not constant time: the inputs are public proof data.
#define BPMI_SC_NQINV 0x5588B13Fu          // -q^-1 mod 2^32
BPMI_HD u32 sc_funnel_r(u32 lo, u32 hi, u32 s) { return (u32)((((u64)hi << 32) | lo) >> s); }      // s in [0, 31]
BPMI_HD void sc_inv(sc &r, const sc &a) {
  const u32 q[8] = BPMI_SC_Q;
  u32 u[8], v[8], x1[8], x2[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { u[i] = a.v[i]; v[i] = q[i]; x1[i] = i ? 0u : 1u; x2[i] = 0u; }
  u32 nz = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) nz |= u[i];
  // every lane of a wave runs the same number of steps (a finished lane's steps change nothing): one uniform branch per step
#if defined(__HIP_DEVICE_COMPILE__)
  while (__builtin_amdgcn_ballot_w64(nz != 0)) {
#else
  while (nz) {
#endif
    const u32 oddm = 0u - (u[0] & 1u);                               // all ones when u is odd
    // u < v ?  (borrow of u - v)
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)sc_sbb(u[i], v[i], br);
    const u32 swm = oddm & (0u - br);                                // exchange the pairs: u odd and u < v
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u32 du = (u[i] ^ v[i]) & swm, dx = (x1[i] ^ x2[i]) & swm;
      u[i] ^= du; v[i] ^= du; x1[i] ^= dx; x2[i] ^= dx;
    }
    // u -= v and x1 = x1 - x2 mod q when u is odd
    u32 bu = 0, bx = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) u[i] = sc_sbb(u[i], v[i] & oddm, bu);
#pragma unroll
    for (int i = 0; i < 8; i++) x1[i] = sc_sbb(x1[i], x2[i] & oddm, bx);
    const u32 addm = 0u - bx;                                        // went below zero: add q back
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) x1[i] = sc_adc(x1[i], q[i] & addm, c);
    nz = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) nz |= u[i];
    // strip the trailing zeros of u: tz = 0 for u = 0 (nothing happens), at most 31 per step (a zero low word)
    u32 tz = u[0] ? (u32)__builtin_ctz(u[0]) : 31u;
    tz = nz ? tz : 0u;
    const u32 m = (x1[0] * BPMI_SC_NQINV) & ((1u << tz) - 1u);      // x1 + m q has tz trailing zero bits
    u32 t9[9];
    u64 cc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { cc += (u64)m * q[i] + x1[i]; t9[i] = (u32)cc; cc >>= 32; }
    t9[8] = (u32)cc;
    u32 y[9];
#pragma unroll
    for (int i = 0; i < 8; i++) { y[i] = sc_funnel_r(t9[i], t9[i + 1], tz); u[i] = sc_funnel_r(u[i], i < 7 ? u[i + 1] : 0u, tz); }
    y[8] = t9[8] >> tz;                                               // (x1 + m q) >> tz < 2 q
    // y -= q when y >= q
    u32 sq_[8], bq = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sq_[i] = sc_sbb(y[i], q[i], bq);
    const u32 gem = 0u - (u32)((y[8] != 0) | (bq == 0));
#pragma unroll
    for (int i = 0; i < 8; i++) x1[i] = (sq_[i] & gem) | (y[i] & ~gem);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = x2[i];
}


// ---- GLV decomposition on secp256k1: k = k1 + k2 lambda (mod q) with |k1|, |k2| < 2^128, where lambda (x, y) = (beta x, y) is the
// curve's efficient endomorphism (lambda^3 = 1 mod q, beta^3 = 1 mod p).  An MSM over n pairs with 256-bit scalars becomes one over
// 2n pairs with 128-bit scalars -- the same number of bucket additions, HALF the windows: half the buckets to reduce, half the
// doublings in the tail (csrc/msm_host.hpp).  The result of the MSM is the same group element, so nothing visible changes.
// Lattice basis (a1, b1), (a2, b2) with a_i + b_i lambda = 0 (mod q); c_i = round(k g_i / 2^384) with g1 = round(2^384 b2 / q),
// g2 = round(2^384 (-b1) / q) (the constants of the well-known decomposition, e.g. libsecp256k1's scalar_split_lambda);
//   k1 = k - c1 a1 - c2 a2,   k2 = c1 (-b1) - c2 b2     as INTEGERS: both are small, so 160-bit wrap-around arithmetic suffices.
// tests/test_csrc_host.py checks k1 + k2 lambda == k (mod q) and the 128-bit bound on random and adversarial scalars.
#define BPMI_GLV_G1  {0x45DBB031u, 0xE893209Au, 0x71E8CA7Fu, 0x3DAA8A14u, 0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u}
#define BPMI_GLV_G2  {0x8AC47F71u, 0x1571B4AEu, 0x9DF506C6u, 0x221208ACu, 0x0ABFE4C4u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u}
#define BPMI_GLV_A1  {0x9284EB15u, 0xE86C90E4u, 0xA7D46BCDu, 0x3086D221u, 0x00000000u}          /* = b2 */
#define BPMI_GLV_MB1 {0x0ABFE4C3u, 0x6F547FA9u, 0x010E8828u, 0xE4437ED6u, 0x00000000u}          /* = -b1 */
#define BPMI_GLV_A2  {0x9D44CFD8u, 0x57C1108Du, 0xA8E2F3F6u, 0x14CA50F7u, 0x00000001u}
// c = round(k g / 2^384): the top 128 bits of the 512-bit product, plus its bit 383
BPMI_HD void glv_mul_shift384(u32 c[4], const u32 k[8], const u32 g[8]) {
  u32 t[16];
#pragma unroll
  for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 cy = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { cy += (u64)k[i] * g[j] + t[i + j]; t[i + j] = (u32)cy; cy >>= 32; }
    t[i + 8] = (u32)cy;
  }
  u64 cy = t[11] >> 31;
#pragma unroll
  for (int i = 0; i < 4; i++) { cy += t[12 + i]; c[i] = (u32)cy; cy >>= 32; }
}
// r (5 words, mod 2^160) -= a (4 words) * b (5 words)
BPMI_HD void glv_submul160(u32 r[5], const u32 a[4], const u32 b[5]) {
  u32 prod[5];
#pragma unroll
  for (int i = 0; i < 5; i++) prod[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u64 cy = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      if (i + j < 5) { cy += (u64)a[i] * b[j] + prod[i + j]; prod[i + j] = (u32)cy; cy >>= 32; }
    }
  }
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) r[i] = sc_sbb(r[i], prod[i], br);
}
// |v| of a 160-bit two's-complement value whose magnitude is known to fit 128 bits; returns the sign
BPMI_HD bool glv_abs160(u32 m[4], const u32 v[5]) {
  const bool neg = (v[4] >> 31) != 0;
  const u32 x = neg ? 0xFFFFFFFFu : 0u;
  u32 c = neg ? 1u : 0u;
#pragma unroll
  for (int i = 0; i < 4; i++) m[i] = sc_adc(v[i] ^ x, 0u, c);
  return neg;
}
// k in [0, q) -> magnitudes (4 words each) and signs of k1, k2
BPMI_HD void glv_split(u32 k1[4], bool &neg1, u32 k2[4], bool &neg2, const sc &k) {
  const u32 G1[8] = BPMI_GLV_G1, G2[8] = BPMI_GLV_G2, A1[5] = BPMI_GLV_A1, MB1[5] = BPMI_GLV_MB1, A2[5] = BPMI_GLV_A2;
  u32 c1[4], c2[4];
  glv_mul_shift384(c1, k.v, G1);
  glv_mul_shift384(c2, k.v, G2);
  u32 v1[5], v2[5];
#pragma unroll
  for (int i = 0; i < 5; i++) { v1[i] = k.v[i]; v2[i] = 0; }       // k mod 2^160
  glv_submul160(v1, c1, A1);
  glv_submul160(v1, c2, A2);                                        // k1 = k - c1 a1 - c2 a2
  glv_submul160(v2, c2, A1);                                        // - c2 b2   (b2 = a1)
  u32 nv[5], z[5] = {0, 0, 0, 0, 0};
  glv_submul160(z, c1, MB1);                                        // z = - c1 (-b1)
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) nv[i] = sc_sbb(v2[i], z[i], br);      // k2 = c1 (-b1) - c2 b2
  neg1 = glv_abs160(k1, v1);
  neg2 = glv_abs160(k2, nv);
}

// ---- the same field on 9 x 29-bit limbs ("sq"): for code whose time is modular MULTIPLICATIONS (the batch-preparation kernel:
// ~400 per proof).  On 8 x 32-bit words every partial product needs its carry handled (sc_mul: ~720 instructions on the
// device, 120 of them multiply-adds); with 29-bit limbs a column of nine 58-bit products fits a 64-bit accumulator, so a
// column is ONE v_mad_u64_u32 chain and carries are resolved once per column (sq_mul: 156 multiply-adds, ~270 instructions).
//   value = sum v[i] 2^(29 i);  "loose": every limb < 2^29 + 2^10 and value < 2^261 + 2^143 -- what every sq_* routine
//   returns and accepts; canonical 32-byte form via sq_to_sc.
//   2^261 == D (mod q), D = 32 (2^256 - q), 134 bits = 5 limbs: high limbs fold as products with D's limbs.
struct sq { u32 v[9]; };
#define BPMI_SQ_D {0x1937D7E0u, 0x0DA1732Fu, 0x1AFE2201u, 0x08C6542Du, 0x00028AA2u}
// 64 q with two units lent from every limb to the one below: limb-wise >= any loose operand, so a + BIAS - b needs no borrow
#define BPMI_SQ_BIAS64 {0x4D905040u, 0x44BD199Eu, 0x4A03BBFBu, 0x4E7357A2u, 0x5FFAEAB9u, 0x5FFFFFFDu, 0x5FFFFFFDu, 0x5FFFFFFDu, 0x3FFFFFFDu}

BPMI_HD void sq_from_sc(sq &r, const sc &a) {
#pragma unroll
  for (int i = 0; i < 9; i++) {
    const int lo = 29 * i, w = lo >> 5, sh = lo & 31;
    u32 v = a.v[w] >> sh;
    if (sh > 3 && w + 1 < 8) v |= a.v[w + 1] << (32 - sh);
    r.v[i] = v & M29;
  }
}
BPMI_HD sq sq_small(u32 x) {             // x < 2^29
  sq r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = i ? 0u : x;
  return r;
}
// columns t[0..9) (each < 2^63, total value < 2^271) -> loose limbs: one carry pass, the carry out of limb 8 (e < 2^10) comes
// back as e * D into limbs 0..4, and a short second pass stops at limb 5
BPMI_HD void sq_norm_cols(sq &r, const u64 t[9]) {
  const u32 D[5] = BPMI_SQ_D;
  u32 u[9];
  u64 c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { c += t[k]; u[k] = (u32)c & M29; c >>= 29; }
  const u32 e = (u32)c;
  c = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) { c += (u64)e * D[k] + u[k]; r.v[k] = (u32)c & M29; c >>= 29; }
  r.v[5] = u[5] + (u32)c;
#pragma unroll
  for (int k = 6; k < 9; k++) r.v[k] = u[k];
}
BPMI_HD void sq_add(sq &r, const sq &a, const sq &b) {
  u64 t[9];
#pragma unroll
  for (int k = 0; k < 9; k++) t[k] = (u64)a.v[k] + b.v[k];
  sq_norm_cols(r, t);
}
BPMI_HD void sq_sub(sq &r, const sq &a, const sq &b) {
  const u32 B[9] = BPMI_SQ_BIAS64;
  u64 t[9];
#pragma unroll
  for (int k = 0; k < 9; k++) t[k] = (u64)a.v[k] + (B[k] - b.v[k]);
  sq_norm_cols(r, t);
}
BPMI_HD void sq_neg(sq &r, const sq &a) { const sq z = sq_small(0); sq_sub(r, z, a); }
// C body of the multiplication (host, and the statement-by-statement model of the generated device body in scalar_gen.hpp):
//   columns 9..16 of a*b, carried -> h[0..9);  columns 0..12 of lo + h*D, carried -> r[0..9), g[0..5);
//   columns 0..8 of r + g*D -> sq_norm_cols
BPMI_HD void sq_mul_c(sq &r, const sq &a, const sq &b) {
  const u32 D[5] = BPMI_SQ_D;
  u32 h[9], lo[9], g[5];
  u64 c = 0;
#pragma unroll
  for (int k = 9; k < 17; k++) {
#pragma unroll
    for (int i = k - 8; i <= 8; i++) c += (u64)a.v[i] * b.v[k - i];
    h[k - 9] = (u32)c & M29;
    c >>= 29;
  }
  h[8] = (u32)c;
  c = 0;
#pragma unroll
  for (int k = 0; k < 13; k++) {
    if (k <= 8) {
#pragma unroll
      for (int i = 0; i <= k; i++) c += (u64)a.v[i] * b.v[k - i];
    }
#pragma unroll
    for (int j = 0; j < 5; j++) { const int i = k - j; if (i >= 0 && i <= 8) c += (u64)h[i] * D[j]; }
    if (k <= 8) lo[k] = (u32)c & M29; else g[k - 9] = (u32)c & M29;
    c >>= 29;
  }
  g[4] = (u32)c;
  u64 t[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    t[k] = lo[k];
#pragma unroll
    for (int j = 0; j < 5; j++) { const int i = k - j; if (i >= 0 && i <= 4) t[k] += (u64)g[i] * D[j]; }
  }
  sq_norm_cols(r, t);
}
#include "scalar_gen.hpp"
BPMI_HD void sq_mul(sq &r, const sq &a, const sq &b) {
#if defined(__HIP_DEVICE_COMPILE__)
  sq_mul_dev(r, a, b);
#else
  sq_mul_c(r, a, b);
#endif
}
// loose limbs -> the canonical value in [0, q) as 8 words
BPMI_HD void sq_to_sc(sc &r, const sq &a) {
  u32 w[9];
#pragma unroll
  for (int i = 0; i < 9; i++) w[i] = 0;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 10; i++) {                 // limb 9 = the carry out of limb 8
    u32 limb;
    if (i < 9) { c += a.v[i]; limb = (u32)c & M29; c >>= 29; } else limb = (u32)c;
    const int lo = 29 * i, k = lo >> 5, sh = lo & 31;
    w[k] |= limb << sh;
    if (sh > 3 && k + 1 < 9) w[k + 1] |= limb >> (32 - sh);
  }
  u32 f[9];
  sc_fold_once<1, 9>(f, w, w + 8);               // bits 256.. (< 2^7) times (2^256 - q)
  sc_cond_sub_q(f);
  sc_cond_sub_q(f);
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = f[i];
}

}  // namespace bpmi
