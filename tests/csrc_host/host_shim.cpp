// Host-side build of the DEVICE arithmetic headers (csrc/field.hpp, csrc/curve.hpp),
// for unit tests only: the headers are plain C++, so their limb arithmetic,
// magnitude discipline and exceptional-case handling can be checked against
// Python integers here, without a GPU.  Not part of the product library.
#include <stdio.h>
#include <string.h>
#include "field.hpp"
#include "curve.hpp"
#include "scalar.hpp"
#include "fold_ops_host.hpp"
#include "../../include/bpmi.h"
#include "rp_batch_host.hpp"
#include "host_tail.hpp"
using namespace bpmi;

static void load_fe(fe &r, const uint8_t *b) { u32 w[8]; memcpy(w, b, 32); fe_from_words(r, w); }
static void store_fe(uint8_t *b, const fe &a) { fe c; fe_canon(c, a); u32 w[8]; fe_to_words(w, c); memcpy(b, w, 32); }
static void load_aff(affine &r, const uint8_t *b) { u32 w[16]; memcpy(w, b, 64); affine_from_words(r, w); }
static void store_aff(uint8_t *b, const affine &a) { u32 w[16]; affine_to_words(w, a); memcpy(b, w, 64); }

extern "C" {
// op: 0 mul, 1 sqr(a), 2 add, 3 sub, 4 neg(a), 5 inv(a), 6 (a+b)*(a+b) lazy, 7 (a-b)*(b) lazy, 8 canon roundtrip
void t_fe_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
  fe x, y, r, t;
  load_fe(x, a); load_fe(y, b);
  switch (op) {
    case 0: fe_mul(r, x, y); break;
    case 1: fe_sqr(r, x); break;
    case 2: fe_add(r, x, y); break;
    case 3: fe_sub(r, x, y); break;
    case 4: fe_neg(r, x); break;
    case 5: fe_inv(r, x); break;
    case 6: fe_add(t, x, y); fe_sqr(r, t); break;
    case 7: fe_sub(t, x, y); fe_mul(r, t, y); break;
    case 8: r = x; break;
    case 9: fe_sub(t, x, y); fe_sub(t, t, y); fe_sub(t, t, y); fe_carry(r, t); break;  // mag 7 carry
    case 10: fe_add(t, x, y); fe_sub_m2(r, x, t); fe_carry(r, r); break;              // x - (x+y) = -y
    case 11: fe_sqrt_candidate(r, x); break;
    default: fe_set_zero(r);
  }
  store_fe(out, r);
}
// the multiplication family on RAW limbs (any lazy magnitude the caller wants to try), raw loose limbs out:
// op 0 mul(a,b), 1 sqr(a), 2 mul_add(a,b,add=c), 3 sqr_add(a,add=c), 4 mul2(a,b,c,d), 5 carry(a), 6 canon(a)
void t_fe_raw(int op, const u32 *a, const u32 *b, const u32 *c, const u32 *d, u32 *out) {
  fe A, B, C, D, r;
  for (int k = 0; k < 9; k++) { A.v[k] = a[k]; B.v[k] = b[k]; C.v[k] = c[k]; D.v[k] = d[k]; }
  switch (op) {
    case 0: fe_mul(r, A, B); break;
    case 1: fe_sqr(r, A); break;
    case 2: fe_mul_add(r, A, B, C); break;
    case 3: fe_sqr_add(r, A, C); break;
    case 4: fe_mul2(r, A, B, C, D); break;
    case 5: fe_carry(r, A); break;
    case 6: fe_canon(r, A); break;
    default: fe_set_zero(r);
  }
  for (int k = 0; k < 9; k++) out[k] = r.v[k];
}
// GLV: k (32 bytes LE, < q) -> |k1|, |k2| (16 bytes LE each) and their signs; beta * x for a field element
void t_glv_split(const uint8_t *k, uint8_t *k1, uint8_t *k2, int *signs) {
  sc s;
  memcpy(s.v, k, 32);
  u32 a[4], b[4];
  bool n1, n2;
  glv_split(a, n1, b, n2, s);
  memcpy(k1, a, 16); memcpy(k2, b, 16);
  signs[0] = n1; signs[1] = n2;
}
// the GLV ladder's operation list for K coefficients (32 bytes LE each, < q): ops out (capacity WNAFG_MAXOPS), returns nops or -1; *tail
int t_glv_fold_ops(const uint8_t *coef, uint32_t K, uint32_t *ops, uint32_t *tail) {
  sc c[WNAFG_MAXK];
  if (K > WNAFG_MAXK) return -1;
  for (uint32_t t = 0; t < K; t++) memcpy(c[t].v, coef + 32 * t, 32);
  static WnafG hw;
  if (!glv_fold_ops(hw, c, K)) return -1;
  memcpy(ops, hw.op, 4 * hw.nops);
  *tail = hw.tail;
  return (int)hw.nops;
}
// SHA-256 compression of one block from the given state: which = 0 the portable code, 1 the CPU's SHA extensions (returns 0 and
// leaves the state alone when the CPU has none), 2 whatever the library dispatches to
int t_sha_block(int which, uint32_t *state, const uint8_t *block) {
  if (which == 0) { rp::sha_block_portable(state, block); return 1; }
#if defined(BPMI_SHA_NI)
  if (which == 1) { if (!rp::g_sha_ni) return 0; rp::sha_block_ni(state, block); return 1; }
#else
  if (which == 1) return 0;
#endif
  rp::sha_block(state, block);
  return 1;
}
void t_fe_mul_beta(const uint8_t *x, uint8_t *out) {
  fe a, r;
  load_fe(a, x);
  fe_mul_beta(r, a);
  store_fe(out, r);
}
// arithmetic mod q (csrc/scalar.hpp) on canonical 32-byte little-endian values: op 0 mul, 1 add, 2 neg(a), 3 inv(a), 4 half(a),
// 5 reduce_once(a)
void t_sc_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
  sc x, y, r;
  memcpy(x.v, a, 32); memcpy(y.v, b, 32);
  switch (op) {
    case 0: sc_mul(r, x, y); break;
    case 1: sc_add(r, x, y); break;
    case 2: sc_neg(r, x); break;
    case 3: sc_inv(r, x); break;
    case 4: r = x; sc_half(r); break;
    case 5: r = x; sc_reduce_once(r); break;
    default: memset(r.v, 0, 32);
  }
  memcpy(out, r.v, 32);
}
// the same field on 9 x 29-bit limbs (sq): op 0 mul, 1 add, 2 sub, 3 neg(a), 4 round trip.  a, b: RAW limbs (any loose
// values the caller wants to try); out_limbs: the raw result limbs; out32: its canonical value
void t_sq_raw(int op, const u32 *a, const u32 *b, u32 *out_limbs, uint8_t *out32) {
  sq x, y, r;
  for (int k = 0; k < 9; k++) { x.v[k] = a[k]; y.v[k] = b[k]; }
  switch (op) {
    case 0: sq_mul(r, x, y); break;
    case 1: sq_add(r, x, y); break;
    case 2: sq_sub(r, x, y); break;
    case 3: sq_neg(r, x); break;
    default: r = x;
  }
  for (int k = 0; k < 9; k++) out_limbs[k] = r.v[k];
  sc c;
  sq_to_sc(c, r);
  memcpy(out32, c.v, 32);
}
void t_sq_from_sc(const uint8_t *a32, u32 *out_limbs) {
  sc x; memcpy(x.v, a32, 32);
  sq r; sq_from_sc(r, x);
  for (int k = 0; k < 9; k++) out_limbs[k] = r.v[k];
}
int t_fe_is_zero(const uint8_t *a, const uint8_t *b) {  // is a - b == 0 ?
  fe x, y; load_fe(x, a); load_fe(y, b);
  return fe_equal(x, y) ? 1 : 0;
}
// acc (affine, may be inf) += sum of n affine points (each optionally negated), via XYZZ madd
void t_madd_chain(const uint8_t *start, const uint8_t *pts, const uint8_t *neg, int n, uint8_t *out) {
  affine s; load_aff(s, start);
  xyzz acc; xyzz_from_affine(acc, s);
  for (int i = 0; i < n; i++) { affine p; load_aff(p, pts + 64 * i); xyzz_madd_signed(acc, p, neg[i] != 0); }
  affine r; xyzz_to_affine(r, acc); store_aff(out, r);
}
// tree: sum n affine points by first converting to XYZZ (through a madd so ZZ != 1 when pre > 0) then xyzz_add
void t_xyzz_sum(const uint8_t *pts, int n, int pre, uint8_t *out) {
  xyzz acc; xyzz_set_inf(acc);
  for (int i = 0; i < n; i++) {
    affine p; load_aff(p, pts + 64 * i);
    xyzz q; xyzz_from_affine(q, p);
    for (int k = 0; k < pre; k++) { xyzz d; xyzz_dbl(d, q); xyzz nq; xyzz_neg(nq, q); xyzz_add(q, d, nq); }  // q = 2q - q, non-trivial Z
    xyzz_add(acc, acc, q);
  }
  affine r; xyzz_to_affine(r, acc); store_aff(out, r);
}
// Jacobian ladder: out = k * P by double-and-add with signed additions of +-P (bits of k,
// `neg` flips every addition and the result is negated back), exercising jac_dbl / jac_madd
void t_jac_mul(const uint8_t *pt, const uint8_t *k32, int neg, uint8_t *out) {
  affine p; load_aff(p, pt);
  jac acc; jac_set_inf(acc);
  if (!affine_is_inf(p)) {
    for (int i = 255; i >= 0; i--) {
      jac_dbl(acc, acc);
      if ((k32[i >> 3] >> (i & 7)) & 1) jac_madd_signed(acc, p.x, p.y, neg != 0);
    }
  }
  affine r; jac_to_affine(r, acc);
  if (neg) { affine nr; affine_neg(nr, r); r = nr; }
  store_aff(out, r);
}
// chain of jac_madd with arbitrary points / signs (exceptional cases)
void t_jac_madd_chain(const uint8_t *pts, const uint8_t *neg, int n, uint8_t *out) {
  jac acc; jac_set_inf(acc);
  for (int i = 0; i < n; i++) { affine p; load_aff(p, pts + 64 * i); if (!affine_is_inf(p)) jac_madd_signed(acc, p.x, p.y, neg[i] != 0); }
  affine r; jac_to_affine(r, acc); store_aff(out, r);
}
void t_xyzz_dbl_n(const uint8_t *pt, int n, uint8_t *out) {
  affine p; load_aff(p, pt);
  xyzz q; xyzz_from_affine(q, p);
  for (int k = 0; k < n; k++) xyzz_dbl(q, q);
  affine r; xyzz_to_affine(r, q); store_aff(out, r);
}
// the host tail's own field arithmetic (host_tail.hpp, 4 x 64-bit limbs): op 0 a*b, 1 a^2, 2 a^-1, 3 a+b, 4 a-b; fully reduced in and out
void t_host_f_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *out) {
  bpmi_host::f64 x, y, r;
  memcpy(x.v, a, 32); memcpy(y.v, b, 32);
  switch (op) {
    case 0: bpmi_host::f_mul(r, x, y); break;
    case 1: bpmi_host::f_sqr(r, x); break;
    case 2: bpmi_host::f_inv(r, x); break;
    case 3: bpmi_host::f_add(r, x, y); break;
    default: bpmi_host::f_sub(r, x, y); break;
  }
  memcpy(out, r.v, 32);
}
// The MSM's host tail on window sums given as AFFINE points scaled by z (pts: W * nv points of 64 bytes; zs: as many 32-byte
// field elements, 0 = the identity record): X = x z^2, Y = y z^3, ZZ = z^2, ZZZ = z^3 in the device's 9 x 29-bit limb records.
void t_host_tail(const uint8_t *pts, const uint8_t *zs, u32 W, u32 c, u32 nv, const u32 *off, u32 top, const u32 *top_off, uint8_t *out) {
  std::vector<u32> E(36 * (size_t)W * nv);
  for (size_t i = 0; i < (size_t)W * nv; i++) {
    fe z, zz, zzz, t;
    load_fe(z, zs + 32 * i);
    affine p; load_aff(p, pts + 64 * i);
    xyzz r;
    if (fe_is_zero(z) || affine_is_inf(p)) xyzz_set_inf(r);
    else {
      fe_sqr(zz, z); fe_mul(zzz, zz, z);
      fe_mul(t, p.x, zz); fe_carry(r.X, t);
      fe_mul(t, p.y, zzz); fe_carry(r.Y, t);
      fe_carry(r.ZZ, zz); fe_carry(r.ZZZ, zzz);
    }
    xyzz_store(E.data() + 36 * i, r);
  }
  TailOffs to;
  memset(&to, 0, sizeof(to));
  to.nv = nv; to.top = top;
  for (int k = 0; k < 4; k++) { to.off[k] = off[k]; to.top_off[k] = top_off[k]; }
  bpmi_host::tail_combine(out, E.data(), W, c, to);
}
}
