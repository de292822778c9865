// host_tail.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).
// The O(256)-doubling window combine of an MSM, for the HOST: the same Horner chain as
// msm_tail_combine (msm_kernels.hpp) on 4 x 64-bit limbs with unsigned __int128 products,
// which is what a CPU core is good at (~12 ns per field multiplication against ~35 ns for
// the 9 x 29-bit device formulation compiled for x86).  Inputs are the device's XYZZ
// records (4 x 9 u32 limbs, weakly reduced); the result is the canonical 64-byte affine
// point.  tests/test_gpu_msm.py runs every multi-level case under both tail = 1 (the
// device kernel built from curve.hpp) and tail = 2 (this file) and requires equal bytes.
#pragma once
#include <stdint.h>
#include <string.h>

#include "curve.hpp"

namespace bpmi_host {

typedef unsigned __int128 u128;
typedef uint64_t u64;
struct f64 { u64 v[4]; };                       // fully reduced, [0, p)
static const u64 P64[4] = {0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL};
static const u64 PC = 0x1000003D1ULL;           // 2^256 - p

static inline bool f_is_zero(const f64 &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static inline bool ge_p(const u64 a[4]) {
  for (int i = 3; i >= 0; i--) if (a[i] != P64[i]) return a[i] > P64[i];
  return true;
}
static inline void sub_p(u64 a[4]) {
  u64 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - P64[i] - br; a[i] = (u64)t; br = (u64)(t >> 64) & 1; }
}
static inline void f_add(f64 &r, const f64 &a, const f64 &b) {
  u64 c = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a.v[i] + b.v[i] + c; r.v[i] = (u64)t; c = (u64)(t >> 64); }
  if (c || ge_p(r.v)) sub_p(r.v);
}
static inline void f_sub(f64 &r, const f64 &a, const f64 &b) {
  u64 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a.v[i] - b.v[i] - br; r.v[i] = (u64)t; br = (u64)(t >> 64) & 1; }
  if (br) { u64 c = 0; for (int i = 0; i < 4; i++) { u128 t = (u128)r.v[i] + P64[i] + c; r.v[i] = (u64)t; c = (u64)(t >> 64); } }
}
// lo + hi * (2^256 - p), twice, then the conditional subtraction: t[0..8) -> r in [0, p)
static inline void f_reduce512(f64 &r, const u64 t[8]) {
  u64 m[5];
  u128 c = 0;
  for (int i = 0; i < 4; i++) { c += (u128)t[4 + i] * PC + t[i]; m[i] = (u64)c; c >>= 64; }
  m[4] = (u64)c;
  c = (u128)m[4] * PC;
  u64 k = 0;
  { u128 a0 = (u128)m[0] + (u64)c; r.v[0] = (u64)a0; k = (u64)(a0 >> 64); }
  { u128 a1 = (u128)m[1] + (u64)(c >> 64) + k; r.v[1] = (u64)a1; k = (u64)(a1 >> 64); }
  { u128 a2 = (u128)m[2] + k; r.v[2] = (u64)a2; k = (u64)(a2 >> 64); }
  { u128 a3 = (u128)m[3] + k; r.v[3] = (u64)a3; k = (u64)(a3 >> 64); }
  if (k) { u128 a0 = (u128)r.v[0] + PC; r.v[0] = (u64)a0; u64 kk = (u64)(a0 >> 64); for (int i = 1; i < 4 && kk; i++) { a0 = (u128)r.v[i] + kk; r.v[i] = (u64)a0; kk = (u64)(a0 >> 64); } }
  if (ge_p(r.v)) sub_p(r.v);
}
static inline void f_mul(f64 &r, const f64 &a, const f64 &b) {
  u64 t[8] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.v[i] * b.v[j] + t[i + j]; t[i + j] = (u64)c; c >>= 64; }
    t[i + 4] = (u64)c;
  }
  f_reduce512(r, t);
}
// Round 5: a squaring of its own (6 cross products doubled + 4 squares instead of 16 products): 1 000 of the tail's 3 300
// field multiplications are squarings.
static inline void f_sqr(f64 &r, const f64 &a) {
  u64 t[8];
  u128 c;
  // cross products a_i a_j, i < j, into t[1..6]
  c = (u128)a.v[0] * a.v[1]; t[1] = (u64)c; c >>= 64;
  c += (u128)a.v[0] * a.v[2]; t[2] = (u64)c; c >>= 64;
  c += (u128)a.v[0] * a.v[3]; t[3] = (u64)c; t[4] = (u64)(c >> 64);
  c = (u128)a.v[1] * a.v[2] + t[3]; t[3] = (u64)c; c >>= 64;
  c += (u128)a.v[1] * a.v[3] + t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
  c = (u128)a.v[2] * a.v[3] + t[5]; t[5] = (u64)c; t[6] = (u64)(c >> 64);
  // double them
  t[7] = t[6] >> 63;
  for (int i = 6; i >= 2; i--) t[i] = (t[i] << 1) | (t[i - 1] >> 63);
  t[1] <<= 1;
  // add the squares a_i^2 at word 2i
  c = (u128)a.v[0] * a.v[0]; t[0] = (u64)c; c >>= 64;
  c += t[1]; t[1] = (u64)c; c >>= 64;
  for (int i = 1; i < 4; i++) {
    const u128 sq = (u128)a.v[i] * a.v[i];
    c += (u128)(u64)sq + t[2 * i]; t[2 * i] = (u64)c; c >>= 64;
    c += (u128)(u64)(sq >> 64) + t[2 * i + 1]; t[2 * i + 1] = (u64)c; c >>= 64;
  }
  f_reduce512(r, t);
}
static inline void f_sqr_n(f64 &r, const f64 &a, int n) { r = a; for (int i = 0; i < n; i++) f_sqr(r, r); }
// a^(p-2) by the addition chain over the runs of ones of p - 2 = 2^256 - 2^32 - 979 (223, 22, 1, 2, 1 ones): 255 squarings and
// 15 multiplications (round 4: 256 + 249 with plain square-and-multiply)
static inline void f_inv(f64 &r, const f64 &a) {
  f64 x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t;
  f_sqr(t, a); f_mul(x2, t, a);
  f_sqr(t, x2); f_mul(x3, t, a);
  f_sqr_n(t, x3, 3); f_mul(x6, t, x3);
  f_sqr_n(t, x6, 3); f_mul(x9, t, x3);
  f_sqr_n(t, x9, 2); f_mul(x11, t, x2);
  f_sqr_n(t, x11, 11); f_mul(x22, t, x11);
  f_sqr_n(t, x22, 22); f_mul(x44, t, x22);
  f_sqr_n(t, x44, 44); f_mul(x88, t, x44);
  f_sqr_n(t, x88, 88); f_mul(x176, t, x88);
  f_sqr_n(t, x176, 44); f_mul(x220, t, x44);
  f_sqr_n(t, x220, 3); f_mul(x223, t, x3);
  f_sqr_n(t, x223, 23); f_mul(t, t, x22);
  f_sqr_n(t, t, 5); f_mul(t, t, a);
  f_sqr_n(t, t, 3); f_mul(t, t, x2);
  f_sqr_n(t, t, 2); f_mul(r, t, a);
}
static inline void f_from_limbs(f64 &r, const bpmi::u32 limbs[9]) {
  bpmi::fe t, c;
  for (int k = 0; k < 9; k++) t.v[k] = limbs[k];
  bpmi::fe_canon(c, t);
  bpmi::u32 w[8];
  bpmi::fe_to_words(w, c);
  memcpy(r.v, w, 32);
}

struct pt { f64 X, Y, ZZ, ZZZ; };                 // XYZZ; identity = ZZ == 0
static inline void pt_set_inf(pt &r) { memset(&r, 0, sizeof(r)); }
static inline void pt_load(pt &r, const bpmi::u32 *rec) {
  f_from_limbs(r.X, rec); f_from_limbs(r.Y, rec + 9); f_from_limbs(r.ZZ, rec + 18); f_from_limbs(r.ZZZ, rec + 27);
}
static inline void pt_dbl(pt &r, const pt &a) {          // dbl-2008-s-1, a = 0
  if (f_is_zero(a.ZZ)) { pt_set_inf(r); return; }
  f64 U, V, W, S, M, t, t2, X3;
  f_add(U, a.Y, a.Y); f_sqr(V, U); f_mul(W, U, V); f_mul(S, a.X, V);
  f_sqr(t, a.X); f_add(M, t, t); f_add(M, M, t);
  f_sqr(X3, M); f_sub(X3, X3, S); f_sub(X3, X3, S);
  f_sub(t, S, X3); f_mul(t, M, t); f_mul(t2, W, a.Y); f_sub(t, t, t2);
  f_mul(r.ZZ, V, a.ZZ); f_mul(r.ZZZ, W, a.ZZZ);
  r.X = X3; r.Y = t;
}
static inline void pt_add(pt &r, const pt &a, const pt &b) {   // add-2008-s, complete
  if (f_is_zero(a.ZZ)) { r = b; return; }
  if (f_is_zero(b.ZZ)) { r = a; return; }
  f64 U1, U2, S1, S2, P, R, PP, PPP, Q, t, t2, X3;
  f_mul(U1, a.X, b.ZZ); f_mul(U2, b.X, a.ZZ); f_mul(S1, a.Y, b.ZZZ); f_mul(S2, b.Y, a.ZZZ);
  f_sub(P, U2, U1); f_sub(R, S2, S1);
  if (f_is_zero(P)) { if (f_is_zero(R)) { pt_dbl(r, a); return; } pt_set_inf(r); return; }
  f_sqr(PP, P); f_mul(PPP, P, PP); f_mul(Q, U1, PP);
  f_sqr(X3, R); f_sub(X3, X3, PPP); f_sub(X3, X3, Q); f_sub(X3, X3, Q);
  f_sub(t, Q, X3); f_mul(t, R, t); f_mul(t2, S1, PPP); f_sub(t, t, t2);
  f_mul(t2, a.ZZ, b.ZZ); f_mul(r.ZZ, t2, PP);
  f_mul(t2, a.ZZZ, b.ZZZ); f_mul(r.ZZZ, t2, PPP);
  r.X = X3; r.Y = t;
}
// k doublings in a row.  Round 5: from two on they run in Jacobian coordinates (dbl-2009-l, a = 0: 2M + 5S against the 6M + 3S of
// the XYZZ doubling; in: (X ZZ, Y ZZZ, ZZ), out: ZZ = Z^2, ZZZ = Z ZZ -- 3M + 1S for the round trip, and the identity Z = 0 stays
// the identity).  The chain is 256 doublings and ~64 additions: the doublings are what the tail costs.
static inline void pt_dbl_run(pt &a, int k) {
  if (k <= 0) return;
  if (k == 1 || f_is_zero(a.ZZ)) { for (int i = 0; i < k; i++) pt_dbl(a, a); return; }
  f64 X, Y, Z, A, B, C, D, E, F, t;
  f_mul(X, a.X, a.ZZ); f_mul(Y, a.Y, a.ZZZ); Z = a.ZZ;
  for (int i = 0; i < k; i++) {
    f_sqr(A, X); f_sqr(B, Y); f_sqr(C, B);
    f_add(t, X, B); f_sqr(t, t); f_sub(t, t, A); f_sub(t, t, C); f_add(D, t, t);
    f_add(E, A, A); f_add(E, E, A);
    f_sqr(F, E);
    f_mul(t, Y, Z); f_add(Z, t, t);
    f_sub(t, F, D); f_sub(X, t, D);
    f_sub(t, D, X); f_mul(t, E, t);
    f_add(C, C, C); f_add(C, C, C); f_add(C, C, C);
    f_sub(Y, t, C);
  }
  a.X = X; a.Y = Y;
  f_sqr(a.ZZ, Z); f_mul(a.ZZZ, a.ZZ, Z);
}
// canonical 64-byte affine form (x || y little-endian, the identity 64 zero bytes) <-> XYZZ
static inline void pt_to_affine(uint8_t out[64], const pt &acc) {
  if (f_is_zero(acc.ZZ)) { memset(out, 0, 64); return; }
  f64 zz_zzz, inv, izz, izzz, x, y;
  f_mul(zz_zzz, acc.ZZ, acc.ZZZ);
  f_inv(inv, zz_zzz);
  f_mul(izz, inv, acc.ZZZ);
  f_mul(izzz, inv, acc.ZZ);
  f_mul(x, acc.X, izz);
  f_mul(y, acc.Y, izzz);
  memcpy(out, x.v, 32);
  memcpy(out + 32, y.v, 32);
}
static inline void pt_from_affine(pt &r, const uint8_t in[64]) {
  memcpy(r.X.v, in, 32);
  memcpy(r.Y.v, in + 32, 32);
  if (f_is_zero(r.X) && f_is_zero(r.Y)) { pt_set_inf(r); return; }
  memset(&r.ZZ, 0, sizeof(f64)); memset(&r.ZZZ, 0, sizeof(f64));
  r.ZZ.v[0] = 1; r.ZZZ.v[0] = 1;
}
// result = sum_w 2^(c w) sum_v 32^v E[w][v] as ONE Horner chain over bit positions
static inline void tail_combine(uint8_t out[64], const bpmi::u32 *E, bpmi::u32 W, bpmi::u32 c, const bpmi::TailOffs &to) {
  pt acc;
  pt_set_inf(acc);
  for (int w = (int)W - 1; w >= 0; w--) {
    const bool wide = to.top && (bpmi::u32)w + to.top >= W;                          // (to.top = Wb: the last Wb windows are c + 1 bits wide)
    int prev = (int)c + (wide ? 1 : 0);
    const bpmi::u32 *offs = wide ? to.top_off : to.off;                              // ... and were split at their own bit offsets
    for (int v = (int)to.nv - 1; v >= 0; v--) {
      pt_dbl_run(acc, prev - (int)offs[v]);
      prev = (int)offs[v];
      pt e;
      pt_load(e, E + ((size_t)w * to.nv + v) * 36);
      pt_add(acc, acc, e);
    }
  }
  pt_to_affine(out, acc);
}

}  // namespace bpmi_host
