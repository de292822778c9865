// field.hpp -- arithmetic in F_p, p = 2^256 - 2^32 - 977 (secp256k1 base field),
// for gfx950.  Replaces what the reference reaches through fastecdsa/GMP on every
// `Point + Point` (/root/reference/src/pippenger/group.py:31-32).
//
// Representation: 9 limbs of 29 bits in u32 (value = sum v[k] * 2^(29k)).
// Why not 8x32: on gfx950 v_mad_u64_u32 adds a full 64-bit value for free but has
// no carry-in, and a carry through VCC costs wait states (profiles/r01_fe_microbench.txt:
// 9x29 carry-free 171 G mul/s vs 8x32 137-165 G).  With 29-bit limbs nine products
// (< 2^58 each) accumulate in one u64 column with no carry handling at all.
//
// Magnitudes.  A value has magnitude m when every limb is < m * 2^29.
//   tight  = magnitude 1, limb 8 <= 2^24 + 2^20        (output of mul/sqr/carry)
//   fe_add / fe_sub are LAZY (no carry): magnitudes add; fe_sub adds 2 (a bias of 2p)
//   fe_mul / fe_sqr need  mag(a) * mag(b) <= 7   (9 * 7 * 2^58 < 2^64)
// Values are kept only weakly reduced (any representative < 2^257); fe_canon gives
// the unique representative in [0, p) for comparison and output.
//
// Everything here is plain C++ so the same header is unit-tested on the host
// (tests/csrc_host) before it ever runs on a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BPMI_HD __host__ __device__ __forceinline__
#else
#define BPMI_HD inline
#endif

namespace bpmi {

typedef uint32_t u32;
typedef uint64_t u64;

struct fe { u32 v[9]; };

constexpr u32 M29 = 0x1FFFFFFFu;
constexpr u32 M24 = 0x00FFFFFFu;

#define BPMI_FE_P     {0x1FFFFC2Fu, 0x1FFFFFF7u, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x00FFFFFFu}
#define BPMI_FE_2P    {0x1FFFF85Eu, 0x1FFFFFEFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x1FFFFFFFu, 0x01FFFFFFu}
// 2p written with every limb >= the largest tight limb, so (a + BIAS2 - b) never borrows
#define BPMI_FE_BIAS2 {0x3FFFF85Eu, 0x3FFFFFEEu, 0x3FFFFFFEu, 0x3FFFFFFEu, 0x3FFFFFFEu, 0x3FFFFFFEu, 0x3FFFFFFEu, 0x3FFFFFFEu, 0x01FFFFFEu}
// 4p written so that a magnitude-2 value can be subtracted
#define BPMI_FE_BIAS4 {0x5FFFF0BCu, 0x5FFFFFDDu, 0x5FFFFFFDu, 0x5FFFFFFDu, 0x5FFFFFFDu, 0x5FFFFFFDu, 0x5FFFFFFDu, 0x5FFFFFFDu, 0x03FFFFFDu}

BPMI_HD void fe_set_zero(fe &r) {
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = 0;
}
BPMI_HD void fe_set_one(fe &r) { fe_set_zero(r); r.v[0] = 1; }

// ---- 32-byte little-endian <-> limbs (the C-ABI layout: 8 x u32 LE words) ----
BPMI_HD void fe_from_words(fe &r, const u32 w[8]) {
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int bit = 29 * k, i = bit >> 5, s = bit & 31;
    u64 x = w[i];
    if (i + 1 < 8) x |= (u64)w[i + 1] << 32;
    r.v[k] = (u32)(x >> s) & M29;
  }
}
// a must be canonical (fe_canon) -- limbs < 2^29, value < 2^256
BPMI_HD void fe_to_words(u32 w[8], const fe &a) {
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int bit = 32 * i, k = bit / 29, s = bit - 29 * k;
    u64 x = (u64)a.v[k] >> s;
    x |= (u64)a.v[k + 1] << (29 - s);
    if (k + 2 < 9) x |= (u64)a.v[k + 2] << (58 - s);
    w[i] = (u32)x;
  }
}

// ---- lazy add / sub ----------------------------------------------------------
BPMI_HD void fe_add(fe &r, const fe &a, const fe &b) {
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = a.v[k] + b.v[k];
}
// r = a - b + 2p ; b must be tight; mag(r) = mag(a) + 2
BPMI_HD void fe_sub(fe &r, const fe &a, const fe &b) {
  const u32 bias[9] = BPMI_FE_BIAS2;
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = a.v[k] + bias[k] - b.v[k];
}
// r = a - b + 4p ; b of magnitude <= 2; mag(r) = mag(a) + 3
BPMI_HD void fe_sub_m2(fe &r, const fe &a, const fe &b) {
  const u32 bias[9] = BPMI_FE_BIAS4;
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = a.v[k] + bias[k] - b.v[k];
}
// r = 2p - a ; a tight; mag 2
BPMI_HD void fe_neg(fe &r, const fe &a) {
  const u32 bias[9] = BPMI_FE_BIAS2;
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = bias[k] - a.v[k];
}

// ---- carry: any limbs (< 2^32) -> tight -----------------------------------------
BPMI_HD void fe_carry(fe &r, const fe &a) {
  const u32 h = a.v[8] >> 24;                      // units of 2^256 == 2^32 + 977
  u64 c = (u64)a.v[0] + (u64)h * 977u;
  r.v[0] = (u32)c & M29; c >>= 29;
  c += (u64)a.v[1] + ((u64)h << 3);                // 2^32 = 2^29 * 8
  r.v[1] = (u32)c & M29; c >>= 29;
#pragma unroll
  for (int k = 2; k < 8; k++) { c += a.v[k]; r.v[k] = (u32)c & M29; c >>= 29; }
  r.v[8] = (a.v[8] & M24) + (u32)c;
}

// acc += x * y (u32 x u32 -> u64, plus the 64-bit accumulator) = ONE v_mad_u64_u32.
// The C bodies below (fe_mul_c / fe_sqr_c) are what the HOST compiles (unit tests, host tail).  On the
// device the generated bodies further down run instead: one inline-asm statement per product column
// chains the multiply-adds through the running accumulator, because the compiler otherwise adds the
// carry of column k with a separate 64-bit addition.  (A first attempt with one asm statement per
// multiply-add was slower -- hipcc pads every asm statement with an s_nop; per column it pays.)
#define BPMI_MAC(acc, x, y) ((acc) += (u64)(x) * (y))

// 2^8 and 2^16 as multiplier operands the optimiser cannot see through: with literal powers of two
// it rewrites the multiply-add as zero-extend + 64-bit shift + 64-bit add (3 instructions); read
// from a (never modified) device variable they stay ONE v_mad_u64_u32 with an SGPR operand.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ u32 bpmi_k256 = 256u, bpmi_k65536 = 65536u;
#define BPMI_K256 bpmi_k256
#define BPMI_K65536 bpmi_k65536
#else
#define BPMI_K256 256u
#define BPMI_K65536 65536u
#endif

// ---- product -> tight result ----------------------------------------------------------
// The 17 product columns are two carry chains.  The HIGH chain (columns 9..16) runs first and
// leaves limbs th[0..7] (29 bits each) plus its carry-out t17; since
//     2^261 == 2^37 + 31264 (mod p)   and   2^37 = 2^8 * 2^29,
// a limb of weight 2^(29 (9 + j)) adds 31264 x itself to column j and 256 x itself to column
// j + 1.  Those two terms are simply two more multiply-adds INTO THE RUNNING 64-bit
// ACCUMULATOR of the low chain's columns, so the fold needs no temporaries, no zero-extended
// copies and no second carry pass: the low chain's own carry propagation normalises products
// and fold together.  (t17 sits at column 17 = 8 + 9: 31264 x t17 goes to column 8, and its
// 256 x t17 part, column 9, folds once more into columns 0 and 1.)  What is left afterwards is
// the carry out of column 8 and the 5 bits of limb 8 above 2^256, both reduced with
// 2^256 == 2^32 + 977 and a 32-bit ripple.  173 instructions per multiplication against 207
// for separate extract / fold / carry passes.
//   bounds: mag(a) mag(b) <= 7  =>  a column of products < 63 * 2^58; the fold terms add < 2^56;
//   t17 < 2^33; carry out of column 8 < 2^36.
BPMI_HD void fe_reduce_tail(fe &r, u32 t[9], u64 c8) {
  // V = units of 2^256 above limb 8's 24 bits: 2^261 = 32 * 2^256
  const u64 V = (u64)(t[8] >> 24) + (c8 << 5);               // < 2^42
  u64 c = (u64)t[0] + V * 977u;
  r.v[0] = (u32)c & M29; c >>= 29;
  c += (u64)t[1] + (V << 3);                                 // 2^32 = 8 * 2^29
  r.v[1] = (u32)c & M29;
  u32 cc = (u32)(c >> 29);                                   // < 2^17: 32-bit from here on
#pragma unroll
  for (int k = 2; k < 8; k++) { const u32 x = t[k] + cc; r.v[k] = x & M29; cc = x >> 29; }
  r.v[8] = (t[8] & M24) + cc;
}

BPMI_HD void fe_mul_c(fe &r, const fe &a, const fe &b) {
  u32 th[8], t[9];
  u64 c = 0;
#pragma unroll
  for (int k = 9; k < 17; k++) {
    u64 s = c;
#pragma unroll
    for (int i = k - 8; i <= 8; i++) BPMI_MAC(s, a.v[i], b.v[k - i]);
    th[k - 9] = (u32)s & M29;
    c = s >> 29;
  }
  const u32 t17 = (u32)c;
  c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    u64 s = c;
    if (k == 0) BPMI_MAC(s, t17, 31264u * 256u);            // 256 t17 at column 9 -> 31264 x at column 0
    if (k == 1) BPMI_MAC(s, t17, BPMI_K65536);               //                    -> 256 x at column 1
    if (k < 8) BPMI_MAC(s, th[k], 31264u); else BPMI_MAC(s, t17, 31264u);
    if (k >= 1) BPMI_MAC(s, th[k - 1], BPMI_K256);
#pragma unroll
    for (int i = 0; i <= k; i++) BPMI_MAC(s, a.v[i], b.v[k - i]);
    t[k] = (u32)s & M29;
    c = s >> 29;
  }
  fe_reduce_tail(r, t, c);
}

// a of magnitude <= 2 (doubled limbs must fit 32 bits and 9 * 2 * m^2 * 2^58 < 2^64)
BPMI_HD void fe_sqr_c(fe &r, const fe &a) {
  u32 d[9];
#pragma unroll
  for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
  u32 th[8], t[9];
  u64 c = 0;
#pragma unroll
  for (int k = 9; k < 17; k++) {
    u64 s = c;
#pragma unroll
    for (int i = k - 8; 2 * i <= k; i++) { if (2 * i == k) BPMI_MAC(s, a.v[i], a.v[i]); else BPMI_MAC(s, d[i], a.v[k - i]); }
    th[k - 9] = (u32)s & M29;
    c = s >> 29;
  }
  const u32 t17 = (u32)c;
  c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    u64 s = c;
    if (k == 0) BPMI_MAC(s, t17, 31264u * 256u);
    if (k == 1) BPMI_MAC(s, t17, BPMI_K65536);
    if (k < 8) BPMI_MAC(s, th[k], 31264u); else BPMI_MAC(s, t17, 31264u);
    if (k >= 1) BPMI_MAC(s, th[k - 1], BPMI_K256);
#pragma unroll
    for (int i = 0; 2 * i <= k; i++) { if (2 * i == k) BPMI_MAC(s, a.v[i], a.v[i]); else BPMI_MAC(s, d[i], a.v[k - i]); }
    t[k] = (u32)s & M29;
    c = s >> 29;
  }
  fe_reduce_tail(r, t, c);
}

// ---- GENERATED by tools/gen_fe_mul_asm.py: device bodies of fe_mul / fe_sqr ----
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void fe_mul_dev(fe &r, const fe &a, const fe &b) {
  u32 th[8], t[9];
  u64 c = 0, sink_;
  const u32 k31264 = 31264u, k256 = 256u, k65536 = 65536u, kf = 31264u * 256u;
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+v"(s), "=&s"(sink_) : "v"(a.v[1]), "v"(b.v[8]), "v"(a.v[2]), "v"(b.v[7]), "v"(a.v[3]), "v"(b.v[6]), "v"(a.v[4]), "v"(b.v[5]), "v"(a.v[5]), "v"(b.v[4]), "v"(a.v[6]), "v"(b.v[3]), "v"(a.v[7]), "v"(b.v[2]), "v"(a.v[8]), "v"(b.v[1]));
    th[0] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+v"(s), "=&s"(sink_) : "v"(a.v[2]), "v"(b.v[8]), "v"(a.v[3]), "v"(b.v[7]), "v"(a.v[4]), "v"(b.v[6]), "v"(a.v[5]), "v"(b.v[5]), "v"(a.v[6]), "v"(b.v[4]), "v"(a.v[7]), "v"(b.v[3]), "v"(a.v[8]), "v"(b.v[2]));
    th[1] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
        : "+v"(s), "=&s"(sink_) : "v"(a.v[3]), "v"(b.v[8]), "v"(a.v[4]), "v"(b.v[7]), "v"(a.v[5]), "v"(b.v[6]), "v"(a.v[6]), "v"(b.v[5]), "v"(a.v[7]), "v"(b.v[4]), "v"(a.v[8]), "v"(b.v[3]));
    th[2] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+v"(s), "=&s"(sink_) : "v"(a.v[4]), "v"(b.v[8]), "v"(a.v[5]), "v"(b.v[7]), "v"(a.v[6]), "v"(b.v[6]), "v"(a.v[7]), "v"(b.v[5]), "v"(a.v[8]), "v"(b.v[4]));
    th[3] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+v"(s), "=&s"(sink_) : "v"(a.v[5]), "v"(b.v[8]), "v"(a.v[6]), "v"(b.v[7]), "v"(a.v[7]), "v"(b.v[6]), "v"(a.v[8]), "v"(b.v[5]));
    th[4] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+v"(s), "=&s"(sink_) : "v"(a.v[6]), "v"(b.v[8]), "v"(a.v[7]), "v"(b.v[7]), "v"(a.v[8]), "v"(b.v[6]));
    th[5] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+v"(s), "=&s"(sink_) : "v"(a.v[7]), "v"(b.v[8]), "v"(a.v[8]), "v"(b.v[7]));
    th[6] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+v"(s), "=&s"(sink_) : "v"(a.v[8]), "v"(b.v[8]));
    th[7] = (u32)s & M29; c = s >> 29; }
  const u32 t17 = (u32)c;
  c = 0;
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+v"(s), "=&s"(sink_) : "v"(t17), "s"(kf), "v"(th[0]), "s"(k31264), "v"(a.v[0]), "v"(b.v[0]));
    t[0] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+v"(s), "=&s"(sink_) : "v"(t17), "s"(k65536), "v"(th[1]), "s"(k31264), "v"(th[0]), "s"(k256), "v"(a.v[0]), "v"(b.v[1]), "v"(a.v[1]), "v"(b.v[0]));
    t[1] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[2]), "s"(k31264), "v"(th[1]), "s"(k256), "v"(a.v[0]), "v"(b.v[2]), "v"(a.v[1]), "v"(b.v[1]), "v"(a.v[2]), "v"(b.v[0]));
    t[2] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[3]), "s"(k31264), "v"(th[2]), "s"(k256), "v"(a.v[0]), "v"(b.v[3]), "v"(a.v[1]), "v"(b.v[2]), "v"(a.v[2]), "v"(b.v[1]), "v"(a.v[3]), "v"(b.v[0]));
    t[3] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[4]), "s"(k31264), "v"(th[3]), "s"(k256), "v"(a.v[0]), "v"(b.v[4]), "v"(a.v[1]), "v"(b.v[3]), "v"(a.v[2]), "v"(b.v[2]), "v"(a.v[3]), "v"(b.v[1]), "v"(a.v[4]), "v"(b.v[0]));
    t[4] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[5]), "s"(k31264), "v"(th[4]), "s"(k256), "v"(a.v[0]), "v"(b.v[5]), "v"(a.v[1]), "v"(b.v[4]), "v"(a.v[2]), "v"(b.v[3]), "v"(a.v[3]), "v"(b.v[2]), "v"(a.v[4]), "v"(b.v[1]), "v"(a.v[5]), "v"(b.v[0]));
    t[5] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[6]), "s"(k31264), "v"(th[5]), "s"(k256), "v"(a.v[0]), "v"(b.v[6]), "v"(a.v[1]), "v"(b.v[5]), "v"(a.v[2]), "v"(b.v[4]), "v"(a.v[3]), "v"(b.v[3]), "v"(a.v[4]), "v"(b.v[2]), "v"(a.v[5]), "v"(b.v[1]), "v"(a.v[6]), "v"(b.v[0]));
    t[6] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[7]), "s"(k31264), "v"(th[6]), "s"(k256), "v"(a.v[0]), "v"(b.v[7]), "v"(a.v[1]), "v"(b.v[6]), "v"(a.v[2]), "v"(b.v[5]), "v"(a.v[3]), "v"(b.v[4]), "v"(a.v[4]), "v"(b.v[3]), "v"(a.v[5]), "v"(b.v[2]), "v"(a.v[6]), "v"(b.v[1]), "v"(a.v[7]), "v"(b.v[0]));
    t[7] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0\n\tv_mad_u64_u32 %0, %1, %16, %17, %0\n\tv_mad_u64_u32 %0, %1, %18, %19, %0\n\tv_mad_u64_u32 %0, %1, %20, %21, %0\n\tv_mad_u64_u32 %0, %1, %22, %23, %0"
        : "+v"(s), "=&s"(sink_) : "v"(t17), "s"(k31264), "v"(th[7]), "s"(k256), "v"(a.v[0]), "v"(b.v[8]), "v"(a.v[1]), "v"(b.v[7]), "v"(a.v[2]), "v"(b.v[6]), "v"(a.v[3]), "v"(b.v[5]), "v"(a.v[4]), "v"(b.v[4]), "v"(a.v[5]), "v"(b.v[3]), "v"(a.v[6]), "v"(b.v[2]), "v"(a.v[7]), "v"(b.v[1]), "v"(a.v[8]), "v"(b.v[0]));
    t[8] = (u32)s & M29; c = s >> 29; }
  (void)sink_;
  fe_reduce_tail(r, t, c);
}
__device__ __forceinline__ void fe_sqr_dev(fe &r, const fe &a) {
  u32 th[8], t[9];
  u64 c = 0, sink_;
  const u32 k31264 = 31264u, k256 = 256u, k65536 = 65536u, kf = 31264u * 256u;
  u32 d[9];
#pragma unroll
  for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+v"(s), "=&s"(sink_) : "v"(d[1]), "v"(a.v[8]), "v"(d[2]), "v"(a.v[7]), "v"(d[3]), "v"(a.v[6]), "v"(d[4]), "v"(a.v[5]));
    th[0] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+v"(s), "=&s"(sink_) : "v"(d[2]), "v"(a.v[8]), "v"(d[3]), "v"(a.v[7]), "v"(d[4]), "v"(a.v[6]), "v"(a.v[5]), "v"(a.v[5]));
    th[1] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+v"(s), "=&s"(sink_) : "v"(d[3]), "v"(a.v[8]), "v"(d[4]), "v"(a.v[7]), "v"(d[5]), "v"(a.v[6]));
    th[2] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+v"(s), "=&s"(sink_) : "v"(d[4]), "v"(a.v[8]), "v"(d[5]), "v"(a.v[7]), "v"(a.v[6]), "v"(a.v[6]));
    th[3] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+v"(s), "=&s"(sink_) : "v"(d[5]), "v"(a.v[8]), "v"(d[6]), "v"(a.v[7]));
    th[4] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0"
        : "+v"(s), "=&s"(sink_) : "v"(d[6]), "v"(a.v[8]), "v"(a.v[7]), "v"(a.v[7]));
    th[5] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+v"(s), "=&s"(sink_) : "v"(d[7]), "v"(a.v[8]));
    th[6] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0"
        : "+v"(s), "=&s"(sink_) : "v"(a.v[8]), "v"(a.v[8]));
    th[7] = (u32)s & M29; c = s >> 29; }
  const u32 t17 = (u32)c;
  c = 0;
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0"
        : "+v"(s), "=&s"(sink_) : "v"(t17), "s"(kf), "v"(th[0]), "s"(k31264), "v"(a.v[0]), "v"(a.v[0]));
    t[0] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+v"(s), "=&s"(sink_) : "v"(t17), "s"(k65536), "v"(th[1]), "s"(k31264), "v"(th[0]), "s"(k256), "v"(d[0]), "v"(a.v[1]));
    t[1] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[2]), "s"(k31264), "v"(th[1]), "s"(k256), "v"(d[0]), "v"(a.v[2]), "v"(a.v[1]), "v"(a.v[1]));
    t[2] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[3]), "s"(k31264), "v"(th[2]), "s"(k256), "v"(d[0]), "v"(a.v[3]), "v"(d[1]), "v"(a.v[2]));
    t[3] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[4]), "s"(k31264), "v"(th[3]), "s"(k256), "v"(d[0]), "v"(a.v[4]), "v"(d[1]), "v"(a.v[3]), "v"(a.v[2]), "v"(a.v[2]));
    t[4] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[5]), "s"(k31264), "v"(th[4]), "s"(k256), "v"(d[0]), "v"(a.v[5]), "v"(d[1]), "v"(a.v[4]), "v"(d[2]), "v"(a.v[3]));
    t[5] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[6]), "s"(k31264), "v"(th[5]), "s"(k256), "v"(d[0]), "v"(a.v[6]), "v"(d[1]), "v"(a.v[5]), "v"(d[2]), "v"(a.v[4]), "v"(a.v[3]), "v"(a.v[3]));
    t[6] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0"
        : "+v"(s), "=&s"(sink_) : "v"(th[7]), "s"(k31264), "v"(th[6]), "s"(k256), "v"(d[0]), "v"(a.v[7]), "v"(d[1]), "v"(a.v[6]), "v"(d[2]), "v"(a.v[5]), "v"(d[3]), "v"(a.v[4]));
    t[7] = (u32)s & M29; c = s >> 29; }
  { u64 s = c;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0\n\tv_mad_u64_u32 %0, %1, %4, %5, %0\n\tv_mad_u64_u32 %0, %1, %6, %7, %0\n\tv_mad_u64_u32 %0, %1, %8, %9, %0\n\tv_mad_u64_u32 %0, %1, %10, %11, %0\n\tv_mad_u64_u32 %0, %1, %12, %13, %0\n\tv_mad_u64_u32 %0, %1, %14, %15, %0"
        : "+v"(s), "=&s"(sink_) : "v"(t17), "s"(k31264), "v"(th[7]), "s"(k256), "v"(d[0]), "v"(a.v[8]), "v"(d[1]), "v"(a.v[7]), "v"(d[2]), "v"(a.v[6]), "v"(d[3]), "v"(a.v[5]), "v"(a.v[4]), "v"(a.v[4]));
    t[8] = (u32)s & M29; c = s >> 29; }
  (void)sink_;
  fe_reduce_tail(r, t, c);
}
#endif
// ---- end GENERATED ----

// The device runs the generated bodies (same columns, same fold, multiply-adds chained by hand);
// the host -- unit tests, host tail helpers -- runs the C bodies above.  tests/test_gpu_*.py and the
// fuzzers compare the device results with the oracle, tests/test_csrc_host.py the C bodies.
BPMI_HD void fe_mul(fe &r, const fe &a, const fe &b) {
#if defined(__HIP_DEVICE_COMPILE__)
  fe_mul_dev(r, a, b);
#else
  fe_mul_c(r, a, b);
#endif
}
BPMI_HD void fe_sqr(fe &r, const fe &a) {
#if defined(__HIP_DEVICE_COMPILE__)
  fe_sqr_dev(r, a);
#else
  fe_sqr_c(r, a);
#endif
}

// r = a * k for a small constant (k * mag(a) must stay < 8), lazy
BPMI_HD void fe_mul_small(fe &r, const fe &a, u32 k) {
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = a.v[i] * k;
}

// ---- canonical form in [0, p) ----------------------------------------------------
BPMI_HD void fe_canon(fe &r, const fe &a) {
  fe t;
  fe_carry(t, a);
  fe_carry(t, t);                                   // now value < 2^256 + tiny, tight
  // t >= p  <=>  t + (2^32 + 977) >= 2^256
  u32 s[9];
  u64 c = (u64)t.v[0] + 977u; s[0] = (u32)c & M29; c >>= 29;
  c += (u64)t.v[1] + 8u;      s[1] = (u32)c & M29; c >>= 29;
#pragma unroll
  for (int k = 2; k < 8; k++) { c += t.v[k]; s[k] = (u32)c & M29; c >>= 29; }
  c += t.v[8]; s[8] = (u32)c;
  const bool ge = (s[8] >> 24) != 0;
#pragma unroll
  for (int k = 0; k < 8; k++) r.v[k] = ge ? s[k] : t.v[k];
  r.v[8] = ge ? (s[8] & M24) : t.v[8];
}

// a tight: is it == 0 (mod p)?  The tight representatives of 0 below 2^257 are 0, p, 2p.
// Kept branch-free: a limb-0 pre-filter with an early return was measured and made the
// kernels that inline the general addition markedly slower (bucket reduce 0.41 -> 0.68 ms).
BPMI_HD bool fe_is_zero_tight(const fe &a) {
  const u32 P1[9] = BPMI_FE_P, P2[9] = BPMI_FE_2P;
  u32 z = 0, d1 = 0, d2 = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) { z |= a.v[k]; d1 |= a.v[k] ^ P1[k]; d2 |= a.v[k] ^ P2[k]; }
  return (z == 0) | (d1 == 0) | (d2 == 0);
}
BPMI_HD bool fe_is_zero(const fe &a) {
  fe t;
  fe_carry(t, a);
  return fe_is_zero_tight(t);
}
BPMI_HD bool fe_equal(const fe &a, const fe &b) {   // both tight
  fe d;
  fe_sub(d, a, b);
  return fe_is_zero(d);
}

// r = a^(p-2): 255 squarings + 15 multiplications (addition chain on the run
// structure of p - 2 = 2^256 - 2^32 - 979: blocks of 223, 22, 1, 1 ones ...)
BPMI_HD void fe_inv(fe &r, const fe &a) {
  fe x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t1;
  fe_sqr(x2, a); fe_mul(x2, x2, a);
  fe_sqr(x3, x2); fe_mul(x3, x3, a);
  x6 = x3; for (int j = 0; j < 3; j++) fe_sqr(x6, x6); fe_mul(x6, x6, x3);
  x9 = x6; for (int j = 0; j < 3; j++) fe_sqr(x9, x9); fe_mul(x9, x9, x3);
  x11 = x9; for (int j = 0; j < 2; j++) fe_sqr(x11, x11); fe_mul(x11, x11, x2);
  x22 = x11; for (int j = 0; j < 11; j++) fe_sqr(x22, x22); fe_mul(x22, x22, x11);
  x44 = x22; for (int j = 0; j < 22; j++) fe_sqr(x44, x44); fe_mul(x44, x44, x22);
  x88 = x44; for (int j = 0; j < 44; j++) fe_sqr(x88, x88); fe_mul(x88, x88, x44);
  x176 = x88; for (int j = 0; j < 88; j++) fe_sqr(x176, x176); fe_mul(x176, x176, x88);
  x220 = x176; for (int j = 0; j < 44; j++) fe_sqr(x220, x220); fe_mul(x220, x220, x44);
  x223 = x220; for (int j = 0; j < 3; j++) fe_sqr(x223, x223); fe_mul(x223, x223, x3);
  // p - 2 = 1^223 0 1^22 0000 101101
  t1 = x223; for (int j = 0; j < 23; j++) fe_sqr(t1, t1); fe_mul(t1, t1, x22);
  for (int j = 0; j < 5; j++) fe_sqr(t1, t1); fe_mul(t1, t1, a);
  for (int j = 0; j < 3; j++) fe_sqr(t1, t1); fe_mul(t1, t1, x2);
  for (int j = 0; j < 2; j++) fe_sqr(t1, t1); fe_mul(r, t1, a);
}

// r = a^((p+1)/4): a square root of a when a is a quadratic residue (p = 3 mod 4); the
// caller checks r^2 == a.  Same addition-chain prefix as fe_inv; (p+1)/4 = 1^223 0 1^22 0000 11 00.
BPMI_HD void fe_sqrt_candidate(fe &r, const fe &a) {
  fe x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t1;
  fe_sqr(x2, a); fe_mul(x2, x2, a);
  fe_sqr(x3, x2); fe_mul(x3, x3, a);
  x6 = x3; for (int j = 0; j < 3; j++) fe_sqr(x6, x6); fe_mul(x6, x6, x3);
  x9 = x6; for (int j = 0; j < 3; j++) fe_sqr(x9, x9); fe_mul(x9, x9, x3);
  x11 = x9; for (int j = 0; j < 2; j++) fe_sqr(x11, x11); fe_mul(x11, x11, x2);
  x22 = x11; for (int j = 0; j < 11; j++) fe_sqr(x22, x22); fe_mul(x22, x22, x11);
  x44 = x22; for (int j = 0; j < 22; j++) fe_sqr(x44, x44); fe_mul(x44, x44, x22);
  x88 = x44; for (int j = 0; j < 44; j++) fe_sqr(x88, x88); fe_mul(x88, x88, x44);
  x176 = x88; for (int j = 0; j < 88; j++) fe_sqr(x176, x176); fe_mul(x176, x176, x88);
  x220 = x176; for (int j = 0; j < 44; j++) fe_sqr(x220, x220); fe_mul(x220, x220, x44);
  x223 = x220; for (int j = 0; j < 3; j++) fe_sqr(x223, x223); fe_mul(x223, x223, x3);
  t1 = x223; for (int j = 0; j < 23; j++) fe_sqr(t1, t1); fe_mul(t1, t1, x22);
  for (int j = 0; j < 6; j++) fe_sqr(t1, t1); fe_mul(t1, t1, x2);
  fe_sqr(t1, t1); fe_sqr(r, t1);
}

}  // namespace bpmi
