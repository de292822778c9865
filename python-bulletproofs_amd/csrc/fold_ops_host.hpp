// fold_ops_host.hpp -- part of libbpmi (included by bpmi.hip before the kernels; plain C++, also compiled for the host by
// tests/csrc_host/host_shim.cpp).  Host-side digit tables of the inner-product prover's 16-way generator fold
// (/root/reference/src/innerproduct/inner_product_prover.py:107-108, sixteen folds at once): the width-4 non-adjacent form of a scalar,
// and the OPERATION LIST of the GLV ladder k_ec_multifold_w4g runs (point_kernels.hpp).
#pragma once
#include <stdint.h>
#include <string.h>

#include "scalar.hpp"

namespace bpmi {

// width-4 NAF of the little-endian scalar k32 (only its low len - 8 bits may be set): dg[pos] in {0, +-1, +-3, +-5, +-7}, at most one
// non-zero digit in four positions; top = the highest position any call has used so far
static void host_wnaf4(const uint8_t k32[32], signed char *dg, int &top, int len = 264) {
  u32 w[9];
  memcpy(w, k32, 32);
  w[8] = 0;
  memset(dg, 0, (size_t)len);
  for (int pos = 0; pos < len - 4; pos++) {
    if (w[0] & 1u) {
      int d = (int)(w[0] & 15u);                     // k mod 16
      if (d > 8) d -= 16;                            // odd digit in [-7, 7]
      dg[pos] = (signed char)d;
      // k -= d
      if (d > 0) { u64 br = (u64)d; for (int i = 0; i < 9 && br; i++) { const u64 t = (u64)w[i] - br; w[i] = (u32)t; br = (t >> 32) & 1u; } }
      else { u64 c = (u64)(-d); for (int i = 0; i < 9 && c; i++) { const u64 t = (u64)w[i] + c; w[i] = (u32)t; c = t >> 32; } }
      if (pos > top) top = pos;
    }
    for (int i = 0; i < 8; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
    w[8] >>= 1;
  }
}

// "double n_dbl times, then add (-)(j-th odd multiple) of (lambda?) point row / 2":
//   op = n_dbl | row << 8 | j << 13 | neg << 16      row = 2 t + half (half 1 = the half that multiplies lambda P), j = 0: P .. 3: 7P
//   tail = doublings after the last addition
#define WNAFG_MAXOPS 2048
#define WNAFG_MAXK 16
struct WnafG { u32 nops, tail; u32 op[WNAFG_MAXOPS]; };
// the K (<= 16) coefficients in two 128-bit halves each (k = k1 + k2 lambda, glv_split), width-4 NAF per half (a negative half with
// its digits negated), positions walked from the top: false only if the list overflows (cannot happen: 32 rows x 34 digits < 2048)
static inline bool glv_fold_ops(WnafG &hw, const sc *coef, u32 K) {
  static thread_local signed char dg[2 * WNAFG_MAXK][136];
  int top = -1;
  if (K > WNAFG_MAXK) return false;
  for (u32 t = 0; t < K; t++) {
    u32 k1[4], k2[4];
    bool n1, n2;
    glv_split(k1, n1, k2, n2, coef[t]);
    for (int hf = 0; hf < 2; hf++) {
      uint8_t k32[32] = {0};
      memcpy(k32, hf ? k2 : k1, 16);
      signed char *row = dg[2 * t + hf];
      host_wnaf4(k32, row, top, 136);
      if (hf ? n2 : n1) for (int q = 0; q < 136; q++) row[q] = (signed char)-row[q];
    }
  }
  hw.nops = 0;
  u32 ndbl = 0;                                        // nothing to double before the first addition
  for (int pos = top; pos >= 0; pos--) {
    for (u32 r = 0; r < 2 * K; r++) {
      const int d = dg[r][pos];
      if (!d) continue;
      if (hw.nops >= WNAFG_MAXOPS) return false;
      const u32 mag = (u32)(d < 0 ? -d : d);
      hw.op[hw.nops++] = ndbl | (r << 8) | ((mag >> 1) << 13) | ((d < 0 ? 1u : 0u) << 16);
      ndbl = 0;
    }
    if (hw.nops) ndbl++;                               // the doubling that moves on to position pos - 1
  }
  hw.tail = hw.nops ? ndbl - 1u : 0u;                  // (the last position has no doubling after it)
  return true;
}

}  // namespace bpmi
