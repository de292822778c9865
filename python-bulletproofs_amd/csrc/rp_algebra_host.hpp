// rp_algebra_host.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).  HOST code.
// The O(n m) scalar algebra of the range-proof provers and verifiers in native code: the reference does it with Python `ModP`
// objects in list comprehensions (src/rangeproofs/rangeproof_aggreg_prover.py:117-146 `_get_polynomial_coeffs` / `_final_compute`,
// rangeproof_prover.py:93-112; rangeproof_aggreg_verifier.py:96-108 `_getP`), which at the 16 384 generators of an aggregated
// 128 x 64-bit proof is most of the proof's wall time once the group operations are on the GPU.  No elliptic-curve arithmetic
// here: vectors of scalars mod q in, vectors of scalars out (32 bytes little-endian each), on `threads` host threads; the Python
// layer (rangeproofs/common.py) keeps an integer path with the same results for any other modulus, and
// tests/test_rp_algebra_cpu.py compares the two.
//   aL: one byte per bit (0 / 1); aR_i = aL_i - 1;  ypow_i = y^i;  zt_i = z^(2 + i / n) 2^(i % n) (aggregated) or z^2 2^i (m = 1)
#pragma once

namespace rpa {

using rp::Sq;
using rp::q_add;
using rp::q_mul;
using rp::q_small;
using rp::q_sub;

static inline void q_pow(Sq &r, const Sq &a, uint64_t e) {
  Sq acc = q_small(1), base = a;
  for (; e; e >>= 1) {
    if (e & 1) q_mul(acc, acc, base);
    q_mul(base, base, base);
  }
  r = acc;
}
struct Walk {                 // the per-index quantities, stepped from i to i + 1
  Sq ypow, zt, zblock, y, z;
  uint32_t n, e;
  bool aggregated;
  void start(uint64_t i, const Sq &y_, const Sq &z_, uint32_t n_, bool agg) {
    y = y_; z = z_; n = n_; aggregated = agg;
    q_pow(ypow, y, i);
    const uint64_t j = agg ? i / n : 0;
    e = agg ? (uint32_t)(i % n) : 0;
    q_pow(zblock, z, 2 + j);
    zt = zblock;
    const uint64_t dbl = agg ? e : i;                       // z^(2+j) 2^(i % n), or z^2 2^i for a single proof
    Sq two = q_small(2), p2;
    q_pow(p2, two, dbl);
    q_mul(zt, zt, p2);
  }
  void next() {
    q_mul(ypow, ypow, y);
    if (aggregated && ++e == n) { e = 0; q_mul(zblock, zblock, z); zt = zblock; }
    else q_add(zt, zt, zt);
  }
};
template <typename F> static inline void parallel(uint64_t count, int threads, F f) {
  if (threads < 1) threads = 1;
  if ((uint64_t)threads > count / 512 + 1) threads = (int)(count / 512 + 1);
  if (threads == 1) { f(0, (uint64_t)0, count); return; }
  hostpool::run(threads, [&](int t) { f(t, count * t / threads, count * (t + 1) / threads); });      // (host_pool.hpp: sleeping workers, not new threads)
}

}  // namespace rpa

extern "C" {

// t1 = sum sL_i (y^i (aR_i + z) + zt_i) + sum (aL_i - z) y^i sR_i ;  t2 = sum sL_i y^i sR_i      (_get_polynomial_coeffs)
int bpmi_rp_poly_coeffs(uint32_t n, uint32_t m, int aggregated, const uint8_t *aL, const uint8_t *sL, const uint8_t *sR, const uint8_t y[32],
                        const uint8_t z[32], int threads, uint8_t t1[32], uint8_t t2[32]) {
  if (!aL || !sL || !sR || !y || !z || !t1 || !t2 || !n || !m) return BPMI_E_ARG;
  const uint64_t nm = (uint64_t)n * m;
  rp::Sq Y, Z;
  rp::q_from_le(Y, y); rp::q_from_le(Z, z);
  std::vector<rp::Sq> p1(64, rp::q_small(0)), p2(64, rp::q_small(0));
  if (threads > 64) threads = 64;
  rpa::parallel(nm, threads, [&](int t, uint64_t lo, uint64_t hi) {
    rpa::Walk w;
    w.start(lo, Y, Z, n, aggregated != 0);
    rp::Sq a1 = rp::q_small(0), a2 = rp::q_small(0), one = rp::q_small(1), sl, sr, u, v, ysr;
    for (uint64_t i = lo; i < hi; i++, w.next()) {
      rp::q_from_le(sl, sL + 32 * i); rp::q_from_le(sr, sR + 32 * i);
      const rp::Sq al = rp::q_small(aL[i] & 1);
      rp::q_sub(u, al, one); rp::q_add(u, u, Z);           // aR_i + z
      rp::q_mul(u, u, w.ypow); rp::q_add(u, u, w.zt);
      rp::q_mul(u, u, sl); rp::q_add(a1, a1, u);
      rp::q_mul(ysr, w.ypow, sr);
      rp::q_sub(v, al, Z); rp::q_mul(v, v, ysr); rp::q_add(a1, a1, v);
      rp::q_mul(v, sl, ysr); rp::q_add(a2, a2, v);
    }
    p1[t] = a1; p2[t] = a2;
  });
  rp::Sq s1 = rp::q_small(0), s2 = rp::q_small(0);
  for (int t = 0; t < 64; t++) { rp::q_add(s1, s1, p1[t]); rp::q_add(s2, s2, p2[t]); }
  rp::q_to_le(t1, s1); rp::q_to_le(t2, s2);
  return BPMI_OK;
}

// l_i = aL_i - z + sL_i x ;  r_i = y^i (aR_i + z + sR_i x) + zt_i ;  t_hat = <l, r>                  (_final_compute)
// and, for P and the inner-product argument over the unscaled hs: yscale_i = y^-i ;  hsc_i = (z y^i + zt_i) y^-i
int bpmi_rp_final_vectors(uint32_t n, uint32_t m, int aggregated, const uint8_t *aL, const uint8_t *sL, const uint8_t *sR, const uint8_t y[32],
                          const uint8_t z[32], const uint8_t x[32], int threads, uint8_t *ls, uint8_t *rs, uint8_t t_hat[32], uint8_t *hsc, uint8_t *yscale) {
  if (!aL || !sL || !sR || !y || !z || !x || !ls || !rs || !t_hat || !hsc || !yscale || !n || !m) return BPMI_E_ARG;
  const uint64_t nm = (uint64_t)n * m;
  rp::Sq Y, Z, X, Yinv;
  rp::q_from_le(Y, y); rp::q_from_le(Z, z); rp::q_from_le(X, x);
  if (rp::q_is_zero(Y)) return BPMI_E_ARG;
  rp::q_inv(Yinv, Y);
  std::vector<rp::Sq> part(64, rp::q_small(0));
  if (threads > 64) threads = 64;
  rpa::parallel(nm, threads, [&](int t, uint64_t lo, uint64_t hi) {
    rpa::Walk w;
    w.start(lo, Y, Z, n, aggregated != 0);
    rp::Sq acc = rp::q_small(0), one = rp::q_small(1), sl, sr, l, r, u, ys;
    rpa::q_pow(ys, Yinv, lo);
    for (uint64_t i = lo; i < hi; i++, w.next()) {
      rp::q_from_le(sl, sL + 32 * i); rp::q_from_le(sr, sR + 32 * i);
      const rp::Sq al = rp::q_small(aL[i] & 1);
      rp::q_mul(l, sl, X); rp::q_add(l, l, al); rp::q_sub(l, l, Z);
      rp::q_mul(r, sr, X); rp::q_add(r, r, al); rp::q_sub(r, r, one); rp::q_add(r, r, Z);
      rp::q_mul(r, r, w.ypow); rp::q_add(r, r, w.zt);
      rp::q_to_le(ls + 32 * i, l); rp::q_to_le(rs + 32 * i, r);
      rp::q_mul(u, l, r); rp::q_add(acc, acc, u);
      rp::q_to_le(yscale + 32 * i, ys);
      rp::q_mul(u, Z, w.ypow); rp::q_add(u, u, w.zt); rp::q_mul(u, u, ys);
      rp::q_to_le(hsc + 32 * i, u);
      rp::q_mul(ys, ys, Yinv);
    }
    part[t] = acc;
  });
  rp::Sq s = rp::q_small(0);
  for (int t = 0; t < 64; t++) rp::q_add(s, s, part[t]);
  rp::q_to_le(t_hat, s);
  return BPMI_OK;
}

// the verifier's half (rangeproof_verifier.py:55-99, rangeproof_aggreg_verifier.py:55-108): yscale_i = y^-i,
// hsc_i = (z y^i + zt_i) y^-i, ysum = sum_{i < n m} y^i
int bpmi_rp_verifier_vectors(uint32_t n, uint32_t m, int aggregated, const uint8_t y[32], const uint8_t z[32], int threads, uint8_t *hsc, uint8_t *yscale,
                             uint8_t ysum[32]) {
  if (!y || !z || !hsc || !yscale || !ysum || !n || !m) return BPMI_E_ARG;
  const uint64_t nm = (uint64_t)n * m;
  rp::Sq Y, Z, Yinv;
  rp::q_from_le(Y, y); rp::q_from_le(Z, z);
  if (rp::q_is_zero(Y)) return BPMI_E_ARG;
  rp::q_inv(Yinv, Y);
  std::vector<rp::Sq> part(64, rp::q_small(0));
  if (threads > 64) threads = 64;
  rpa::parallel(nm, threads, [&](int t, uint64_t lo, uint64_t hi) {
    rpa::Walk w;
    w.start(lo, Y, Z, n, aggregated != 0);
    rp::Sq acc = rp::q_small(0), u, ys;
    rpa::q_pow(ys, Yinv, lo);
    for (uint64_t i = lo; i < hi; i++, w.next()) {
      rp::q_add(acc, acc, w.ypow);
      rp::q_to_le(yscale + 32 * i, ys);
      rp::q_mul(u, Z, w.ypow); rp::q_add(u, u, w.zt); rp::q_mul(u, u, ys);
      rp::q_to_le(hsc + 32 * i, u);
      rp::q_mul(ys, ys, Yinv);
    }
    part[t] = acc;
  });
  rp::Sq s = rp::q_small(0);
  for (int t = 0; t < 64; t++) rp::q_add(s, s, part[t]);
  rp::q_to_le(ysum, s);
  return BPMI_OK;
}

}  // extern "C"
