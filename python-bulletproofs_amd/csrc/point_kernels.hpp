// point_kernels.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).
// Batched point kernels: scalar multiplication, shared-scalar folds, deferred-fold helpers of the IPA (/root/reference/src/innerproduct/inner_product_prover.py:107-108, src/utils/utils.py:43-44).
#pragma once

// ------------------------------------------------------------------------------------
// batched point kernels
// ------------------------------------------------------------------------------------
// out[i] = k_i * P_i   (left-to-right double-and-add in Jacobian coordinates; the scalars
// differ per lane, so lanes diverge on the addition only)
__global__ void __launch_bounds__(256, 3) k_ec_mul_batch(const u32 *__restrict__ pts, const u32 *__restrict__ scs, u32 n, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine P;
  load_affine(P, pts + 16ull * i);
  sc s;
  load_words8(s.v, scs + 8ull * i);
  sc_reduce_once(s);
  const bool neg = sc_is_high(s);
  if (neg) sc_neg(s, s);
  jac acc;
  jac_set_inf(acc);
  const bool pinf = affine_is_inf(P);
  for (int word = 7; word >= 0; word--) {
    // static word selection keeps s.v[] in registers
    u32 wv = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (k == word) wv = s.v[k];
    for (int bit = 31; bit >= 0; bit--) {
      jac_dbl(acc, acc);
      if (((wv >> bit) & 1u) && !pinf) jac_madd_signed(acc, P.x, P.y, neg);
    }
  }
  affine r;
  jac_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16(out + 16ull * i, w16);
}

// The same product for many points at once, twice as fast (20 -> 10 ms per 2^20; the path of rangeproof_prover.py:77's
// hsp and of every test / bench input):
//  * GLV: k = k1 + k2 lambda with |k1|, |k2| < 2^128 (scalar.hpp glv_split) and lambda (x, y) = (beta x, y): 128 doublings
//    instead of 256;
//  * every lane of a wave executes every addition anyway (the scalars differ per lane), so sparse recodings buy nothing:
//    FIXED windows with SIGNED ODD digits instead -- for an odd magnitude k < 2^129, E = (k - 1) / 2 + 2^128 and
//    k = sum_i (2 e_i - 7) 8^i with e_i the 43 three-bit windows of E: every digit is one of +-1, +-3, +-5, +-7, none is
//    zero, 43 additions per half-scalar.  An even magnitude uses k | 1 and takes the surplus point off at the end;
//  * 3P, 5P, 7P of every point in AFFINE form from k_ec_odd_multiples (one inversion per 16 points), so every addition is
//    a mixed one; the multiples of lambda P are the same entries with beta x.
//   tab: entry j (0: 3P, 1: 5P, 2: 7P) of point i at tab + ((j * n) + i) * 18 limbs
#define MULB_STEPS 43
#define MULB_MIN_N 32768      // below: both forms are one latency-bound wave per SIMD or less, and the old one is a single launch
__device__ __forceinline__ void mulb_window_register(u32 R[5], const u32 k[4]) {      // ((k & ~1) << 30) | 2^159: E, left-aligned
  const u32 k0 = k[0] & ~1u;
  R[0] = k0 << 30;
  R[1] = (k[1] << 30) | (k0 >> 2);
  R[2] = (k[2] << 30) | (k[1] >> 2);
  R[3] = (k[3] << 30) | (k[2] >> 2);
  R[4] = (k[3] >> 2) | 0x80000000u;
}
__device__ __forceinline__ u32 mulb_next_window(u32 R[5]) {                           // the top three bits, then R <<= 3
  const u32 e = R[4] >> 29;
#pragma unroll
  for (int w = 4; w > 0; w--) R[w] = (R[w] << 3) | (R[w - 1] >> 29);
  R[0] <<= 3;
  return e;
}
// acc += +-(|d| P) or +-(|d| lambda P), d = 2e - 7
__device__ __forceinline__ void mulb_add_digit(jac &acc, u32 e, bool neg_scalar, bool endo, const affine &P, const u32 *__restrict__ tab, u32 n, u32 i) {
  const bool neg_digit = e < 4u;
  const u32 idx = neg_digit ? 3u - e : e - 4u;                          // (|d| - 1) / 2
  const uint2 *q = reinterpret_cast<const uint2 *>(tab + ((u64)(idx ? idx - 1u : 0u) * n + i) * 18u);
  u32 w[18];
#pragma unroll
  for (int l = 0; l < 9; l++) { const uint2 t = q[l]; w[2 * l] = t.x; w[2 * l + 1] = t.y; }
  fe x, y;
#pragma unroll
  for (int l = 0; l < 9; l++) { x.v[l] = idx ? w[l] : P.x.v[l]; y.v[l] = idx ? w[9 + l] : P.y.v[l]; }
  if (endo) { fe bx; fe_mul_beta(bx, x); x = bx; }                      // wave-uniform
  jac_madd_signed(acc, x, y, neg_digit != neg_scalar);
}
__global__ void __launch_bounds__(256, 3) k_ec_mul_batch_glv(const u32 *__restrict__ pts, const u32 *__restrict__ scs, u32 n, const u32 *__restrict__ tab,
                                                            u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine P;
  load_affine(P, pts + 16ull * i);
  u32 w16[16];
  if (affine_is_inf(P)) {                                               // k * identity
#pragma unroll
    for (int k = 0; k < 16; k++) w16[k] = 0;
    store_words16(out + 16ull * i, w16);
    return;
  }
  sc s;
  load_words8(s.v, scs + 8ull * i);
  sc_reduce_once(s);
  u32 k1[4], k2[4], R1[5], R2[5];
  bool n1, n2;
  glv_split(k1, n1, k2, n2, s);
  const bool even1 = !(k1[0] & 1u), even2 = !(k2[0] & 1u);
  mulb_window_register(R1, k1);
  mulb_window_register(R2, k2);
  jac acc;
  jac_set_inf(acc);
  for (int step = 0; step < MULB_STEPS; step++) {
    if (step) { jac_dbl(acc, acc); jac_dbl(acc, acc); jac_dbl(acc, acc); }
    mulb_add_digit(acc, mulb_next_window(R1), n1, false, P, tab, n, i);
    mulb_add_digit(acc, mulb_next_window(R2), n2, true, P, tab, n, i);
  }
  // (k | 1) was used for an even magnitude: one sign(k) P (lambda P) too many
  {
    fe y;
#pragma unroll
    for (int l = 0; l < 9; l++) y.v[l] = P.y.v[l];
    jac t = acc;
    jac_madd_signed(t, P.x, y, !n1);
    if (even1) acc = t;
    fe bx;
    fe_mul_beta(bx, P.x);
    t = acc;
    jac_madd_signed(t, bx, y, !n2);
    if (even2) acc = t;
  }
  affine r;
  jac_to_affine(r, acc);
  affine_to_words(w16, r);
  store_words16(out + 16ull * i, w16);
}

struct Sc2 { u32 k1[8]; u32 k2[8]; };

// Non-adjacent forms of the two shared scalars, computed once on the host: bit i of nz*
// says digit i is non-zero, bit i of sg* says it is -1.  257 positions each.
struct NafPair { u32 nz1[9], sg1[9], nz2[9], sg2[9]; int top; };
static void host_naf(const uint8_t k32[32], u32 nz[9], u32 sg[9], int &top) {
  u32 w[9];
  memcpy(w, k32, 32);
  w[8] = 0;
  for (int i = 0; i < 9; i++) nz[i] = sg[i] = 0;
  for (int pos = 0; pos < 257; pos++) {
    if (w[0] & 1u) {
      const bool minus = (w[0] & 3u) == 3u;          // k mod 4 == 3 -> digit -1, k += 1
      nz[pos >> 5] |= 1u << (pos & 31);
      if (minus) {
        sg[pos >> 5] |= 1u << (pos & 31);
        for (int i = 0; i < 9; i++) { if (++w[i] != 0) break; }
      } else {
        w[0] &= ~1u;
      }
      if (pos > top) top = pos;
    }
    for (int i = 0; i < 8; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
    w[8] >>= 1;
  }
}

// out[i] = k1 * P1_i + k2 * P2_i with k1, k2 shared by all i (the generator fold,
// inner_product_prover.py:107-108).  Jacobian ladder driven by the NAF digits: every
// branch is on a kernel argument, hence wave-uniform; only mixed additions; the two input
// points of each thread are parked in LDS ([word][thread], conflict-free) to keep the
// register count at the ladder's working set.
struct LincombJob { const u32 *p1, *p2; u32 *out; u32 n; };
// Two independent jobs share one launch (threads [0, A.n) do job A, the next B.n do job
// B): a ladder thread is ~2 ms of serial issue, so per-launch latency, not throughput,
// bounds the small rounds of the IPA -- g and h are therefore folded together.
__global__ void __launch_bounds__(256, 3) k_ec_lincomb2(LincombJob ja, NafPair nfa, LincombJob jb, NafPair nfb) {
  __shared__ u32 s_pts[36 * 256];
  const u32 tid = threadIdx.x;
  u32 i = blockIdx.x * blockDim.x + tid;
  const bool second = i >= ja.n;          // may differ inside one wave only at the seam
  if (second) i -= ja.n;
  const u32 n = second ? jb.n : ja.n;
  if (i >= n) return;
  const u32 *p1 = second ? jb.p1 : ja.p1;
  const u32 *p2 = second ? jb.p2 : ja.p2;
  u32 *out = second ? jb.out : ja.out;
  bool inf1, inf2;
  {
    affine A;
    load_affine(A, p1 + 16ull * i);
    inf1 = affine_is_inf(A);
#pragma unroll
    for (int k = 0; k < 9; k++) { s_pts[k * 256 + tid] = A.x.v[k]; s_pts[(9 + k) * 256 + tid] = A.y.v[k]; }
    load_affine(A, p2 + 16ull * i);
    inf2 = affine_is_inf(A);
#pragma unroll
    for (int k = 0; k < 9; k++) { s_pts[(18 + k) * 256 + tid] = A.x.v[k]; s_pts[(27 + k) * 256 + tid] = A.y.v[k]; }
  }
  jac acc;
  jac_set_inf(acc);
  const int top = nfa.top > nfb.top ? nfa.top : nfb.top;
  for (int pos = top; pos >= 0; pos--) {
    jac_dbl(acc, acc);
    const u32 m = 1u << (pos & 31);
    const int wd = pos >> 5;
    // one inlined copy of the addition serves both points (the loop is kept rolled)
#pragma unroll 1
    for (int which = 0; which < 2; which++) {
      const u32 nzw = second ? (which ? nfb.nz2[wd] : nfb.nz1[wd]) : (which ? nfa.nz2[wd] : nfa.nz1[wd]);
      const u32 sgw = second ? (which ? nfb.sg2[wd] : nfb.sg1[wd]) : (which ? nfa.sg2[wd] : nfa.sg1[wd]);
      const bool isinf = which ? inf2 : inf1;
      if ((nzw & m) && !isinf) {
        fe x, y;
        const u32 base = which ? 18u * 256u : 0u;
#pragma unroll
        for (int k = 0; k < 9; k++) { x.v[k] = s_pts[base + k * 256 + tid]; y.v[k] = s_pts[base + (9 + k) * 256 + tid]; }
        jac_madd_signed(acc, x, y, (sgw & m) != 0);
      }
    }
  }
  affine r;
  jac_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16(out + 16ull * i, w16);
}

// ---- deferred generator folding (IPA) ----------------------------------------------------
// After d deferred folds the logical generator i (i < m) is sum_t coef[t] * G[i + t*m],
// t < 2^d, over the UNFOLDED base array G of length M = m << d; the newest fold is the
// least significant bit of t:  coef'[2t + s] = coef[t] * (s ? hi_factor : lo_factor).
__global__ void __launch_bounds__(256) k_ipa_coef_update(const u32 *cg, const u32 *ch, Sc2 x_xinv, u32 K, u32 *cg2, u32 *ch2) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;     // new index in [0, 2K)
  if (j >= 2u * K) return;
  sc X, XI, c, r;
#pragma unroll
  for (int k = 0; k < 8; k++) { X.v[k] = x_xinv.k1[k]; XI.v[k] = x_xinv.k2[k]; }
  // g' = x^-1 g_lo + x g_hi ;  h' = x h_lo + x^-1 h_hi   (inner_product_prover.py:107-108)
  load_words8(c.v, cg + 8ull * (j >> 1));
  sc_mul(r, c, (j & 1u) ? X : XI);
  store_words8(cg2 + 8ull * j, r.v);
  load_words8(c.v, ch + 8ull * (j >> 1));
  sc_mul(r, c, (j & 1u) ? XI : X);
  store_words8(ch2 + 8ull * j, r.v);
}
// scalars of the L (right = 0) or R (right = 1) MSM over the unfolded bases:
//   L = <a_lo, g_hi> + <b_hi, h_lo>,  R = <a_hi, g_lo> + <b_lo, h_hi>   (:98-99)
// both in one launch: blockIdx.y = right
struct ExpandOut { u32 *eg[2], *eh[2]; };
__global__ void __launch_bounds__(256) k_ipa_expand(const u32 *__restrict__ a, const u32 *__restrict__ b, const u32 *__restrict__ cg,
                                                    const u32 *__restrict__ ch, const u32 *__restrict__ hscale, u32 M, u32 logm, ExpandOut o) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  const int right = (int)blockIdx.y;
  u32 *__restrict__ eg = o.eg[blockIdx.y], *__restrict__ eh = o.eh[blockIdx.y];
  const u32 m = 1u << logm, half = m >> 1;
  const u32 i = k & (m - 1u), t = k >> logm;
  const bool hi = i >= half;
  sc z;
#pragma unroll
  for (int q = 0; q < 8; q++) z.v[q] = 0;
  sc rg = z, rh = z;
  // g side uses the g-half OPPOSITE to the a-half: L pairs a_lo with g_hi
  if (hi != (right != 0)) {
    sc av, c;
    load_words8(av.v, a + 8ull * (right ? half + i : i - half));
    load_words8(c.v, cg + 8ull * t);
    sc_mul(rg, av, c);
  }
  if (hi == (right != 0)) {
    sc bv, c;
    load_words8(bv.v, b + 8ull * (right ? i - half : half + i));
    load_words8(c.v, ch + 8ull * t);
    sc_mul(rh, bv, c);
    if (hscale) {                       // the h generators are hscale[k] * H[k] (never materialised)
      load_words8(c.v, hscale + 8ull * k);
      sc_mul(rh, rh, c);
    }
  }
  store_words8(eg + 8ull * k, rg.v);
  store_words8(eh + 8ull * k, rh.v);
}
// scalars that pick the current (folded) generator number `pos` out of the unfolded bases:
//   g'[pos] = sum_t cg[t] * G[pos + t*m],  h'[pos] = sum_t ch[t] * hscale[.] * H[pos + t*m]
__global__ void __launch_bounds__(256) k_ipa_export_scalars(const u32 *__restrict__ cg, const u32 *__restrict__ ch,
                                                            const u32 *__restrict__ hscale, u32 M, u32 logm, u32 pos,
                                                            u32 *__restrict__ eg, u32 *__restrict__ eh) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  const u32 i = k & ((1u << logm) - 1u), t = k >> logm;
  sc rg, rh;
#pragma unroll
  for (int q = 0; q < 8; q++) rg.v[q] = rh.v[q] = 0;
  if (i == pos) {
    load_words8(rg.v, cg + 8ull * t);
    load_words8(rh.v, ch + 8ull * t);
    if (hscale) {
      sc c;
      load_words8(c.v, hscale + 8ull * k);
      sc_mul(rh, rh, c);
    }
  }
  store_words8(eg + 8ull * k, rg.v);
  store_words8(eh + 8ull * k, rh.v);
}
// ---- the second fold: through PRODUCTS (round 4) ---------------------------------------------------------------------------
// The shared-scalar ladder above is one thread per OUTPUT running 256 doublings + K x 51 additions in a row: for the 2 x 4 096
// outputs of a fold of 2^16 bases that is 128 waves and ~2 ms of pure latency -- why bases below 2^18 points were never folded and
// rounds 5 .. 20 of a 2^20-element proof all ran pairs of 65 537-pair MSMs (0.56 ms each).  Here every TERM is a thread:
//   k_ipa_fold_scalars   e[k] = coef[k >> log m] (x hscale[k]): the scalar of base point k in its output
//   bpmi_ec_mul_batch    prod[k] = e[k] * base[k]  (GLV + fixed signed windows: 126 doublings + 88 additions deep, 2 M threads)
//   k_ec_sum_strided     out[i] = sum_t prod[i + t m]: K mixed additions and one inversion per output
// ~0.5 ms for 2 x 2^16 bases; afterwards the argument runs on 4 096 generators per side and its MSMs on the one-launch kernel.
__global__ void __launch_bounds__(256) k_ipa_fold_scalars(const u32 *__restrict__ cg, const u32 *__restrict__ ch, const u32 *__restrict__ hscale, u32 M, u32 logm,
                                                          u32 *__restrict__ eg, u32 *__restrict__ eh) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  const u32 t = k >> logm;
  sc rg, rh;
  load_words8(rg.v, cg + 8ull * t);
  load_words8(rh.v, ch + 8ull * t);
  if (hscale) {
    sc c;
    load_words8(c.v, hscale + 8ull * k);
    sc_mul(rh, rh, c);
  }
  store_words8(eg + 8ull * k, rg.v);
  store_words8(eh + 8ull * k, rh.v);
}
// prod: [side 0: M points | side 1: M points] (affine, wire form); out_a[i] / out_b[i] = sum over t < K of prod[side M + i + t m]
__global__ void __launch_bounds__(256) k_ec_sum_strided(const u32 *__restrict__ prod, u32 m, u32 K, u32 *__restrict__ out_a, u32 *__restrict__ out_b) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = i >= m;
  if (second) i -= m;
  if (i >= m) return;
  const u32 *base = prod + (second ? 16ull * m * K : 0ull);
  xyzz acc;
  xyzz_set_inf(acc);
#pragma unroll 1
  for (u32 t = 0; t < K; t++) {
    affine P;
    load_affine(P, base + 16ull * ((u64)i + (u64)t * m));
    xyzz_madd_signed(acc, P, false);
  }
  affine r;
  xyzz_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16((second ? out_b : out_a) + 16ull * i, w16);
}

// ---- the product fold with SHARED scalars: GLV halves, two terms per thread (round 4) ------------------------------------------
// When the K coefficients of a fold are the same for every output (no per-generator scale rides along), the products need no
// per-lane recoding: every coefficient is split on the host into two 128-bit halves (k = k1 + k2 lambda, scalar.hpp glv_split),
// their non-adjacent forms go to the device as bit masks, and every branch is wave-uniform.  Thread (side, group, i) adds up TWO
// terms of output i -- four half-scalars over P, lambda P (= (beta x, y)) of two base points held in registers: 128 doublings
// + ~172 mixed additions, against 126 + 88 PER TERM on the per-lane product path (k_ec_mul_batch_glv) -- and leaves an XYZZ
// partial; k_ec_sum_partials adds the K / 2 partials of an output and makes it affine.
//   2 x 2^16 bases -> 2 x 4 096 generators: 0.9 ms against 1.5 (profiles/r04_C3_product_fold_shared_scalars.txt)
#define GLVF_MAXK 32
#define GLVF_TERMS 2
struct GlvFoldK { u32 nz[2][2 * GLVF_MAXK][5]; u32 sg[2][2 * GLVF_MAXK][5]; int top; };      // [side][2 t + half][160 bits]
__global__ void __launch_bounds__(256, 2) k_ec_fold_glv(const u32 *__restrict__ base_a, const u32 *__restrict__ base_b, u32 m, u32 K,
                                                        const GlvFoldK *__restrict__ dk, u32 *__restrict__ partial) {
  const u32 G = K / GLVF_TERMS;
  const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= 2u * G * m) return;
  const u32 i = tid % m, grp = (tid / m) % G, side = tid / (m * G);          // m is a multiple of 64: side and group are wave-uniform
  const u32 *base = side ? base_b : base_a;
  // (two terms, written out: arrays of points indexed in an unrolled loop went to scratch)
  affine P0, P1;
  fe bx0, bx1;
  load_affine(P0, base + 16ull * ((u64)i + (u64)(grp * GLVF_TERMS) * m));
  load_affine(P1, base + 16ull * ((u64)i + (u64)(grp * GLVF_TERMS + 1u) * m));
  const bool inf0 = affine_is_inf(P0), inf1 = affine_is_inf(P1);
  fe_mul_beta(bx0, P0.x); fe_carry(bx0, bx0);
  fe_mul_beta(bx1, P1.x); fe_carry(bx1, bx1);
  const u32 row0 = 2u * (grp * GLVF_TERMS);
  xyzz acc;
  xyzz_set_inf(acc);
  for (int pos = dk->top; pos >= 0; pos--) {
    xyzz_dbl(acc, acc);
    const u32 msk = 1u << (pos & 31);
    const int wd = pos >> 5;
#pragma unroll 1
    for (u32 r = 0; r < 4u; r++) {                       // ONE inlined copy of the addition; r is wave-uniform
      if (!(dk->nz[side][row0 + r][wd] & msk) || (r < 2u ? inf0 : inf1)) continue;
      affine Q;
#pragma unroll
      for (int l = 0; l < 9; l++) {
        Q.x.v[l] = r == 0u ? P0.x.v[l] : (r == 1u ? bx0.v[l] : (r == 2u ? P1.x.v[l] : bx1.v[l]));
        Q.y.v[l] = r < 2u ? P0.y.v[l] : P1.y.v[l];
      }
      xyzz_madd_signed(acc, Q, (dk->sg[side][row0 + r][wd] & msk) != 0);
    }
  }
  xyzz_store_g(partial + (u64)tid * XYZZ_WORDS, acc);
}
// out_a[i] / out_b[i] = sum over the G partials of output i (partial index = (side G + grp) m + i), canonical affine
__global__ void __launch_bounds__(256) k_ec_sum_partials(const u32 *__restrict__ partial, u32 m, u32 G, u32 *__restrict__ out_a, u32 *__restrict__ out_b) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = i >= m;
  if (second) i -= m;
  if (i >= m) return;
  xyzz acc;
  xyzz_load_g(acc, partial + ((u64)(second ? G : 0u) * m + i) * XYZZ_WORDS);
#pragma unroll 1
  for (u32 grp = 1; grp < G; grp++) {
    xyzz x;
    xyzz_load_g(x, partial + ((u64)((second ? G : 0u) + grp) * m + i) * XYZZ_WORDS);
    xyzz_add(acc, acc, x);
  }
  affine r;
  xyzz_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16((second ? out_b : out_a) + 16ull * i, w16);
}

// materialise 2^d-way folded generators: out[i] = sum_t coef[t] * G[i + t*m], i < m, as an
// interleaved NAF ladder (shared scalars -> wave-uniform branches); two jobs (g and h) per launch
#define MULTIFOLD_MAXK 16
struct NafK { u32 nz[MULTIFOLD_MAXK][9]; u32 sg[MULTIFOLD_MAXK][9]; int top; };
struct MultifoldJob { const u32 *base; u32 *out; };
__global__ void __launch_bounds__(256, 3) k_ec_multifold(MultifoldJob ja, MultifoldJob jb, const NafK *__restrict__ nfa, const NafK *__restrict__ nfb,
                                                         u32 m, u32 K) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = i >= m;
  if (second) i -= m;
  if (i >= m) return;
  const u32 *base = second ? jb.base : ja.base;
  u32 *out = second ? jb.out : ja.out;
  const NafK *nf = second ? nfb : nfa;
  jac acc;
  jac_set_inf(acc);
  const int top = nf->top;
  for (int pos = top; pos >= 0; pos--) {
    jac_dbl(acc, acc);
    const u32 msk = 1u << (pos & 31);
    const int wd = pos >> 5;
#pragma unroll 1
    for (u32 t = 0; t < K; t++) {
      if (nf->nz[t][wd] & msk) {
        affine P;
        load_affine(P, base + 16ull * ((u64)i + (u64)t * m));
        if (!affine_is_inf(P)) jac_madd_signed(acc, P.x, P.y, (nf->sg[t][wd] & msk) != 0);
      }
    }
  }
  affine r;
  jac_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16(out + 16ull * i, w16);
}

// ---- the same fold with width-4 non-adjacent forms ------------------------------------------------------------------
// A plain NAF has one non-zero digit in three, a width-4 NAF (digits 0, +-1, +-3, +-5, +-7) one in five: 16 x 51 instead
// of 16 x 86 mixed additions per output -- if 3P, 5P, 7P of every base point exist in AFFINE form.  They are built
// once per fold by k_ec_odd_multiples (2P, 3P = 2P + P, 4P, 5P = 4P + P, 6P = 2 (3P), 7P = 6P + P: three doublings and
// three mixed additions per point in Jacobian coordinates, then ONE inversion per thread for the 48 Z coordinates of its
// 16 points -- Montgomery's trick, prefix products in a scratch column) and read by k_ec_multifold_w4.
//   tab: entry j (0: 3P, 1: 5P, 2: 7P) of point k at tab + ((j * npts) + k) * 18 limbs (x, y; all zero = identity)
//   scratch: 4 x 9 limbs (X, Y, Z, prefix) per (point, j), column-major per thread
#define ODDMUL_PER_THREAD 16        // the fold: 2^21 points, 131 072 threads
#define ODDMUL_PER_THREAD_MULB 4    // k_ec_mul_batch_glv's slices of 196 608 points: 49 152 threads (16 per thread left most SIMDs idle)
//   tabx (optional, the GLV ladder k_ec_multifold_w4g): beta x of P, 3P, 5P, 7P -- the x of lambda (jP) = (beta x, y) -- entry j
//   (0: P .. 3: 7P) of point k at tabx + ((j * npts) + k) * 9 limbs
template <int PER> __global__ void __launch_bounds__(256, 3) k_ec_odd_multiples(const u32 *__restrict__ base_a, const u32 *__restrict__ base_b, u32 npts,
                                                            u32 *__restrict__ tab_a, u32 *__restrict__ tab_b, u32 *__restrict__ scratch,
                                                            u32 *__restrict__ tabx_a = nullptr, u32 *__restrict__ tabx_b = nullptr) {
  const u32 nthreads_per = (npts + PER - 1) / PER;
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = tid >= nthreads_per;
  if (second) tid -= nthreads_per;
  if (tid >= nthreads_per || (second && !base_b)) return;              // base_b == nullptr: one array only (k_ec_mul_batch_glv)
  const u32 *base = second ? base_b : base_a;
  u32 *tab = second ? tab_b : tab_a;
  u32 *tabx = second ? tabx_b : tabx_a;
  const u32 col = (second ? nthreads_per : 0u) + tid;                 // scratch column of this thread
  const u32 ncols = 2u * nthreads_per;
  auto rec = [&](u32 e) { return scratch + ((u64)e * ncols + col) * 36u; };      // e = r * 3 + j
  fe pref;
  fe_set_one(pref);
  for (u32 r = 0; r < PER; r++) {
    const u32 k = tid + r * nthreads_per;                               // strided: consecutive threads read consecutive points
    affine P;
    if (k < npts) load_affine(P, base + 16ull * k); else { fe_set_zero(P.x); fe_set_zero(P.y); }
    const bool inf = affine_is_inf(P);
    if (tabx && k < npts) {
      fe bx;
      fe_mul_beta(bx, P.x); fe_carry(bx, bx);
#pragma unroll
      for (int l = 0; l < 9; l++) tabx[(u64)k * 9u + l] = bx.v[l];
    }
    jac two, m[3];
    if (!inf) {
      jac one; one.X = P.x; one.Y = P.y; fe_set_one(one.Z);
      jac_dbl(two, one);
      m[0] = two; jac_madd(m[0], P.x, P.y);                             // 3P
      jac four; jac_dbl(four, two);
      m[1] = four; jac_madd(m[1], P.x, P.y);                            // 5P
      jac six; jac_dbl(six, m[0]);
      m[2] = six; jac_madd(m[2], P.x, P.y);                             // 7P
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      if (inf) { fe_set_zero(m[j].X); fe_set_zero(m[j].Y); fe_set_one(m[j].Z); }      // Z = 1 keeps the product chain alive
      u32 *q = rec(r * 3 + j);
#pragma unroll
      for (int l = 0; l < 9; l++) { q[l] = m[j].X.v[l]; q[9 + l] = m[j].Y.v[l]; q[18 + l] = m[j].Z.v[l]; q[27 + l] = pref.v[l]; }
      fe_mul(pref, pref, m[j].Z);
    }
  }
  fe inv;
  fe_inv(inv, pref);                                                     // 1 / (product of all 48 Z)
  for (int e = PER * 3 - 1; e >= 0; e--) {
    const u32 r = (u32)e / 3u, j = (u32)e % 3u, k = tid + r * nthreads_per;
    const u32 *q = rec((u32)e);
    fe X, Y, Z, pj, zi, zi2, zi3, x, y;
#pragma unroll
    for (int l = 0; l < 9; l++) { X.v[l] = q[l]; Y.v[l] = q[9 + l]; Z.v[l] = q[18 + l]; pj.v[l] = q[27 + l]; }
    fe_mul(zi, inv, pj);                                                 // 1 / Z_e
    fe_mul(inv, inv, Z);
    fe_sqr(zi2, zi);
    fe_mul(zi3, zi2, zi);
    fe_mul(x, X, zi2);
    fe_mul(y, Y, zi3);
    fe_carry(x, x); fe_carry(y, y);                                      // tight limbs: the ladder negates y lazily
    if (k < npts) {
      u32 *o = tab + ((u64)j * npts + k) * 18u;
#pragma unroll
      for (int l = 0; l < 9; l++) { o[l] = x.v[l]; o[9 + l] = y.v[l]; }   // the identity gives X = Y = 0 -> (0, 0)
      if (tabx) {
        fe bx;
        fe_mul_beta(bx, x); fe_carry(bx, bx);
        u32 *ox = tabx + ((u64)(j + 1u) * npts + k) * 9u;
#pragma unroll
        for (int l = 0; l < 9; l++) ox[l] = bx.v[l];
      }
    }
  }
}
// digits of the K (<= 16) shared scalars, width-4 NAF: dg[t][pos] in {0, +-1, +-3, +-5, +-7}, pos <= top
struct WnafK { signed char dg[MULTIFOLD_MAXK][264]; int top; };
__global__ void __launch_bounds__(256, 3) k_ec_multifold_w4(MultifoldJob ja, MultifoldJob jb, const u32 *__restrict__ tab_a, const u32 *__restrict__ tab_b,
                                                            const WnafK *__restrict__ wa, const WnafK *__restrict__ wb, u32 m, u32 K) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = i >= m;
  if (second) i -= m;
  if (i >= m) return;
  const u32 *base = second ? jb.base : ja.base;
  const u32 *tab = second ? tab_b : tab_a;
  u32 *out = second ? jb.out : ja.out;
  const WnafK *nf = second ? wb : wa;
  const u32 npts = m * K;
  // XYZZ accumulator (round 4): the ladder is 256 doublings and ~816 mixed additions per output -- additions dominate, and the
  // XYZZ mixed addition is 9 reductions against the Jacobian one's 10 (its doubling 8 against 7): 9 392 reductions per output
  // instead of 9 952
  xyzz acc;
  xyzz_set_inf(acc);
  for (int pos = nf->top; pos >= 0; pos--) {
    xyzz_dbl(acc, acc);
#pragma unroll 1
    for (u32 t = 0; t < K; t++) {
      const int d = nf->dg[t][pos];                                       // the same for every thread of the job
      if (d == 0) continue;
      const u32 mag = (u32)(d < 0 ? -d : d), k = i + t * m;
      affine P;
      if (mag == 1u) load_affine(P, base + 16ull * k);
      else {
        const u32 *q = tab + ((u64)((mag >> 1) - 1u) * npts + k) * 18u;
#pragma unroll
        for (int l = 0; l < 9; l++) { P.x.v[l] = q[l]; P.y.v[l] = q[9 + l]; }
      }
      xyzz_madd_signed(acc, P, d < 0);
    }
  }
  affine r;
  xyzz_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16(out + 16ull * i, w16);
}

// ---- ... and with the coefficients split in two (round 4) -------------------------------------------------------------------
// k = k1 + k2 lambda (scalar.hpp glv_split), lambda (x, y) = (beta x, y): the 16 coefficients become 32 half-scalars of 128 bits,
// the ladder 129 doublings instead of 257 with the same ~816 additions -- 8 370 reductions per output instead of 9 392.  The x of
// lambda (jP) comes from k_ec_odd_multiples' beta-x table (a multiplication per addition would give half of the saving back);
// a negative half has its digits' signs flipped on the host.
// The ladder as a flat list of operations, built on the host (the digits are the same for every output): "double n times, then
// add (-)(j-th odd multiple) of (lambda?) point row / 2".  The kernel fetches the point of operation q + 1 before it computes
// operation q: 2^17 outputs are 2 048 waves -- two per SIMD, too few to hide a load behind the other waves' arithmetic.
//   op = n_dbl | row << 8 | j << 13 | neg << 16          tail = doublings after the last addition
// (WnafG and its builder: fold_ops_host.hpp)
__device__ __forceinline__ void multifold_fetch(affine &P, u32 op, u32 i, u32 m, u32 npts, const u32 *base, const u32 *tab, const u32 *tabx) {
  const u32 r = (op >> 8) & 31u, j = (op >> 13) & 7u, k = i + (r >> 1) * m;      // j = 0: P, 1: 3P, 2: 5P, 3: 7P
  const bool lam = (r & 1u) != 0u;                                               // wave-uniform: the half that multiplies lambda (jP)
  if (j == 0u) {
    if (lam) load_affine_y(P.y, base + 16ull * k); else load_affine(P, base + 16ull * k);
  } else {
    const u32 *q = tab + ((u64)(j - 1u) * npts + k) * 18u;
#pragma unroll
    for (int l = 0; l < 9; l++) P.y.v[l] = q[9 + l];
    if (!lam) {
#pragma unroll
      for (int l = 0; l < 9; l++) P.x.v[l] = q[l];
    }
  }
  if (lam) {
    const u32 *qx = tabx + ((u64)j * npts + k) * 9u;
#pragma unroll
    for (int l = 0; l < 9; l++) P.x.v[l] = qx[l];
  }
}
__global__ void __launch_bounds__(256, 2) k_ec_multifold_w4g(MultifoldJob ja, MultifoldJob jb, const u32 *__restrict__ tab_a, const u32 *__restrict__ tab_b,
                                                             const u32 *__restrict__ tabx_a, const u32 *__restrict__ tabx_b,
                                                             const WnafG *__restrict__ wa, const WnafG *__restrict__ wb, u32 m, u32 K) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = i >= m;
  if (second) i -= m;
  if (i >= m) return;
  const u32 *base = second ? jb.base : ja.base;
  const u32 *tab = second ? tab_b : tab_a;
  const u32 *tabx = second ? tabx_b : tabx_a;
  u32 *out = second ? jb.out : ja.out;
  const WnafG *nf = second ? wb : wa;
  const u32 npts = m * K;
  const u32 nops = __builtin_amdgcn_readfirstlane(nf->nops);
  xyzz acc;
  xyzz_set_inf(acc);
  affine Pn;
  u32 opn = 0;
  if (nops) { opn = __builtin_amdgcn_readfirstlane(nf->op[0]); multifold_fetch(Pn, opn, i, m, npts, base, tab, tabx); }
#pragma unroll 1
  for (u32 q = 0; q < nops; q++) {
    const affine P = Pn;
    const u32 op = opn;
    if (q + 1u < nops) { opn = __builtin_amdgcn_readfirstlane(nf->op[q + 1u]); multifold_fetch(Pn, opn, i, m, npts, base, tab, tabx); }
#pragma unroll 1
    for (u32 t = op & 255u; t; t--) xyzz_dbl(acc, acc);
    xyzz_madd_signed(acc, P, ((op >> 16) & 1u) != 0u);
  }
#pragma unroll 1
  for (u32 t = __builtin_amdgcn_readfirstlane(nf->tail); t; t--) xyzz_dbl(acc, acc);
  affine r;
  xyzz_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16(out + 16ull * i, w16);
}

// Batch decompression of SEC1 compressed points: in = n x 33 B (0x02 | 0x03, x big-endian;
// 33 zero bytes = identity), out = n x 64 B wire format, ok[i] = 1 when the encoding is
// valid.  Replaces bytes_to_point (/root/reference/src/utils/utils.py:119-131):
// y = (x^3 + 7)^((p+1)/4), then the root with the requested parity.
__device__ __forceinline__ bool ec_decompress_one(const uint8_t *src, u32 w16[16]) {
  const u32 tag = src[0];
  u32 w[8];
  u32 any = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {               // word k (little-endian) = bytes 32-4k-3 .. 32-4k of the big-endian x
    const uint8_t *q = src + 1 + 28 - 4 * k;
    w[k] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    any |= w[k];
  }
#pragma unroll
  for (int k = 0; k < 16; k++) w16[k] = 0;
  bool valid;
  if (tag == 0u) {
    valid = (any == 0);                       // identity
  } else if (tag == 2u || tag == 3u) {
    // x must be < p:  x + (2^32 + 977) must not carry out of 256 bits
    u64 c = (u64)w[0] + 977u; c >>= 32;
    c += (u64)w[1] + 1u; c >>= 32;
#pragma unroll
    for (int k = 2; k < 8; k++) { c += w[k]; c >>= 32; }
    valid = (c == 0);
    fe x, y, a, t;
    fe_from_words(x, w);
    fe_sqr(t, x); fe_mul(a, t, x);
    fe seven; fe_set_zero(seven); seven.v[0] = 7;
    fe_add(a, a, seven); fe_carry(a, a);      // a = x^3 + 7
    fe_sqrt_candidate(y, a);
    fe_sqr(t, y);
    valid = valid && fe_equal(t, a);          // a is a quadratic residue
    fe cy;
    fe_canon(cy, y);
    if ((cy.v[0] & 1u) != (tag & 1u)) { fe_neg(t, cy); fe_canon(cy, t); }
    fe cx;
    fe_canon(cx, x);
    if (valid) { fe_to_words(w16, cx); fe_to_words(w16 + 8, cy); }
  } else {
    valid = false;
  }
  return valid;
}
// The same point from its encoding AND its y coordinate (wire format 3, rangeproofs/codec.py: 32 bytes big-endian per point behind
// the proof): no square root -- y is taken when it is below p, has the parity the encoding's tag asks for and lies on the curve
// with x, which makes it the ONE y ec_decompress_one would have computed; anything else is an invalid proof.  (The root is 253
// squarings + 13 multiplications per point; at 2^14 64-bit proofs it was a quarter of a batch verification's device time.)
__device__ __forceinline__ bool ec_hinted_one(const uint8_t *src, const uint8_t *hint, u32 w16[16]) {
  const u32 tag = src[0];
  u32 wx[8], wy[8];
  u32 any = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint8_t *q = src + 1 + 28 - 4 * k, *h = hint + 28 - 4 * k;
    wx[k] = ((u32)q[0] << 24) | ((u32)q[1] << 16) | ((u32)q[2] << 8) | (u32)q[3];
    wy[k] = ((u32)h[0] << 24) | ((u32)h[1] << 16) | ((u32)h[2] << 8) | (u32)h[3];
    any |= wx[k] | wy[k];
  }
#pragma unroll
  for (int k = 0; k < 16; k++) w16[k] = 0;
  if (tag == 0u) return any == 0;             // identity: 33 zero bytes and a zero hint
  if (tag != 2u && tag != 3u) return false;
  // x, y < p:  v + (2^32 + 977) must not carry out of 256 bits
  u64 cx = (u64)wx[0] + 977u, cy = (u64)wy[0] + 977u;
  cx >>= 32; cy >>= 32;
  cx += (u64)wx[1] + 1u; cy += (u64)wy[1] + 1u;
  cx >>= 32; cy >>= 32;
#pragma unroll
  for (int k = 2; k < 8; k++) { cx += wx[k]; cx >>= 32; cy += wy[k]; cy >>= 32; }
  bool valid = (cx | cy) == 0 && (wy[0] & 1u) == (tag & 1u);
  fe x, y, a, t;
  fe_from_words(x, wx);
  fe_from_words(y, wy);
  fe_sqr(t, x); fe_mul(a, t, x);
  fe seven; fe_set_zero(seven); seven.v[0] = 7;
  fe_add(a, a, seven); fe_carry(a, a);        // a = x^3 + 7
  fe_sqr(t, y);
  valid = valid && fe_equal(t, a);
  if (valid) {
#pragma unroll
    for (int k = 0; k < 8; k++) { w16[k] = wx[k]; w16[8 + k] = wy[k]; }
  }
  return valid;
}
__global__ void __launch_bounds__(256, 3) k_ec_decompress(const uint8_t *__restrict__ in, u32 n, u32 *__restrict__ out, uint8_t *__restrict__ ok) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w16[16];
  const bool valid = ec_decompress_one(in + 33ull * i, w16);
  store_words16(out + 16ull * i, w16);
  ok[i] = valid ? 1 : 0;
}
// the same for the points of wire proofs where they lie inside the blobs (rangeproofs/codec.py: (6 + 2k) encodings after the
// 6-byte header and the 5 + k scalars).  It needs nothing from the preparation kernel -- it checks by itself that the blob
// is long enough to hold the encodings and carries the header of a k-round proof -- so it runs beside it on the second
// lane; a blob that fails that check is left to the preparation's verdict (its points are zero), an invalid encoding
// records its proof in *bad (atomicMin; `first` = batch index of proof 0 of this launch)
__global__ void __launch_bounds__(256, 3) k_ec_decompress_wire(const uint8_t *__restrict__ blobs, const u64 *__restrict__ off, u32 k, u32 n_proofs, u64 first,
                                                               u32 max_len, u32 *__restrict__ out, unsigned long long *bad) {
  const u32 per = 6 + 2 * k;
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_proofs * per) return;
  const u32 g = i / per, t = i - g * per;
  u32 w16[16];
#pragma unroll
  for (int j = 0; j < 16; j++) w16[j] = 0;
  const uint8_t *blob = blobs + off[g];
  const u64 len = off[g + 1] - off[g];
  const u32 pts_at = 6 + 32 * (5 + k);
  // (all wire formats hold the encodings at the same offset; which one a batch is in, and that every proof is in it, is the
  // preparation's business.  Format 3 ends with the points' y coordinates: checked, not computed)
  if (len >= pts_at + 33ull * per + 2 && len <= max_len && blob[0] == 'B' && blob[1] == 'P' && blob[2] == 'R' && blob[3] == 'P' && blob[5] == k) {
    if (blob[4] == '1' || blob[4] == '2') {
      const bool valid = ec_decompress_one(blob + pts_at + 33 * t, w16);
      if (!valid) atomicMin(bad, (unsigned long long)(first + g));
    } else if (blob[4] == '3' && len >= pts_at + 33ull * per + 132 + 32ull * per) {
      const bool valid = ec_hinted_one(blob + pts_at + 33 * t, blob + (len - 32ull * per) + 32 * t, w16);
      if (!valid) atomicMin(bad, (unsigned long long)(first + g));
    }
  }
  store_words16(out + 16ull * i, w16);
}

// C-ABI input check (bpmi.hip validate_*): *first_bad = min(*first_bad, tag | i) over the points i that are neither the identity nor
// on the curve (curve.hpp wire_point_valid)
__global__ void __launch_bounds__(256) k_ec_validate(const u32 *__restrict__ pts, u32 n, u32 tag, u32 *first_bad) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 w[16];
  load_words16(w, pts + 16ull * i);
  if (!wire_point_valid(w)) atomicMin(first_bad, tag | i);
}

// out = sum of n affine points (one block)
__global__ void __launch_bounds__(256) k_ec_sum(const u32 *__restrict__ pts, u32 n, u32 *__restrict__ out) {
  __shared__ u32 s_val[256 * LDS_STRIDE];
  xyzz acc;
  xyzz_set_inf(acc);
  for (u32 i = threadIdx.x; i < n; i += 256u) {
    affine P;
    load_affine(P, pts + 16ull * i);
    xyzz_madd_signed(acc, P, false);
  }
  block_tree_sum(acc, s_val);
  if (threadIdx.x == 0) {
    affine r;
    xyzz_to_affine(r, acc);
    u32 w16[16];
    affine_to_words(w16, r);
#pragma unroll
    for (int k = 0; k < 16; k++) out[k] = w16[k];
  }
}
