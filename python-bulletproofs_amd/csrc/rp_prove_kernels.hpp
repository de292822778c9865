// rp_prove_kernels.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).  DEVICE code.
// A BATCH of single-value range proofs over shared generators, proved on the GPU from the first blinding scalar to the wire bytes
// (SURVEY.md section 2.1 K11).  It replaces a loop of NIRangeProver.prove (/root/reference/src/rangeproofs/rangeproof_prover.py:35-91)
// with NIProver.prove / FastNIProver2.prove inside (/root/reference/src/innerproduct/inner_product_prover.py:27-44, :84-110): the same
// transcripts (/root/reference/src/utils/transcript.py:13-33), the same seeded scalars (mod_hash, /root/reference/src/utils/utils.py:84-97),
// the same proof, byte for byte (tests/test_gpu_prove_batch.py compares every proof with the single-proof prover's).
//
// Shape of the work.  One proof is ~16 multi-scalar multiplications of 2 .. 129 terms and a dozen Fiat-Shamir hashes between them:
// per-proof launches made it 2.1 ms a proof (474 proofs/s, profiles/r04_rocprofv3_kernel_stats_C5_batch_verify_2e14.csv).  Here
// every step is ONE launch over all proofs of the batch:
//   * the generators are fixed for a prover, so every scalar multiplication is a FIXED-BASE one: table[b][k][d - 1] = d 2^(8k) base_b
//     for the 3 + 2n generators, windows of signed tw-bit digits (k_pv_table_scalars + the engine's batched multiplication; tw = 8: 32
//     windows, 34 MB for 64-bit proofs, built once per prover), and a term costs one mixed addition per window with no doubling.  The inner-product rounds never
//     fold a generator: L and R are sums over the ORIGINAL generators with the fold coefficients in the scalars (cg, hf below) -- 65
//     terms per side and round, whatever the round;
//   * a multi-scalar multiplication is a JOB of k_pv_msm: 2^G lanes take its terms round-robin, each walks the 32 windows of its
//     terms with the next table entry in flight, a shuffle butterfly adds the lanes' sums; k_pv_affine turns the results into affine
//     points (one inversion per point, all lanes busy);
//   * the transcripts are hashed where they are needed, one lane per proof (k_pv_chal_*): SHA-256 over the text the reference builds
//     (base64 of the compressed points, decimal numbers, '&'), kept in a per-proof byte buffer; the 2n + 2 blinding scalars of a
//     proof are one hash each, one lane per hash (k_pv_blind);
//   * the O(n) scalar algebra between the steps runs one lane per proof (mod-q arithmetic of scalar.hpp).
// The proofs leave as wire format 2 (rangeproofs/codec.py): what the batch verifier takes.
#pragma once

namespace rpp {

using bpmi::sc;
using bpmi::u32;
using bpmi::u64;
using bpmi::fe;
using bpmi::affine;
using bpmi::xyzz;
typedef unsigned char u8;

// geometry of the fixed-base tables: windows of tw bits (signed digits), wt = ceil(256 / tw) windows per scalar, bt = 2^(tw-1) entries
// per (base, window): entry (b, k, d) = d 2^(tw k) base_b at table + 64 ((b wt + k) bt + d - 1) bytes
struct Tab { const u32 *p; u32 tw, wt, bt; };
// the next tw-bit window of a magnitude that is shifted down as it is consumed (static register indexing), recoded to a signed digit:
// d in [0, bt], sg = 1 for a negative digit, the carry goes into the next window (the top one cannot carry out: |s| < 2^255)
__device__ __forceinline__ void next_digit(sc &s, const Tab &T, u32 &carry, u32 &d, u32 &sg) {
  const u32 tt = (s.v[0] & ((1u << T.tw) - 1u)) + carry;
#pragma unroll
  for (int i = 0; i < 7; i++) s.v[i] = (u32)((((u64)s.v[i + 1] << 32) | s.v[i]) >> T.tw);
  s.v[7] >>= T.tw;
  if (tt > T.bt) { d = (1u << T.tw) - tt; sg = 1; carry = 1; } else { d = tt; sg = 0; carry = 0; }
}

// ---- mod q helpers (out of line: dozens of call sites) --------------------------------------------------------------------------
__device__ __noinline__ sc mulq(const sc a, const sc b) { sc r; bpmi::sc_mul(r, a, b); return r; }
__device__ __forceinline__ sc addq(const sc &a, const sc &b) { sc r; bpmi::sc_add(r, a, b); return r; }
__device__ __forceinline__ sc negq(const sc &a) { sc r; bpmi::sc_neg(r, a); return r; }
__device__ __forceinline__ sc subq(const sc &a, const sc &b) { return addq(a, negq(b)); }
__device__ __noinline__ sc invq(const sc a) { sc r; bpmi::sc_inv(r, a); return r; }
__device__ __forceinline__ sc sc_u32(u32 x) { sc r; r.v[0] = x; for (int i = 1; i < 8; i++) r.v[i] = 0; return r; }
__device__ __forceinline__ sc ld_sc(const u32 *p) { sc r; ::load_words8(r.v, p); return r; }
__device__ __forceinline__ void st_sc(u32 *p, const sc &a) { ::store_words8(p, a.v); }

// ---- SHA-256 of (up to 8 prefix bytes) || msg[0, mlen) ------------------------------------------------------------------------------
// The prefix travels in a register (byte k = bits 8k ..), the message is read from memory a byte at a time: these hashes are a few
// hundred blocks per proof against ~30 000 point additions.
__device__ __noinline__ rpd::H8 sha256_msg(u64 pre, u32 plen, const u8 *msg, u32 mlen) {
  u32 h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  const u32 total = plen + mlen;
  const u32 nblocks = (total + 9u + 63u) / 64u;
  for (u32 blk = 0; blk < nblocks; blk++) {
    u32 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      u32 word = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 pos = blk * 64u + (u32)(i * 4 + k);
        u32 byte = 0;
        if (pos < plen) byte = (u32)(pre >> (8u * pos)) & 0xFFu;
        else if (pos < total) byte = msg[pos - plen];
        else if (pos == total) byte = 0x80u;
        word = (word << 8) | byte;
      }
      w[i] = word;
    }
    if (blk + 1u == nblocks) { w[14] = total >> 29; w[15] = total << 3; }
    const rpd::H8 r = rpd::sha_compress_v(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9],
                                          w[10], w[11], w[12], w[13], w[14], w[15]);
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = r.v[i];
  }
  rpd::H8 out;
#pragma unroll
  for (int i = 0; i < 8; i++) out.v[i] = h[i];
  return out;
}
// mod_hash(tag || msg, q) (utils.py:84-97): the first counter c >= 1 for which SHA-256(str(c) || tag || msg), read as a big-endian
// number (q has 256 bits: the mask keeps all of it), lies in [1, q).  tag: up to 5 bytes in a register.
__device__ __noinline__ sc mod_hash_q(u64 tag, u32 taglen, const u8 *msg, u32 mlen) {
  for (u32 c = 1;; c++) {
    u64 pre; u32 plen;
    if (c < 10u) { pre = (u64)('0' + c); plen = 1; }
    else if (c < 100u) { pre = (u64)('0' + c / 10u) | ((u64)('0' + c % 10u) << 8); plen = 2; }
    else { pre = (u64)('0' + (c / 100u) % 10u) | ((u64)('0' + (c / 10u) % 10u) << 8) | ((u64)('0' + c % 10u) << 16); plen = 3; }
    pre |= tag << (8u * plen);
    const rpd::H8 d = sha256_msg(pre, plen + taglen, msg, mlen);
    sc r, t;
#pragma unroll
    for (int k = 0; k < 8; k++) r.v[k] = d.v[7 - k];
    t = r;
    bpmi::sc_reduce_once(t);
    bool same = true;
#pragma unroll
    for (int k = 0; k < 8; k++) same = same && (t.v[k] == r.v[k]);
    if (same && !bpmi::sc_is_zero(r)) return r;          // (a retry has probability 2^-128; after c = 999 the prefix would not fit: never)
    if (c >= 999u) return r;
  }
}
__device__ __forceinline__ u64 tag_of_index(u32 i, u32 &len) {            // str(i), i < 1000
  if (i < 10u) { len = 1; return (u64)('0' + i); }
  if (i < 100u) { len = 2; return (u64)('0' + i / 10u) | ((u64)('0' + i % 10u) << 8); }
  len = 3;
  return (u64)('0' + i / 100u) | ((u64)('0' + (i / 10u) % 10u) << 8) | ((u64)('0' + i % 10u) << 16);
}

// ---- text of a transcript ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u8 b64c(u32 v) { return (u8)(v < 26u ? 'A' + v : v < 52u ? 'a' + (v - 26u) : v < 62u ? '0' + (v - 52u) : v == 62u ? '+' : '/'); }
// dst <- base64(point_to_bytes(P)) || '&' (utils.py:100-112): 02 / 03 and x big-endian, 44 characters; the identity is b"\x00" = "AA=="
__device__ __noinline__ u32 put_point(u8 *dst, const u32 *aff) {
  u32 w[16];
  ::load_words16(w, aff);
  u32 any = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) any |= w[i];
  if (!any) { dst[0] = 'A'; dst[1] = 'A'; dst[2] = '='; dst[3] = '='; dst[4] = '&'; return 5; }
  u8 c[33];
  c[0] = (u8)(2u + (w[8] & 1u));
  for (int i = 0; i < 32; i++) c[1 + i] = (u8)(w[7 - (i >> 2)] >> (8 * (3 - (i & 3))));
  for (int g = 0; g < 11; g++) {
    const u32 v = ((u32)c[3 * g] << 16) | ((u32)c[3 * g + 1] << 8) | c[3 * g + 2];
    dst[4 * g] = b64c(v >> 18); dst[4 * g + 1] = b64c((v >> 12) & 63u); dst[4 * g + 2] = b64c((v >> 6) & 63u); dst[4 * g + 3] = b64c(v & 63u);
  }
  dst[44] = '&';
  return 45;
}
// dst <- str(v) || '&' (transcript.py:27-30)
__device__ __noinline__ u32 put_number(u8 *dst, const sc v) {
  u32 x[8], chunk[9], n = 0;
  for (int i = 0; i < 8; i++) x[i] = v.v[i];
  for (;;) {
    u32 nz = 0;
    for (int i = 0; i < 8; i++) nz |= x[i];
    if (!nz) break;
    u64 rem = 0;
    for (int i = 7; i >= 0; i--) { const u64 cur = (rem << 32) | x[i]; x[i] = (u32)(cur / 1000000000ull); rem = cur % 1000000000ull; }
    chunk[n++] = (u32)rem;
  }
  u32 len = 0;
  if (n == 0) { dst[len++] = '0'; }
  else {
    u8 tmp[10]; u32 t = 0, top = chunk[n - 1];
    while (top) { tmp[t++] = (u8)('0' + top % 10u); top /= 10u; }
    while (t) dst[len++] = tmp[--t];
    for (u32 k = n - 1; k-- > 0;) {
      u32 c = chunk[k];
      for (int d = 8; d >= 0; d--) { dst[len + d] = (u8)('0' + c % 10u); c /= 10u; }
      len += 9;
    }
  }
  dst[len++] = '&';
  return len;
}

// ---- parameters of a batch ----------------------------------------------------------------------------------------------------------------
// Device arrays of one batch (P proofs of n bits, k = log2 n).  Scalars: 8 words little-endian.  Points: 16 words (x, y) little-endian.
// Round 6: AGGREGATED proofs (AggregNIRangeProver, /root/reference/src/rangeproofs/rangeproof_aggreg_prover.py:36-146): a proof covers m values of
// nb bits, its vectors have n = nb m elements (element i = bit i % nb of value i / nb), z^2 2^i becomes z^(2 + i / nb) 2^(i % nb) (:85, :120-131)
// and the blinding of taux is sum_j z^(2 + j) gamma_j (:138-142).  m = 1 is the single-value prover (rangeproof_prover.py:35-91).
struct Batch {
  u32 P, n, k;
  u32 nb, m;                 // bits per value, values per proof: n = nb m
  Tab table;                 // [(3 + 2n) bases][wt windows][bt] affine points: base 0 g, 1 h, 2 u, 3 + i gs_i, 3 + n + i hs_i
  const u8 *dig0;  u32 dig0_stride; const u32 *dig0_len;       // base64(seed) || '&' of every proof
  const u32 *values;         // P x m scalars: v (only its low nb bits are used, as in rangeproof_prover.py:40)
  const u32 *gammas;         // P x m scalars
  const u8 *ip_prefix; u32 ip_prefix_len;                      // "&&" || str(x_ip) || "&": the Protocol-2 transcript before the first round
  sc x_ip;
  u32 u_new[16];             // x_ip u (inner_product_prover.py:37), the same for every proof
  u8 *tr;  u32 tr_stride;  u32 *tr_len;                        // the range-proof transcript, then (from k_pv_final on) the Protocol-2 transcript
  u32 *slr;                  // P x (2n + 1): sL, sR, rho     (the scalars of S in base order gs, hs, h)
  u32 *alpha;                // P
  u32 *chal;                 // P x 4: y, z, x, 1 / y
  u32 *tau;                  // P x 2
  u32 *tsc;                  // P x 4: t1, tau1, t2, tau2   (the scalars of T1 and T2 over g, h)
  u32 *res;                  // P x 5: taux, mu, t_hat, a, b
  u32 *xs;                   // P x k
  u32 *xr;                   // P x 2: the current round's challenge and its inverse
  u32 *a, *b, *cg, *hf;      // P x n each: the inner-product state (cg / hf: coefficient of gs_j / hs_j in the folded generators, y^-j included)
  u32 *jsc;                  // job scalars: P x (2n + 1) (P_new), or 2P x (n + 1) (the L / R of a round)
  u32 *jout;                 // job results, XYZZ: up to 2P x 36 words
  u32 *pts;                  // P x (6 + 2k) affine points in wire order: T1 T2 A S u_new P_new L_0.. R_0..
};
#define PV_PT_T1 0u
#define PV_PT_T2 1u
#define PV_PT_A 2u
#define PV_PT_S 3u
#define PV_PT_UNEW 4u
#define PV_PT_PNEW 5u

// ---- building the table (round 6) ---------------------------------------------------------------------------------------------------------
// Round 5 built every entry (b, k, d) = d 2^(tw k) base_b as a scalar multiplication of its own (k_pv_table_scalars + bpmi_ec_mul_batch:
// 72 ms for the 5.6 M entries of 12-bit windows; 16-bit windows -- 69 M entries -- would have taken 0.9 s).  Now only the nb x wt WINDOW
// BASES 2^(tw k) base_b are scalar multiplications (k_pv_window_scalars), and the multiples of a window base W come level by level:
//   d in (2^j, 2^(j+1)]:   d W = 2^j W + (d - 2^j) W        one complete mixed addition + one inversion per entry,
// every (base, window) pair in the same launch (k_pv_table_level, tw - 1 launches).  ~285 field multiplications per entry: 7 ms for 12-bit
// windows, ~90 ms for 16-bit ones (4.4 GB for 64-bit proofs: a sixtieth of the HBM).
__global__ void __launch_bounds__(256) k_pv_window_scalars(const u32 *__restrict__ bases, u32 nbases, Tab T, u32 *__restrict__ pts, u32 *__restrict__ scal) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nbases * T.wt) return;
  const u32 k = t % T.wt, b = t / T.wt;
  u32 w[16];
  ::load_words16(w, bases + 16ull * b);
  ::store_words16(pts + 16ull * t, w);
  u32 s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const u32 bit = T.tw * k;                     // 2^(tw k), tw k <= 255
  s[bit >> 5] = 1u << (bit & 31u);
  ::store_words8(scal + 8ull * t, s);
}
// entry d = 1 of every (base, window): the window bases themselves
__global__ void __launch_bounds__(256) k_pv_table_seed(const u32 *__restrict__ wbase, u32 nbk, Tab T, u32 *__restrict__ table) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nbk) return;
  u32 w[16];
  ::load_words16(w, wbase + 16ull * t);
  ::store_words16(table + 16ull * ((size_t)t * T.bt), w);
}
// level j: entries d = 2^j + i, i = 1 .. 2^j, of every (base, window) pair bk: slot(d) = d - 1
__global__ void __launch_bounds__(256) k_pv_table_level(u32 nbk, Tab T, u32 j, u32 *__restrict__ table) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)1 << j;
  if (t >= (size_t)nbk * per) return;
  const size_t bk = t >> j, i = (t & (per - 1)) + 1;
  u32 *row = table + 16ull * (bk * T.bt);
  affine Q, Pi;
  ::load_affine(Q, row + 16ull * (per - 1));
  ::load_affine(Pi, row + 16ull * (i - 1));
  xyzz acc;
  bpmi::xyzz_set_inf(acc);
  bpmi::xyzz_madd_signed(acc, Q, false);
  bpmi::xyzz_madd_signed(acc, Pi, false);          // complete: i = 2^j is the doubling 2 (2^j W)
  affine r;
  bpmi::xyzz_to_affine(r, acc);
  u32 w[16];
  bpmi::affine_to_words(w, r);
  ::store_words16(row + 16ull * (per + i - 1), w);
}

// element i of a proof's bit vector aL: bit i % nb of value i / nb
__device__ __forceinline__ u32 bit_of(const Batch &B, size_t p, u32 i) {
  const u32 jv = i / B.nb, jb = i % B.nb;
  return (B.values[8ull * (p * B.m + jv) + (jb >> 5)] >> (jb & 31u)) & 1u;
}
// z^(2 + e), e < m (a short loop: m is 1 for single-value proofs and a handful for aggregated ones)
__device__ __forceinline__ sc z_pow2p(const sc &z, const sc &zz, u32 e) {
  sc r = zz;
  for (u32 t = 0; t < e; t++) r = mulq(r, z);
  return r;
}

// ---- blinding scalars: one hash per lane ----------------------------------------------------------------------------------------------------
// lane (p, i): i < 2n: s_i = mod_hash(str(i) || digest) (rangeproof_prover.py:50-57); i = 2n: rho = mod_hash(str(2 nb) || digest) (:58 -- the
// aggregated prover hashes str(2 * n) with n the bits PER VALUE, rangeproof_aggreg_prover.py:62: rho then equals one of the s_i; kept as it is);
// i = 2n + 1: alpha = mod_hash(b"alpha" || digest) (:48)
__global__ void __launch_bounds__(256) k_pv_blind(Batch B) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 per = 2u * B.n + 2u;
  if (t >= B.P * per) return;
  const u32 p = t / per, i = t % per;
  const u8 *dg = B.dig0 + (size_t)p * B.dig0_stride;
  const u32 dl = B.dig0_len[p];
  if (i == 2u * B.n + 1u) {
    const u64 tag = (u64)'a' | ((u64)'l' << 8) | ((u64)'p' << 16) | ((u64)'h' << 24) | ((u64)'a' << 32);
    st_sc(B.alpha + 8ull * p, mod_hash_q(tag, 5, dg, dl));
    return;
  }
  u32 tl;
  const u64 tag = tag_of_index(i == 2u * B.n ? 2u * B.nb : i, tl);
  st_sc(B.slr + 8ull * ((size_t)p * (2u * B.n + 1u) + i), mod_hash_q(tag, tl, dg, dl));
}

// ---- fixed-base multi-scalar multiplication: one job = 2^GL lanes ---------------------------------------------------------------------------
struct MsmJobs {
  u32 njobs;               // jobs; job j uses base list j % ntypes
  u32 ntypes, T;           // base lists, terms per job
  const unsigned short *bases;     // [ntypes][T]
  const u32 *scalars;      // job j, term t: scalars + 8 (j * stride + t)
  u32 stride;
  u32 *out;                // XYZZ of job j at out + 36 j
};
// signed tw-bit digits of |s| (s folded to s or q - s: the top digit cannot carry out)
// Round 6: the terms that do not fill a last round of the job's lanes (T mod 2^GL of them: ONE for every job of the prover -- the u term of
// 2^k + 1 terms) are shared out by WINDOWS, lane l taking the windows l, l + 2^GL, ... of each: with whole terms round-robin lane 0 of a
// round's 16 lanes walked 5 terms and the other 15 lanes 4 -- the job took 5 x wt additions for 4.06 x wt of work per lane (81 %).
template <int GL> __global__ void __launch_bounds__(256) k_pv_msm(MsmJobs J, Tab T) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 job = t >> GL, l = t & ((1u << GL) - 1u);
  const bool live = job < J.njobs;
  xyzz acc;
  bpmi::xyzz_set_inf(acc);
  if (live) {
    const unsigned short *bl = J.bases + (size_t)(job % J.ntypes) * J.T;
    const u32 Tfull = J.T & ~((1u << GL) - 1u);
    for (u32 term = l; term < Tfull; term += (1u << GL)) {
      sc s = ld_sc(J.scalars + 8ull * ((size_t)job * J.stride + term));
      if (bpmi::sc_is_zero(s)) continue;
      const bool neg = bpmi::sc_is_high(s);
      if (neg) bpmi::sc_neg(s, s);
      const u32 *tb = T.p + 16ull * ((size_t)bl[term] * T.wt * T.bt);
      // window k + 1's entry is requested before window k's addition
      u32 carry = 0;
      u32 wv[16];
      u32 d_cur, sg_cur;
      next_digit(s, T, carry, d_cur, sg_cur);
      ::load_words16(wv, tb + 16ull * (d_cur ? d_cur - 1u : 0u));
      for (u32 k = 0; k < T.wt; k++) {
        affine Pt;
        bpmi::affine_from_words(Pt, wv);
        const u32 d = d_cur, sg = sg_cur;
        if (k + 1u < T.wt) {
          next_digit(s, T, carry, d_cur, sg_cur);
          ::load_words16(wv, tb + 16ull * ((size_t)(k + 1u) * T.bt + (d_cur ? d_cur - 1u : 0u)));
        }
        if (d) bpmi::xyzz_madd_signed(acc, Pt, (sg != 0u) != neg);
      }
    }
    for (u32 term = Tfull; term < J.T; term++) {            // the remainder terms: every lane recodes the scalar, and adds its own windows
      sc s = ld_sc(J.scalars + 8ull * ((size_t)job * J.stride + term));
      if (bpmi::sc_is_zero(s)) continue;
      const bool neg = bpmi::sc_is_high(s);
      if (neg) bpmi::sc_neg(s, s);
      const u32 *tb = T.p + 16ull * ((size_t)bl[term] * T.wt * T.bt);
      // lane l: windows l, l + 2^GL, ...  -- every lane reaches ITS window by a walk of digits (a few instructions each) and all lanes add
      // in the same iteration: an `if (window is mine)` inside ONE loop over the windows made the wave run wt additions with a lane in
      // sixteen active, i.e. exactly the time the fifth term had cost
      u32 carry = 0, walked = 0;
      for (u32 k = l; k < T.wt; k += (1u << GL)) {
        u32 d = 0, sg = 0;
        do { next_digit(s, T, carry, d, sg); walked++; } while (walked <= k);
        affine Pt;
        ::load_affine(Pt, tb + 16ull * ((size_t)k * T.bt + (d ? d - 1u : 0u)));
        if (d) bpmi::xyzz_madd_signed(acc, Pt, (sg != 0u) != neg);
      }
    }
  }
#pragma unroll 1
  for (u32 m = 1; m < (1u << GL); m <<= 1) {
    xyzz o;
    ::xyzz_shfl_xor(o, acc, (int)m);
    bpmi::xyzz_add(acc, acc, o);
  }
  if (live && l == 0) ::xyzz_store_g(J.out + 36ull * job, acc);
}

// A = sum_i (bit_i ? gs_i : -hs_i) + alpha h (rangeproof_prover.py:40-49: aL the bits, aR = aL - 1): 16 lanes per proof, n / 16 bit
// positions and every sixteenth window of alpha h each
__global__ void __launch_bounds__(256) k_pv_commit_A(Batch B, u32 *__restrict__ out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 p = t >> 4, l = t & 15u;
  const bool live = p < B.P;
  xyzz acc;
  bpmi::xyzz_set_inf(acc);
  if (live) {
    for (u32 i = l; i < B.n; i += 16u) {
      const u32 bit = bit_of(B, p, i);
      const u32 base = bit ? 3u + i : 3u + B.n + i;
      affine Pt;
      ::load_affine(Pt, B.table.p + 16ull * ((size_t)base * B.table.wt * B.table.bt));          // window 0, d = 1: the generator itself
      bpmi::xyzz_madd_signed(acc, Pt, bit == 0u);
    }
    sc s = ld_sc(B.alpha + 8ull * p);
    const bool neg = bpmi::sc_is_high(s);
    if (neg) bpmi::sc_neg(s, s);
    u32 carry = 0, walked = 0;                       // (windows l, l + 16, ...: see k_pv_msm's remainder terms)
    for (u32 k = l; k < B.table.wt; k += 16u) {
      u32 d = 0, sg = 0;
      do { next_digit(s, B.table, carry, d, sg); walked++; } while (walked <= k);
      affine Pt;
      ::load_affine(Pt, B.table.p + 16ull * ((size_t)(1u * B.table.wt + k) * B.table.bt + (d ? d - 1u : 0u)));
      if (d) bpmi::xyzz_madd_signed(acc, Pt, (sg != 0u) != neg);
    }
  }
#pragma unroll 1
  for (u32 m = 1; m < 16u; m <<= 1) {
    xyzz o;
    ::xyzz_shfl_xor(o, acc, (int)m);
    bpmi::xyzz_add(acc, acc, o);
  }
  if (live && l == 0) ::xyzz_store_g(out + 36ull * p, acc);
}

// XYZZ results -> affine points: result j goes to pts[(j / per) * pt_stride + slot0 + (j % per) * slot_step]
__global__ void __launch_bounds__(256) k_pv_affine(const u32 *__restrict__ in, u32 count, u32 per, u32 *__restrict__ pts, u32 pt_stride, u32 slot0, u32 slot_step) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  xyzz a;
  ::xyzz_load_g(a, in + 36ull * j);
  affine r;
  bpmi::xyzz_to_affine(r, a);
  u32 w[16];
  bpmi::affine_to_words(w, r);
  ::store_words16(pts + 16ull * ((size_t)(j / per) * pt_stride + slot0 + (j % per) * slot_step), w);
}

// ---- challenges y, z and the blinding factors of T1, T2 (rangeproof_prover.py:60-67) ------------------------------------------------------
__global__ void __launch_bounds__(64) k_pv_chal_yz(Batch B) {
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B.P) return;
  u8 *tr = B.tr + (size_t)p * B.tr_stride;
  const u8 *dg = B.dig0 + (size_t)p * B.dig0_stride;
  u32 len = B.dig0_len[p];
  for (u32 i = 0; i < len; i++) tr[i] = dg[i];
  const u32 *pt = B.pts + 16ull * (size_t)p * (6u + 2u * B.k);
  len += put_point(tr + len, pt + 16u * PV_PT_A);
  len += put_point(tr + len, pt + 16u * PV_PT_S);
  const sc y = mod_hash_q(0, 0, tr, len);
  len += put_number(tr + len, y);
  const sc z = mod_hash_q(0, 0, tr, len);
  len += put_number(tr + len, z);
  B.tr_len[p] = len;
  st_sc(B.chal + 32ull * p, y);
  st_sc(B.chal + 32ull * p + 8, z);
  const u64 t1 = (u64)'t' | ((u64)'a' << 8) | ((u64)'u' << 16) | ((u64)'1' << 24), t2 = (u64)'t' | ((u64)'a' << 8) | ((u64)'u' << 16) | ((u64)'2' << 24);
  st_sc(B.tau + 16ull * p, mod_hash_q(t1, 4, tr, len));
  st_sc(B.tau + 16ull * p + 8, mod_hash_q(t2, 4, tr, len));
}

// ---- the vector steps with n lanes per proof (block = 256 threads = 256 / n proofs, n <= 128 a power of two) -------------------------------
// With ONE lane per proof these two steps ran 2^14 lanes of 64 serial iterations on a quarter of the SIMDs (0.65 + 0.88 ms of a 28 ms
// batch); here lane j owns element j: the powers y^j / y^-j are an inclusive PRODUCT scan over the proof's lanes in LDS (log2 n
// multiplications per lane), the sums over j an LDS tree.  The hashes and the inversion of y stay one lane per proof (k_pv_final_chal).
// v[tid] <- product of v[base .. tid] over the proof's n lanes (every thread of the block calls it)
__device__ __forceinline__ sc lane_product_scan(u32 *s_v, u32 tid, u32 j, u32 n, sc v) {
  for (u32 d = 1; d < n; d <<= 1) {
    st_sc(s_v + 8u * tid, v);
    __syncthreads();
    if (j >= d) v = mulq(v, ld_sc(s_v + 8u * (tid - d)));
    __syncthreads();
  }
  return v;
}
// v summed over the proof's n lanes; valid in lane j == 0 (every thread of the block calls it)
__device__ __forceinline__ sc lane_tree_sum(u32 *s_v, u32 tid, u32 j, u32 n, sc v) {
  st_sc(s_v + 8u * tid, v);
  __syncthreads();
  for (u32 d = n >> 1; d > 0u; d >>= 1) {
    if (j < d) st_sc(s_v + 8u * tid, addq(ld_sc(s_v + 8u * tid), ld_sc(s_v + 8u * (tid + d))));
    __syncthreads();
  }
  return ld_sc(s_v + 8u * tid);
}
__device__ __forceinline__ sc sc_pow2(u32 e) {                // 2^e, e < 128
  sc r = sc_u32(0);
  r.v[e >> 5] = 1u << (e & 31u);
  return r;
}
// ---- t1, t2 (rangeproof_prover.py:93-101) -------------------------------------------------------------------------------------------------
//   t1 = sum sL_i (y^i (aR_i + z) + z^(2 + i / nb) 2^(i % nb)) + sum (aL_i - z) y^i sR_i,   t2 = sum sL_i y^i sR_i;   y^j is left in cg_j for k_pv_final_wide
__global__ void __launch_bounds__(256) k_pv_poly(Batch B) {
  __shared__ u32 s_v[256 * 8], s_w[256 * 8];
  const u32 n = B.n, tid = threadIdx.x;
  const u32 p = blockIdx.x * (256u / n) + tid / n, j = tid & (n - 1u);
  const bool live = p < B.P;
  const size_t pc = live ? p : 0;                             // (lanes past the batch compute on proof 0 and store nothing)
  const sc y = ld_sc(B.chal + 32ull * pc), z = ld_sc(B.chal + 32ull * pc + 8);
  const sc one = sc_u32(1), zz = mulq(z, z);
  const sc yp = lane_product_scan(s_v, tid, j, n, j ? y : one);                  // y^j
  const u32 *slr = B.slr + 8ull * pc * (2u * n + 1u);
  const u32 bit = bit_of(B, pc, j);
  const sc sL = ld_sc(slr + 8ull * j), sR = ld_sc(slr + 8ull * (n + j));
  const sc ysr = mulq(yp, sR);
  // aR + z = z - 1 + bit;  aL - z = bit - z;  the power of two: z^(2 + j / nb) 2^(j % nb)
  sc c1 = mulq(sL, addq(mulq(yp, bit ? z : subq(z, one)), mulq(z_pow2p(z, zz, j / B.nb), sc_pow2(j % B.nb))));
  c1 = addq(c1, mulq(bit ? subq(one, z) : negq(z), ysr));
  const sc t1 = lane_tree_sum(s_v, tid, j, n, c1);
  const sc t2 = lane_tree_sum(s_w, tid, j, n, mulq(sL, ysr));
  if (!live) return;
  st_sc(B.cg + 8ull * ((size_t)p * n + j), yp);
  if (j == 0u) {
    u32 *o = B.tsc + 32ull * p;
    st_sc(o, t1); st_sc(o + 8, ld_sc(B.tau + 16ull * p)); st_sc(o + 16, t2); st_sc(o + 24, ld_sc(B.tau + 16ull * p + 8));
  }
}

// ---- x; l, r, t_hat, taux, mu; the scalars of P_new and the state of Protocol 2 (rangeproof_prover.py:68-90, inner_product_prover.py:33-37)
//   P + (-mu) h = <l, gs> + <r, hsp>,  hsp_i = y^-i hs_i   ->   P_new = sum l_i gs_i + sum (r_i y^-i) hs_i + (x_ip t_hat) u
// k_pv_final_chal, one lane per proof: the hash, 1 / y, taux, mu, the Protocol-2 transcript's start; k_pv_final_wide: the vectors
__global__ void __launch_bounds__(64) k_pv_final_chal(Batch B) {
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B.P) return;
  u8 *tr = B.tr + (size_t)p * B.tr_stride;
  u32 len = B.tr_len[p];
  const u32 npt = 6u + 2u * B.k;
  const u32 *pt = B.pts + 16ull * (size_t)p * npt;
  len += put_point(tr + len, pt + 16u * PV_PT_T1);
  len += put_point(tr + len, pt + 16u * PV_PT_T2);
  const sc x = mod_hash_q(0, 0, tr, len);
  st_sc(B.chal + 32ull * p + 16, x);
  const sc y = ld_sc(B.chal + 32ull * p), z = ld_sc(B.chal + 32ull * p + 8);
  st_sc(B.chal + 32ull * p + 24, invq(y));
  const u32 n = B.n;
  const u32 *slr = B.slr + 8ull * (size_t)p * (2u * n + 1u);
  const sc tau1 = ld_sc(B.tau + 16ull * p), tau2 = ld_sc(B.tau + 16ull * p + 8), alpha = ld_sc(B.alpha + 8ull * p), rho = ld_sc(slr + 8ull * (2u * n));
  sc taux = addq(mulq(tau2, mulq(x, x)), mulq(tau1, x));
  {
    sc zp = mulq(z, z);                                           // sum_j z^(2 + j) gamma_j (rangeproof_aggreg_prover.py:138-142; m = 1: z^2 gamma)
    for (u32 jv = 0; jv < B.m; jv++) { taux = addq(taux, mulq(zp, ld_sc(B.gammas + 8ull * ((size_t)p * B.m + jv)))); zp = mulq(zp, z); }
  }
  const sc mu = addq(alpha, mulq(rho, x));
  u32 *res = B.res + 40ull * p;
  st_sc(res, taux); st_sc(res + 8, mu);
  // the Protocol-2 transcript starts here: "&" (its own empty seed) || "&" || str(x_ip) || "&" (inner_product_prover.py:60-63)
  for (u32 i = 0; i < B.ip_prefix_len; i++) tr[i] = B.ip_prefix[i];
  B.tr_len[p] = B.ip_prefix_len;
}
__global__ void __launch_bounds__(256) k_pv_final_wide(Batch B) {
  __shared__ u32 s_v[256 * 8];
  const u32 n = B.n, tid = threadIdx.x;
  const u32 p = blockIdx.x * (256u / n) + tid / n, j = tid & (n - 1u);
  const bool live = p < B.P;
  const size_t pc = live ? p : 0;
  const sc y_inv = ld_sc(B.chal + 32ull * pc + 24), z = ld_sc(B.chal + 32ull * pc + 8), x = ld_sc(B.chal + 32ull * pc + 16);
  const sc one = sc_u32(1);
  const sc yip = lane_product_scan(s_v, tid, j, n, j ? y_inv : one);             // y^-j
  const sc yp = ld_sc(B.cg + 8ull * (pc * n + j));                               // y^j (k_pv_poly)
  const u32 *slr = B.slr + 8ull * pc * (2u * n + 1u);
  const u32 bit = bit_of(B, pc, j);
  const sc sL = ld_sc(slr + 8ull * j), sR = ld_sc(slr + 8ull * (n + j));
  const sc l = addq(bit ? subq(one, z) : negq(z), mulq(sL, x));                                          // aL - z + sL x
  const sc r = addq(mulq(yp, addq(bit ? z : subq(z, one), mulq(sR, x))), mulq(z_pow2p(z, mulq(z, z), j / B.nb), sc_pow2(j % B.nb)));   // y^j (aR + z + sR x) + z^(2 + j / nb) 2^(j % nb)
  const sc t_hat = lane_tree_sum(s_v, tid, j, n, mulq(l, r));
  if (!live) return;
  const size_t e = (size_t)p * n + j;
  st_sc(B.a + 8ull * e, l); st_sc(B.b + 8ull * e, r);
  st_sc(B.cg + 8ull * e, one); st_sc(B.hf + 8ull * e, yip);
  u32 *js = B.jsc + 8ull * (size_t)p * (2u * n + 1u);
  st_sc(js + 8ull * j, l); st_sc(js + 8ull * (n + j), mulq(r, yip));
  if (j == 0u) {
    st_sc(js + 8ull * (2u * n), mulq(B.x_ip, t_hat));
    st_sc(B.res + 40ull * p + 16, t_hat);
  }
}

// ---- one round of Protocol 2 (inner_product_prover.py:94-110) over the UNFOLDED generators ----------------------------------------------------
// State of length len = n >> round, half = len / 2; generator j of the original n sits in folded position j mod len with coefficient
// cg_j (gs side) / hf_j (hs side, y^-j included).  Job 2p is L, job 2p + 1 is R, n + 1 terms each in the base order of `bases`
// (host: the gs_j with (j mod len) >= half, then the hs_j with (j mod len) < half, then u, for L; the complements for R):
//   L = sum a_(i-half) cg_j gs_j + sum b_(i+half) hf_j hs_j + (x_ip cl) u,   i = j mod len
// Two kernels per round: k_pv_round_chal, ONE lane per proof (the transcript, its hash, the inversion: serial work, dense waves),
// and k_pv_round_wide, n lanes per proof (the folds and the next round's scalars: 2-6 multiplications per lane).  With one lane per
// proof for everything a round cost 1.4 ms beside its 3.1 ms of additions (profiles/r05_batch_prover_first_run.txt).

// L, R -> transcript -> x, 1 / x (inner_product_prover.py:100-106)
__global__ void __launch_bounds__(64) k_pv_round_chal(Batch B, u32 round) {
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B.P) return;
  const u32 npt = 6u + 2u * B.k;
  u8 *tr = B.tr + (size_t)p * B.tr_stride;
  u32 tl = B.tr_len[p];
  const u32 *pt = B.pts + 16ull * (size_t)p * npt;
  tl += put_point(tr + tl, pt + 16u * (6u + round));
  tl += put_point(tr + tl, pt + 16u * (6u + B.k + round));
  const sc x = mod_hash_q(0, 0, tr, tl);
  tl += put_number(tr + tl, x);
  B.tr_len[p] = tl;
  st_sc(B.xs + 8ull * ((size_t)p * B.k + round), x);
  st_sc(B.xr + 16ull * p, x);
  st_sc(B.xr + 16ull * p + 8, invq(x));
}
// n lanes per proof (block = 256 threads = 256 / n proofs; n <= 128).  round = the round whose challenge was just drawn: the state is
// folded with it (:107-110), then the scalars of the NEXT round's L and R are written; first = 1: no fold, the state is the one
// k_pv_final left (the scalars of round 0).  After the last fold a[0], b[0] are the proof's scalars.
__global__ void __launch_bounds__(256) k_pv_round_wide(Batch B, u32 round, u32 first) {
  __shared__ u32 s_a[256 * 8], s_b[256 * 8], s_l[256 * 8], s_r[256 * 8];
  const u32 n = B.n, tid = threadIdx.x;
  const u32 p = blockIdx.x * (256u / n) + tid / n, j = tid & (n - 1u);
  const bool live = p < B.P;
  const u32 base = tid - j;                                  // first thread of this proof in the block
  u32 len = first ? n : (n >> round);                        // length BEFORE this call's fold
  sc cgj = sc_u32(0), hfj = sc_u32(0);
  if (live) {
    u32 *a = B.a + 8ull * (size_t)p * n, *b = B.b + 8ull * (size_t)p * n;
    cgj = ld_sc(B.cg + 8ull * ((size_t)p * n + j)); hfj = ld_sc(B.hf + 8ull * ((size_t)p * n + j));
    if (!first) {
      const u32 half = len >> 1;
      const sc x = ld_sc(B.xr + 16ull * p), xi = ld_sc(B.xr + 16ull * p + 8);
      if (j < half) {
        const sc a0 = ld_sc(a + 8ull * j), a1 = ld_sc(a + 8ull * (half + j)), b0 = ld_sc(b + 8ull * j), b1 = ld_sc(b + 8ull * (half + j));
        const sc an = addq(mulq(x, a0), mulq(xi, a1)), bn = addq(mulq(xi, b0), mulq(x, b1));
        st_sc(s_a + 8u * tid, an); st_sc(s_b + 8u * tid, bn);
      }
      const bool low = (j & (len - 1u)) < half;
      cgj = mulq(cgj, low ? xi : x); hfj = mulq(hfj, low ? x : xi);
      st_sc(B.cg + 8ull * ((size_t)p * n + j), cgj); st_sc(B.hf + 8ull * ((size_t)p * n + j), hfj);
      len = half;
    } else {
      st_sc(s_a + 8u * tid, ld_sc(a + 8ull * j)); st_sc(s_b + 8u * tid, ld_sc(b + 8ull * j));
    }
  }
  __syncthreads();                                           // s_a / s_b [base + i], i < len: the folded state
  if (live && !first && j < len) {                           // (the folded halves go back in place: nobody reads the old ones any more)
    st_sc(B.a + 8ull * ((size_t)p * n + j), ld_sc(s_a + 8u * tid));
    st_sc(B.b + 8ull * ((size_t)p * n + j), ld_sc(s_b + 8u * tid));
  }
  if (len == 1u) {
    if (live && j == 0u) { u32 *res = B.res + 40ull * p; st_sc(res + 24, ld_sc(s_a + 8u * tid)); st_sc(res + 32, ld_sc(s_b + 8u * tid)); }
    return;                                                  // (block-uniform: len depends on the arguments only)
  }
  const u32 half = len >> 1, i = j & (len - 1u);
  // the products of c_L = <a_lo, b_hi>, c_R = <a_hi, b_lo>, summed over the proof's lanes in LDS
  sc pl = sc_u32(0), pr = sc_u32(0);
  if (live && j < half) {
    pl = mulq(ld_sc(s_a + 8u * (base + j)), ld_sc(s_b + 8u * (base + half + j)));
    pr = mulq(ld_sc(s_a + 8u * (base + half + j)), ld_sc(s_b + 8u * (base + j)));
  }
  st_sc(s_l + 8u * tid, pl); st_sc(s_r + 8u * tid, pr);
  __syncthreads();
  for (u32 d = n >> 1; d > 0u; d >>= 1) {
    if (j < d) {
      st_sc(s_l + 8u * tid, addq(ld_sc(s_l + 8u * tid), ld_sc(s_l + 8u * (tid + d))));
      st_sc(s_r + 8u * tid, addq(ld_sc(s_r + 8u * tid), ld_sc(s_r + 8u * (tid + d))));
    }
    __syncthreads();
  }
  if (!live) return;
  u32 *jl = B.jsc + 8ull * (size_t)(2u * p) * (n + 1u), *jr = jl + 8ull * (n + 1u);
  if (j == 0u) { st_sc(jl + 8ull * n, mulq(B.x_ip, ld_sc(s_l + 8u * tid))); st_sc(jr + 8ull * n, mulq(B.x_ip, ld_sc(s_r + 8u * tid))); }
  // generator j: rank among the generators of its side of the split = (j / len) half + (i mod half)
  const bool up = i >= half;
  const u32 rank = (j / len) * half + (up ? i - half : i);
  const sc ga = ld_sc(s_a + 8u * (base + (up ? i - half : i + half))), hb = ld_sc(s_b + 8u * (base + (up ? i - half : i + half)));
  st_sc((up ? jl : jr) + 8ull * rank, mulq(ga, cgj));                       // L: a_(i-half) cg_j for i >= half; R: a_(i+half) cg_j for i < half
  st_sc((up ? jr : jl) + 8ull * ((n >> 1) + rank), mulq(hb, hfj));          // L: b_(i+half) hf_j for i < half; R: b_(i-half) hf_j for i >= half
}

// ---- the proofs as wire format 2 or 3 (rangeproofs/codec.py; option "prover_wire_format"): "BPRP2" k | taux mu t_hat a b | xs | 6 + 2k compressed points | y z x x_ip |
// len seed | len Protocol-1 seed (empty) ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void put_be32(u8 *dst, const sc &v) {
  for (int i = 0; i < 32; i++) dst[i] = (u8)(v.v[7 - (i >> 2)] >> (8 * (3 - (i & 3))));
}
__global__ void __launch_bounds__(64) k_pv_emit(Batch B, const u8 *__restrict__ seeds, const u64 *__restrict__ seed_off, u8 *__restrict__ out, const u64 *__restrict__ out_off,
                                                u32 fmt) {
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= B.P) return;
  u8 *o = out + out_off[p];
  const u32 k = B.k, npt = 6u + 2u * k;
  u8 *ys = fmt == 3u ? out + out_off[p + 1] - 32u * npt : nullptr;              // format 3: the points' y coordinates end the proof
  o[0] = 'B'; o[1] = 'P'; o[2] = 'R'; o[3] = 'P'; o[4] = (u8)('0' + fmt); o[5] = (u8)k;
  u32 pos = 6;
  for (u32 j = 0; j < 5u; j++) { put_be32(o + pos, ld_sc(B.res + 40ull * p + 8u * j)); pos += 32; }
  for (u32 j = 0; j < k; j++) { put_be32(o + pos, ld_sc(B.xs + 8ull * ((size_t)p * k + j))); pos += 32; }
  const u32 *pt = B.pts + 16ull * (size_t)p * npt;
  for (u32 j = 0; j < npt; j++) {
    u32 w[16];
    if (j == PV_PT_UNEW) { for (int i = 0; i < 16; i++) w[i] = B.u_new[i]; }
    else ::load_words16(w, pt + 16u * j);
    u32 any = 0;
    for (int i = 0; i < 16; i++) any |= w[i];
    if (!any) { for (int i = 0; i < 33; i++) o[pos + i] = 0; }
    else {
      o[pos] = (u8)(2u + (w[8] & 1u));
      for (int i = 0; i < 32; i++) o[pos + 1 + i] = (u8)(w[7 - (i >> 2)] >> (8 * (3 - (i & 3))));
    }
    if (ys) for (int i = 0; i < 32; i++) ys[32u * j + i] = (u8)(w[15 - (i >> 2)] >> (8 * (3 - (i & 3))));      // (zeros for the identity)
    pos += 33;
  }
  for (u32 j = 0; j < 3u; j++) { put_be32(o + pos, ld_sc(B.chal + 32ull * p + 8u * j)); pos += 32; }
  put_be32(o + pos, B.x_ip); pos += 32;
  const u64 s0 = seed_off[p], sl = seed_off[p + 1] - s0;
  o[pos] = (u8)(sl >> 8); o[pos + 1] = (u8)sl; pos += 2;
  for (u64 i = 0; i < sl; i++) o[pos + i] = seeds[s0 + i];
  pos += (u32)sl;
  o[pos] = 0; o[pos + 1] = 0;
}

}  // namespace rpp
