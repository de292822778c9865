// rp_batch_host.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).  HOST code.
// Per-proof preparation of the random-linear-combination batch verifier for range
// proofs (single-value or aggregated) in wire format (python-bulletproofs_amd/rangeproofs/batch.py, codec.py): parsing, the
// three byte-level transcript checks of the reference's verifiers, and the weighted scalars of the
// ONE MSM that the GPU then evaluates.  It is the native twin of BatchRangeVerifier.add(): the
// Python version costs ~150 us per proof, the GPU ~1 us, so at 2^14 proofs the interpreter was
// the whole bill.  tests/test_batch_native_cpu.py checks this file against the Python path
// scalar by scalar (same weights in, same numbers out) and verdict by verdict.
//
// What is checked per proof (reference call sites in brackets):
//   range-proof transcript   items 1,2,5,6 are base64(A), base64(S), base64(T1), base64(T2);
//                            y, z, x are READ from items 3,4,7 (rangeproof_verifier.py:42-53)
//   Protocol-1 transcript    item 1 == str(mod_hash(item 0 + "&"))        (inner_product_verifier.py:31-43)
//   Protocol-2 transcript    per round i: items s+3i, s+3i+1 are base64(L_i), base64(R_i) and
//                            str(x_i) == item s+3i+2 == str(mod_hash(prefix))   (:104-125)
// Numbers in transcripts must be plain decimal digits without leading zeros (what str(int) prints).
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <cpuid.h>
#include <immintrin.h>
#endif

#include <thread>
#include <vector>

namespace rp {

typedef unsigned __int128 u128;
typedef uint64_t u64;

// ---- arithmetic mod q (group order), 4 x 64-bit limbs, values in [0, q) -----------------------
struct Sq { u64 v[4]; };
static const u64 QW[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};
static const u64 QC[3] = {0x402DA1732FC9BEBFULL, 0x4551231950B75FC4ULL, 1ULL};      // 2^256 - q

static inline bool q_is_zero(const Sq &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static inline bool ge_q(const u64 a[4]) {
  for (int i = 3; i >= 0; i--) if (a[i] != QW[i]) return a[i] > QW[i];
  return true;
}
static inline void sub_q(u64 a[4]) {
  u64 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - QW[i] - br; a[i] = (u64)t; br = (u64)(t >> 64) & 1; }
}
static inline void q_add(Sq &r, const Sq &a, const Sq &b) {
  u64 c = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a.v[i] + b.v[i] + c; r.v[i] = (u64)t; c = (u64)(t >> 64); }
  if (c || ge_q(r.v)) sub_q(r.v);
}
static inline void q_sub(Sq &r, const Sq &a, const Sq &b) {
  u64 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a.v[i] - b.v[i] - br; r.v[i] = (u64)t; br = (u64)(t >> 64) & 1; }
  if (br) { u64 c = 0; for (int i = 0; i < 4; i++) { u128 t = (u128)r.v[i] + QW[i] + c; r.v[i] = (u64)t; c = (u64)(t >> 64); } }
}
static inline void q_neg(Sq &r, const Sq &a) { Sq z = {{0, 0, 0, 0}}; q_sub(r, z, a); }
// t[0..n) (n <= 8 limbs) -> value mod q: fold the limbs above 256 bits with 2^256 == QC until none are left
static inline void q_reduce_wide(Sq &r, u64 t[8], int n) {
  while (n > 4) {
    const int nh = n - 4;
    u64 m[8] = {0};
    // m = hi * QC  (nh x 3 limbs)
    for (int i = 0; i < nh; i++) {
      u128 c = 0;
      for (int j = 0; j < 3; j++) { c += (u128)t[4 + i] * QC[j] + m[i + j]; m[i + j] = (u64)c; c >>= 64; }
      int k = i + 3;
      while (c) { c += m[k]; m[k] = (u64)c; c >>= 64; k++; }
    }
    // t = lo + m
    u128 c = 0;
    int top = 0;
    for (int i = 0; i < 8; i++) {
      c += (u128)(i < 4 ? t[i] : 0) + m[i];
      t[i] = (u64)c;
      c >>= 64;
      if (t[i]) top = i + 1;
    }
    n = top > 4 ? top : 4;
  }
  memcpy(r.v, t, 32);
  while (ge_q(r.v)) sub_q(r.v);
}
static inline void q_mul(Sq &r, const Sq &a, const Sq &b) {
  u64 t[8] = {0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a.v[i] * b.v[j] + t[i + j]; t[i + j] = (u64)c; c >>= 64; }
    t[i + 4] = (u64)c;
  }
  q_reduce_wide(r, t, 8);
}
static inline void q_sqr(Sq &r, const Sq &a) { q_mul(r, a, a); }
static inline void q_inv(Sq &r, const Sq &a) {            // a^(q-2); called once per chunk of proofs
  static const u64 E[4] = {0xBFD25E8CD036413FULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};
  Sq acc = {{1, 0, 0, 0}};
  for (int i = 255; i >= 0; i--) {
    q_sqr(acc, acc);
    if ((E[i >> 6] >> (i & 63)) & 1) q_mul(acc, acc, a);
  }
  r = acc;
}
static inline Sq q_small(u64 x) { Sq r = {{x, 0, 0, 0}}; return r; }
static inline void q_from_be(Sq &r, const uint8_t *b, bool &lt_q) {          // 32 bytes big-endian
  for (int i = 0; i < 4; i++) { u64 w = 0; for (int k = 0; k < 8; k++) w = (w << 8) | b[8 * (3 - i) + k]; r.v[i] = w; }
  lt_q = !ge_q(r.v);
}
static inline void q_from_le(Sq &r, const uint8_t *b) { memcpy(r.v, b, 32); while (ge_q(r.v)) sub_q(r.v); }
static inline void q_to_le(uint8_t *b, const Sq &a) { memcpy(b, a.v, 32); }
static inline bool q_eq(const Sq &a, const Sq &b) { return memcmp(a.v, b.v, 32) == 0; }

// ---- SHA-256 (FIPS 180-4), incremental with copyable state -------------------------------------
struct Sha {
  uint32_t h[8];
  uint8_t buf[64];
  u64 len;
  uint32_t fill;
};
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define BPMI_SHA_NI 1
// The same compression function on the SHA extensions of the host CPU (sha256rnds2 / sha256msg1 / sha256msg2), chosen at run time
// (CPUID leaf 7, EBX bit 29).  The provers' seeded blinding vectors are 2 n m hashes (16 384 for a 128 x 64-bit proof: 0.68 ms of a
// 8.6 ms proof on eight threads with the portable code), every challenge of every transcript is two more.
// tests/test_csrc_host.py::test_sha_block_on_cpu_extensions_equals_the_portable_one pins it to sha_block_portable.
__attribute__((target("sha,sse4.1,ssse3"))) static inline void sha_block_ni(uint32_t h[8], const uint8_t *p) {
  const __m128i flip = _mm_set_epi64x(0x0c0d0e0f08090a0bULL, 0x0405060700010203ULL);
  __m128i t = _mm_loadu_si128((const __m128i *)&h[0]);                // DCBA (a in lane 0)
  __m128i s1 = _mm_loadu_si128((const __m128i *)&h[4]);               // HGFE
  t = _mm_shuffle_epi32(t, 0xB1);                                     // CDAB
  s1 = _mm_shuffle_epi32(s1, 0x1B);                                   // EFGH
  __m128i s0 = _mm_alignr_epi8(t, s1, 8);                             // ABEF
  s1 = _mm_blend_epi16(s1, t, 0xF0);                                  // CDGH
  const __m128i save0 = s0, save1 = s1;
  __m128i m[4];
  for (int g = 0; g < 16; g++) {
    if (g < 4) m[g] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(p + 16 * g)), flip);
    else {
      const __m128i w4 = m[g & 3], w3 = m[(g + 1) & 3], w2 = m[(g + 2) & 3], w1 = m[(g + 3) & 3];      // W[g-4], W[g-3], W[g-2], W[g-1] (4 words each)
      m[g & 3] = _mm_sha256msg2_epu32(_mm_add_epi32(_mm_sha256msg1_epu32(w4, w3), _mm_alignr_epi8(w1, w2, 4)), w1);
    }
    __m128i x = _mm_add_epi32(m[g & 3], _mm_loadu_si128((const __m128i *)&SHA_K[4 * g]));
    s1 = _mm_sha256rnds2_epu32(s1, s0, x);
    x = _mm_shuffle_epi32(x, 0x0E);
    s0 = _mm_sha256rnds2_epu32(s0, s1, x);
  }
  s0 = _mm_add_epi32(s0, save0);
  s1 = _mm_add_epi32(s1, save1);
  t = _mm_shuffle_epi32(s0, 0x1B);                                    // FEBA
  s1 = _mm_shuffle_epi32(s1, 0xB1);                                   // DCHG
  s0 = _mm_blend_epi16(t, s1, 0xF0);                                  // DCBA
  s1 = _mm_alignr_epi8(s1, t, 8);                                     // HGFE
  _mm_storeu_si128((__m128i *)&h[0], s0);
  _mm_storeu_si128((__m128i *)&h[4], s1);
}
static inline bool cpu_has_sha_extensions() {
  unsigned a = 0, b = 0, c = 0, d = 0;
  if (!__get_cpuid_count(7, 0, &a, &b, &c, &d)) return false;
  const bool sha = (b >> 29) & 1u;
  if (!__get_cpuid(1, &a, &b, &c, &d)) return false;
  return sha && ((c >> 19) & 1u) && ((c >> 9) & 1u);                  // + SSE4.1, SSSE3
}
static const bool g_sha_ni = cpu_has_sha_extensions();
#endif
static inline void sha_block_portable(uint32_t h[8], const uint8_t *p) {
  uint32_t w[64];
  for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
  for (int i = 16; i < 64; i++) {
    const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
    const uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; i++) {
    const uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
    const uint32_t ch = (e & f) ^ (~e & g);
    const uint32_t t1 = hh + S1 + ch + SHA_K[i] + w[i];
    const uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
    const uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    const uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
static inline void sha_block(uint32_t h[8], const uint8_t *p) {
#if defined(BPMI_SHA_NI)
  if (g_sha_ni) { sha_block_ni(h, p); return; }
#endif
  sha_block_portable(h, p);
}
static inline void sha_init(Sha &s) {
  static const uint32_t H0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(s.h, H0, 32);
  s.len = 0; s.fill = 0;
}
static inline void sha_update(Sha &s, const uint8_t *p, size_t n) {
  s.len += n;
  while (n) {
    const size_t take = (64 - s.fill < n) ? 64 - s.fill : n;
    memcpy(s.buf + s.fill, p, take);
    s.fill += (uint32_t)take; p += take; n -= take;
    if (s.fill == 64) { sha_block(s.h, s.buf); s.fill = 0; }
  }
}
static inline void sha_final_copy(const Sha &s0, uint8_t out[32]) {     // digest of a COPY: s0 can go on absorbing
  Sha s = s0;
  const u64 bits = s.len * 8;
  uint8_t pad[72] = {0x80};
  const size_t padlen = (s.fill < 56) ? 56 - s.fill : 120 - s.fill;
  sha_update(s, pad, padlen);
  uint8_t lb[8];
  for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
  sha_update(s, lb, 8);
  for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(s.h[i] >> 24); out[4 * i + 1] = (uint8_t)(s.h[i] >> 16); out[4 * i + 2] = (uint8_t)(s.h[i] >> 8); out[4 * i + 3] = (uint8_t)s.h[i]; }
}
// mod_hash(msg, q): first i >= 1 with SHA-256(str(i) || msg) in [1, q)   (src/utils/utils.py:84-97).
// `one` is the state after absorbing "1" and the part of msg hashed so far (the common case i = 1).
static inline void mod_hash_q(Sq &r, const Sha &one, const uint8_t *msg, size_t n) {
  uint8_t d[32];
  sha_final_copy(one, d);
  bool lt;
  q_from_be(r, d, lt);
  for (unsigned i = 2; !lt || q_is_zero(r); i++) {            // probability ~2^-128 per challenge
    char pre[16];
    const int pl = snprintf(pre, sizeof(pre), "%u", i);
    Sha s;
    sha_init(s);
    sha_update(s, (const uint8_t *)pre, (size_t)pl);
    sha_update(s, msg, n);
    sha_final_copy(s, d);
    q_from_be(r, d, lt);
  }
}

// ---- small codecs ------------------------------------------------------------------------------
static inline size_t b64_encode(uint8_t *out, const uint8_t *in, size_t n) {
  static const char T[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  size_t o = 0;
  for (size_t i = 0; i < n; i += 3) {
    const uint32_t b0 = in[i], b1 = i + 1 < n ? in[i + 1] : 0, b2 = i + 2 < n ? in[i + 2] : 0;
    const uint32_t v = (b0 << 16) | (b1 << 8) | b2;
    out[o++] = T[(v >> 18) & 63];
    out[o++] = T[(v >> 12) & 63];
    out[o++] = i + 1 < n ? T[(v >> 6) & 63] : '=';
    out[o++] = i + 2 < n ? T[v & 63] : '=';
  }
  return o;
}
// the transcript item of a point: base64 of its encoding (33 zero bytes in the wire format = identity = b"\x00")
static inline size_t point_item(uint8_t out[48], const uint8_t comp[33]) {
  bool zero = true;
  for (int i = 0; i < 33; i++) zero &= comp[i] == 0;
  const uint8_t z = 0;
  return zero ? b64_encode(out, &z, 1) : b64_encode(out, comp, 33);
}
// canonical decimal -> value mod q (false: not canonical decimal, or >= 2^256)
static inline bool parse_decimal(Sq &r, const uint8_t *p, size_t n) {
  if (n == 0 || n > 78 || (n > 1 && p[0] == '0')) return false;
  u64 t[5] = {0, 0, 0, 0, 0};
  for (size_t i = 0; i < n; i++) {
    if (p[i] < '0' || p[i] > '9') return false;
    u128 c = p[i] - '0';
    for (int k = 0; k < 5; k++) { c += (u128)t[k] * 10; t[k] = (u64)c; c >>= 64; }
  }
  if (t[4]) return false;
  memcpy(r.v, t, 32);
  while (ge_q(r.v)) sub_q(r.v);
  return true;
}

// items of a '&'-separated transcript: item j = [off[j], off[j+1] - 1)
struct Items {
  const uint8_t *base;
  std::vector<uint32_t> off;      // off[j] = start of item j; off.back() = len + 1
  void split(const uint8_t *p, size_t n) {
    base = p;
    off.clear();
    off.push_back(0);
    for (size_t i = 0; i < n; i++) if (p[i] == '&') off.push_back((uint32_t)i + 1);
    off.push_back((uint32_t)n + 1);
  }
  size_t count() const { return off.size() - 1; }
  const uint8_t *ptr(size_t j) const { return base + off[j]; }
  size_t len(size_t j) const { return off[j + 1] - 1 - off[j]; }
  bool equals(size_t j, const uint8_t *q, size_t n) const { return j < count() && len(j) == n && memcmp(ptr(j), q, n) == 0; }
};

// ---- one proof ------------------------------------------------------------------------------------
struct Parsed {
  uint32_t k;
  Sq taux, mu, t_hat, a, b;
  Sq xs[16];
  const uint8_t *comp;            // (6 + 2k) x 33 bytes: T1 T2 A S u_new P_new Ls Rs
  uint32_t start;
  const uint8_t *ts[3];
  uint32_t tl[3];
};
static inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
#define RP_HOST_MAX_PROOF_BYTES 32768u      // = RP_MAX_PROOF_BYTES of the device twin: a longer wire proof is invalid
static inline bool parse_blob(Parsed &P, const uint8_t *blob, size_t n) {
  if (n < 6 || n > RP_HOST_MAX_PROOF_BYTES || memcmp(blob, "BPRP1", 5) != 0) return false;
  P.k = blob[5];
  if (P.k > 16) return false;
  size_t o = 6;
  if (n < o + 32 * (5 + P.k) + 33 * (6 + 2 * P.k) + 2) return false;
  Sq *dst[5] = {&P.taux, &P.mu, &P.t_hat, &P.a, &P.b};
  bool lt;
  for (int j = 0; j < 5; j++) { q_from_be(*dst[j], blob + o, lt); if (!lt) return false; o += 32; }
  for (uint32_t j = 0; j < P.k; j++) { q_from_be(P.xs[j], blob + o, lt); if (!lt) return false; o += 32; }
  P.comp = blob + o;
  o += 33 * (6 + 2 * P.k);
  P.start = ((uint32_t)blob[o] << 8) | blob[o + 1];
  o += 2;
  for (int t = 0; t < 3; t++) {
    if (n < o + 4) return false;
    P.tl[t] = be32(blob + o);
    o += 4;
    if (n < o + P.tl[t]) return false;
    P.ts[t] = blob + o;
    o += P.tl[t];
  }
  return o == n;
}

struct Challenges { Sq x, y, z, x_ip; };

// the three transcript checks; fills the challenges
static inline bool check_transcripts(const Parsed &P, Challenges &C, Items &it) {
  uint8_t item[48];
  const uint8_t *T1 = P.comp, *T2 = P.comp + 33, *A = P.comp + 66, *S = P.comp + 99;
  // range-proof transcript
  it.split(P.ts[0], P.tl[0]);
  if (it.count() < 8) return false;
  size_t l;
  l = point_item(item, A);  if (!it.equals(1, item, l)) return false;
  l = point_item(item, S);  if (!it.equals(2, item, l)) return false;
  if (!parse_decimal(C.y, it.ptr(3), it.len(3))) return false;
  if (!parse_decimal(C.z, it.ptr(4), it.len(4))) return false;
  l = point_item(item, T1); if (!it.equals(5, item, l)) return false;
  l = point_item(item, T2); if (!it.equals(6, item, l)) return false;
  if (!parse_decimal(C.x, it.ptr(7), it.len(7))) return false;
  // Protocol 1: item 1 is the decimal of mod_hash(item 0 + "&")
  it.split(P.ts[1], P.tl[1]);
  if (it.count() < 2) return false;
  {
    Sha s;
    sha_init(s);
    sha_update(s, (const uint8_t *)"1", 1);
    sha_update(s, P.ts[1], it.off[1]);                      // item 0 and its '&'
    Sq h;
    mod_hash_q(h, s, P.ts[1], it.off[1]);
    if (!parse_decimal(C.x_ip, it.ptr(1), it.len(1)) || !q_eq(C.x_ip, h)) return false;
  }
  // Protocol 2: L_i, R_i, x_i per round, x_i re-hashed from the prefix that ends after R_i's '&'
  it.split(P.ts[2], P.tl[2]);
  const uint8_t *Ls = P.comp + 33 * 6, *Rs = Ls + 33 * P.k;
  Sha run;
  sha_init(run);
  sha_update(run, (const uint8_t *)"1", 1);
  size_t hashed = 0;
  for (uint32_t i = 0; i < P.k; i++) {
    const size_t j = (size_t)P.start + 3 * i;
    if (j + 2 >= it.count()) return false;
    l = point_item(item, Ls + 33 * i); if (!it.equals(j, item, l)) return false;
    l = point_item(item, Rs + 33 * i); if (!it.equals(j + 1, item, l)) return false;
    const size_t upto = it.off[j + 2];                       // prefix incl. the '&' after item j+1
    sha_update(run, P.ts[2] + hashed, upto - hashed);
    hashed = upto;
    Sq h, xi;
    mod_hash_q(h, run, P.ts[2], upto);
    if (!parse_decimal(xi, it.ptr(j + 2), it.len(j + 2))) return false;
    if (!q_eq(xi, h) || !q_eq(xi, P.xs[i])) return false;
  }
  return true;
}

// Everything add() accumulates for one proof.  out_v: 1 scalar (for V); out_p: (6 + 2k) scalars in the
// wire order of the points (T1 T2 A S u_new P_new Ls Rs); acc: c_g c_h c_u gs_const hs_const c_gs[n] c_hs[n].
struct Work { std::vector<Sq> sg, sh, tmp; };
// m = values per proof (1: single proof; > 1: aggregated, n = m * bits per value)
static inline void accumulate(const Parsed &P, const Challenges &C, const Sq w[4], const Sq *xinvs, const Sq &yinv, uint32_t n, uint32_t m,
                              Sq *acc, Sq *out_v, Sq *out_p, Work &W) {
  const uint32_t k = P.k;
  Sq t, u;
  // s-vector by doubling with the weights folded in (batch.py add()): sg_i = w4 a s_i, sh_i = w4 b s_i^-1 y^-i
  W.sg.resize(n); W.sh.resize(n);
  q_mul(W.sg[0], w[3], P.a);
  q_mul(W.sh[0], w[3], P.b);
  Sq ypow2 = yinv;
  uint32_t len = 1;
  for (int j = (int)k - 1; j >= 0; j--) {
    const Sq &xv = P.xs[j], &xi = xinvs[j];
    Sq hi_h;
    q_mul(hi_h, xi, ypow2);
    for (uint32_t i = 0; i < len; i++) {
      q_mul(W.sg[len + i], W.sg[i], xv);
      q_mul(W.sg[i], W.sg[i], xi);
      q_mul(W.sh[len + i], W.sh[i], hi_h);
      q_mul(W.sh[i], W.sh[i], xv);
    }
    q_sqr(ypow2, ypow2);
    len <<= 1;
  }
  Sq z2, w2z, geo, r2;
  q_sqr(z2, C.z);
  q_mul(w2z, w[1], C.z);
  q_add(acc[3], acc[3], w2z);                                   // gs_const
  q_sub(acc[4], acc[4], w2z);                                   // hs_const
  q_add(r2, yinv, yinv);                                        // 2 / y
  const uint32_t bits = n / m;
  Sq yn_inv = q_small(1);
  for (uint32_t i = 0; i < bits; i++) q_mul(yn_inv, yn_inv, yinv);     // y^-bits
  W.tmp.resize(m);                                               // z^(2 + j)
  W.tmp[0] = z2;
  for (uint32_t j = 1; j < m; j++) q_mul(W.tmp[j], W.tmp[j - 1], C.z);
  Sq *c_gs = acc + 5, *c_hs = acc + 5 + n;
  Sq blk = q_small(1);
  for (uint32_t j = 0, i = 0; j < m; j++) {
    q_mul(geo, w[1], W.tmp[j]);
    q_mul(geo, geo, blk);                                       // w2 z^(2+j) 2^(i % bits) y^-i at i = bits j
    for (uint32_t e = 0; e < bits; e++, i++) {
      q_add(c_gs[i], c_gs[i], W.sg[i]);
      q_sub(t, W.sh[i], geo);
      q_add(c_hs[i], c_hs[i], t);
      q_mul(geo, geo, r2);
    }
    q_mul(blk, blk, yn_inv);
  }
  // sum_{i<n} y^i by doubling; delta = (z - z^2) ysum - z^3 (2^n - 1)
  Sq ysum = q_small(1), ypw = C.y, one = q_small(1);
  for (uint32_t l2 = 1; l2 < n; l2 <<= 1) {
    q_add(t, one, ypw);
    q_mul(ysum, ysum, t);
    q_sqr(ypw, ypw);
  }
  Sq two_n = q_small(1), two = q_small(2);
  for (uint32_t i = 0; i < bits; i++) q_mul(two_n, two_n, two);  // 2^bits mod q
  q_sub(two_n, two_n, one);
  // delta = (z - z^2) ysum - (2^bits - 1) sum_{j=1..m} z^(j+2)
  Sq delta, zsum = q_small(0), zp;
  q_sub(t, C.z, z2);
  q_mul(delta, t, ysum);
  q_mul(zp, z2, C.z);                                            // z^3
  for (uint32_t j = 1; j <= m; j++) { q_add(zsum, zsum, zp); q_mul(zp, zp, C.z); }
  q_mul(t, zsum, two_n);
  q_sub(delta, delta, t);
  // c_g += w1 (t_hat - delta); c_h += w1 taux + w2 mu; c_u -= w2 x_ip t_hat + w3 x_ip
  q_sub(t, P.t_hat, delta); q_mul(t, t, w[0]); q_add(acc[0], acc[0], t);
  q_mul(t, w[0], P.taux); q_mul(u, w[1], P.mu); q_add(t, t, u); q_add(acc[1], acc[1], t);
  q_mul(t, w[1], C.x_ip); q_mul(t, t, P.t_hat); q_mul(u, w[2], C.x_ip); q_add(t, t, u); q_sub(acc[2], acc[2], t);
  // per-proof points: V: -w1 z^2 | T1: -w1 x | T2: -w1 x^2 | A: -w2 | S: -w2 x | u_new: w3 + w4 a b | P_new: w2 - w4
  for (uint32_t j = 0; j < m; j++) { q_mul(t, w[0], W.tmp[j]); q_neg(out_v[j], t); }
  q_mul(t, w[0], C.x); q_neg(out_p[0], t);
  q_mul(t, t, C.x); q_neg(out_p[1], t);
  q_neg(out_p[2], w[1]);
  q_mul(t, w[1], C.x); q_neg(out_p[3], t);
  q_mul(t, w[3], P.a); q_mul(t, t, P.b); q_add(out_p[4], w[2], t);
  q_sub(out_p[5], w[1], w[3]);
  for (uint32_t j = 0; j < k; j++) {
    q_mul(t, w[3], P.xs[j]); q_mul(t, t, P.xs[j]); q_neg(out_p[6 + j], t);
    q_mul(t, w[3], xinvs[j]); q_mul(t, t, xinvs[j]); q_neg(out_p[6 + k + j], t);
  }
}

// proofs [lo, hi): returns false at the first invalid proof (its index in *bad)
// weights == nullptr: proof g's four weights are SHA-256(seed || LE64(g) || t), t = 0..3, cut to 248 bits (< q by
// construction; 0 is replaced by 1) -- a fresh 32-byte seed per batch makes them unpredictable to whoever made the proofs
static inline void derive_weight(Sq &w, const uint8_t seed[32], u64 g, int t) {
  uint8_t msg[41], d[32];
  memcpy(msg, seed, 32);
  for (int i = 0; i < 8; i++) msg[32 + i] = (uint8_t)(g >> (8 * i));
  msg[40] = (uint8_t)t;
  Sha s;
  sha_init(s);
  sha_update(s, msg, 41);
  sha_final_copy(s, d);
  d[31] = 0;
  memcpy(w.v, d, 32);
  if (q_is_zero(w)) w = q_small(1);
}
static inline bool run_chunk(uint32_t n, uint32_t k, uint32_t m, const uint8_t *blobs, const u64 *off, const uint8_t *weights, u64 lo, u64 hi,
                             const u64 *pt_off, uint8_t *v_scalars, uint8_t *pt_scalars, uint8_t *comp_out, Sq *acc, u64 *bad,
                             const uint8_t *seed = nullptr) {
  const u64 cnt = hi - lo;
  std::vector<Parsed> P(cnt);
  std::vector<Challenges> C(cnt);
  Items it;
  for (u64 j = 0; j < cnt; j++) {
    const u64 g = lo + j;
    if (!parse_blob(P[j], blobs + off[g], (size_t)(off[g + 1] - off[g])) || P[j].k != k || !check_transcripts(P[j], C[j], it)) { *bad = g; return false; }
    if (comp_out) memcpy(comp_out + 33 * pt_off[g], P[j].comp, 33 * (6 + 2 * (size_t)k));
  }
  // one inversion for every x_j and y of the chunk (Montgomery's trick)
  const u64 per = k + 1, total = cnt * per;
  std::vector<Sq> val(total), pre(total), inv(total);
  Sq run = q_small(1);
  for (u64 j = 0; j < cnt; j++) {
    for (uint32_t t = 0; t <= k; t++) {
      const Sq &v = t < k ? P[j].xs[t] : C[j].y;
      if (q_is_zero(v)) { *bad = lo + j; return false; }          // a zero challenge cannot come out of mod_hash
      val[j * per + t] = v;
      pre[j * per + t] = run;
      q_mul(run, run, v);
    }
  }
  Sq rinv;
  q_inv(rinv, run);
  for (u64 idx = total; idx-- > 0;) {
    q_mul(inv[idx], rinv, pre[idx]);
    q_mul(rinv, rinv, val[idx]);
  }
  Work W;
  std::vector<Sq> outp(6 + 2 * k), ov(m);
  for (u64 j = 0; j < cnt; j++) {
    const u64 g = lo + j;
    Sq w[4];
    for (int t = 0; t < 4; t++) { if (weights) q_from_le(w[t], weights + (g * 4 + t) * 32); else derive_weight(w[t], seed, g, t); }
    accumulate(P[j], C[j], w, &inv[j * per], inv[j * per + k], n, m, acc, ov.data(), outp.data(), W);
    for (uint32_t t = 0; t < m; t++) q_to_le(v_scalars + 32 * (g * m + t), ov[t]);
    for (uint32_t t = 0; t < 6 + 2 * k; t++) q_to_le(pt_scalars + 32 * (pt_off[g] + t), outp[t]);
  }
  return true;
}

}  // namespace rp
