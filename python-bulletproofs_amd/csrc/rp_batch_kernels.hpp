// rp_batch_kernels.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).  DEVICE code.
// The per-proof preparation of the batch range-proof verifier on the GPU: one lane per proof parses the wire blob,
// re-hashes the three Fiat-Shamir transcripts (SHA-256), runs the byte-level checks of the reference's verifiers
// (rangeproof_verifier.py:42-53, inner_product_verifier.py:31-43 and :104-125) and produces the weighted scalars of
// the batch's one MSM.  It is the device twin of rp_batch_host.hpp (same checks, same numbers: tests/test_gpu_batch_dev.py
// compares the two byte for byte on the same weights); see that file for what each check means.
//
// Layout: the (5 + 2n) shared-generator contributions of proof g live in cells (col, g) of nine 29-bit limbs, limb-major
// (contrib[(col * 9 + limb) * P + g]: the lanes of a wave touch neighbouring dwords), summed over g by k_rp_colsum.  The s-vector doubling
// works in place inside those columns, so a lane needs no O(n) private memory.
// Occupancy: a wave issues ~230 000 instructions here at the wave64 minimum of 4 cycles each (profiles/
// r02_C5_prepare_kernel_pmc.txt), and an instruction costs those 4 cycles whatever its active-lane count; 2^14 proofs are
// only 2 x 256 full waves on 1024 SIMDs.  So full waves (`lanes` = 64, the default) are the fastest launch: 32 proofs per
// wave take the same time, 16 x1.9, 8 x8 (profiles/r02_C5_prepare_kernel.txt).
#pragma once

namespace rpd {

using bpmi::sc;
using bpmi::sq;
using bpmi::u32;
using bpmi::u64;

// ---- SHA-256 with the 64-byte block as a 16-word shift register (no indexed private memory) -------------------
struct Sha {
  u32 h[8];
  u32 w[16];
  u32 cur;        // bytes of the word being assembled
  u32 fill;       // bytes absorbed into the current block
  u32 len;        // total bytes absorbed
};
__device__ __constant__ const u32 SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
__device__ __forceinline__ u32 rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }
// One compression, out of line, everything in registers: 24 scalar arguments (8 chaining words, 16 message words -- scalars
// always travel in VGPRs, an aggregate of 24 words would go through scratch) and the 8 new chaining words back.  ONE copy of
// the ~2000-instruction round function in the whole kernel: with the rounds inlined at every feeding site the hashing role's
// hot code (~60 KB) did not fit the instruction cache that a pair of CUs share.
struct H8 { u32 v[8]; };
__device__ __noinline__ H8 sha_compress_v(u32 h0, u32 h1, u32 h2, u32 h3, u32 h4, u32 h5, u32 h6, u32 h7, u32 w0, u32 w1, u32 w2, u32 w3, u32 w4, u32 w5,
                                          u32 w6, u32 w7, u32 w8, u32 w9, u32 w10, u32 w11, u32 w12, u32 w13, u32 w14, u32 w15) {
  u32 w[16] = {w0, w1, w2, w3, w4, w5, w6, w7, w8, w9, w10, w11, w12, w13, w14, w15};
  u32 a = h0, b = h1, c = h2, d = h3, e = h4, f = h5, g = h6, hh = h7;
#pragma unroll
  for (int i = 0; i < 64; i++) {
    if (i >= 16) {
      const u32 x = w[(i - 15) & 15], y = w[(i - 2) & 15];
      const u32 s0 = rotr(x, 7) ^ rotr(x, 18) ^ (x >> 3);
      const u32 s1 = rotr(y, 17) ^ rotr(y, 19) ^ (y >> 10);
      w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
    }
    const u32 S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
    const u32 ch = (e & f) ^ (~e & g);
    const u32 t1 = hh + S1 + ch + SHA_K[i] + w[i & 15];
    const u32 S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
    const u32 mj = (a & b) ^ (a & c) ^ (b & c);
    const u32 t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  H8 r = {{h0 + a, h1 + b, h2 + c, h3 + d, h4 + e, h5 + f, h6 + g, h7 + hh}};
  return r;
}
__device__ __forceinline__ void sha_compress(Sha &s) {
  const H8 r = sha_compress_v(s.h[0], s.h[1], s.h[2], s.h[3], s.h[4], s.h[5], s.h[6], s.h[7], s.w[0], s.w[1], s.w[2], s.w[3], s.w[4], s.w[5], s.w[6],
                              s.w[7], s.w[8], s.w[9], s.w[10], s.w[11], s.w[12], s.w[13], s.w[14], s.w[15]);
#pragma unroll
  for (int i = 0; i < 8; i++) s.h[i] = r.v[i];
}
__device__ __forceinline__ void sha_init(Sha &s) {
  const u32 H0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
#pragma unroll
  for (int i = 0; i < 8; i++) s.h[i] = H0[i];
#pragma unroll
  for (int i = 0; i < 16; i++) s.w[i] = 0;
  s.cur = 0; s.fill = 0; s.len = 0;
}
__device__ __forceinline__ void sha_byte_inl(Sha &s, u32 b) {
  s.cur = (s.cur << 8) | b;
  s.fill++;
  s.len++;
  if ((s.fill & 3u) == 0) {
#pragma unroll
    for (int i = 0; i < 15; i++) s.w[i] = s.w[i + 1];
    s.w[15] = s.cur;
    if (s.fill == 64u) { sha_compress(s); s.fill = 0; }
  }
}
// The three routines below work on a private COPY of the state: the bytes come in through a char pointer, which may alias
// anything, so updating the caller's object in place would force a store of the shift register per byte.
__device__ __noinline__ void sha_byte(Sha &s, u32 b) {          // odd bytes (prefixes, the weight derivation)
  Sha t = s;
  sha_byte_inl(t, b);
  s = t;
}
// Unaligned 8-byte load.  Byte-at-a-time loops over proof data pay one memory round trip (~0.3 us, nothing else in the wave
// to hide it) per byte; everything below that walks the wire bytes fetches eight at a time.  The load may cover up to 7 bytes
// beyond the item it is used for: the staged batch has 64 bytes of slack after the last proof, and bytes outside the item
// are never interpreted.
__device__ __forceinline__ u64 ld8(const uint8_t *p) { u64 w; __builtin_memcpy(&w, p, 8); return w; }
__device__ __forceinline__ void sha_word_inl(Sha &s, u32 w) {            // four bytes at a word boundary of the message
#pragma unroll
  for (int i = 0; i < 15; i++) s.w[i] = s.w[i + 1];
  s.w[15] = w;
  s.fill += 4;
  s.len += 4;
  if (s.fill == 64u) { sha_compress(s); s.fill = 0; }
}
__device__ __noinline__ void sha_update(Sha &s, const uint8_t *p, u32 n) {
  Sha t = s;
  u32 i = 0;
  while (i < n && (t.fill & 3u)) sha_byte_inl(t, p[i++]);                // up to the next word boundary of the message
  for (; i + 8 <= n; i += 8) {
    const u64 x = ld8(p + i);
    sha_word_inl(t, __builtin_bswap32((u32)x));
    sha_word_inl(t, __builtin_bswap32((u32)(x >> 32)));
  }
  if (i < n) {
    const u64 x = ld8(p + i);
    for (u32 j = 0; i < n; i++, j++) sha_byte_inl(t, (u32)(x >> (8 * j)) & 0xFFu);
  }
  s = t;
}
// digest of a COPY of the state (the caller's state can go on absorbing), as the big-endian number it spells, and
// whether that number is in [0, q)
__device__ __noinline__ void sha_final_number(const Sha &s0, sc &r, bool &lt_q) {
  Sha s = s0;
  const u32 bits_hi = s.len >> 29, bits_lo = s.len << 3;
  // padding: 0x80, zeros up to 56 mod 64, the bit length as 8 big-endian bytes -- one feeding site
  u32 tail = 0;                                     // 0: before the length field; 1..8: length bytes emitted
  for (u32 i = 0; tail < 8; i++) {
    u32 b;
    if (tail == 0 && (i == 0 || s.fill != 56u)) b = i == 0 ? 0x80u : 0u;
    else { b = tail < 4 ? (bits_hi >> (8 * (3 - tail))) & 0xFFu : (bits_lo >> (8 * (7 - tail))) & 0xFFu; tail++; }
    sha_byte_inl(s, b);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = s.h[7 - i];
  const u32 q[8] = BPMI_SC_Q;
  u32 t[8];
  lt_q = bpmi::words_sub(t, r.v, q) != 0;
}
// mod_hash(msg, q): the first i >= 1 with SHA-256(str(i) || msg) in [1, q) (src/utils/utils.py:84-97); `one` has
// absorbed "1" and msg already (the case i = 1); the retry (probability ~2^-128) re-hashes msg[0, n) from scratch
__device__ __noinline__ void mod_hash_q(sc &r, const Sha &one, const uint8_t *msg, u32 n) {
  bool lt;
  sha_final_number(one, r, lt);
  for (u32 i = 2; !lt || bpmi::sc_is_zero(r); i++) {
    Sha s;
    sha_init(s);
    u32 div = 1000000000u;
    while (div > i) div /= 10;
    for (u32 v = i; div; div /= 10) { sha_byte(s, '0' + v / div); v %= div; }
    sha_update(s, msg, n);
    sha_final_number(s, r, lt);
  }
}

// ---- small codecs ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 b64_char(u32 v) {
  return v < 26 ? 'A' + v : v < 52 ? 'a' + (v - 26) : v < 62 ? '0' + (v - 52) : v == 62 ? '+' : '/';
}
// does the item [p, p + n) spell base64(encoding of the point)?  33 zero bytes in the wire format = identity = b"\x00"
__device__ __forceinline__ u32 byte_of(const u64 w[], int i) { return (u32)(w[i >> 3] >> (8 * (i & 7))) & 0xFFu; }
__device__ __noinline__ bool point_item_equals(const uint8_t *comp, const uint8_t *p, u32 n) {
  u64 c[5], t[6];
#pragma unroll
  for (int i = 0; i < 5; i++) c[i] = ld8(comp + 8 * i);
#pragma unroll
  for (int i = 0; i < 6; i++) t[i] = ld8(p + 8 * i);
  if (!(c[0] | c[1] | c[2] | c[3] | (c[4] & 0xFFu))) return n == 4 && (u32)t[0] == 0x3D3D4141u;      // "AA=="
  if (n != 44) return false;
  bool same = true;
#pragma unroll
  for (int i = 0; i < 11; i++) {
    const u32 v = (byte_of(c, 3 * i) << 16) | (byte_of(c, 3 * i + 1) << 8) | byte_of(c, 3 * i + 2);
    same &= byte_of(t, 4 * i) == b64_char((v >> 18) & 63) && byte_of(t, 4 * i + 1) == b64_char((v >> 12) & 63) &&
            byte_of(t, 4 * i + 2) == b64_char((v >> 6) & 63) && byte_of(t, 4 * i + 3) == b64_char(v & 63);
  }
  return same;
}
// canonical decimal -> value mod q (false: not canonical decimal, or >= 2^256); eight digits per load and multiplication pass
__device__ __noinline__ bool parse_decimal(sc &r, const uint8_t *p, u32 n) {
  if (n == 0 || n > 78 || (n > 1 && p[0] == '0')) return false;
  u32 t[9];
#pragma unroll
  for (int k = 0; k < 9; k++) t[k] = 0;
  u32 i = 0, take = n & 7u;
  if (take == 0) take = 8;
  bool digits = true;
  while (i < n) {
    const u64 x = ld8(p + i);
    u32 val = 0, scale = 1;
    for (u32 j = 0; j < take; j++) {
      const u32 d = ((u32)(x >> (8 * j)) & 0xFFu) - '0';
      digits &= d <= 9u;
      val = val * 10u + d;
      scale *= 10u;
    }
    i += take;
    u64 c = val;
#pragma unroll
    for (int k = 0; k < 9; k++) { c += (u64)t[k] * scale; t[k] = (u32)c; c >>= 32; }
    take = 8;
  }
  if (!digits || t[8]) return false;
#pragma unroll
  for (int k = 0; k < 8; k++) r.v[k] = t[k];
  bpmi::sc_reduce_once(r);
  return true;
}
__device__ __forceinline__ bool sc_from_be(sc &r, const uint8_t *b) {          // 32 bytes big-endian; false when >= q
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const u64 x = __builtin_bswap64(ld8(b + 8 * j));
    r.v[7 - 2 * j] = (u32)(x >> 32);
    r.v[6 - 2 * j] = (u32)x;
  }
  const u32 qq[8] = BPMI_SC_Q;
  u32 t[8];
  return bpmi::words_sub(t, r.v, qq) != 0;
}
__device__ __forceinline__ bool sc_eq(const sc &a, const sc &b) {
  u32 d = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) d |= a.v[i] ^ b.v[i];
  return d == 0;
}
__device__ __forceinline__ sc sc_small(u32 x) {
  sc r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = i ? 0 : x;
  return r;
}
// by value: two 8-word arguments and the 8-word result travel in registers (a by-reference version goes through scratch)
__device__ __noinline__ sc mulq_v(const sc a, const sc b) { sc r; bpmi::sc_mul(r, a, b); return r; }
__device__ __forceinline__ void mulq(sc &r, const sc &a, const sc &b) { r = mulq_v(a, b); }
__device__ __forceinline__ void addq(sc &r, const sc &a, const sc &b) { bpmi::sc_add(r, a, b); }
__device__ __forceinline__ void negq(sc &r, const sc &a) { bpmi::sc_neg(r, a); }
__device__ __forceinline__ void subq(sc &r, const sc &a, const sc &b) { sc t; bpmi::sc_neg(t, b); bpmi::sc_add(r, a, t); }
// mod-q arithmetic of the algebra role: 9 x 29-bit limbs (scalar.hpp "sq").  The multiplication is the one out-of-line routine
// (~270 instructions, ~70 call sites); its operands travel in registers: one struct (9 words) plus nine scalars -- two
// structs would exceed the 16 registers the ABI gives to aggregate arguments and the second would go through scratch.
__device__ __noinline__ sq mq_v(const sq a, u32 b0, u32 b1, u32 b2, u32 b3, u32 b4, u32 b5, u32 b6, u32 b7, u32 b8) {
  const sq b = {{b0, b1, b2, b3, b4, b5, b6, b7, b8}};
  sq r;
  bpmi::sq_mul(r, a, b);
  return r;
}
__device__ __forceinline__ void mq(sq &r, const sq &a, const sq &b) { r = mq_v(a, b.v[0], b.v[1], b.v[2], b.v[3], b.v[4], b.v[5], b.v[6], b.v[7], b.v[8]); }
__device__ __forceinline__ sq to_sq(const sc &a) { sq r; bpmi::sq_from_sc(r, a); return r; }
__device__ __noinline__ sc to_sc(const sq a) { sc r; bpmi::sq_to_sc(r, a); return r; }
__device__ __forceinline__ void store_canon(u32 *p, const sq &a) { const sc c = to_sc(a); ::store_words8(p, c.v); }
__device__ __noinline__ sc invq_v(const sc a) { sc r; bpmi::sc_inv(r, a); return r; }        // binary extended Euclid (scalar.hpp)
__device__ __forceinline__ void invq(sc &r, const sc &a) { r = invq_v(a); }

// ---- items of a '&'-separated transcript, walked front to back ---------------------------------------------------
struct Walk {
  const uint8_t *p;
  u32 n;
  u32 pos;        // start of the current item
  bool have;      // there is a current item
};
__device__ __forceinline__ void walk_init(Walk &w, const uint8_t *p, u32 n) { w.p = p; w.n = n; w.pos = 0; w.have = true; }
__device__ __forceinline__ u32 walk_end(const Walk &w) {        // end of the current item: the next '&' or the end of the transcript
  u32 e = w.pos;
  while (e < w.n) {
    const u64 x = ld8(w.p + e) ^ 0x2626262626262626ull;                                      // '&' bytes become zero
    const u64 z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;               // lowest set bit marks the FIRST zero byte
    if (z) { e += (u32)__builtin_ctzll(z) >> 3; break; }
    e += 8;
  }
  return e < w.n ? e : w.n;
}
__device__ __forceinline__ void walk_next(Walk &w, u32 e) { w.have = e < w.n; w.pos = e + 1; }   // e = walk_end(w)

struct Params {
  const uint8_t *blobs;
  const u64 *off;            // P + 1 offsets into blobs
  const uint8_t *weights;    // P x 4 x 32 bytes, or null: derived from `seed`
  u32 seed[8];               // the 32 seed bytes as big-endian words
  u32 n, k, m, P, lanes;     // P: proofs of this launch
  int only_role;             // profiling: -1 both roles (product), 0 / 1 = run only that role (the other reports success)
  u32 Pall;                  // proofs of the whole call (the two status arrays are Pall bytes apart)
  u64 first;                 // index of proof 0 of this launch inside the whole batch (seed weights depend on it)
  u32 *contrib;              // (5 + 2n) x P cells of 9 limbs, limb-major (see cell_load)
  u32 *v_scalars, *pt_scalars;
  uint8_t *status;           // [g] verdict of the transcript role, [Pall + g] of the algebra role (1 = passed); points at this launch's proof 0
  unsigned long long *bad;   // atomicMin of the failing proof indices (whole-batch numbering)
};

// cell (col, g) = 9 limbs of a loose sq; limb w lives at contrib[(col * 9 + w) * P + g]: every access of a wave is one
// coalesced row of dwords
typedef __attribute__((address_space(1))) u32 gu32;              // known-global pointer: global_load / global_store, not flat
struct Cells { gu32 *base; size_t P; };                          // base = contrib + g
__device__ __forceinline__ Cells cells_of(const Params &q, u32 g) { Cells c; c.base = (gu32 *)q.contrib + g; c.P = q.P; return c; }
__device__ __forceinline__ void cell_load(sq &r, const Cells &c, u32 col) {
  const gu32 *p = c.base + (size_t)col * 9 * c.P;
#pragma unroll
  for (int w = 0; w < 9; w++) r.v[w] = p[w * c.P];
}
__device__ __forceinline__ void cell_store(const Cells &c, u32 col, const sq &a) {
  gu32 *p = c.base + (size_t)col * 9 * c.P;
#pragma unroll
  for (int w = 0; w < 9; w++) p[w * c.P] = a.v[w];
}

// SHA-256(seed || LE64(g) || t) with byte 31 cleared, read little-endian; 0 -> 1   (rp::derive_weight).  The 41-byte message
// is one padded block, built in place: 8 seed words, g, t and the 0x80 marker, zeros, the bit length 328.
__device__ __noinline__ void derive_weight(sc &w, const u32 seed[8], u64 g, u32 t) {
  Sha s;
  sha_init(s);
#pragma unroll
  for (int i = 0; i < 8; i++) s.w[i] = seed[i];
  s.w[8] = __builtin_bswap32((u32)g);
  s.w[9] = __builtin_bswap32((u32)(g >> 32));
  s.w[10] = (t << 24) | 0x00800000u;
#pragma unroll
  for (int i = 11; i < 15; i++) s.w[i] = 0;
  s.w[15] = 41u * 8u;
  sha_compress(s);
#pragma unroll
  for (int i = 0; i < 8; i++) w.v[i] = __builtin_bswap32(s.h[i]);       // digest bytes read as a little-endian number
  w.v[7] &= 0x00FFFFFFu;
  if (bpmi::sc_is_zero(w)) w = sc_small(1);
}

// ---- one proof, structurally (rp::parse_blob): header, scalars < q, the three transcripts exactly fill the rest
struct Parsed {
  sc taux, mu, t_hat, a, b, xs[16];
  const uint8_t *comp;
  const uint8_t *ts[3];
  u32 tl[3], start;
};
__device__ __forceinline__ bool parse_proof(Parsed &P, const uint8_t *blob, u32 blen, u32 k) {
  const u32 fixed = 6 + 32 * (5 + k) + 33 * (6 + 2 * k) + 2;
  if (!(blen >= fixed && blob[0] == 'B' && blob[1] == 'P' && blob[2] == 'R' && blob[3] == 'P' && blob[4] == '1' && blob[5] == k)) return false;
  bool ok = true;
  ok &= sc_from_be(P.taux, blob + 6);
  ok &= sc_from_be(P.mu, blob + 38);
  ok &= sc_from_be(P.t_hat, blob + 70);
  ok &= sc_from_be(P.a, blob + 102);
  ok &= sc_from_be(P.b, blob + 134);
  for (u32 j = 0; j < k; j++) ok &= sc_from_be(P.xs[j], blob + 166 + 32 * j);
  P.comp = blob + 6 + 32 * (5 + k);
  u32 o = fixed - 2;
  P.start = ((u32)blob[o] << 8) | blob[o + 1];
  o += 2;
  for (int t = 0; t < 3; t++) {
    if (!ok || blen < o + 4) return false;
    P.tl[t] = ((u32)blob[o] << 24) | ((u32)blob[o + 1] << 16) | ((u32)blob[o + 2] << 8) | blob[o + 3];
    o += 4;
    if (P.tl[t] > blen || blen - P.tl[t] < o) return false;
    P.ts[t] = blob + o;
    o += P.tl[t];
  }
  return ok && o == blen;
}

// ---- role 0: the byte-level transcript checks (rp::check_transcripts) ------------------------------------------------------
__device__ __noinline__ bool check_transcripts(const Parsed &P, u32 k) {
  const uint8_t *comp = P.comp;
  bool ok = true;
  {
    const uint8_t *T1 = comp, *T2 = comp + 33, *A = comp + 66, *S = comp + 99;
    Walk w;
    walk_init(w, P.ts[0], P.tl[0]);
    u32 e = walk_end(w);                                             // item 0: not checked
    walk_next(w, e);
    for (u32 j = 1; j < 8 && ok; j++) {
      if (!w.have) return false;
      e = walk_end(w);
      const uint8_t *ip = w.p + w.pos;
      const u32 il = e - w.pos;
      sc num;
      if (j == 1) ok = point_item_equals(A, ip, il);
      else if (j == 2) ok = point_item_equals(S, ip, il);
      else if (j == 5) ok = point_item_equals(T1, ip, il);
      else if (j == 6) ok = point_item_equals(T2, ip, il);
      else ok = parse_decimal(num, ip, il);                           // y, z, x: read by the algebra role
      walk_next(w, e);
    }
    if (!ok) return false;
  }
  {                                                                  // Protocol 1: item 1 = str(mod_hash(item 0 + "&"))
    Walk w;
    walk_init(w, P.ts[1], P.tl[1]);
    u32 e = walk_end(w);
    walk_next(w, e);
    if (!w.have) return false;
    Sha s;
    sha_init(s);
    sha_byte(s, '1');
    sha_update(s, P.ts[1], w.pos);
    sc h, x_ip;
    mod_hash_q(h, s, P.ts[1], w.pos);
    e = walk_end(w);
    if (!(parse_decimal(x_ip, w.p + w.pos, e - w.pos) && sc_eq(x_ip, h))) return false;
  }
  {                                                                  // Protocol 2: L_i, R_i, x_i per round
    Walk w;
    walk_init(w, P.ts[2], P.tl[2]);
    for (u32 j = 0; j < P.start && w.have; j++) walk_next(w, walk_end(w));
    const uint8_t *Ls = comp + 33 * 6, *Rs = Ls + 33 * k;
    Sha run;
    sha_init(run);
    sha_byte(run, '1');
    u32 hashed = 0;
    for (u32 i = 0; i < k; i++) {
      if (!w.have) return false;
      u32 e = walk_end(w);
      ok = point_item_equals(Ls + 33 * i, w.p + w.pos, e - w.pos);
      walk_next(w, e);
      if (!ok || !w.have) return false;
      e = walk_end(w);
      ok = point_item_equals(Rs + 33 * i, w.p + w.pos, e - w.pos);
      walk_next(w, e);
      if (!ok || !w.have) return false;
      const u32 upto = w.pos;                                        // prefix incl. the '&' after R_i
      sha_update(run, P.ts[2] + hashed, upto - hashed);
      hashed = upto;
      sc h, xi;
      mod_hash_q(h, run, P.ts[2], upto);
      e = walk_end(w);
      if (!(parse_decimal(xi, w.p + w.pos, e - w.pos) && sc_eq(xi, h) && sc_eq(xi, P.xs[i]))) return false;
      walk_next(w, e);
    }
  }
  return true;
}

// ---- role 1: the weighted scalars (rp::accumulate); the challenges are READ here, role 0 checks where they come from ---------
__device__ __forceinline__ bool read_challenges(const Parsed &P, sc &cx, sc &cy, sc &cz, sc &x_ip) {
  Walk w;
  walk_init(w, P.ts[0], P.tl[0]);
  for (u32 j = 0; j < 8; j++) {
    if (!w.have) return false;
    const u32 e = walk_end(w);
    bool ok = true;
    if (j == 3) ok = parse_decimal(cy, w.p + w.pos, e - w.pos);
    else if (j == 4) ok = parse_decimal(cz, w.p + w.pos, e - w.pos);
    else if (j == 7) ok = parse_decimal(cx, w.p + w.pos, e - w.pos);
    if (!ok) return false;
    walk_next(w, e);
  }
  walk_init(w, P.ts[1], P.tl[1]);
  walk_next(w, walk_end(w));
  if (!w.have) return false;
  return parse_decimal(x_ip, w.p + w.pos, walk_end(w) - w.pos);
}
__device__ __forceinline__ void zero_outputs(const Params &q, u32 g) {
  const sc z = sc_small(0);
  const sq zq = bpmi::sq_small(0);
  for (u32 c = 0; c < 5 + 2 * q.n; c++) cell_store(cells_of(q, g), c, zq);
  for (u32 j = 0; j < q.m; j++) ::store_words8(q.v_scalars + ((size_t)g * q.m + j) * 8, z.v);
  for (u32 j = 0; j < 6 + 2 * q.k; j++) ::store_words8(q.pt_scalars + ((size_t)g * (6 + 2 * q.k) + j) * 8, z.v);
}
__device__ __noinline__ bool weighted_scalars(const Params &q, u32 g, const Parsed &P) {
  const u32 n = q.n, k = q.k, m = q.m;
  const Cells C = cells_of(q, g);
  sc cxs, cys, czs, xips;
  if (!read_challenges(P, cxs, cys, czs, xips)) return false;
  const sq cx = to_sq(cxs), cy = to_sq(cys), cz = to_sq(czs), x_ip = to_sq(xips);
  // inverses of x_1 .. x_k and y: one inversion per proof (Montgomery's trick inside the lane)
  sq xs[16], xinv[16], yinv;
  {
    sq pre[17], run = bpmi::sq_small(1);
    bool nonzero = !bpmi::sc_is_zero(cys);                 // a zero challenge cannot come out of mod_hash
    for (u32 t = 0; t < k; t++) { nonzero &= !bpmi::sc_is_zero(P.xs[t]); xs[t] = to_sq(P.xs[t]); }
    if (!nonzero) return false;
    for (u32 t = 0; t <= k; t++) {
      pre[t] = run;
      mq(run, run, t < k ? xs[t] : cy);
    }
    sc rinv_c;
    invq(rinv_c, to_sc(run));
    sq rinv = to_sq(rinv_c);
    for (int t = (int)k; t >= 0; t--) {
      sq iv;
      mq(iv, rinv, pre[t]);
      if (t == (int)k) { yinv = iv; mq(rinv, rinv, cy); }
      else { xinv[t] = iv; mq(rinv, rinv, xs[t]); }
    }
  }
  sq w[4];
  for (u32 t = 0; t < 4; t++) {
    sc ws;
    if (q.weights) {
      const uint8_t *src = q.weights + ((size_t)(q.first + g) * 4 + t) * 32;
#pragma unroll
      for (int i = 0; i < 4; i++) { const u64 x = ld8(src + 8 * i); ws.v[2 * i] = (u32)x; ws.v[2 * i + 1] = (u32)(x >> 32); }
      bpmi::sc_reduce_once(ws);
    } else {
      derive_weight(ws, q.seed, q.first + g, t);
    }
    w[t] = to_sq(ws);
  }
  const sq pa = to_sq(P.a), pb = to_sq(P.b), t_hat = to_sq(P.t_hat);
  sq t, u;
  const u32 SG = 5, SH = 5 + n;
  // s-vector by doubling with the weights folded in: sg_i = w4 a s_i, sh_i = w4 b s_i^-1 y^-i, in place in the columns
  mq(t, w[3], pa); cell_store(C, SG, t);
  mq(t, w[3], pb); cell_store(C, SH, t);
  {
    sq ypow2 = yinv;
    u32 len = 1;
    for (int j = (int)k - 1; j >= 0; j--) {
      const sq xv = xs[j], xi = xinv[j];
      sq hi_h;
      mq(hi_h, xi, ypow2);
      for (u32 i = 0; i < len; i++) {
        sq s;
        cell_load(s, C, SG + i);
        mq(t, s, xv); cell_store(C, SG + len + i, t);
        mq(t, s, xi); cell_store(C, SG + i, t);
        cell_load(s, C, SH + i);
        mq(t, s, hi_h); cell_store(C, SH + len + i, t);
        mq(t, s, xv); cell_store(C, SH + i, t);
      }
      mq(ypow2, ypow2, ypow2);
      len <<= 1;
    }
  }
  sq z2, w2z, geo, r2;
  mq(z2, cz, cz);
  mq(w2z, w[1], cz);
  cell_store(C, 3, w2z);                                           // gs_const
  bpmi::sq_neg(t, w2z); cell_store(C, 4, t);                       // hs_const
  bpmi::sq_add(r2, yinv, yinv);                                       // 2 / y
  const u32 bits = n / m;
  sq yn_inv = bpmi::sq_small(1);                                      // y^-bits, square and multiply
  for (int i = 31 - __clz(bits); i >= 0; i--) {
    mq(yn_inv, yn_inv, yn_inv);
    if ((bits >> i) & 1u) mq(yn_inv, yn_inv, yinv);
  }
  {
    sq blk = bpmi::sq_small(1), zp = z2;                              // zp = z^(2 + j)
    for (u32 j = 0, i = 0; j < m; j++) {
      mq(geo, w[1], zp);
      mq(geo, geo, blk);                                              // w2 z^(2+j) 2^(i % bits) y^-i at i = bits j
      mq(t, w[0], zp); bpmi::sq_neg(t, t);
      store_canon(q.v_scalars + ((size_t)g * m + j) * 8, t);          // V_j: -w1 z^(2+j)
      for (u32 e = 0; e < bits; e++, i++) {
        sq s;
        cell_load(s, C, SH + i);
        bpmi::sq_sub(s, s, geo);
        cell_store(C, SH + i, s);
        mq(geo, geo, r2);
      }
      mq(blk, blk, yn_inv);
      mq(zp, zp, cz);
    }
  }
  // sum_{i<n} y^i by doubling; delta = (z - z^2) ysum - (2^bits - 1) sum_{j=1..m} z^(j+2)
  sq ysum = bpmi::sq_small(1), ypw = cy;
  const sq one = bpmi::sq_small(1);
  for (u32 l2 = 1; l2 < n; l2 <<= 1) {
    bpmi::sq_add(t, one, ypw);
    mq(ysum, ysum, t);
    mq(ypw, ypw, ypw);
  }
  sq two_n = bpmi::sq_small(1);
  for (u32 i = 0; i < bits; i++) bpmi::sq_add(two_n, two_n, two_n);   // 2^bits mod q
  bpmi::sq_sub(two_n, two_n, one);
  sq delta, zsum = bpmi::sq_small(0), zp;
  bpmi::sq_sub(t, cz, z2);
  mq(delta, t, ysum);
  mq(zp, z2, cz);                                                     // z^3
  for (u32 j = 1; j <= m; j++) { bpmi::sq_add(zsum, zsum, zp); mq(zp, zp, cz); }
  mq(t, zsum, two_n);
  bpmi::sq_sub(delta, delta, t);
  // c_g: w1 (t_hat - delta); c_h: w1 taux + w2 mu; c_u: -(w2 x_ip t_hat + w3 x_ip)
  bpmi::sq_sub(t, t_hat, delta); mq(t, t, w[0]); cell_store(C, 0, t);
  mq(t, w[0], to_sq(P.taux)); mq(u, w[1], to_sq(P.mu)); bpmi::sq_add(t, t, u); cell_store(C, 1, t);
  mq(t, w[1], x_ip); mq(t, t, t_hat); mq(u, w[2], x_ip); bpmi::sq_add(t, t, u); bpmi::sq_neg(t, t); cell_store(C, 2, t);
  // per-proof points in wire order: T1: -w1 x | T2: -w1 x^2 | A: -w2 | S: -w2 x | u_new: w3 + w4 a b | P_new: w2 - w4 | Ls | Rs
  u32 *op = q.pt_scalars + (size_t)g * (6 + 2 * k) * 8;
  mq(t, w[0], cx); bpmi::sq_neg(u, t); store_canon(op, u);
  mq(t, t, cx); bpmi::sq_neg(u, t); store_canon(op + 8, u);
  bpmi::sq_neg(u, w[1]); store_canon(op + 16, u);
  mq(t, w[1], cx); bpmi::sq_neg(u, t); store_canon(op + 24, u);
  mq(t, w[3], pa); mq(t, t, pb); bpmi::sq_add(u, w[2], t); store_canon(op + 32, u);
  bpmi::sq_sub(u, w[1], w[3]); store_canon(op + 40, u);
  for (u32 j = 0; j < k; j++) {
    mq(t, w[3], xs[j]); mq(t, t, xs[j]); bpmi::sq_neg(u, t); store_canon(op + (6 + j) * 8, u);
    mq(t, w[3], xinv[j]); mq(t, t, xinv[j]); bpmi::sq_neg(u, t); store_canon(op + (6 + k + j) * 8, u);
  }
  return true;
}

// Two waves per group of `lanes` proofs: even blocks run role 0 (hashing and byte checks), odd blocks role 1 (algebra).  The
// roles share nothing but the input -- the challenges a proof CLAIMS are all the algebra needs, and role 0 verifies the claims
// -- so they run side by side and the critical path of a proof is the longer role, not their sum.  status[g] / status[Pall + g]
// = verdict of role 0 / role 1.  A proof whose algebra role fails gets all-zero outputs.
__global__ void __launch_bounds__(64) k_rp_prepare(Params q) {
  if (threadIdx.x >= q.lanes) return;
  const u32 role = blockIdx.x & 1u;
  const u32 g = (blockIdx.x >> 1) * q.lanes + threadIdx.x;
  if (g >= q.P) return;
  const uint8_t *blob = q.blobs + q.off[g];
  const u32 blen = (u32)(q.off[g + 1] - q.off[g]);
  if (q.only_role >= 0 && (u32)q.only_role != role) { q.status[(size_t)role * q.Pall + g] = 1; return; }
  Parsed P;
  bool ok = parse_proof(P, blob, blen, q.k);
  if (role == 0) {
    ok = ok && check_transcripts(P, q.k);
  } else {
    ok = ok && weighted_scalars(q, g, P);
    if (!ok) zero_outputs(q, g);
  }
  q.status[(size_t)role * q.Pall + g] = ok ? 1 : 0;
  if (!ok) atomicMin(q.bad, (unsigned long long)(q.first + g));
}

// shared[col] += sum over the P proofs of cell (col, .): the nine limb rows are summed as plain 64-bit integers (P < 2^22 loose
// limbs cannot overflow) and reduced mod q once; one block per column
__global__ void __launch_bounds__(256) k_rp_colsum(const u32 *__restrict__ contrib, u32 P, u32 *__restrict__ shared) {
  __shared__ u64 sh[256 * 9];
  const u32 col = blockIdx.x;
  u64 acc[9];
#pragma unroll
  for (int w = 0; w < 9; w++) acc[w] = 0;
  const u32 *base = contrib + (size_t)col * 9 * P;
  for (u32 i = threadIdx.x; i < P; i += 256u) {
#pragma unroll
    for (int w = 0; w < 9; w++) acc[w] += base[(size_t)w * P + i];
  }
  for (u32 d = 128; d > 0; d >>= 1) {
#pragma unroll
    for (int w = 0; w < 9; w++) sh[threadIdx.x * 9 + w] = acc[w];
    __syncthreads();
    if (threadIdx.x < d) {
#pragma unroll
      for (int w = 0; w < 9; w++) acc[w] += sh[(threadIdx.x + d) * 9 + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sq total;
    bpmi::sq_norm_cols(total, acc);              // limb 5 may end above "loose" here (the carry is < 2^22): sq_to_sc takes any 32-bit limbs
    sc sum, cur;
    bpmi::sq_to_sc(sum, total);
    ::load_words8(cur.v, shared + 8ull * col);
    addq(cur, cur, sum);
    ::store_words8(shared + 8ull * col, cur.v);
  }
}

}  // namespace rpd
