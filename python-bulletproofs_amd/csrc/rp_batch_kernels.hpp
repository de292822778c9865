// rp_batch_kernels.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).  DEVICE code.
// The per-proof preparation of the batch range-proof verifier on the GPU: the wire blob of every proof is parsed, its Fiat-Shamir
// transcripts are re-hashed (SHA-256), the byte-level checks of the reference's verifiers run (rangeproof_verifier.py:42-53,
// inner_product_verifier.py:31-43 and :104-125) and the weighted scalars of the batch's one MSM are produced.  It is the device
// twin of rp_batch_host.hpp (same checks, same numbers: tests/test_gpu_batch_dev.py compares the two byte for byte on the same
// weights); see that file for what each check means.
//
// Shape of the work (round 3).  A proof is a handful of SERIAL chains (a SHA-256 chain over the Protocol-2 transcript, the text
// checks, one modular inversion with the tables that hang off it, the inverse-free scalars) plus 2n independent elements of the
// s-vector.  A wave64 instruction costs its issue cycles whatever its active-lane count, so every chain runs with one lane per
// proof in FULL waves (64 proofs), one wave per chain kind ("role"), and nothing is computed twice:
//   k_rp_roles     role 0  hash chain of the Protocol-2 transcript (every x_i re-hashed from its prefix)
//                  role 1  range-proof / Protocol-1 transcripts and the text half of Protocol 2 (L_i / R_i items vs the points)
//                  role 2  the inversion of (x_1 .. x_k, y) and what needs it: the per-bit factor tables of the s-vector and of the
//                          y^-i / 2^i progressions (the "context" of a proof, limb-major in global memory), the R_j scalars
//                  role 3  everything that needs no inverse: the four weights, delta(y, z), the five scalar columns, the scalars of
//                          V_j, T1, T2, A, S, u_new, P_new, L_j
//   k_rp_elements  the 2n shared-generator contributions of every proof, EIGHT consecutive elements per lane: a lane multiplies
//                  the proof's base value by the factors of the set high index bits (wave-uniform: a block is 64 proofs at ONE
//                  element range), then walks its 8 elements with 7 multiplications (a binary tree over the three low bits, all
//                  in registers) -- 2 * n/8 waves per 64 proofs instead of two, no LDS, no private arrays.
// The roles share nothing but the input: the algebra needs only the challenges a proof CLAIMS, roles 0 / 1 verify the claims.
// Nothing here lives in private memory: states and operands travel BY VALUE in registers (an object passed by reference to an
// out-of-line routine lives in scratch, and a lone wave per SIMD has nothing to hide a scratch round trip behind); the one
// indexed per-lane array (the prefix products of the batched inversion) lives in LDS.
// Input: k_rp_transpose first re-lays the batch as 8-byte words, word-major over the proofs (see BPtr), so that lanes walking
// their own proofs read neighbouring words.
// Output: the (5 + 2n) shared-generator contributions of proof g go to cells (col, g) of nine 29-bit limbs, limb-major
// (contrib[(col * 9 + limb) * P + g]: a wave stores coalesced dword rows), summed over g by k_rp_colsum; per-proof scalars go
// straight to the MSM's scalar array.
// Measurements: profiles/r03_C5_prepare_kernel.txt; round 2's four-role kernel: profiles/r02_C5_prepare_kernel*.txt.
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
// the ~1750-instruction round function in the whole kernel instead of one per feeding site (the transcript roles' code is
// 12 K instructions this way, 45 K with the rounds inlined).
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
// Where the proof bytes are read from.  Every lane walks its own proof, so against the wire buffer a wave's load is 64
// scattered cache lines.  k_rp_transpose therefore re-lays the batch as 8-byte words, word-major over the proofs:
// T[w * stride + g] = bytes [8w, 8w + 8) of proof g, zero beyond the proof's end.  Lanes that read the same word index of their
// proofs -- the normal case, the proofs have the same layout -- then touch neighbouring words: a load is a few coalesced
// 512-byte rows.  BPtr is a byte position inside one proof of that array and stands in for `const uint8_t *` (p + n, p[i],
// ld8(p)); loads may run up to 80 bytes past the proof's end: the array carries 16 zero rows of padding.
struct BPtr {
  const u64 *base;      // T + g
  u32 stride;           // words between consecutive 8-byte words of one proof (= proofs in the array)
  u32 off;              // byte offset inside the proof
  __device__ __forceinline__ BPtr operator+(u32 n) const { BPtr r = *this; r.off += n; return r; }
  __device__ __forceinline__ u32 operator[](u32 i) const {
    const u32 o = off + i;
    return (u32)(base[(size_t)(o >> 3) * stride] >> (8 * (o & 7u))) & 0xFFu;
  }
};
__device__ __forceinline__ u64 ld8(const BPtr p) {                       // the 8 bytes at p, any alignment
  const u64 *q = p.base + (size_t)(p.off >> 3) * p.stride;
  const u32 s = 8 * (p.off & 7u);
  const u64 a = q[0], b = q[p.stride];
  return (a >> s) | ((b << (63 - s)) << 1);
}
__device__ __forceinline__ u64 ld8_raw(const uint8_t *p) { u64 w; __builtin_memcpy(&w, p, 8); return w; }      // plain memory (the weights)

// Feeding WITHOUT compressing at every step: the lanes of a wave hash prefixes of slightly different lengths, so their 64-byte
// blocks fill at different steps; a compression call at every feeding step would run once per distinct phase in the wave.
// sha_feed and sha_digest_number therefore fill a lane's block completely (however many words that takes for the lane) and
// compress at ONE point per block, where every lane of the wave that has a full block takes part.
__device__ __forceinline__ void sha_push_byte(Sha &s, u32 b) {           // the caller guarantees fill < 64
  s.cur = (s.cur << 8) | b;
  s.fill++;
  s.len++;
  if ((s.fill & 3u) == 0) {
#pragma unroll
    for (int i = 0; i < 15; i++) s.w[i] = s.w[i + 1];
    s.w[15] = s.cur;
  }
}
__device__ __forceinline__ void sha_push_word(Sha &s, u32 w) {           // fill is a multiple of 4 and < 64
#pragma unroll
  for (int i = 0; i < 15; i++) s.w[i] = s.w[i + 1];
  s.w[15] = w;
  s.fill += 4;
  s.len += 4;
}
// absorb p[0, n): inlined at its (few) call sites, so the state never leaves the registers
__device__ __forceinline__ void sha_feed(Sha &t, const BPtr p, u32 n) {
  u32 i = 0;
  while (i < n) {
    while (i < n && (t.fill & 3u)) sha_push_byte(t, p[i++]);              // up to the next word boundary of the message (<= 3 bytes)
    if ((t.fill & 3u) == 0 && t.fill < 64u) {
      const u32 nw = min((64u - t.fill) >> 2, (n - i) >> 2);             // words that fit this block and exist
      u64 x[8];                                                          // loads in flight together
#pragma unroll
      for (int j = 0; j < 8; j++) x[j] = ld8(p + i + 8 * j);
#pragma unroll
      for (int j = 0; j < 16; j++)
        if ((u32)j < nw) sha_push_word(t, __builtin_bswap32((u32)(x[j >> 1] >> (32 * (j & 1)))));
      i += 4 * nw;
      if (t.fill < 64u && n - i < 4u)                                    // the input ends inside this block: its last (< 4) bytes
        while (i < n) sha_push_byte(t, p[i++]);
    }
    if (t.fill == 64u) { sha_compress(t); t.fill = 0; }
  }
}
// digest of a COPY of the state (the caller's state can go on absorbing), as the big-endian number it spells, and
// whether that number is in [0, q)
__device__ __forceinline__ void sha_digest_number(const Sha &s0, sc &r, bool &lt_q) {
  Sha s = s0;
  const u32 bits_hi = s.len >> 29, bits_lo = s.len << 3;
  // padding: 0x80, zeros up to 56 mod 64, the bit length as 8 big-endian bytes; one or two blocks, compressed at one point
  u32 emitted = 0;                                  // 0: nothing yet; 1: the 0x80 is out; 2..9: length bytes out (9 = done)
  for (int blk = 0; blk < 2; blk++) {
    while (emitted < 9u && s.fill < 64u) {
      u32 b;
      if (emitted == 0) { b = 0x80u; emitted = 1; }
      else if (emitted == 1 && s.fill != 56u) b = 0u;
      else { const u32 k = emitted - 1; b = k < 4 ? (bits_hi >> (8 * (3 - k))) & 0xFFu : (bits_lo >> (8 * (7 - k))) & 0xFFu; emitted++; }
      sha_push_byte(s, b);
    }
    if (s.fill == 64u) { sha_compress(s); s.fill = 0; }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = s.h[7 - i];
  const u32 q[8] = BPMI_SC_Q;
  u32 t[8];
  lt_q = bpmi::words_sub(t, r.v, q) != 0;
}
// mod_hash(msg, q): the first i >= 1 with SHA-256(str(i) || msg) in [1, q) (src/utils/utils.py:84-97); `one` has
// absorbed "1" and msg already (the case i = 1); the retry (probability ~2^-128) re-hashes msg[0, n) from scratch
__device__ __forceinline__ void mod_hash_q(sc &r, const Sha &one, const BPtr msg, u32 n) {
  bool lt;
  sha_digest_number(one, r, lt);
  for (u32 i = 2; !lt || bpmi::sc_is_zero(r); i++) {
    Sha s;
    sha_init(s);
    u32 div = 1000000000u;
    while (div > i) div /= 10;
    for (u32 v = i; div; div /= 10) { sha_push_byte(s, '0' + v / div); v %= div; }       // <= 10 bytes: the block cannot fill
    sha_feed(s, msg, n);
    sha_digest_number(s, r, lt);
  }
}

// ---- small codecs ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 b64_char(u32 v) {
  return v < 26 ? 'A' + v : v < 52 ? 'a' + (v - 26) : v < 62 ? '0' + (v - 52) : v == 62 ? '+' : '/';
}
// does the item [p, p + n) spell base64(encoding of the point)?  33 zero bytes in the wire format = identity = b"\x00"
__device__ __forceinline__ u32 byte_of(const u64 w[], int i) { return (u32)(w[i >> 3] >> (8 * (i & 7))) & 0xFFu; }
__device__ __noinline__ bool point_item_equals(const BPtr comp, const BPtr p, u32 n) {
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
// canonical decimal -> value mod q (ok = false: not canonical decimal, or >= 2^256).  All (up to 80) bytes are loaded at once;
// eight digits become a number with three multiplications (first character in the lowest byte):
//   pairs (x * 10 + (x >> 8)) & 0x00FF.., fours (* 100, >> 16), eight (* 10000, >> 32)
__device__ __forceinline__ bool eight_digits(u64 x, u32 count, u32 &val) {          // the first `count` (1..8) bytes of x
  x -= 0x3030303030303030ull;
  if (count < 8) {                                   // right-align: the missing leading digits are zeros
    const u32 sh = 8 * (8 - count);
    x <<= sh;
  }
  const bool ok = (((x + 0x7676767676767676ull) | x) & 0x8080808080808080ull) == 0;
  x = (x * 10 + (x >> 8)) & 0x00FF00FF00FF00FFull;
  x = (x * 100 + (x >> 16)) & 0x0000FFFF0000FFFFull;
  x = (x * 10000 + (x >> 32)) & 0x00000000FFFFFFFFull;
  val = (u32)x;
  return ok;
}
struct Dec { sc v; u32 ok; };            // returned BY VALUE: nine registers, no scratch
__device__ __noinline__ Dec parse_decimal(const BPtr p, u32 n) {
  Dec out;
#pragma unroll
  for (int k = 0; k < 8; k++) out.v.v[k] = 0;
  out.ok = 0;
  if (n == 0 || n > 78 || (n > 1 && p[0] == '0')) return out;
  u64 x[10];
#pragma unroll
  for (int c = 0; c < 10; c++) x[c] = ld8(p + 8 * c);                  // 80 bytes; only the first n are looked at
  const u32 full = n >> 3, rem = n & 7u;
  u32 t[9];
#pragma unroll
  for (int k = 0; k < 9; k++) t[k] = 0;
  bool digits = true;
#pragma unroll
  for (int c = 0; c < 10; c++) {
    const u32 count = (u32)c < full ? 8u : ((u32)c == full ? rem : 0u);
    if (count) {
      u32 val;
      digits &= eight_digits(x[c], count, val);
      u32 scale = 100000000u;
      if (count < 8) { scale = 1; for (u32 j = 0; j < count; j++) scale *= 10u; }
      u64 cy = val;
#pragma unroll
      for (int k = 0; k < 9; k++) { cy += (u64)t[k] * scale; t[k] = (u32)cy; cy >>= 32; }
    }
  }
  if (!digits || t[8]) return out;
#pragma unroll
  for (int k = 0; k < 8; k++) out.v.v[k] = t[k];
  bpmi::sc_reduce_once(out.v);
  out.ok = 1;
  return out;
}
__device__ __forceinline__ bool sc_from_be(sc &r, const BPtr b) {          // 32 bytes big-endian; false when >= q
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
// mod-q arithmetic of the algebra roles: 9 x 29-bit limbs (scalar.hpp "sq").  The multiplication is ONE out-of-line routine
// (~270 instructions, dozens of call sites); its operands travel in registers: one struct (9 words) plus nine scalars -- two
// structs would exceed the 16 registers the ABI gives to aggregate arguments and the second would go through scratch.
__device__ __noinline__ sq mq_v(const sq a, u32 b0, u32 b1, u32 b2, u32 b3, u32 b4, u32 b5, u32 b6, u32 b7, u32 b8) {
  const sq b = {{b0, b1, b2, b3, b4, b5, b6, b7, b8}};
  sq r;
  bpmi::sq_mul(r, a, b);
  return r;
}
__device__ __forceinline__ void mq_inl(sq &r, const sq &a, const sq &b) { sq t; bpmi::sq_mul(t, a, b); r = t; }
__device__ __forceinline__ void mq(sq &r, const sq &a, const sq &b) { r = mq_v(a, b.v[0], b.v[1], b.v[2], b.v[3], b.v[4], b.v[5], b.v[6], b.v[7], b.v[8]); }
__device__ __forceinline__ sq to_sq(const sc &a) { sq r; bpmi::sq_from_sc(r, a); return r; }
__device__ __noinline__ sc to_sc(const sq a) { sc r; bpmi::sq_to_sc(r, a); return r; }
__device__ __forceinline__ void store_canon(u32 *p, const sq &a) { const sc c = to_sc(a); ::store_words8(p, c.v); }
__device__ __noinline__ sc invq_v(const sc a) { sc r; bpmi::sc_inv(r, a); return r; }        // binary extended Euclid (scalar.hpp)

// ---- items of a '&'-separated transcript, walked front to back ---------------------------------------------------
struct Walk {
  BPtr p;
  u32 n;
  u32 pos;        // start of the current item
  bool have;      // there is a current item
};
__device__ __forceinline__ void walk_init(Walk &w, const BPtr p, u32 n) { w.p = p; w.n = n; w.pos = 0; w.have = true; }
__device__ __forceinline__ u32 first_amp(u64 x) {               // index of the first '&' among the 8 bytes of x, 8 if none
  x ^= 0x2626262626262626ull;                                                              // '&' bytes become zero
  const u64 z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;               // lowest set bit marks the FIRST zero byte
  return z ? (u32)__builtin_ctzll(z) >> 3 : 8u;
}
__device__ __forceinline__ u32 walk_end(const Walk &w) {        // end of the current item: the next '&' or the end of the transcript
  u32 e = w.pos;
  while (e < w.n) {
    u64 x[4];                                                   // four loads in flight per wait
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = ld8(w.p + e + 8 * j);
    u32 k = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { if (k == 8u * j) k += first_amp(x[j]); }
    e += k;
    if (k < 32) break;
  }
  return e < w.n ? e : w.n;
}
__device__ __forceinline__ void walk_next(Walk &w, u32 e) { w.have = e < w.n; w.pos = e + 1; }   // e = walk_end(w)

struct Params {
  const u64 *T;              // the proofs of this launch as 8-byte words, word-major (k_rp_transpose); T points at proof 0 of the launch
  u32 Tstride;               // proofs in the array (= Pall)
  const u64 *off;            // P + 1 offsets into the wire buffer (only the lengths are used here)
  const u32 *lens;           // format 2: the lengths of the EXPANDED proofs (k_rp_expand_v2), or null: the lengths are the offsets' differences
  const uint8_t *weights;    // P x 4 x 32 bytes, or null: derived from `seed`
  u32 seed[8];               // the 32 seed bytes as big-endian words
  u32 n, k, m, P, lanes;     // P: proofs of this launch; lanes: proofs per wave of k_rp_roles
  int only_role;             // profiling: -1 all roles (product), 0..3 = run only that role (the others report success)
  u32 Pall;                  // proofs of the whole call (the status arrays are Pall bytes apart)
  u64 first;                 // index of proof 0 of this launch inside the whole batch (seed weights depend on it)
  u32 *contrib;              // (5 + 2n) x P cells of 9 limbs, limb-major (see cell_load)
  u32 *ctx;                  // (2 + 3k + m) x P context slots of 9 limbs, limb-major (role 2 -> k_rp_elements)
  u32 *v_scalars, *pt_scalars;
  uint8_t *status;           // [role * Pall + g] = verdict of a role (1 = passed); points at this launch's proof 0
  unsigned long long *bad;   // atomicMin of the failing proof indices (whole-batch numbering)
  u32 prio;                  // the preparation's kernels raise their waves' issue priority (option "rp_priority"; raise_priority, msm_kernels.hpp)
};
// context slots of one proof (9 loose limbs each)
#define CTX_BASE_G 0u                          // w4 a prod x_d^-1
#define CTX_BASE_H 1u                          // w4 b prod x_d
#define CTX_FG(k_, b_) (2u + (b_))             // x_(k-1-b)^2: factor of index bit b on the gs side
#define CTX_FH(k_, b_) (2u + (k_) + (b_))      // x_(k-1-b)^-2 y^-(2^b)
#define CTX_GEO(k_, b_) (2u + 2u * (k_) + (b_))      // (2/y)^(2^b) inside a value block, y^-(2^b) for the block-index bits
#define CTX_W2Z(k_, j_) (2u + 3u * (k_) + (j_))      // w2 z^(2+j)
#define CTX_SLOTS(k_, m_) (2u + 3u * (k_) + (m_))

// cell (col, g) = 9 limbs of a loose sq; limb w lives at base[(col * 9 + w) * P + g]: every access of a wave is one
// coalesced row of dwords
typedef __attribute__((address_space(1))) u32 gu32;              // known-global pointer: global_load / global_store, not flat
struct Cells { gu32 *base; size_t P; };                          // base = array + g
__device__ __forceinline__ Cells cells_of(u32 *arr, u32 P, u32 g) { Cells c; c.base = (gu32 *)arr + g; c.P = P; return c; }
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
__device__ __forceinline__ sc derive_weight(const u32 seed[8], u64 g, u32 t) {
  const u32 H0[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  const H8 d = sha_compress_v(H0[0], H0[1], H0[2], H0[3], H0[4], H0[5], H0[6], H0[7], seed[0], seed[1], seed[2], seed[3], seed[4], seed[5], seed[6], seed[7],
                              __builtin_bswap32((u32)g), __builtin_bswap32((u32)(g >> 32)), (t << 24) | 0x00800000u, 0, 0, 0, 0, 41u * 8u);
  sc w;
#pragma unroll
  for (int i = 0; i < 8; i++) w.v[i] = __builtin_bswap32(d.v[i]);       // digest bytes read as a little-endian number
  w.v[7] &= 0x00FFFFFFu;
  if (bpmi::sc_is_zero(w)) w = sc_small(1);
  return w;
}
__device__ __forceinline__ sq weight_of(const Params &q, u32 g, u32 t) {
  sc ws;
  if (q.weights) {
    const uint8_t *src = q.weights + ((size_t)(q.first + g) * 4 + t) * 32;
#pragma unroll
    for (int i = 0; i < 4; i++) { const u64 x = ld8_raw(src + 8 * i); ws.v[2 * i] = (u32)x; ws.v[2 * i + 1] = (u32)(x >> 32); }
    bpmi::sc_reduce_once(ws);
  } else {
    ws = derive_weight(q.seed, q.first + g, t);
  }
  return to_sq(ws);
}

// ---- one proof, structurally (rp::parse_blob): header, the three transcripts exactly fill the rest.  The scalars' ranges are
// checked by the roles that read them (role 3: taux, mu, t_hat, a, b; role 0: the x_i).
struct Layout {
  BPtr blob;
  u32 comp;            // byte offset of the (6 + 2k) x 33 compressed points
  u32 ts[3], tl[3];    // byte offsets and lengths of the three transcripts
  u32 start;
};
__device__ __forceinline__ bool parse_layout(Layout &L, const BPtr blob, u32 blen, u32 k) {
  const u32 fixed = 6 + 32 * (5 + k) + 33 * (6 + 2 * k) + 2;
  L.blob = blob;
  if (!(blen >= fixed && blob[0] == 'B' && blob[1] == 'P' && blob[2] == 'R' && blob[3] == 'P' && blob[4] == '1' && blob[5] == k)) return false;
  L.comp = 6 + 32 * (5 + k);
  u32 o = fixed - 2;
  L.start = ((u32)blob[o] << 8) | blob[o + 1];
  o += 2;
#pragma unroll
  for (int t = 0; t < 3; t++) {
    if (blen < o + 4) return false;
    const u64 x = ld8(blob + o);
    L.tl[t] = __builtin_bswap32((u32)x);
    o += 4;
    if (L.tl[t] > blen || blen - L.tl[t] < o) return false;
    L.ts[t] = o;
    o += L.tl[t];
  }
  return o == blen;
}
__device__ __forceinline__ BPtr xs_at(const Layout &L, u32 j) { return L.blob + (166 + 32 * j); }

// ---- role 0: the hash chain of the Protocol-2 transcript: per round i the prefix that ends after R_i's '&' is re-hashed and
// must spell x_i, which must also be the x_i of the proof's scalar section (inner_product_verifier.py:104-125, the hash half)
__device__ __forceinline__ bool role_hash_chain(const Layout &L, u32 k) {
  const BPtr ts2 = L.blob + L.ts[2];
  Walk w;
  walk_init(w, ts2, L.tl[2]);
  for (u32 j = 0; j < L.start && w.have; j++) walk_next(w, walk_end(w));
  Sha run;
  sha_init(run);
  sha_push_byte(run, '1');
  u32 hashed = 0;
  bool ok = true;
  for (u32 i = 0; i < k; i++) {
    if (!w.have) return false;
    walk_next(w, walk_end(w));                                     // L_i and R_i: compared with the proof's points by role 1
    if (!w.have) return false;
    walk_next(w, walk_end(w));
    if (!w.have) return false;
    const u32 upto = w.pos;                                        // prefix incl. the '&' after R_i
    sha_feed(run, ts2 + hashed, upto - hashed);
    hashed = upto;
    sc h, xi;
    mod_hash_q(h, run, ts2, upto);
    const u32 e = walk_end(w);
    const Dec d = parse_decimal(w.p + w.pos, e - w.pos);
    ok = sc_from_be(xi, xs_at(L, i));                              // the scalar section's x_i, < q
    if (!(ok && d.ok && sc_eq(d.v, h) && sc_eq(d.v, xi))) return false;
    walk_next(w, e);
  }
  return true;
}

// ---- role 1: the range-proof and Protocol-1 transcripts and the text half of Protocol 2 (rangeproof_verifier.py:42-53,
// inner_product_verifier.py:31-43, :104-125)
__device__ __forceinline__ bool role_text_checks(const Layout &L, u32 k) {
  const BPtr comp = L.blob + L.comp;
  bool ok = true;
  {
    const BPtr T1 = comp, T2 = comp + 33, A = comp + 66, S = comp + 99;
    Walk w;
    walk_init(w, L.blob + L.ts[0], L.tl[0]);
    u32 e = walk_end(w);                                             // item 0: not checked
    walk_next(w, e);
    for (u32 j = 1; j < 8 && ok; j++) {
      if (!w.have) return false;
      e = walk_end(w);
      const BPtr ip = w.p + w.pos;
      const u32 il = e - w.pos;
      if (j == 1) ok = point_item_equals(A, ip, il);
      else if (j == 2) ok = point_item_equals(S, ip, il);
      else if (j == 5) ok = point_item_equals(T1, ip, il);
      else if (j == 6) ok = point_item_equals(T2, ip, il);
      else ok = parse_decimal(ip, il).ok != 0;                       // y, z, x: read by the algebra roles
      walk_next(w, e);
    }
    if (!ok) return false;
  }
  {                                                                  // Protocol 1: item 1 = str(mod_hash(item 0 + "&"))
    const BPtr ts1 = L.blob + L.ts[1];
    Walk w;
    walk_init(w, ts1, L.tl[1]);
    u32 e = walk_end(w);
    walk_next(w, e);
    if (!w.have) return false;
    Sha s;
    sha_init(s);
    sha_push_byte(s, '1');
    sha_feed(s, ts1, w.pos);
    sc h;
    mod_hash_q(h, s, ts1, w.pos);
    e = walk_end(w);
    const Dec d = parse_decimal(w.p + w.pos, e - w.pos);
    if (!(d.ok && sc_eq(d.v, h))) return false;
  }
  {                                                                  // Protocol 2, the text half: items s+3i, s+3i+1 are base64(L_i), base64(R_i)
    Walk w;
    walk_init(w, L.blob + L.ts[2], L.tl[2]);
    for (u32 j = 0; j < L.start && w.have; j++) walk_next(w, walk_end(w));
    const BPtr Ls = comp + 33 * 6, Rs = Ls + 33 * k;
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
      walk_next(w, walk_end(w));                                     // x_i: role 0
    }
  }
  return true;
}

// items 3, 4 (and 7) of the range-proof transcript: y, z (, x); item 1 of the Protocol-1 transcript: x_ip
__device__ __forceinline__ bool read_yz(const Layout &L, sc &cy, sc &cz, sc *cx) {
  Walk w;
  walk_init(w, L.blob + L.ts[0], L.tl[0]);
  for (u32 j = 0; j < 8; j++) {
    if (!w.have) return false;
    const u32 e = walk_end(w);
    if (j == 3 || j == 4 || (j == 7 && cx)) {
      const Dec d = parse_decimal(w.p + w.pos, e - w.pos);
      if (!d.ok) return false;
      if (j == 3) cy = d.v; else if (j == 4) cz = d.v; else *cx = d.v;
    }
    if (j == 4 && !cx) return true;
    walk_next(w, e);
  }
  return true;
}

// ---- role 2: the inversion and what hangs off it.  Montgomery's trick inside the lane: prefix products of (x_1 .. x_k, y) (kept
// in LDS: k + 1 slots of 9 limbs per lane, the one indexed array of this file), ONE inversion (binary Euclid stripping all
// trailing zero bits per step, branch-free, ~195 steps), then the backward pass, which hands out x_t^-1 for t = k-1 .. 0, i.e.
// for the index bits b = 0 .. k-1 in ascending order -- exactly the order in which the y^-(2^b) and (2/y)^(2^b) squaring
// chains grow, so every table entry is produced in registers and stored once:
//   FG[b]  = x_(k-1-b)^2                      gs side: s_i = base_g * prod over set bits b of i of FG[b]      (stored by role 3)
//   FH[b]  = x_(k-1-b)^-2 y^-(2^b)            hs side: w4 b s_i^-1 y^-i = base_h * prod over set bits of FH[b]
//   GEO[b] = (2/y)^(2^b) for b < log2(bits), y^-(2^b) above: w2 z^(2+j) 2^(i % bits) y^-i = W2Z[j] * prod over set bits of GEO[b]
//            (W2Z[j] = w2 z^(2+j): role 3)
// and the scalar of R_t: -w4 x_t^-2.
__device__ __forceinline__ bool role_inverse_tables(const Params &q, u32 g, const Layout &L) {
  const u32 n = q.n, k = q.k, m = q.m;
  extern __shared__ u32 lds_slots[];
  const u32 lane = threadIdx.x;
  auto slot_store = [&](u32 slot, const sq &x) {
#pragma unroll
    for (int wd = 0; wd < 9; wd++) lds_slots[(slot * 9 + wd) * 64 + lane] = x.v[wd];
  };
  auto slot_load = [&](sq &x, u32 slot) {
#pragma unroll
    for (int wd = 0; wd < 9; wd++) x.v[wd] = lds_slots[(slot * 9 + wd) * 64 + lane];
  };
  auto load_x = [&](u32 t) { sc v; (void)sc_from_be(v, xs_at(L, t)); return v; };
  sc cys, czs;
  if (!read_yz(L, cys, czs, nullptr)) return false;
  bool nonzero = !bpmi::sc_is_zero(cys);                   // a zero challenge cannot come out of mod_hash
  const sq cy = to_sq(cys);
  sq run = bpmi::sq_small(1);
  for (u32 t = 0; t < k; t++) {
    const sc xv = load_x(t);
    nonzero &= !bpmi::sc_is_zero(xv);
    slot_store(t, run);
    mq(run, run, to_sq(xv));
  }
  if (!nonzero) return false;
  const sq prodx = run;                                    // prod x_d
  slot_store(k, run);
  mq(run, run, cy);
  sq rinv = to_sq(invq_v(to_sc(run)));
  sq yinv, pre;
  slot_load(pre, k);
  mq(yinv, rinv, pre);
  mq(rinv, rinv, cy);                                      // (prod x_d)^-1
  const sq w4 = weight_of(q, g, 3);
  sc as, bs;
  (void)sc_from_be(as, L.blob + 102);                      // a, b: range-checked by role 3
  (void)sc_from_be(bs, L.blob + 134);
  const Cells X = cells_of(q.ctx, q.P, g);
  sq t, u;
  mq(t, w4, to_sq(as)); mq(t, t, rinv); cell_store(X, CTX_BASE_G, t);
  mq(t, w4, to_sq(bs)); mq(t, t, prodx); cell_store(X, CTX_BASE_H, t);
  u32 lb = 0;
  while ((1u << lb) < n / m) lb++;                         // log2(bits per value)
  sq yp = yinv, rp;
  bpmi::sq_add(rp, yinv, yinv);                            // 2 / y
  u32 *op = q.pt_scalars + (size_t)g * (6 + 2 * k) * 8;
  for (u32 b = 0; b < k; b++) {
    const u32 d = k - 1 - b;
    const sq xd = to_sq(load_x(d));
    sq xi;
    slot_load(pre, d);
    mq(xi, rinv, pre);                                     // x_d^-1
    if (d) mq(rinv, rinv, xd);
    mq(t, xi, xi);                                         // x_d^-2        (FG[b] = x_d^2 needs no inverse: role 3)
    mq(u, w4, t); bpmi::sq_neg(u, u); store_canon(op + (6 + k + d) * 8, u);
    mq(t, t, yp); cell_store(X, CTX_FH(k, b), t);
    cell_store(X, CTX_GEO(k, b), b < lb ? rp : yp);
    if (b + 1 < k) {
      mq(yp, yp, yp);
      if (b + 1 < lb) mq(rp, rp, rp);
    }
  }
  return true;
}

// ---- role 3: everything that needs no inverse (rp::accumulate, the scalar half).  Cells 0..4 of the proof: c_g, c_h, c_u,
// gs_const, hs_const; the scalars of V_j and of T1, T2, A, S, u_new, P_new, L_j in wire order.
__device__ __forceinline__ bool role_scalars(const Params &q, u32 g, const Layout &L) {
  const u32 n = q.n, k = q.k, m = q.m;
  // operands are loaded (and range-checked) where they are used and weights derived when first needed: few values live at a time
  sc cxs, cys, czs;
  if (!read_yz(L, cys, czs, &cxs)) return false;
  bool ok = true;
  const Cells C = cells_of(q.contrib, q.P, g), X = cells_of(q.ctx, q.P, g);
  const u32 bits = n / m;
  u32 *op = q.pt_scalars + (size_t)g * (6 + 2 * k) * 8;
  const sq w2 = weight_of(q, g, 1);
  sq t, u;
  {
    const sq w1 = weight_of(q, g, 0), cz = to_sq(czs);
    sq z2, delta;
    mq(z2, cz, cz);
    mq(t, w2, cz);
    cell_store(C, 3, t);                                              // gs_const: w2 z
    bpmi::sq_neg(t, t); cell_store(C, 4, t);                          // hs_const
    {
      sq zp = z2;                                                     // V_j: -w1 z^(2+j); context W2Z[j] = w2 z^(2+j)
      for (u32 j = 0; j < m; j++) {
        mq(t, w1, zp); bpmi::sq_neg(t, t);
        store_canon(q.v_scalars + ((size_t)g * m + j) * 8, t);
        mq(t, w2, zp); cell_store(X, CTX_W2Z(k, j), t);
        mq(zp, zp, cz);
      }
    }
    {
      // sum_{i<n} y^i by doubling; delta = (z - z^2) ysum - (2^bits - 1) sum_{j=1..m} z^(j+2)
      sq ysum = bpmi::sq_small(1), ypw = to_sq(cys);
      const sq one = bpmi::sq_small(1);
      for (u32 l2 = 1; l2 < n; l2 <<= 1) {
        bpmi::sq_add(t, one, ypw);
        mq(ysum, ysum, t);
        mq(ypw, ypw, ypw);
      }
      bpmi::sq_sub(t, cz, z2);
      mq(delta, t, ysum);
      sq two_n = bpmi::sq_small(1);
      for (u32 i = 0; i < bits; i++) bpmi::sq_add(two_n, two_n, two_n);   // 2^bits mod q
      bpmi::sq_sub(two_n, two_n, one);
      sq zsum = bpmi::sq_small(0), zp;
      mq(zp, z2, cz);                                                 // z^3
      for (u32 j = 1; j <= m; j++) { bpmi::sq_add(zsum, zsum, zp); mq(zp, zp, cz); }
      mq(t, zsum, two_n);
      bpmi::sq_sub(delta, delta, t);
    }
    sc v;
    // c_g: w1 (t_hat - delta); c_h: w1 taux + w2 mu
    ok &= sc_from_be(v, L.blob + 70);
    bpmi::sq_sub(t, to_sq(v), delta); mq(t, t, w1); cell_store(C, 0, t);
    ok &= sc_from_be(v, L.blob + 6);
    mq(t, w1, to_sq(v));
    ok &= sc_from_be(v, L.blob + 38);
    mq(u, w2, to_sq(v)); bpmi::sq_add(t, t, u); cell_store(C, 1, t);
    // per-proof points in wire order: T1: -w1 x | T2: -w1 x^2 | A: -w2 | S: -w2 x | u_new: w3 + w4 a b | P_new: w2 - w4 | Ls | (Rs: role 2)
    const sq cx = to_sq(cxs);
    mq(t, w1, cx); bpmi::sq_neg(u, t); store_canon(op, u);
    mq(t, t, cx); bpmi::sq_neg(u, t); store_canon(op + 8, u);
    bpmi::sq_neg(u, w2); store_canon(op + 16, u);
    mq(t, w2, cx); bpmi::sq_neg(u, t); store_canon(op + 24, u);
  }
  const sq w3 = weight_of(q, g, 2);
  {
    // c_u: -(w2 x_ip t_hat + w3 x_ip) = -(w2 t_hat + w3) x_ip
    Walk w;
    walk_init(w, L.blob + L.ts[1], L.tl[1]);
    walk_next(w, walk_end(w));
    if (!w.have) return false;
    const Dec d = parse_decimal(w.p + w.pos, walk_end(w) - w.pos);
    if (!d.ok) return false;
    sc v;
    (void)sc_from_be(v, L.blob + 70);
    mq(t, w2, to_sq(v)); bpmi::sq_add(t, t, w3); mq(t, t, to_sq(d.v)); bpmi::sq_neg(t, t); cell_store(C, 2, t);
  }
  const sq w4 = weight_of(q, g, 3);
  {
    sc v;
    ok &= sc_from_be(v, L.blob + 102);                                // a
    mq(t, w4, to_sq(v));
    ok &= sc_from_be(v, L.blob + 134);                                // b
    mq(t, t, to_sq(v)); bpmi::sq_add(u, w3, t); store_canon(op + 32, u);
    bpmi::sq_sub(u, w2, w4); store_canon(op + 40, u);
  }
  for (u32 j = 0; j < k; j++) {
    sc xv;
    (void)sc_from_be(xv, xs_at(L, j));                                // range and origin: role 0
    const sq xj = to_sq(xv);
    mq(u, xj, xj); cell_store(X, CTX_FG(k, k - 1 - j), u);            // context: the factor of index bit k-1-j on the gs side
    mq(t, w4, u); bpmi::sq_neg(u, t); store_canon(op + (6 + j) * 8, u);
  }
  return ok;
}

// Four waves per group of `lanes` proofs, one per ROLE (see the head of the file).  status[role * Pall + g] = verdict of a role.
#define RP_ROLES 4
// A wire proof longer than this is invalid (both here and in the host twin): it bounds the transposed array.  A 64-bit proof
// is 2.6 KB, the largest shape the format allows (k = 16) under 8 KB.
#define RP_MAX_PROOF_BYTES 32768u
__global__ void __launch_bounds__(64) k_rp_roles(Params q) {
  ::raise_priority(q.prio);
  if (threadIdx.x >= q.lanes) return;
  const u32 role = blockIdx.x & (RP_ROLES - 1);
  const u32 g = (blockIdx.x / RP_ROLES) * q.lanes + threadIdx.x;
  if (g >= q.P) return;
  if (q.only_role >= 0 && (u32)q.only_role != role) { q.status[(size_t)role * q.Pall + g] = 1; return; }
  BPtr blob;
  blob.base = q.T + g; blob.stride = q.Tstride; blob.off = 0;
  const u64 blen64 = q.lens ? (u64)q.lens[g] : q.off[g + 1] - q.off[g];
  Layout L;
  bool ok = blen64 <= RP_MAX_PROOF_BYTES && parse_layout(L, blob, (u32)blen64, q.k);
  if (ok) {
    if (role == 0) ok = role_hash_chain(L, q.k);
    else if (role == 1) ok = role_text_checks(L, q.k);
    else if (role == 2) ok = role_inverse_tables(q, g, L);
    else ok = role_scalars(q, g, L);
  }
  q.status[(size_t)role * q.Pall + g] = ok ? 1 : 0;
  if (!ok) atomicMin(q.bad, (unsigned long long)(q.first + g));
}


#if defined(BPMI_ROLE_PROBE)
template <int ROLE> __global__ void __launch_bounds__(64) k_rp_role_probe(Params q) {
  const u32 g = blockIdx.x * 64 + threadIdx.x;
  BPtr blob;
  blob.base = q.T + g; blob.stride = q.Tstride; blob.off = 0;
  Layout L;
  bool ok = parse_layout(L, blob, (u32)(q.off[g + 1] - q.off[g]), q.k);
  if (ok) {
    if (ROLE == 0) ok = role_hash_chain(L, q.k);
    else if (ROLE == 1) ok = role_text_checks(L, q.k);
    else if (ROLE == 2) ok = role_inverse_tables(q, g, L);
    else ok = role_scalars(q, g, L);
  }
  q.status[g] = ok;
}
template __global__ void k_rp_role_probe<0>(Params);
template __global__ void k_rp_role_probe<1>(Params);
template __global__ void k_rp_role_probe<2>(Params);
template __global__ void k_rp_role_probe<3>(Params);
#endif

// ---- the 2n shared-generator contributions.  Block = 64 consecutive proofs at ONE (side, element range): lane = proof, the
// range is wave-uniform.  value(i) = base * prod over the set bits b of i of F[b]; the range's high bits select factors once
// (uniform branches), its EL = min(8, bits per value) elements are a binary tree over the low bits:
//   v0 | v1 = v0 F0 | v2 = v0 F1 | v3 = v2 F0 | v4 = v0 F2 | v5 = v4 F0 | v6 = v4 F1 | v7 = v6 F0          (7 multiplications)
// side 0 (gs): cell 5 + i = w4 a s_i.  side 1 (hs): cell 5 + n + i = w4 b s_i^-1 y^-i - w2 z^(2+j) 2^(i % bits) y^-i, both
// progressions walk the same tree.  A proof whose role 2 failed gets zeros (its context is not valid).
struct ElemGeom { u32 el_log, ranges; };       // elements per lane = 2^el_log; ranges per side = n >> el_log
__global__ void __launch_bounds__(64) k_rp_elements(Params q, ElemGeom eg) {
  ::raise_priority(q.prio);
  const u32 n = q.n, k = q.k;
  const u32 groups = (q.P + 63u) / 64u;
  const u32 grp = blockIdx.x % groups, rs = blockIdx.x / groups;          // rs = side * ranges + range
  const u32 side = rs / eg.ranges, range = rs % eg.ranges;
  const u32 g = grp * 64u + threadIdx.x;
  if (g >= q.P) return;
  const u32 i0 = range << eg.el_log, EL = 1u << eg.el_log;
  const Cells C = cells_of(q.contrib, q.P, g), X = cells_of(q.ctx, q.P, g);
  const u32 col0 = 5u + side * n + i0;
  if (!(q.status[(size_t)2 * q.Pall + g] & q.status[(size_t)3 * q.Pall + g])) {          // its context is not valid
    const sq z = bpmi::sq_small(0);
    for (u32 e = 0; e < EL; e++) cell_store(C, col0 + e, z);
    return;
  }
  sq f;
  if (side == 0) {
    sq a0;
    cell_load(a0, X, CTX_BASE_G);
    for (u32 b = eg.el_log; b < k; b++)
      if ((i0 >> b) & 1u) { cell_load(f, X, CTX_FG(k, b)); mq_inl(a0, a0, f); }
    sq F0, F1, F2;
    if (EL > 1) cell_load(F0, X, CTX_FG(k, 0));
    if (EL > 2) cell_load(F1, X, CTX_FG(k, 1));
    if (EL > 4) cell_load(F2, X, CTX_FG(k, 2));
    sq v, a2, a4, a6;
    cell_store(C, col0, a0);
    if (EL > 1) { mq_inl(v, a0, F0); cell_store(C, col0 + 1, v); }
    if (EL > 2) { mq_inl(a2, a0, F1); cell_store(C, col0 + 2, a2); mq_inl(v, a2, F0); cell_store(C, col0 + 3, v); }
    if (EL > 4) {
      mq_inl(a4, a0, F2); cell_store(C, col0 + 4, a4);
      mq_inl(v, a4, F0); cell_store(C, col0 + 5, v);
      mq_inl(a6, a4, F1); cell_store(C, col0 + 6, a6);
      mq_inl(v, a6, F0); cell_store(C, col0 + 7, v);
    }
    return;
  }
  // hs side: two progressions on the same tree
  const u32 bits = n / q.m;
  sq a0, b0;
  cell_load(a0, X, CTX_BASE_H);
  cell_load(b0, X, CTX_W2Z(k, i0 / bits));
  for (u32 b = eg.el_log; b < k; b++)
    if ((i0 >> b) & 1u) {
      cell_load(f, X, CTX_FH(k, b)); mq_inl(a0, a0, f);
      cell_load(f, X, CTX_GEO(k, b)); mq_inl(b0, b0, f);
    }
  sq F0, F1, F2, G0, G1, G2;
  if (EL > 1) { cell_load(F0, X, CTX_FH(k, 0)); cell_load(G0, X, CTX_GEO(k, 0)); }
  if (EL > 2) { cell_load(F1, X, CTX_FH(k, 1)); cell_load(G1, X, CTX_GEO(k, 1)); }
  if (EL > 4) { cell_load(F2, X, CTX_FH(k, 2)); cell_load(G2, X, CTX_GEO(k, 2)); }
  sq v, w, d, a2, a4, a6, b2, b4, b6;
  bpmi::sq_sub(d, a0, b0); cell_store(C, col0, d);
  if (EL > 1) { mq_inl(v, a0, F0); mq_inl(w, b0, G0); bpmi::sq_sub(d, v, w); cell_store(C, col0 + 1, d); }
  if (EL > 2) {
    mq_inl(a2, a0, F1); mq_inl(b2, b0, G1); bpmi::sq_sub(d, a2, b2); cell_store(C, col0 + 2, d);
    mq_inl(v, a2, F0); mq_inl(w, b2, G0); bpmi::sq_sub(d, v, w); cell_store(C, col0 + 3, d);
  }
  if (EL > 4) {
    mq_inl(a4, a0, F2); mq_inl(b4, b0, G2); bpmi::sq_sub(d, a4, b4); cell_store(C, col0 + 4, d);
    mq_inl(v, a4, F0); mq_inl(w, b4, G0); bpmi::sq_sub(d, v, w); cell_store(C, col0 + 5, d);
    mq_inl(a6, a4, F1); mq_inl(b6, b4, G1); bpmi::sq_sub(d, a6, b6); cell_store(C, col0 + 6, d);
    mq_inl(v, a6, F0); mq_inl(w, b6, G0); bpmi::sq_sub(d, v, w); cell_store(C, col0 + 7, d);
  }
}

// T[w * P + g] = bytes [8w, 8w + 8) of proof g for w < W, zero beyond the proof's end (or beyond RP_MAX_PROOF_BYTES).  64 x 64
// tiles through LDS: rows (proofs) are read as 512 contiguous bytes, columns (word index) written as 512 contiguous bytes.
__global__ void __launch_bounds__(256) k_rp_transpose(const uint8_t *__restrict__ blobs, const u64 *__restrict__ off, u32 P, u32 W, u64 *__restrict__ T) {
  __shared__ u64 tile[64][65];
  const u32 g0 = blockIdx.x * 64, w0 = blockIdx.y * 64, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (u32 r = wave; r < 64; r += 4) {
    const u32 g = g0 + r;
    u64 x = 0;
    if (g < P) {
      const u64 beg = off[g];
      u64 len = off[g + 1] - beg;
      if (len > RP_MAX_PROOF_BYTES) len = RP_MAX_PROOF_BYTES;
      const u64 o = 8ull * (w0 + lane);
      if (o < len) {
        __builtin_memcpy(&x, blobs + beg + o, 8);
        const u64 left = len - o;
        if (left < 8) x &= (1ull << (8 * left)) - 1ull;
      }
    }
    tile[r][lane] = x;
  }
  __syncthreads();
  for (u32 c = wave; c < 64; c += 4) {
    const u32 w = w0 + c, g = g0 + lane;
    if (w < W && g < P) T[(size_t)w * P + g] = tile[lane][c];
  }
}

// ---- wire format 2 -> the transposed format-1 array (round 4; host twin and the format: rp_wire_v2_host.hpp) -------------------------
// SIXTEEN lanes per proof rebuild what format 2 leaves out -- the three transcripts: base64 of the seeds and of the points, the
// challenges in decimal -- and write the format-1 proof straight into the word-major array the roles read (T[w * P + g]), so a
// batch of format-2 proofs needs no k_rp_transpose.  The expansion is a list of ITEMS (64-byte pieces of the unchanged head, the
// length fields, base64 items, decimal items); a first pass has the lanes of a proof measure the items whose length depends on
// the data (the digits of a challenge, "AA==" for the identity point) into LDS, then every lane walks the list, adding up
// lengths, and writes the items it owns (item i: lane i mod 16) at their byte offsets -- whole words with plain stores, the
// partial first / last word of an item with atomicOr into the zeroed array.  (One lane per proof took 0.27 ms of a batch's
// latency: 256 waves, ~100 000 dependent instructions each; this form is 4 096 waves of ~5 000.)
// lens[g] = length of the expansion, 0 for a proof that is not a well-formed format-2 proof (the roles then report it).
struct TWriter {
  u64 *base;        // T + g
  u32 stride;       // P
  u32 rows;         // W: never written beyond
  u32 pos;          // byte position of the next byte
  u32 first;        // word index of the item's first (possibly shared) word
  u64 cur;          // the word being assembled
};
__device__ __forceinline__ TWriter tw_at(u64 *base, u32 stride, u32 rows, u32 at) {
  TWriter t;
  t.base = base; t.stride = stride; t.rows = rows; t.pos = at; t.first = at >> 3; t.cur = 0;
  return t;
}
__device__ __forceinline__ void tw_store(TWriter &t, u32 w, bool shared) {
  if (w >= t.rows) return;
  u64 *p = t.base + (size_t)w * t.stride;
  if (shared) atomicOr((unsigned long long *)p, (unsigned long long)t.cur); else *p = t.cur;
}
__device__ __forceinline__ void tw_put(TWriter &t, u32 b) {
  t.cur |= (u64)(b & 0xFFu) << (8 * (t.pos & 7u));
  t.pos++;
  if ((t.pos & 7u) == 0) {
    const u32 w = (t.pos >> 3) - 1u;
    tw_store(t, w, w == t.first);           // the item's first word may hold the tail of the item before it
    t.cur = 0;
  }
}
__device__ __forceinline__ void tw_flush(TWriter &t) {
  if (t.pos & 7u) tw_store(t, t.pos >> 3, true);
}
__device__ __forceinline__ void tw_be32(TWriter &t, u32 v) { tw_put(t, v >> 24); tw_put(t, v >> 16); tw_put(t, v >> 8); tw_put(t, v); }
// base64 of n raw bytes at p, then '&'
__device__ __forceinline__ void tw_b64(TWriter &t, const uint8_t *p, u32 n) {
  for (u32 i = 0; i < n; i += 3) {
    const u32 b0 = p[i], b1 = i + 1 < n ? p[i + 1] : 0u, b2 = i + 2 < n ? p[i + 2] : 0u;
    const u32 v = (b0 << 16) | (b1 << 8) | b2;
    tw_put(t, b64_char((v >> 18) & 63u));
    tw_put(t, b64_char((v >> 12) & 63u));
    tw_put(t, i + 1 < n ? b64_char((v >> 6) & 63u) : '=');
    tw_put(t, i + 2 < n ? b64_char(v & 63u) : '=');
  }
  tw_put(t, '&');
}
__device__ __forceinline__ bool comp_is_identity(const uint8_t *comp) {
  u32 any = 0;
  for (int i = 0; i < 33; i++) any |= comp[i];
  return any == 0;
}
// the transcript item of a point (33 zero bytes = the identity = base64(b"\x00")), then '&'
__device__ __forceinline__ void tw_point(TWriter &t, const uint8_t *comp) {
  const uint8_t zero = 0;
  if (comp_is_identity(comp)) tw_b64(t, &zero, 1); else tw_b64(t, comp, 33);
}
// a 256-bit big-endian value at p -> base 10^9, least significant first; false when the value is >= q
__device__ __forceinline__ bool dec_chunks(u32 c[9], const uint8_t *p) {
  u32 w[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { const uint8_t *s = p + 28 - 4 * k; w[k] = ((u32)s[0] << 24) | ((u32)s[1] << 16) | ((u32)s[2] << 8) | s[3]; }
  const u32 Qw[8] = BPMI_SC_Q;
  bool lt = false, decided = false;
#pragma unroll
  for (int k = 7; k >= 0; k--) { if (!decided && w[k] != Qw[k]) { lt = w[k] < Qw[k]; decided = true; } }
#pragma unroll
  for (int j = 0; j < 9; j++) {
    u64 rem = 0;
#pragma unroll
    for (int k = 7; k >= 0; k--) { const u64 cur = (rem << 32) | w[k]; w[k] = (u32)(cur / 1000000000ull); rem = cur % 1000000000ull; }
    c[j] = (u32)rem;
  }
  return lt;
}
// digits of the value (no leading zeros; "0" for zero) + 1 for the '&'; 0 when the value is >= q
__device__ __forceinline__ u32 dec_item_len(const uint8_t *p) {
  u32 c[9];
  if (!dec_chunks(c, p)) return 0;
  u32 top = 0, tv = c[0];
#pragma unroll
  for (int j = 1; j < 9; j++) if (c[j]) { top = (u32)j; tv = c[j]; }
  u32 d = 1;
  for (u32 lim = 10u; d < 9u && tv >= lim; lim *= 10u) d++;
  return 9u * top + d + 1u;
}
__device__ __forceinline__ void tw_decimal(TWriter &t, const uint8_t *p) {
  u32 c[9];
  (void)dec_chunks(c, p);
  bool started = false;
#pragma unroll
  for (int j = 8; j >= 0; j--) {
    u32 div = 100000000u;
#pragma unroll
    for (int d = 0; d < 9; d++) {
      const u32 digit = (c[j] / div) % 10u;
      div /= 10u;
      if (digit || started || (j == 0 && d == 8)) { tw_put(t, '0' + digit); started = true; }
    }
  }
  tw_put(t, '&');
}
#define V2_LPP 16u                 // lanes per proof
#define V2_MAXLEN 64u              // measured items per proof: 4 + k challenges, 4 + 2 k points  (k <= 16)
// fmt = '2' or '3': the format of the call.  A format-3 proof is a format-2 proof followed by the y coordinates of its 6 + 2k points
// (32 bytes each; k_ec_decompress_wire checks them instead of taking square roots): only its length differs here.
__global__ void __launch_bounds__(64) k_rp_expand_v2(const uint8_t *__restrict__ blobs, const u64 *__restrict__ off, u32 P, u32 k, u32 W, u64 *__restrict__ T,
                                                     u32 *__restrict__ lens, u32 fmt, u32 prio) {
  ::raise_priority(prio);
  // Lanes of a wave only run side by side when they run the SAME code: the items are therefore written kind by kind -- all decimal
  // items of a proof at once (one per lane), then all point items, then the pieces of the unchanged head -- each at the byte
  // offset the group's first lane has worked out from the measured lengths.  (A first parallel version gave every lane "its"
  // items of the mixed list: at any item only one lane in sixteen was active, every wave ran the whole list serially, and the
  // kernel was slower than one lane per proof.)
  __shared__ u32 s_len[4][V2_MAXLEN];       // measured lengths: challenges y z x x_ip x_0 .., then points A S T1 T2 L_0 .. R_0 ..
  __shared__ u32 s_dec[4][24];              // byte offsets of the decimal items in layout order: y z x x_ip x_ip(again) x_0 ..
  __shared__ u32 s_pt[4][40];               // byte offsets of the point items: A S T1 T2 L_0 R_0 L_1 R_1 ..
  __shared__ u32 s_fix[4][8];               // offsets of: start, len_rp, seed, len_1, seed1, len_2, '&', seed1 (again)
  const u32 pw = threadIdx.x >> 4, lane = threadIdx.x & (V2_LPP - 1u);
  const u32 g = blockIdx.x * 4u + pw;
  const bool live = g < P;
  const uint8_t *b = blobs + (live ? off[g] : 0ull);
  const u64 n_wire = live ? off[g + 1] - off[g] : 0ull;
  const u32 body = 6 + 32 * (5 + k) + 33 * (6 + 2 * k), hints = fmt == (u32)'3' ? 32u * (6 + 2 * k) : 0u;
  const u64 n = n_wire >= hints ? n_wire - hints : 0ull;          // the format-2 part
  bool ok = live && n >= body + 132ull && n_wire <= RP_MAX_PROOF_BYTES && b[0] == 'B' && b[1] == 'P' && b[2] == 'R' && b[3] == 'P' && b[4] == fmt && b[5] == k;
  u32 sl = 0, sl1 = 0;
  if (ok) {
    sl = ((u32)b[body + 128] << 8) | b[body + 129];
    ok = n >= (u64)body + 132 + sl;
    if (ok) { sl1 = ((u32)b[body + 130 + sl] << 8) | b[body + 131 + sl]; ok = n == (u64)body + 132 + sl + sl1; }
  }
  const uint8_t *sc = b + 6, *pts = sc + 32 * (5 + k), *ch = b + body, *seed = b + body + 130, *seed1 = seed + sl + 2;
  const u32 nch = 4 + k, npt = 4 + 2 * k, nmeas = nch + npt;
  auto chal = [&](u32 j) { return j < 4 ? ch + 32 * j : sc + 32 * (5 + (j - 4)); };                     // y z x x_ip x_0 ..
  auto point = [&](u32 j) { return j == 0 ? pts + 66 : (j == 1 ? pts + 99 : (j == 2 ? pts : (j == 3 ? pts + 33 : pts + 33 * (6 + (j - 4))))); };      // A S T1 T2 L.. R..
  if (ok)
    for (u32 j = lane; j < nmeas; j += V2_LPP) s_len[pw][j] = j < nch ? dec_item_len(chal(j)) : (comp_is_identity(point(j - nch)) ? 5u : 45u);
  __syncthreads();
  const u32 *L = s_len[pw];
  if (ok) for (u32 j = 0; j < nch; j++) ok = ok && L[j] != 0;                   // a challenge >= q
  u32 out_len = 0, len_rp = 0, len_1 = 0, len_2 = 0;
  const u32 b64s = 4 * ((sl + 2) / 3) + 1, b64s1 = 4 * ((sl1 + 2) / 3) + 1;
  if (ok) {
    len_rp = b64s + L[nch] + L[nch + 1] + L[0] + L[1] + L[nch + 2] + L[nch + 3] + L[2];
    len_1 = b64s1 + L[3];
    len_2 = 1 + len_1;
    for (u32 j = 0; j < k; j++) len_2 += L[nch + 4 + j] + L[nch + 4 + k + j] + L[4 + j];
    out_len = body + 2 + 12 + len_rp + len_1 + len_2;
    ok = out_len <= RP_MAX_PROOF_BYTES && (out_len + 7u) / 8u <= W;
  }
  if (ok && lane == 0) {                                        // the layout, once per proof
    u32 pos = body;
    s_fix[pw][0] = pos; pos += 2;                               // start_transcript
    s_fix[pw][1] = pos; pos += 4;                               // length of the range-proof transcript
    s_fix[pw][2] = pos; pos += b64s;                            // seed
    s_pt[pw][0] = pos; pos += L[nch];                           // A
    s_pt[pw][1] = pos; pos += L[nch + 1];                       // S
    s_dec[pw][0] = pos; pos += L[0];                            // y
    s_dec[pw][1] = pos; pos += L[1];                            // z
    s_pt[pw][2] = pos; pos += L[nch + 2];                       // T1
    s_pt[pw][3] = pos; pos += L[nch + 3];                       // T2
    s_dec[pw][2] = pos; pos += L[2];                            // x
    s_fix[pw][3] = pos; pos += 4;
    s_fix[pw][4] = pos; pos += b64s1;
    s_dec[pw][3] = pos; pos += L[3];                            // x_ip
    s_fix[pw][5] = pos; pos += 4;
    s_fix[pw][6] = pos; pos += 1;
    s_fix[pw][7] = pos; pos += b64s1;
    s_dec[pw][4] = pos; pos += L[3];                            // x_ip again (Protocol 2 starts with the Protocol-1 transcript)
    for (u32 j = 0; j < k; j++) {
      s_pt[pw][4 + 2 * j] = pos; pos += L[nch + 4 + j];         // L_j
      s_pt[pw][5 + 2 * j] = pos; pos += L[nch + 4 + k + j];     // R_j
      s_dec[pw][5 + j] = pos; pos += L[4 + j];                  // x_j
    }
  }
  __syncthreads();
  if (ok) {
    u64 *base = T + g;
    // decimal items: y z x x_ip x_ip x_0 .. (5 + k of them), one per lane
    for (u32 j = lane; j < 5 + k; j += V2_LPP) {
      TWriter t = tw_at(base, P, W, s_dec[pw][j]);
      tw_decimal(t, j < 4 ? chal(j) : (j == 4 ? chal(3) : chal(j - 1)));
      tw_flush(t);
    }
    // point items: A S T1 T2 L_0 R_0 L_1 R_1 ..
    for (u32 j = lane; j < npt; j += V2_LPP) {
      const uint8_t *c = j < 4 ? point(j) : pts + 33 * (6 + ((j - 4) >> 1) + ((j - 4) & 1u) * k);
      TWriter t = tw_at(base, P, W, s_pt[pw][j]);
      tw_point(t, c);
      tw_flush(t);
    }
    // the head (the same bytes in both formats but for the '1') in pieces of 64 bytes
    for (u32 o = 64u * lane; o < body; o += 64u * V2_LPP) {
      const u32 cnt = body - o < 64u ? body - o : 64u;
      TWriter t = tw_at(base, P, W, o);
      for (u32 i = 0; i < cnt; i++) tw_put(t, (o + i) == 4u ? (u32)'1' : (u32)b[o + i]);
      tw_flush(t);
    }
    // the few fixed items, one lane each
    if (lane == 1) { TWriter t = tw_at(base, P, W, s_fix[pw][0]); tw_put(t, 0); tw_put(t, 3); tw_be32(t, len_rp); tw_flush(t); }      // start | len_rp: adjacent
    if (lane == 2) { TWriter t = tw_at(base, P, W, s_fix[pw][2]); tw_b64(t, seed, sl); tw_flush(t); }
    if (lane == 3) { TWriter t = tw_at(base, P, W, s_fix[pw][3]); tw_be32(t, len_1); tw_flush(t); }
    if (lane == 4) { TWriter t = tw_at(base, P, W, s_fix[pw][4]); tw_b64(t, seed1, sl1); tw_flush(t); }
    if (lane == 5) { TWriter t = tw_at(base, P, W, s_fix[pw][5]); tw_be32(t, len_2); tw_put(t, '&'); tw_flush(t); }                   // len_2 | '&': adjacent
    if (lane == 6) { TWriter t = tw_at(base, P, W, s_fix[pw][7]); tw_b64(t, seed1, sl1); tw_flush(t); }
  }
  if (live && lane == 0) lens[g] = ok ? out_len : 0u;
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
    bpmi::sc_add(cur, cur, sum);
    ::store_words8(shared + 8ull * col, cur.v);
  }
}

// out[0..3 + 2n): the scalars of g, h, u, gs_i, hs_i of the batch's MSM from the 5 + 2n summed columns:
// c_g, c_h, c_u, c_gs[i] + gs_const, c_hs[i] + hs_const  (BatchRangeVerifier.partial does the same on Python integers)
__global__ void __launch_bounds__(256) k_rp_shared_scalars(const u32 *__restrict__ shared, u32 n, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 + 2 * n) return;
  sc v;
  if (i < 3) {
    ::load_words8(v.v, shared + 8ull * i);
  } else {
    const u32 j = i - 3, side = j >= n ? 1u : 0u;
    sc a, b;
    ::load_words8(a.v, shared + 8ull * (5 + j));
    ::load_words8(b.v, shared + 8ull * (3 + side));
    bpmi::sc_add(v, a, b);
  }
  ::store_words8(out + 8ull * i, v.v);
}

}  // namespace rpd
