// rp_wire_v2_host.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).  HOST code.
// Wire format 2 of a range proof: format 1 (rangeproofs/codec.py) without what a verifier can rebuild.  A format-1 proof carries
// its three Fiat-Shamir transcripts (/root/reference/src/utils/transcript.py:13-33) -- base64 of points that are in the proof
// anyway, the challenges in decimal -- and is 2.56 KB for a 64-bit proof, 3.7 x its information; at 2^14 proofs the 42 MB upload
// is the largest term of a batch's latency.  Format 2 keeps what cannot be derived:
//   "BPRP2" | k | taux mu t_hat a b xs[0..k) | T1 T2 A S u_new P_new Ls Rs        (the SAME bytes as format 1 up to here)
//   y | z | x | x_ip                     4 x 32 B big-endian: the challenges the transcripts would spell out in decimal
//   seed_len (2 B) | seed                item 0 of the range-proof transcript (raw; the transcript holds its base64)
//   seed1_len (2 B) | seed1              item 0 of the Protocol-1 transcript (empty for the reference's provers)
// 1.09 KB for a 64-bit proof.  Format 3 (round 6) is a format-2 proof with "BPRP3" for its magic, followed by
//   ys[0 .. 6 + 2k)                      32 B big-endian each: the y coordinate of the proof's points, in their order (0 for the identity)
// 1.67 KB: a verifier CHECKS each y (on the curve with x, the parity of the encoding's tag) instead of computing it -- the square
// roots were a quarter of a batch verification's device time.  A format-3 proof is valid exactly when its format-2 part is AND
// every y is the one the encoding stands for; a wrong y is an invalid proof.  expand_v2() writes the format-1 proof those fields stand for -- the transcripts are the canonical
// ones: rangeproof_prover.py:62-92, inner_product_prover.py:25-47, :94-110 -- and every verifier then runs on format 1 as before:
// a format-2 proof is valid exactly when its expansion is (tests: same verdicts on valid, corrupted and mutated proofs).  The
// device twin is rpd::k_rp_expand_v2 (rp_batch_kernels.hpp), compared with this byte for byte.
#pragma once
#include "host_tail.hpp"

namespace rpw {

static inline void put_be32(std::vector<uint8_t> &o, size_t at, uint32_t v) { o[at] = (uint8_t)(v >> 24); o[at + 1] = (uint8_t)(v >> 16); o[at + 2] = (uint8_t)(v >> 8); o[at + 3] = (uint8_t)v; }
static inline void put_b64(std::vector<uint8_t> &o, const uint8_t *p, size_t n) {
  std::vector<uint8_t> t(4 * ((n + 2) / 3) + 4);
  const size_t l = rp::b64_encode(t.data(), p, n);
  o.insert(o.end(), t.begin(), t.begin() + l);
  o.push_back('&');
}
static inline void put_point(std::vector<uint8_t> &o, const uint8_t comp[33]) {
  uint8_t item[48];
  const size_t l = rp::point_item(item, comp);
  o.insert(o.end(), item, item + l);
  o.push_back('&');
}
// length of the format-2 or format-3 proof that starts at b (0: not a well-formed one within n bytes)
static inline size_t v2_length(const uint8_t *b, size_t n) {
  if (n < 6 || (memcmp(b, "BPRP2", 5) != 0 && memcmp(b, "BPRP3", 5) != 0)) return 0;
  const uint32_t k = b[5];
  if (k > 16) return 0;
  size_t o = 6 + 32 * (size_t)(5 + k) + 33 * (size_t)(6 + 2 * k) + 128;
  for (int s = 0; s < 2; s++) {
    if (n < o + 2) return 0;
    o += 2 + (((size_t)b[o] << 8) | b[o + 1]);
  }
  if (b[4] == '3') o += 32 * (size_t)(6 + 2 * k);
  return o <= n ? o : 0;
}
// Format 3: y (32 B big-endian) is THE y coordinate of the encoding comp (SEC1 compressed; 33 zero bytes = identity, y = 0)?
// The device twin is ec_hinted_one (point_kernels.hpp).
static inline bool hint_ok(const uint8_t comp[33], const uint8_t y[32]) {
  using namespace bpmi_host;
  u64 wx[4], wy[4];
  for (int i = 0; i < 4; i++) {
    wx[i] = wy[i] = 0;
    for (int j = 0; j < 8; j++) { wx[i] = (wx[i] << 8) | comp[1 + 8 * (3 - i) + j]; wy[i] = (wy[i] << 8) | y[8 * (3 - i) + j]; }
  }
  if (comp[0] == 0) return (wx[0] | wx[1] | wx[2] | wx[3] | wy[0] | wy[1] | wy[2] | wy[3]) == 0;
  if ((comp[0] != 2 && comp[0] != 3) || ge_p(wx) || ge_p(wy) || (wy[0] & 1u) != (comp[0] & 1u)) return false;
  f64 fx, fy, a, t, seven = {{7, 0, 0, 0}};
  memcpy(fx.v, wx, 32); memcpy(fy.v, wy, 32);
  f_sqr(t, fx); f_mul(a, t, fx); f_add(a, a, seven);
  f_sqr(t, fy);
  return memcmp(t.v, a.v, 32) == 0;
}
// format 2 -> format 1; false: not a format-2 proof (bad magic / lengths, a challenge >= q)
static inline bool expand_v2(const uint8_t *b, size_t n, std::vector<uint8_t> &out) {
  out.clear();
  // (v2_length answers 0 for "not a proof": an EMPTY blob, n = 0, must not pass as its own length -- round 5: found by the sanitizer
  // harness once its random sequence changed; the device expander checks n >= body + 132 on its own)
  if (n == 0 || v2_length(b, n) != n) return false;
  const uint32_t k = b[5];
  const size_t body = 6 + 32 * (size_t)(5 + k) + 33 * (size_t)(6 + 2 * k);
  if (b[4] == '3') {                               // the hints must be the points' y coordinates; the rest is a format-2 proof
    n -= 32 * (size_t)(6 + 2 * k);
    for (uint32_t j = 0; j < 6 + 2 * k; j++)
      if (!hint_ok(b + 6 + 32 * (size_t)(5 + k) + 33 * (size_t)j, b + n + 32 * (size_t)j)) return false;
  }
  const uint8_t *sc = b + 6, *pts = sc + 32 * (size_t)(5 + k), *ch = b + body;
  const uint8_t *T1 = pts, *T2 = pts + 33, *A = pts + 66, *S = pts + 99, *Ls = pts + 33 * 6, *Rs = Ls + 33 * (size_t)k;
  rp::Sq y, z, x, xip, xi[16];
  bool lt;
  rp::q_from_be(y, ch, lt); if (!lt) return false;
  rp::q_from_be(z, ch + 32, lt); if (!lt) return false;
  rp::q_from_be(x, ch + 64, lt); if (!lt) return false;
  rp::q_from_be(xip, ch + 96, lt); if (!lt) return false;
  for (uint32_t j = 0; j < k; j++) { rp::q_from_be(xi[j], sc + 32 * (size_t)(5 + j), lt); if (!lt) return false; }
  size_t o = body + 128;
  const size_t sl = ((size_t)b[o] << 8) | b[o + 1];
  const uint8_t *seed = b + o + 2;
  o += 2 + sl;
  const size_t sl1 = ((size_t)b[o] << 8) | b[o + 1];
  const uint8_t *seed1 = b + o + 2;
  out.assign(b, b + body);
  out[4] = '1';
  out.push_back(0); out.push_back(3);                              // start_transcript: the items of "b64(seed1)&x_ip&" split at '&'
  // range-proof transcript (rangeproof_prover.py:62-92)
  size_t at = out.size();
  out.resize(at + 4);
  put_b64(out, seed, sl); put_point(out, A); put_point(out, S);
  rpt::append_decimal(out, y); rpt::append_decimal(out, z);
  put_point(out, T1); put_point(out, T2);
  rpt::append_decimal(out, x);
  put_be32(out, at, (uint32_t)(out.size() - at - 4));
  // Protocol 1 (inner_product_prover.py:25-47)
  at = out.size();
  out.resize(at + 4);
  const size_t t1_at = out.size();
  put_b64(out, seed1, sl1);
  rpt::append_decimal(out, xip);
  const size_t t1_len = out.size() - t1_at;
  put_be32(out, at, (uint32_t)t1_len);
  // Protocol 2 (:62-67, :94-110): "&" + the Protocol-1 transcript, then L_i, R_i, x_i per round
  at = out.size();
  out.resize(at + 4);
  out.push_back('&');
  { const std::vector<uint8_t> t1(out.begin() + t1_at, out.begin() + t1_at + t1_len); out.insert(out.end(), t1.begin(), t1.end()); }
  for (uint32_t j = 0; j < k; j++) { put_point(out, Ls + 33 * (size_t)j); put_point(out, Rs + 33 * (size_t)j); rpt::append_decimal(out, xi[j]); }
  put_be32(out, at, (uint32_t)(out.size() - at - 4));
  return true;
}

}  // namespace rpw

extern "C" {

// Format-2 proofs (n_proofs of them, proof g = blobs[off[g], off[g + 1])) -> format-1 proofs packed into out[0, cap) with out_off[0 ..
// n_proofs] (host code; the batch verifiers do this on the device).  *first_bad = the first proof that is not a well-formed
// format-2 proof (-1: none; then out / out_off are complete).  BPMI_E_ARG: null arguments, offsets that leave the buffer, cap too small.
int bpmi_rp_wire_v2_to_v1(const uint8_t *blobs, uint64_t blobs_len, const uint64_t *off, uint64_t n_proofs, uint8_t *out, uint64_t cap, uint64_t *out_off,
                          int64_t *first_bad) {
  if (!blobs || !off || !out || !out_off || !first_bad) return BPMI_E_ARG;
  *first_bad = -1;
  if (off[0] > blobs_len) return BPMI_E_ARG;
  for (uint64_t g = 0; g < n_proofs; g++) if (off[g] > off[g + 1] || off[g + 1] > blobs_len) return BPMI_E_ARG;      // the whole table, before a byte is read
  std::vector<uint8_t> one;
  uint64_t o = 0;
  out_off[0] = 0;
  for (uint64_t g = 0; g < n_proofs; g++) {
    if (!rpw::expand_v2(blobs + off[g], (size_t)(off[g + 1] - off[g]), one)) { *first_bad = (int64_t)g; return BPMI_OK; }
    if (o + one.size() > cap) return BPMI_E_ARG;
    memcpy(out + o, one.data(), one.size());
    o += one.size();
    out_off[g + 1] = o;
  }
  return BPMI_OK;
}

}  // extern "C"
