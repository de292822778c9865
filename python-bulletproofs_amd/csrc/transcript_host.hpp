// transcript_host.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).  HOST code.
// The Fiat-Shamir edge of the inner-product prover in native code: the transcript bytes utils/transcript.py builds
// (/root/reference/src/utils/transcript.py:13-33: base64 of the compressed point, '&'; the challenge in decimal, '&') and
// the challenge x = mod_hash(transcript, q) (src/utils/utils.py:84-97), used by bpmi_ipa_prove_rounds; and the reference's
// seeded "randomness" in bulk (bpmi_mod_hash_range).  No GPU call in here: tests/csrc_host/host_native_fuzz.cpp builds this
// file with ASan / UBSan / TSan.
#pragma once

namespace rpt {

// item of one point: base64(SEC1 compressed) + '&'; pt = x || y, 32 bytes little-endian each, 64 zero bytes = the identity
static inline void append_point(std::vector<uint8_t> &dg, const uint8_t pt[64]) {
  uint8_t comp[33] = {0}, item[48];
  bool zero = true;
  for (int k = 0; k < 64; k++) zero &= pt[k] == 0;
  if (!zero) {
    comp[0] = (pt[32] & 1) ? 3 : 2;                              // y is little-endian: its parity is in byte 0
    for (int k = 0; k < 32; k++) comp[1 + k] = pt[31 - k];        // x big-endian
  }
  const size_t il = rp::point_item(item, comp);
  dg.insert(dg.end(), item, item + il);
  dg.push_back('&');
}
// decimal digits of x (no leading zeros; "0" for zero) + '&'
static inline void append_decimal(std::vector<uint8_t> &dg, const rp::Sq &x) {
  uint64_t v[4] = {x.v[0], x.v[1], x.v[2], x.v[3]};
  char buf[80];
  int pos = 80;
  for (;;) {
    unsigned __int128 rem = 0;                                   // v /= 10^19
    const uint64_t D = 10000000000000000000ULL;
    for (int k = 3; k >= 0; k--) { const unsigned __int128 cur = (rem << 64) | v[k]; v[k] = (uint64_t)(cur / D); rem = cur % D; }
    uint64_t chunk = (uint64_t)rem;
    const bool more = (v[0] | v[1] | v[2] | v[3]) != 0;
    for (int k = 0; k < 19 && (more || chunk || k == 0); k++) { buf[--pos] = (char)('0' + chunk % 10); chunk /= 10; }
    if (!more) break;
  }
  dg.insert(dg.end(), buf + pos, buf + 80);
  dg.push_back('&');
}
// x = mod_hash(transcript so far, q); its decimal item is appended
static inline void challenge(rp::Sq &x, std::vector<uint8_t> &dg) {
  rp::Sha one;
  rp::sha_init(one);
  rp::sha_update(one, (const uint8_t *)"1", 1);
  rp::sha_update(one, dg.data(), dg.size());
  rp::mod_hash_q(x, one, dg.data(), dg.size());
  append_decimal(dg, x);
}
// the finished transcript into the caller's buffer of `cap` bytes: false (nothing written) when it does not fit
static inline bool export_digest(const std::vector<uint8_t> &dg, uint8_t *out, uint64_t cap, uint64_t *out_len) {
  if (dg.size() > cap) return false;
  if (!dg.empty()) memcpy(out, dg.data(), dg.size());
  *out_len = dg.size();
  return true;
}

}  // namespace rpt

extern "C" {

// out[i - lo] = mod_hash(str(i) || tail, q) for i in [lo, hi), 32 bytes little-endian each: the reference's seeded "randomness"
// (src/utils/utils.py:84-97; the provers draw their blinding vectors sL, sR this way, rangeproof_prover.py:57-60, one hash per
// element) in native code.  Host code, no GPU involved; `threads` host threads share the range.
int bpmi_mod_hash_range(const uint8_t *tail, uint64_t tail_len, uint64_t lo, uint64_t hi, int threads, uint8_t *out) {
  if ((!tail && tail_len) || !out || hi < lo) return BPMI_E_ARG;
  const uint64_t count = hi - lo;
  if (threads < 1) threads = 1;
  if ((uint64_t)threads > count) threads = count ? (int)count : 1;
  auto work = [&](int t) {
    std::vector<uint8_t> msg(21 + tail_len);        // 20 digits of a 64-bit counter + snprintf's terminator
    for (uint64_t i = lo + count * t / threads, e = lo + count * (t + 1) / threads; i < e; i++) {
      const int dl = snprintf((char *)msg.data(), 21, "%llu", (unsigned long long)i);
      if (tail_len) memcpy(msg.data() + dl, tail, tail_len);
      rp::Sha one;
      rp::sha_init(one);
      rp::sha_update(one, (const uint8_t *)"1", 1);
      rp::sha_update(one, msg.data(), dl + tail_len);
      rp::Sq v;
      rp::mod_hash_q(v, one, msg.data(), dl + tail_len);
      rp::q_to_le(out + 32 * (i - lo), v);
    }
  };
  hostpool::run(threads, work);
  return BPMI_OK;
}

}  // extern "C"
