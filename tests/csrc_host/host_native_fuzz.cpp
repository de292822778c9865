// Sanitizer harness for the native HOST code of libbpmi that is not the wire parser (that one: rp_fuzz.cpp):
//   rp_algebra_host.hpp   bpmi_rp_poly_coeffs / _final_vectors / _verifier_vectors (threaded O(n m) scalar algebra;
//                         /root/reference/src/rangeproofs/rangeproof_aggreg_prover.py:117-146, rangeproof_aggreg_verifier.py:96-108)
//   transcript_host.hpp   the transcript builder of bpmi_ipa_prove_rounds (/root/reference/src/utils/transcript.py:13-33) with its
//                         bounded export, bpmi_mod_hash_range (src/utils/utils.py:84-97)
//   host_tail.hpp         the MSM's window combine on the host
//   rp_wire_v2_host.hpp   wire formats 2 and 3 -> format 1 (untrusted bytes in: every malformed shape must be refused, never overrun;
//                         format 3's y coordinates checked with host_tail.hpp's field arithmetic)
//   host_pool.hpp         the sleeping worker pool those loops run on: concurrent callers, nested loops, every width
// Built by tests/test_host_native_sanitizers.py with -fsanitize=address,undefined (every output buffer is a heap block of
// EXACTLY the documented size, so an overrun of one byte is a report) and again with -fsanitize=thread (the threaded entry
// points with 1..8 threads).  Besides "no report" it checks what can be checked without an oracle: results do not depend on the
// thread count, t_hat == <l, r>, the export refuses a buffer one byte short and fills one that fits exactly.
//   host_native_fuzz <iterations> [seed]        exit 0 = every check held
#include <atomic>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/bpmi.h"
#include "rp_batch_host.hpp"
#include "host_pool.hpp"
#include "rp_algebra_host.hpp"
#include "transcript_host.hpp"
#include "rp_wire_v2_host.hpp"
#include "host_tail.hpp"

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static int fails = 0;
#define CHECK(c) do { if (!(c)) { if (fails++ < 20) fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); } } while (0)

// a heap block of exactly n bytes (n = 0: one byte nobody may touch ... ASan poisons around both)
struct Buf {
  uint8_t *p; size_t n;
  explicit Buf(size_t n_) : p((uint8_t *)malloc(n_ ? n_ : 1)), n(n_) { for (size_t i = 0; i < n; i++) p[i] = (uint8_t)rnd(); }
  ~Buf() { free(p); }
  Buf(const Buf &) = delete;
};
static void rand_scalar(uint8_t out[32], int kind) {
  rp::Sq v;
  for (int k = 0; k < 4; k++) v.v[k] = rnd();
  if (kind == 1) { v = rp::q_small(rnd() % 3); }
  else if (kind == 2) { for (int k = 0; k < 4; k++) v.v[k] = rp::QW[k]; v.v[0] -= 1 + rnd() % 3; }      // q - 1 .. q - 3
  else { v.v[3] &= 0x7FFFFFFFFFFFFFFFULL; }                                                                // < 2^255 < q
  rp::q_to_le(out, v);
}

static void test_algebra() {
  static const uint32_t shapes[][2] = {{1, 1}, {2, 1}, {8, 1}, {64, 1}, {3, 1}, {2, 2}, {4, 8}, {16, 4}, {64, 9}, {8, 130}, {1, 700}};
  const uint32_t n = shapes[rnd() % 11][0], m = shapes[rnd() % 11][1];
  const int aggregated = (m > 1) ? 1 : (int)(rnd() & 1);
  const uint64_t nm = (uint64_t)n * m;
  Buf aL(nm), sL(32 * nm), sR(32 * nm), y(32), z(32), x(32);
  const int kind = (int)(rnd() % 4);
  for (uint64_t i = 0; i < nm; i++) { rand_scalar(sL.p + 32 * i, kind); rand_scalar(sR.p + 32 * i, (int)(rnd() % 3)); }
  rand_scalar(y.p, 0); rand_scalar(z.p, (int)(rnd() % 3)); rand_scalar(x.p, 0);
  if ((y.p[0] | y.p[1]) == 0) y.p[0] = 5;                       // y = 0 is a refused argument (checked below)
  Buf t1a(32), t2a(32), t1b(32), t2b(32);
  const int th = 1 + (int)(rnd() % 8);
  CHECK(bpmi_rp_poly_coeffs(n, m, aggregated, aL.p, sL.p, sR.p, y.p, z.p, 1, t1a.p, t2a.p) == BPMI_OK);
  CHECK(bpmi_rp_poly_coeffs(n, m, aggregated, aL.p, sL.p, sR.p, y.p, z.p, th, t1b.p, t2b.p) == BPMI_OK);
  CHECK(!memcmp(t1a.p, t1b.p, 32) && !memcmp(t2a.p, t2b.p, 32));
  Buf ls(32 * nm), rs(32 * nm), that(32), hsc(32 * nm), ysc(32 * nm), ls2(32 * nm), rs2(32 * nm), that2(32), hsc2(32 * nm), ysc2(32 * nm);
  CHECK(bpmi_rp_final_vectors(n, m, aggregated, aL.p, sL.p, sR.p, y.p, z.p, x.p, 1, ls.p, rs.p, that.p, hsc.p, ysc.p) == BPMI_OK);
  CHECK(bpmi_rp_final_vectors(n, m, aggregated, aL.p, sL.p, sR.p, y.p, z.p, x.p, th, ls2.p, rs2.p, that2.p, hsc2.p, ysc2.p) == BPMI_OK);
  CHECK(!memcmp(ls.p, ls2.p, 32 * nm) && !memcmp(rs.p, rs2.p, 32 * nm) && !memcmp(that.p, that2.p, 32) && !memcmp(hsc.p, hsc2.p, 32 * nm) &&
        !memcmp(ysc.p, ysc2.p, 32 * nm));
  {                                                             // t_hat == <l, r>; yscale_i y^i == 1
    rp::Sq acc = rp::q_small(0), l, r, u, Y, yp = rp::q_small(1), one = rp::q_small(1);
    rp::q_from_le(Y, y.p);
    bool ys_ok = true;
    for (uint64_t i = 0; i < nm; i++) {
      rp::q_from_le(l, ls.p + 32 * i); rp::q_from_le(r, rs.p + 32 * i);
      rp::q_mul(u, l, r); rp::q_add(acc, acc, u);
      rp::q_from_le(u, ysc.p + 32 * i); rp::q_mul(u, u, yp);
      ys_ok &= !memcmp(u.v, one.v, 32);
      rp::q_mul(yp, yp, Y);
    }
    uint8_t le[32];
    rp::q_to_le(le, acc);
    CHECK(!memcmp(le, that.p, 32));
    CHECK(ys_ok);
  }
  Buf hsc3(32 * nm), ysc3(32 * nm), ysum(32), hsc4(32 * nm), ysc4(32 * nm), ysum4(32);
  CHECK(bpmi_rp_verifier_vectors(n, m, aggregated, y.p, z.p, 1, hsc3.p, ysc3.p, ysum.p) == BPMI_OK);
  CHECK(bpmi_rp_verifier_vectors(n, m, aggregated, y.p, z.p, th, hsc4.p, ysc4.p, ysum4.p) == BPMI_OK);
  CHECK(!memcmp(hsc3.p, hsc4.p, 32 * nm) && !memcmp(ysc3.p, ysc4.p, 32 * nm) && !memcmp(ysum.p, ysum4.p, 32));
  CHECK(!memcmp(hsc3.p, hsc.p, 32 * nm) && !memcmp(ysc3.p, ysc.p, 32 * nm));        // prover's and verifier's halves agree
  // refused arguments never touch the outputs' neighbours
  uint8_t zero[32] = {0};
  CHECK(bpmi_rp_verifier_vectors(n, m, aggregated, zero, z.p, th, hsc4.p, ysc4.p, ysum4.p) == BPMI_E_ARG);
  CHECK(bpmi_rp_final_vectors(n, m, aggregated, aL.p, sL.p, sR.p, zero, z.p, x.p, th, ls2.p, rs2.p, that2.p, hsc2.p, ysc2.p) == BPMI_E_ARG);
  CHECK(bpmi_rp_poly_coeffs(0, m, aggregated, aL.p, sL.p, sR.p, y.p, z.p, th, t1b.p, t2b.p) == BPMI_E_ARG);
  CHECK(bpmi_rp_poly_coeffs(n, m, aggregated, nullptr, sL.p, sR.p, y.p, z.p, th, t1b.p, t2b.p) == BPMI_E_ARG);
}

static void test_transcript() {
  // what bpmi_ipa_prove_rounds does between two GPU calls, for 1 .. 24 rounds, from a digest of 0 .. 300 bytes
  const size_t dl = rnd() % 301;
  Buf digest(dl);
  std::vector<uint8_t> dg(digest.p, digest.p + dl);
  const int rounds = 1 + (int)(rnd() % 24);
  for (int r = 0; r < rounds; r++) {
    uint8_t pt[64];
    for (int side = 0; side < 2; side++) {
      const int kind = (int)(rnd() % 6);
      for (int k = 0; k < 64; k++) pt[k] = kind == 0 ? 0 : (kind == 1 ? 0xFF : (uint8_t)rnd());
      const size_t before = dg.size();
      rpt::append_point(dg, pt);
      CHECK(dg.size() == before + (kind == 0 ? 5 : 45) && dg.back() == '&');     // base64 of 33 bytes = 44 characters; the identity is b"\x00" -> "AA=="
    }
    rp::Sq x;
    const size_t before = dg.size();
    rpt::challenge(x, dg);
    CHECK(dg.size() > before + 1 && dg.size() <= before + 79 && dg.back() == '&');
    CHECK(dg[before] != '0' || dg.size() == before + 2);        // no leading zero
    // the decimal item parses back to x
    rp::Sq back = rp::q_small(0), ten = rp::q_small(10);
    for (size_t i = before; i + 1 < dg.size(); i++) { rp::q_mul(back, back, ten); rp::q_add(back, back, rp::q_small(dg[i] - '0')); }
    CHECK(!memcmp(back.v, x.v, 32));
  }
  // bounded export: a buffer one byte short is refused and untouched, one that fits exactly is filled
  uint64_t out_len = 12345;
  {
    Buf out(dg.size() - 1);
    std::vector<uint8_t> copy(out.p, out.p + out.n);
    CHECK(!rpt::export_digest(dg, out.p, out.n, &out_len));
    CHECK(out_len == 12345 && !memcmp(copy.data(), out.p, out.n));
  }
  {
    Buf out(dg.size());
    CHECK(rpt::export_digest(dg, out.p, out.n, &out_len));
    CHECK(out_len == dg.size() && !memcmp(out.p, dg.data(), dg.size()));
  }
  // decimal items at the chunk seams of the conversion (10^19 per division)
  static const uint64_t seams[] = {0ULL, 1ULL, 9ULL, 10ULL, 9999999999999999999ULL, 10000000000000000000ULL, ~0ULL};
  for (uint64_t lo : seams) for (uint64_t hi : {0ULL, 1ULL, 5421010862427522170ULL}) {
    rp::Sq v = rp::q_small(0);
    v.v[0] = lo; v.v[1] = hi;
    std::vector<uint8_t> item;
    rpt::append_decimal(item, v);
    rp::Sq back = rp::q_small(0), ten = rp::q_small(10);
    for (size_t i = 0; i + 1 < item.size(); i++) { rp::q_mul(back, back, ten); rp::q_add(back, back, rp::q_small(item[i] - '0')); }
    CHECK(!memcmp(back.v, v.v, 32) && item.back() == '&' && (item[0] != '0' || item.size() == 2));
  }
}

static void test_mod_hash_range() {
  const size_t tl = rnd() % 200;
  Buf tail(tl);
  static const uint64_t starts[] = {0ULL, 7ULL, 99ULL, 999999ULL, 9999999999999999990ULL, 18446744073709551000ULL};
  const uint64_t lo = starts[rnd() % 6] + rnd() % 5, count = rnd() % 40;
  const int th = 1 + (int)(rnd() % 8);
  Buf a(32 * count), b(32 * count);
  CHECK(bpmi_mod_hash_range(tail.p, tl, lo, lo + count, 1, a.p) == BPMI_OK);
  CHECK(bpmi_mod_hash_range(tail.p, tl, lo, lo + count, th, b.p) == BPMI_OK);
  CHECK(!memcmp(a.p, b.p, 32 * count));
  if (count) {                                                   // element i is the hash of str(lo + i) || tail, whatever the range around it
    const uint64_t i = rnd() % count;
    Buf one(32);
    CHECK(bpmi_mod_hash_range(tail.p, tl, lo + i, lo + i + 1, 3, one.p) == BPMI_OK);
    CHECK(!memcmp(one.p, a.p + 32 * i, 32));
    rp::Sq v;
    rp::q_from_le(v, one.p);
    CHECK(!rp::ge_q(v.v));
  }
  CHECK(bpmi_mod_hash_range(tail.p, tl, lo + 1, lo, th, b.p) == BPMI_E_ARG);
  CHECK(bpmi_mod_hash_range(nullptr, 0, lo, lo + count, th, b.p) == BPMI_OK);       // an empty tail may be a null pointer
  CHECK(bpmi_mod_hash_range(nullptr, 3, lo, lo + count, th, b.p) == BPMI_E_ARG);
}

static void test_host_tail() {
  // window sums as the GPU leaves them: W x nv records of 4 x 9 limbs; the arithmetic is total on any tight limbs (a random
  // record is just not a curve point), the identity is the all-zero record
  const uint32_t c = 2 + (uint32_t)(rnd() % 15), W = 255u / c + 1u;
  bpmi::TailOffs to;
  to.nv = (rnd() & 1) ? 4u : 1u;
  to.off[0] = 0;
  for (uint32_t v = 1; v < 4; v++) to.off[v] = to.nv == 4 ? to.off[v - 1] + 1 + (uint32_t)(rnd() % 3) : 0;
  if (to.nv == 4 && to.off[3] >= c) { to.nv = 1; to.off[1] = to.off[2] = to.off[3] = 0; }
  if (to.nv == 1 && (rnd() & 1)) to.nv = 2 + (uint32_t)(rnd() % 3);      // several sums per window at offset 0 (the parts of k_msm_mid)
  if (to.nv == 4 && (rnd() & 1)) {                                // wide windows at the top with their own offsets (round 5): any count, any increasing offsets below c
    to.top = 1 + (uint32_t)(rnd() % W);
    to.top_off[0] = 0;
    for (uint32_t v = 1; v < 4; v++) to.top_off[v] = to.top_off[v - 1] + 1 + (uint32_t)(rnd() % 3);
    if (to.top_off[3] >= c) to.top = 0;
  }
  Buf E(4 * 36 * (size_t)W * to.nv);
  bpmi::u32 *e = (bpmi::u32 *)E.p;
  const int kind = (int)(rnd() % 4);
  for (size_t r = 0; r < (size_t)W * to.nv; r++)
    for (int k = 0; k < 36; k++) {
      bpmi::u32 v = (bpmi::u32)rnd() & 0x1FFFFFFFu;
      if (k % 9 == 8) v &= 0xFFFFFFu;                           // limb 8 holds 24 bits
      if (kind == 0 || (kind == 1 && (rnd() & 1))) v = 0;       // identities
      e[36 * r + k] = v;
    }
  if (kind == 1) for (size_t r = 0; r < (size_t)W * to.nv; r++) { bool z = e[36 * r + 18] == 0; for (int k = 0; k < 36; k++) if (z) e[36 * r + k] = 0; }
  Buf out(64), out2(64);
  bpmi_host::tail_combine(out.p, e, W, c, to);
  bpmi_host::tail_combine(out2.p, e, W, c, to);
  CHECK(!memcmp(out.p, out2.p, 64));
  if (kind == 0) { uint8_t z[64] = {0}; CHECK(!memcmp(out.p, z, 64)); }
}

// a structurally valid format-2 proof with random content (the expander never decodes a point: any 33 bytes will do), or a
// format-3 one: its points are then the identity, G or -G, followed by their y coordinates
static const uint8_t GX[32] = {0x79, 0xBE, 0x66, 0x7E, 0xF9, 0xDC, 0xBB, 0xAC, 0x55, 0xA0, 0x62, 0x95, 0xCE, 0x87, 0x0B, 0x07,
                               0x02, 0x9B, 0xFC, 0xDB, 0x2D, 0xCE, 0x28, 0xD9, 0x59, 0xF2, 0x81, 0x5B, 0x16, 0xF8, 0x17, 0x98};
static const uint8_t GY[32] = {0x48, 0x3A, 0xDA, 0x77, 0x26, 0xA3, 0xC4, 0x65, 0x5D, 0xA4, 0xFB, 0xFC, 0x0E, 0x11, 0x08, 0xA8,
                               0xFD, 0x17, 0xB4, 0x48, 0xA6, 0x85, 0x54, 0x19, 0x9C, 0x47, 0xD0, 0x8F, 0xFB, 0x10, 0xD4, 0xB8};
static std::vector<uint8_t> random_v2(uint32_t k, bool v3 = false) {
  std::vector<uint8_t> b = {'B', 'P', 'R', 'P', (uint8_t)(v3 ? '3' : '2'), (uint8_t)k};
  std::vector<uint8_t> ys;
  uint8_t sc32[32];
  auto put_scalar = [&] { rand_scalar(sc32, (int)(rnd() % 3)); for (int i = 31; i >= 0; i--) b.push_back(sc32[i]); };      // big-endian, < q
  for (uint32_t j = 0; j < 5 + k; j++) put_scalar();
  for (uint32_t j = 0; j < 6 + 2 * k; j++) {
    const bool inf = rnd() % 9 == 0;
    if (!v3) { for (int i = 0; i < 33; i++) b.push_back(inf ? 0 : (uint8_t)rnd()); continue; }
    if (inf) { for (int i = 0; i < 33; i++) b.push_back(0); for (int i = 0; i < 32; i++) ys.push_back(0); continue; }
    const bool neg = rnd() & 1;                      // -G: y = p - Gy (odd)
    b.push_back(neg ? 3 : 2);
    for (int i = 0; i < 32; i++) b.push_back(GX[i]);
    bpmi_host::f64 y, z = {{0, 0, 0, 0}};
    for (int w = 0; w < 4; w++) { y.v[w] = 0; for (int i = 0; i < 8; i++) y.v[w] = (y.v[w] << 8) | GY[8 * (3 - w) + i]; }
    if (neg) bpmi_host::f_sub(y, z, y);
    for (int w = 3; w >= 0; w--) for (int i = 7; i >= 0; i--) ys.push_back((uint8_t)(y.v[w] >> (8 * i)));
  }
  for (int j = 0; j < 4; j++) put_scalar();
  for (int s = 0; s < 2; s++) {
    const size_t sl = rnd() % 3 == 0 ? 0 : rnd() % 40;
    b.push_back((uint8_t)(sl >> 8)); b.push_back((uint8_t)sl);
    for (size_t i = 0; i < sl; i++) b.push_back((uint8_t)rnd());
  }
  b.insert(b.end(), ys.begin(), ys.end());
  return b;
}
static void test_wire_v2() {
  const uint32_t k = (uint32_t)(rnd() % 17);
  std::vector<std::vector<uint8_t>> proofs;
  const int count = 1 + (int)(rnd() % 4);
  for (int i = 0; i < count; i++) proofs.push_back(random_v2(k, rnd() % 2 == 0));
  const int victim = (int)(rnd() % count), kind = (int)(rnd() % 8);
  std::vector<uint8_t> &v = proofs[victim];
  const size_t body = 6 + 32 * (size_t)(5 + k) + 33 * (size_t)(6 + 2 * k);
  if (kind == 1) v[rnd() % v.size()] ^= (uint8_t)(1u << (rnd() % 8));
  else if (kind == 2) v.resize(rnd() % (v.size() + 1));
  else if (kind == 3) { const size_t extra = 1 + rnd() % 9; for (size_t i = 0; i < extra; i++) v.push_back((uint8_t)rnd()); }
  else if (kind == 4) { v[body + 128] = (uint8_t)rnd(); v[body + 129] = (uint8_t)rnd(); }                       // a seed length that lies
  else if (kind == 5) { for (int i = 0; i < 32; i++) v[body + 32 * (rnd() % 4) + i] = 0xFF; }                    // a challenge >= q
  else if (kind == 6) v[5] = (uint8_t)rnd();
  else if (kind == 7 && v[4] == '3') {                                                                           // a y that is not its point's
    const size_t npt = 6 + 2 * (size_t)k, at = v.size() - 32 * npt, t = rnd() % npt;
    const int how = (int)(rnd() % 3);
    if (how == 0) v[at + 32 * t + rnd() % 32] ^= (uint8_t)(1u << (rnd() % 8));
    else if (how == 1) for (int i = 0; i < 32; i++) v[at + 32 * t + i] = 0xFF;
    else v[6 + 32 * (5 + (size_t)k) + 33 * t] ^= 1;                                                              // the tag: the other root (or 0 <-> 1: no encoding)
    std::vector<uint8_t> dummy;
    CHECK(!rpw::expand_v2(v.data(), v.size(), dummy));
  }
  // one exact-size heap block for the batch, offsets
  size_t total = 0;
  std::vector<uint64_t> off = {0};
  for (auto &p : proofs) { total += p.size(); off.push_back(total); }
  Buf in(total);
  { size_t o = 0; for (auto &p : proofs) { if (!p.empty()) memcpy(in.p + o, p.data(), p.size()); o += p.size(); } }
  std::vector<uint64_t> out_off(count + 1, 777);
  int64_t bad = 5;
  Buf big(4 * total + 4096);
  const int rc = bpmi_rp_wire_v2_to_v1(in.p, total, off.data(), (uint64_t)count, big.p, big.n, out_off.data(), &bad);
  CHECK(rc == BPMI_OK);
  std::vector<uint8_t> one;
  bool all = true;
  for (int i = 0; i < count; i++) {
    const bool ok = rpw::expand_v2(in.p + off[i], (size_t)(off[i + 1] - off[i]), one);
    if (!ok && all) { CHECK(bad == i); all = false; }
    if (ok && all) { CHECK(out_off[i + 1] - out_off[i] == one.size() && !memcmp(big.p + out_off[i], one.data(), one.size())); CHECK(one[4] == '1' && one.size() > body); }
  }
  if (all) {
    CHECK(bad == -1);
    const size_t need = (size_t)out_off[count];
    Buf exact(need), shorter(need - 1);
    CHECK(bpmi_rp_wire_v2_to_v1(in.p, total, off.data(), (uint64_t)count, exact.p, need, out_off.data(), &bad) == BPMI_OK && bad == -1 && !memcmp(exact.p, big.p, need));
    CHECK(bpmi_rp_wire_v2_to_v1(in.p, total, off.data(), (uint64_t)count, shorter.p, need - 1, out_off.data(), &bad) == BPMI_E_ARG);
  }
  if (kind == 0) CHECK(all);
  // an offset table that leaves the buffer is refused before anything is read
  off[count] = total + 1;
  CHECK(bpmi_rp_wire_v2_to_v1(in.p, total, off.data(), (uint64_t)count, big.p, big.n, out_off.data(), &bad) == BPMI_E_ARG);
}

// host_pool.hpp: several application threads run data-parallel loops at the same time (one gets the pool, the others fall back
// to threads of their own), a loop body that starts a loop of its own (nested: the pool is busy, so it must not wait for it),
// widths from 1 to more than the pool has had so far -- every index exactly once, nothing lost, nothing run twice
static void test_host_pool() {
  std::atomic<int> bad{0};
  auto caller = [&bad](unsigned seed) {
    for (int rep = 0; rep < 24; rep++) {
      const int n = 1 + (int)((seed * 2654435761u + (unsigned)rep * 40503u) % 13u);
      std::vector<std::atomic<int>> hit(n);
      for (auto &h : hit) h = 0;
      hostpool::run(n, [&](int t) {
        hit[t]++;
        if (rep % 6 == 0 && t == 1) {                             // nested
          std::atomic<int> inner{0};
          hostpool::run(3, [&](int) { inner++; });
          if (inner != 3) bad++;
        }
      });
      for (auto &h : hit) if (h != 1) bad++;
    }
  };
  std::vector<std::thread> callers;
  for (unsigned c = 0; c < 4; c++) callers.emplace_back(caller, c + 1u);
  for (auto &x : callers) x.join();
  CHECK(bad == 0);
}

int main(int argc, char **argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 200;
  if (argc > 2) rng_state ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ULL;
  for (long it = 0; it < iters; it++) {
    test_algebra();
    test_transcript();
    test_mod_hash_range();
    test_host_tail();
    if (it % 8 == 0) test_host_pool();
    for (int r = 0; r < 8; r++) test_wire_v2();
  }
  printf("host_native_fuzz: %ld iterations, %d failed checks\n", iters, fails);
  return fails ? 1 : 0;
}
