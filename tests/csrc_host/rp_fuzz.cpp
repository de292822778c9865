// Sanitizer harness for python-bulletproofs_amd/csrc/rp_batch_host.hpp (the parser of untrusted proof
// bytes): built with -fsanitize=address,undefined by tests/test_batch_native_cpu.py, fed valid wire
// proofs plus random corruptions (bit flips, truncations, extensions, length-field edits, garbage).
// Exit code 0 = no sanitizer report; prints how many inputs were accepted.
#include <stdio.h>
#include <stdlib.h>

#include "rp_batch_host.hpp"

static uint64_t rng_state = 88172645463325252ULL;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  const uint32_t n_gens = (uint32_t)atoi(argv[2]);
  const long iters = atol(argv[3]);
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 2;
  std::vector<uint8_t> file;
  uint8_t buf[4096];
  size_t got;
  while ((got = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + got);
  fclose(f);
  // file = u32 count, then count x (u32 length, bytes)
  std::vector<std::vector<uint8_t>> proofs;
  size_t o = 4;
  uint32_t count;
  memcpy(&count, file.data(), 4);
  for (uint32_t i = 0; i < count; i++) { uint32_t l; memcpy(&l, file.data() + o, 4); o += 4; proofs.emplace_back(file.begin() + o, file.begin() + o + l); o += l; }
  uint32_t k = 0;
  while ((1u << k) < n_gens) k++;
  long accepted = 0;
  for (long it = 0; it < iters; it++) {
    std::vector<uint8_t> b = proofs[rnd() % proofs.size()];
    const int kind = (int)(rnd() % 8);
    if (kind == 1) b[rnd() % b.size()] ^= (uint8_t)(1u << (rnd() % 8));
    else if (kind == 2) b.resize(rnd() % (b.size() + 1));
    else if (kind == 3) { const size_t extra = rnd() % 64; for (size_t i = 0; i < extra; i++) b.push_back((uint8_t)rnd()); }
    else if (kind == 4 && b.size() > 8) { const size_t p = rnd() % (b.size() - 4); const uint32_t v = (uint32_t)rnd() >> (rnd() % 32); memcpy(&b[p], &v, 4); }
    else if (kind == 5) { for (auto &x : b) if (rnd() % 50 == 0) x = (uint8_t)rnd(); }
    else if (kind == 6) { b[5] = (uint8_t)rnd(); }
    else if (kind == 7) { const size_t p = rnd() % b.size(); b[p] = '&'; }
    std::vector<uint64_t> off = {0, (uint64_t)b.size()}, pt_off = {0, 6 + 2 * (uint64_t)k};
    std::vector<uint8_t> w(128, 7), vs(32), ps(32 * (6 + 2 * k)), comp(33 * (6 + 2 * k));
    std::vector<rp::Sq> acc(5 + 2 * n_gens, rp::q_small(0));
    uint64_t bad = 0;
    if (rp::run_chunk(n_gens, k, 1, b.data(), off.data(), w.data(), 0, 1, pt_off.data(), vs.data(), ps.data(), comp.data(), acc.data(), &bad)) accepted++;
  }
  printf("accepted %ld of %ld\n", accepted, iters);
  return 0;
}
