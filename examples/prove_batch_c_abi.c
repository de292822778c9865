/* Many range proofs proved in ONE call and verified as ONE batch, through the C-ABI alone (no Python, no torch, no HIP headers):
 *
 *   gcc -O2 -std=c99 -Iinclude examples/prove_batch_c_abi.c -o prove_batch_c_abi \
 *       python-bulletproofs_amd/libbpmi.so -Wl,-rpath,$PWD/python-bulletproofs_amd
 *   ./prove_batch_c_abi [n_proofs [bits [spoil [values_per_proof]]]]
 *
 * What a prover service does, and what it replaces: a loop of NIRangeProver(v, bits, g, h, gs, hs, gamma, u, group, seed).prove()
 * (/root/reference/src/rangeproofs/rangeproof_prover.py:35-91).
 *   1. bpmi_rp_prover_create: fixed-base tables of the deployment's generators, once;
 *   2. bpmi_rp_prove_batch: values, blinding factors and transcript seeds in, wire-format-2 proofs out -- one device call;
 *   3. (the other side) the commitments V_i = v_i g + gamma_i h and bpmi_rp_batch_verify_dev over the same bytes.
 * The generators here are multiples of the curve's base point (a demonstration, not a setup ceremony).  spoil = 1 changes one
 * commitment after proving: the batch must be rejected.  values_per_proof > 1 (round 6): AGGREGATED proofs -- a loop of
 * AggregNIRangeProver (/root/reference/src/rangeproofs/rangeproof_aggreg_prover.py:36-146) over proofs of that many values of `bits`
 * bits each (bits x values <= 128), bpmi_rp_prover_create_aggregated.  Exit code 0: proved and verified; 1: the batch did not verify; 2: error. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bpmi.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd(void) { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static void small_scalar(uint8_t s[32], int bytes) { int i; memset(s, 0, 32); for (i = 0; i < bytes; i++) s[i] = (uint8_t)rnd(); }

/* the base point of secp256k1, x || y, little-endian coordinates */
static const uint8_t G_LE[64] = {
  0x98, 0x17, 0xF8, 0x16, 0x5B, 0x81, 0xF2, 0x59, 0xD9, 0x28, 0xCE, 0x2D, 0xDB, 0xFC, 0x9B, 0x02, 0x07, 0x0B, 0x87, 0xCE, 0x95, 0x62, 0xA0, 0x55,
  0xAC, 0xBB, 0xDC, 0xF9, 0x7E, 0x66, 0xBE, 0x79, 0xB8, 0xD4, 0x10, 0xFB, 0x8F, 0xD0, 0x47, 0x9C, 0x19, 0x54, 0x85, 0xA6, 0x48, 0xB4, 0x17, 0xFD,
  0xA8, 0x08, 0x11, 0x0E, 0xFC, 0xFB, 0xA4, 0x5D, 0x65, 0xC4, 0xA3, 0x26, 0x77, 0xDA, 0x3A, 0x48};

#define CK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bpmi_last_error(ctx)); return 2; } } while (0)

int main(int argc, char **argv) {
  const uint64_t P = argc > 1 ? strtoull(argv[1], NULL, 10) : 1024;
  const uint32_t bits = argc > 2 ? (uint32_t)atoi(argv[2]) : 64;
  const int spoil = argc > 3 ? atoi(argv[3]) : 0;
  const uint32_t m = argc > 4 ? (uint32_t)atoi(argv[4]) : 1;
  const uint32_t nel = bits * m;                       /* elements of a proof's vectors */
  uint32_t k = 0;
  uint64_t i;
  while ((1u << k) < nel) k++;
  bpmi_ctx *ctx = bpmi_ctx_create(0, NULL);
  if (!ctx) { fprintf(stderr, "bpmi_ctx_create: %s\n", bpmi_last_error(NULL)); return 2; }
  /* generators: 3 + 2 bits m multiples of G */
  const uint64_t ng = 3 + 2ull * nel;
  uint8_t *gpts = malloc(64 * ng), *gin = malloc(64 * ng), *gsc = malloc(32 * ng);
  for (i = 0; i < ng; i++) { memcpy(gin + 64 * i, G_LE, 64); small_scalar(gsc + 32 * i, 31); }
  CK(bpmi_ec_mul_batch(ctx, gin, gsc, ng, gpts));
  const uint8_t *g = gpts, *h = gpts + 64, *u = gpts + 128, *gs = gpts + 192, *hs = gpts + 192 + 64 * (uint64_t)nel;
  double t0 = now();
  bpmi_rp_prover *pv = NULL;
  CK(bpmi_rp_prover_create_aggregated(ctx, bits, m, g, h, u, gs, hs, &pv));
  printf("prover for proofs of %u x %u bits: tables built in %.1f ms\n", m, bits, (now() - t0) * 1e3);
  /* inputs: values below 2^bits, blinding factors, seeds "proof-<i>" */
  const uint64_t NV = P * m;                           /* values: proof i owns entries i m .. i m + m - 1 */
  uint8_t *vals = calloc(NV, 32), *gams = malloc(32 * NV), *seeds = malloc(24 * P);
  uint64_t *soff = malloc(8 * (P + 1)), *ooff = malloc(8 * (P + 1));
  uint64_t spos = 0;
  for (i = 0; i < NV; i++) {
    const uint64_t v = bits >= 64 ? rnd() : (rnd() & ((1ull << bits) - 1));
    memcpy(vals + 32 * i, &v, 8);                      /* (little-endian host) */
    small_scalar(gams + 32 * i, 31);
  }
  for (i = 0; i < P; i++) {
    soff[i] = spos;
    spos += (uint64_t)sprintf((char *)seeds + spos, "proof-%llu", (unsigned long long)i);
  }
  soff[P] = spos;
  const uint64_t cap = P * bpmi_rp_prove_batch_proof_bytes(pv, 24);
  uint8_t *wire = NULL;
  CK(bpmi_host_alloc(ctx, cap, (void **)&wire));       /* page-locked: the verifier's upload runs at link speed */
  CK(bpmi_rp_prove_batch(pv, P, vals, gams, seeds, soff, wire, cap, ooff));      /* warm */
  t0 = now();
  CK(bpmi_rp_prove_batch(pv, P, vals, gams, seeds, soff, wire, cap, ooff));
  const double dt = now() - t0;
  double ms[7];
  CK(bpmi_rp_prover_last_ms(pv, ms));
  printf("%llu proofs in %.2f ms (%.0f proofs/s; device %.2f ms), %llu wire bytes\n", (unsigned long long)P, dt * 1e3, P / dt, ms[6],
         (unsigned long long)ooff[P]);
  /* the verifier's side: V_i = v_i g + gamma_i h, then one batch verification over the same bytes */
  uint8_t *rep = malloc(64 * NV), *vg = malloc(64 * NV), *rh = malloc(64 * NV), *V = malloc(64 * NV), one[32] = {1};
  for (i = 0; i < NV; i++) memcpy(rep + 64 * i, g, 64);
  CK(bpmi_ec_mul_batch(ctx, rep, vals, NV, vg));
  for (i = 0; i < NV; i++) memcpy(rep + 64 * i, h, 64);
  CK(bpmi_ec_mul_batch(ctx, rep, gams, NV, rh));
  CK(bpmi_ec_lincomb2_batch(ctx, vg, rh, one, one, NV, V));
  if (spoil && NV > 1) memcpy(V + 64 * (NV / 2), V, 64);
  void *d_gens = NULL, *d_pts = NULL, *d_sc = NULL;
  const uint64_t pairs = P * (m + 6 + 2ull * k);
  CK(bpmi_malloc(ctx, 64 * ng, &d_gens));
  CK(bpmi_malloc(ctx, 64 * pairs, &d_pts));
  CK(bpmi_malloc(ctx, 32 * pairs, &d_sc));
  CK(bpmi_upload(ctx, d_gens, gpts, 64 * ng));
  uint8_t seed[32], out[64], zero[64] = {0};
  for (i = 0; i < 32; i++) seed[i] = (uint8_t)rnd();   /* (a real verifier draws this from the system's CSPRNG) */
  int64_t bad = -1;
  t0 = now();
  CK(bpmi_rp_batch_verify_dev(ctx, nel, m, P, wire, ooff[P], ooff, NULL, seed, V, d_gens, d_pts, d_sc, out, &bad));
  const int valid = bad < 0 && !memcmp(out, zero, 64);
  printf("batch verification: %s in %.2f ms\n", valid ? "VALID" : "INVALID", (now() - t0) * 1e3);
  bpmi_rp_prover_destroy(pv);
  bpmi_free(ctx, d_gens); bpmi_free(ctx, d_pts); bpmi_free(ctx, d_sc);
  bpmi_host_free(ctx, wire);
  bpmi_ctx_destroy(ctx);
  return valid ? 0 : 1;
}
