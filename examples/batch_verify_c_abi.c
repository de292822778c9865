/* Batch verification of range proofs through the C-ABI alone (no Python, no torch, no HIP headers):
 *
 *   gcc -O2 -std=c99 -Iinclude examples/batch_verify_c_abi.c -o batch_verify_c_abi \
 *       python-bulletproofs_amd/libbpmi.so -Wl,-rpath,$PWD/python-bulletproofs_amd
 *   ./batch_verify_c_abi batch.bin [repeat]
 *
 * batch.bin (little-endian; tests/test_gpu_abi_errors.py writes one from proofs made by the Python prover):
 *   u32 n_gens | u32 values_per_proof m | u32 n_proofs | u64 blobs_len
 *   g, h, u, gs[n_gens], hs[n_gens]            64-byte points (x || y, 32-byte little-endian coordinates)
 *   V[n_proofs * m]                            the commitments, same format
 *   u64 offsets[n_proofs + 1]                  positions of the wire proofs (rangeproofs/codec.py) inside blobs
 *   blobs[blobs_len]
 *
 * What a verifier service does per batch, and what this program does `repeat` times:
 *   1. the wire proofs sit in a page-locked receive buffer (bpmi_host_alloc);
 *   2. bpmi_rp_batch_prepare_dev: one upload; parsing, the byte-level transcript checks of the three verifiers and the
 *      weighted scalars of every proof on the GPU; the proofs' points decoded into the MSM's point array; back come
 *      5 + 2 n_gens shared coefficients and the index of the first bad proof (-1: none);
 *   3. the shared coefficients become the scalars of g, h, u, gs_i, hs_i (two constants are added to every gs_i / hs_i);
 *   4. ONE MSM over [g h u gs hs | V .. | proof points ..] (bpmi_msm_segs_dev) -- the identity iff every proof is valid.
 *   (and steps 2-4 again as one call, bpmi_rp_batch_verify_dev, which must give the same verdict)
 * Exit code 0: the batch verifies; 1: it does not; 2: usage / I/O / library error. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bpmi.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* r = a + b mod q on 32-byte little-endian values in [0, q) */
static const uint8_t Q_LE[32] = {0x41, 0x41, 0x36, 0xD0, 0x8C, 0x5E, 0xD2, 0xBF, 0x3B, 0xA0, 0x48, 0xAF, 0xE6, 0xDC, 0xAE, 0xBA,
                                 0xFE, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
static void add_mod_q(uint8_t r[32], const uint8_t a[32], const uint8_t b[32]) {
  uint8_t s[33], d[32];
  unsigned c = 0;
  int i;
  for (i = 0; i < 32; i++) { c += (unsigned)a[i] + b[i]; s[i] = (uint8_t)c; c >>= 8; }
  s[32] = (uint8_t)c;
  int br = 0;
  for (i = 0; i < 32; i++) { int t = (int)s[i] - Q_LE[i] - br; d[i] = (uint8_t)t; br = t < 0; }
  memcpy(r, (s[32] || !br) ? d : s, 32);
}

#define CK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bpmi_last_error(ctx)); return 2; } } while (0)

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s batch.bin [repeat]\n", argv[0]); return 2; }
  const int repeat = argc > 2 ? atoi(argv[2]) : 1;
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  uint32_t hdr[3];
  uint64_t blobs_len;
  if (fread(hdr, 4, 3, f) != 3 || fread(&blobs_len, 8, 1, f) != 1) { fprintf(stderr, "short header\n"); return 2; }
  const uint32_t n = hdr[0], m = hdr[1], P = hdr[2];
  uint32_t k = 0;
  while ((1u << k) < n) k++;
  const uint64_t n_shared = 3 + 2ull * n, nv = (uint64_t)P * m, npts = (uint64_t)P * (6 + 2 * k);
  uint8_t *shared_pts = malloc(64 * n_shared), *vpts = malloc(64 * nv);
  uint64_t *off = malloc(8 * ((size_t)P + 1));
  if (fread(shared_pts, 64, n_shared, f) != n_shared || fread(vpts, 64, nv, f) != nv || fread(off, 8, (size_t)P + 1, f) != (size_t)P + 1) {
    fprintf(stderr, "short file\n");
    return 2;
  }
  bpmi_ctx *ctx = bpmi_ctx_create(0, NULL);
  if (!ctx) { fprintf(stderr, "no GPU: %s\n", bpmi_last_error(NULL)); return 2; }
  void *recv = NULL;                                       /* the receive buffer: page-locked, handed to the library as it is */
  CK(bpmi_host_alloc(ctx, blobs_len ? blobs_len : 1, &recv));
  if (fread(recv, 1, blobs_len, f) != blobs_len) { fprintf(stderr, "short blobs\n"); return 2; }
  fclose(f);

  /* device arrays of the MSM: segment 0 = the shared generators, segment 1 = commitments followed by the proofs' points */
  void *d_shared_pts, *d_shared_sc, *d_pts, *d_sc;
  CK(bpmi_malloc(ctx, 64 * n_shared, &d_shared_pts));
  CK(bpmi_malloc(ctx, 32 * n_shared, &d_shared_sc));
  CK(bpmi_malloc(ctx, 64 * (nv + npts), &d_pts));
  CK(bpmi_malloc(ctx, 32 * (nv + npts), &d_sc));
  CK(bpmi_upload(ctx, d_shared_pts, shared_pts, 64 * n_shared));
  CK(bpmi_upload(ctx, d_pts, vpts, 64 * nv));

  uint8_t *coef = malloc(32 * (5 + 2ull * n)), *sc = malloc(32 * n_shared), seed[32], out[64];
  int verdict = 1;
  for (int rep = 0; rep < repeat; rep++) {
    FILE *ur = fopen("/dev/urandom", "rb");                /* fresh weights per batch: nobody who made the proofs may know them */
    if (!ur || fread(seed, 1, 32, ur) != 32) { fprintf(stderr, "no randomness\n"); return 2; }
    fclose(ur);
    const double t0 = now();
    int64_t first_bad = -1;
    CK(bpmi_rp_batch_prepare_dev(ctx, n, m, P, recv, blobs_len, off, NULL, seed, d_sc, (char *)d_sc + 32 * nv, (char *)d_pts + 64 * nv, coef,
                                 &first_bad));
    if (first_bad >= 0) {
      printf("proof %lld is invalid (parsing, transcript check or point encoding)\n", (long long)first_bad);
      verdict = 1;
      break;
    }
    /* coef: c_g c_h c_u | constant of every gs_i | constant of every hs_i | c_gs[n] | c_hs[n] */
    memcpy(sc, coef, 96);
    for (uint32_t i = 0; i < n; i++) {
      add_mod_q(sc + 32 * (3 + i), coef + 32 * (5 + i), coef + 32 * 3);
      add_mod_q(sc + 32 * (3 + n + i), coef + 32 * (5 + n + i), coef + 32 * 4);
    }
    CK(bpmi_upload(ctx, d_shared_sc, sc, 32 * n_shared));
    const void *pts[2] = {d_shared_pts, d_pts}, *scs[2] = {d_shared_sc, d_sc};
    const uint64_t cnt[2] = {n_shared, nv + npts};
    CK(bpmi_msm_segs_dev(ctx, 2, pts, scs, cnt, out));
    int zero = 1;
    for (int i = 0; i < 64; i++) zero &= out[i] == 0;
    verdict = zero ? 0 : 1;
    printf("batch of %u proofs (%llu MSM pairs): %s in %.3f ms\n", P, (unsigned long long)(n_shared + nv + npts), zero ? "VALID" : "INVALID",
           (now() - t0) * 1e3);
    /* The same in ONE call (bpmi_rp_batch_verify_dev): the upload in slices with the point decoding beside it, the preparation, the
     * shared coefficients folded on the device, the MSM -- nothing comes back to the host in between. */
    const double t1 = now();
    uint8_t out1[64];
    int64_t bad1 = -1;
    CK(bpmi_rp_batch_verify_dev(ctx, n, m, P, recv, blobs_len, off, NULL, seed, vpts, d_shared_pts, d_pts, d_sc, out1, &bad1));
    int zero1 = bad1 < 0;
    for (int i = 0; i < 64 && zero1; i++) zero1 &= out1[i] == 0;
    printf("one call: %s in %.3f ms\n", zero1 ? "VALID" : "INVALID", (now() - t1) * 1e3);
    if (zero1 != zero) { fprintf(stderr, "the two paths disagree\n"); return 3; }
  }
  bpmi_free(ctx, d_shared_pts); bpmi_free(ctx, d_shared_sc); bpmi_free(ctx, d_pts); bpmi_free(ctx, d_sc);
  bpmi_host_free(ctx, recv);
  bpmi_ctx_destroy(ctx);
  free(shared_pts); free(vpts); free(off); free(coef); free(sc);
  return verdict;
}
