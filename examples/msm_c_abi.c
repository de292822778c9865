/* Stand-alone use of libbpmi through its C-ABI only (no Python, no torch, no HIP headers):
 *
 *   gcc -O2 -std=c99 -Iinclude examples/msm_c_abi.c -o msm_c_abi \
 *       python-bulletproofs_amd/libbpmi.so -Wl,-rpath,$PWD/python-bulletproofs_amd
 *   ./msm_c_abi 20            # log2 n
 *
 * Builds n points k_i * G on the GPU (bpmi_ec_mul_batch_dev), keeps them and n scalars resident in
 * device memory obtained from the library (bpmi_malloc / bpmi_upload), times bpmi_msm_dev and the
 * asynchronous pair (bpmi_msm_dev_enqueue / bpmi_msm_finish, two MSMs in flight on two lanes), and
 * checks every result against the known answer (sum e_i k_i mod q) * G computed with a 1-term MSM.
 * Scalars come from a small xorshift generator; q-reduction is done by clearing the top bit. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bpmi.h"

static uint64_t s = 88172645463325252ULL;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static const uint8_t G_LE[64] = {
    0x98, 0x17, 0xF8, 0x16, 0x5B, 0x81, 0xF2, 0x59, 0xD9, 0x28, 0xCE, 0x2D, 0xDB, 0xFC, 0x9B, 0x02,
    0x07, 0x0B, 0x87, 0xCE, 0x95, 0x62, 0xA0, 0x55, 0xAC, 0xBB, 0xDC, 0xF9, 0x7E, 0x66, 0xBE, 0x79,
    0xB8, 0xD4, 0x10, 0xFB, 0x8F, 0xD0, 0x47, 0x9C, 0x19, 0x54, 0x85, 0xA6, 0x48, 0xB4, 0x17, 0xFD,
    0xA8, 0x08, 0x11, 0x0E, 0xFC, 0xFB, 0xA4, 0x5D, 0x65, 0xC4, 0xA3, 0x26, 0x77, 0xDA, 0x3A, 0x48};

#define CK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, bpmi_last_error(ctx)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int logn = argc > 1 ? atoi(argv[1]) : 16;
  const uint64_t n = 1ull << logn;
  bpmi_ctx *ctx = bpmi_ctx_create(0, NULL);
  if (!ctx) { fprintf(stderr, "no GPU: %s\n", bpmi_last_error(NULL)); return 2; }
  uint8_t *ks = malloc(32 * n), *es = malloc(32 * n), *gs = malloc(64 * n);
  for (uint64_t i = 0; i < n; i++) {
    for (int w = 0; w < 4; w++) { uint64_t a = rnd(), b = rnd(); memcpy(ks + 32 * i + 8 * w, &a, 8); memcpy(es + 32 * i + 8 * w, &b, 8); }
    ks[32 * i + 31] &= 0x3F; es[32 * i + 31] &= 0x3F;        /* < 2^254 < q */
    memcpy(gs + 64 * i, G_LE, 64);
  }
  void *d_k, *d_g, *d_p, *d_e;
  CK(bpmi_malloc(ctx, 32 * n, &d_k)); CK(bpmi_malloc(ctx, 64 * n, &d_g)); CK(bpmi_malloc(ctx, 64 * n, &d_p)); CK(bpmi_malloc(ctx, 32 * n, &d_e));
  CK(bpmi_upload(ctx, d_k, ks, 32 * n)); CK(bpmi_upload(ctx, d_g, gs, 64 * n)); CK(bpmi_upload(ctx, d_e, es, 32 * n));
  CK(bpmi_ec_mul_batch_dev(ctx, d_g, d_k, n, d_p));          /* P_i = k_i G */
  CK(bpmi_sync(ctx));
  uint8_t out[64], dot[32], want[64];
  for (int i = 0; i < 3; i++) CK(bpmi_msm_dev(ctx, d_p, d_e, n, out));
  const int reps = 20;
  const double t0 = now();
  for (int i = 0; i < reps; i++) CK(bpmi_msm_dev(ctx, d_p, d_e, n, out));
  const double dt = (now() - t0) / reps;
  CK(bpmi_sc_dot_dev(ctx, d_k, d_e, n, dot));                /* sum e_i k_i mod q */
  CK(bpmi_msm(ctx, G_LE, dot, 1, want));
  printf("n=2^%d  %.3f ms per MSM  %.3e pairs/s  known-answer %s\n", logn, dt * 1e3, n / dt, memcmp(out, want, 64) ? "MISMATCH" : "ok");
  /* the same MSM with two in flight: MSM j + 1 is queued before MSM j is finished (n <= 2^23 per call) */
  int bad = memcmp(out, want, 64) != 0;
  if (n <= (1ull << 23)) {
    uint8_t res[64];
    CK(bpmi_set_option(ctx, "async_lanes", 1));
    CK(bpmi_msm_dev_enqueue(ctx, 0, d_p, d_e, n));
    const double t1 = now();
    for (int j = 0; j < reps; j++) {
      if (j + 1 < reps) CK(bpmi_msm_dev_enqueue(ctx, (j + 1) & 1, d_p, d_e, n));
      CK(bpmi_msm_finish(ctx, j & 1, res));
      bad |= memcmp(res, want, 64) != 0;
    }
    const double dp = (now() - t1) / reps;
    printf("n=2^%d  %.3f ms per MSM  %.3e pairs/s  two in flight, known-answer %s\n", logn, dp * 1e3, n / dp, bad ? "MISMATCH" : "ok");
  }
  bpmi_free(ctx, d_k); bpmi_free(ctx, d_g); bpmi_free(ctx, d_p); bpmi_free(ctx, d_e);
  bpmi_ctx_destroy(ctx);
  free(ks); free(es); free(gs);
  return bad;
}
