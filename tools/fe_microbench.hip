// Microbenchmark: which 256-bit modmul formulation is fastest on gfx950?
// Standalone (hipcc --offload-arch=gfx950 -O3 tools/fe_microbench.hip -o fe_microbench).
// Validates every variant against a host __int128 reference, then times it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../python-bulletproofs_amd/csrc/field.hpp"      // the product's own fe_mul / fe_sqr (variants 8, 9)
typedef uint32_t u32; typedef uint64_t u64; typedef unsigned __int128 u128;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct fe8 { u32 v[8]; };

// ---------------------------------------------------------------- V0: 8x32 operand scanning (compiler)
__device__ __forceinline__ void mulwide0(u32 t[16], const u32 a[8], const u32 b[8]) {
#pragma unroll
  for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (u64)a[i] * b[j] + t[i + j]; t[i + j] = (u32)c; c >>= 32; }
    t[i + 8] = (u32)c;
  }
}
// ---------------------------------------------------------------- V4: 8x32 product scanning, asm mad + carry
__device__ __forceinline__ void mac4(u64 &acc, u32 &acc2, u32 a, u32 b) {
  asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(acc2) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mulwide4(u32 t[16], const u32 a[8], const u32 b[8]) {
  u64 acc = 0; u32 acc2 = 0;
#pragma unroll
  for (int k = 0; k < 15; k++) {
#pragma unroll
    for (int i = 0; i < 8; i++) { int j = k - i; if (j < 0 || j > 7) continue; mac4(acc, acc2, a[i], b[j]); }
    t[k] = (u32)acc; acc = (acc >> 32) | ((u64)acc2 << 32); acc2 = 0;
  }
  t[15] = (u32)acc;
}
// V6: like V4 but two interleaved column accumulators so the nop slots are filled with useful work
__device__ __forceinline__ void mac6(u64 &accA, u32 &accA2, u32 a0, u32 b0, u64 &accB, u32 &accB2, u32 a1, u32 b1) {
  asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
               "v_mad_u64_u32 %2, s[2:3], %6, %7, %2\n\t"
               "s_nop 0\n\t"
               "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
               "v_addc_co_u32 %3, s[2:3], 0, %3, s[2:3]"
               : "+v"(accA), "+v"(accA2), "+v"(accB), "+v"(accB2) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc", "s2", "s3");
}
__device__ __forceinline__ void reduce8(u32 r[8], const u32 t[16]) {
  u32 s[8]; u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    c += (u64)t[8 + i] * 977u + t[i];
    if (i > 0) c += t[8 + i - 1];
    s[i] = (u32)c; c >>= 32;
  }
  c += t[15];
  u64 top = c;                              // < 2^34
  u64 lo = (top & 0xffffffffu) * 977u;
  u64 hi = (top >> 32) * 977u;
  u64 d = (u64)s[0] + (u32)lo; r[0] = (u32)d; d >>= 32;
  d += (u64)s[1] + (lo >> 32) + (u32)hi + (u32)top; r[1] = (u32)d; d >>= 32;
  d += (u64)s[2] + (hi >> 32) + (top >> 32); r[2] = (u32)d; d >>= 32;
#pragma unroll
  for (int i = 3; i < 8; i++) { d += s[i]; r[i] = (u32)d; d >>= 32; }
  u64 e = (u64)r[0] + d * 977u; r[0] = (u32)e; e >>= 32;
  e += (u64)r[1] + d; r[1] = (u32)e; e >>= 32;
#pragma unroll
  for (int i = 2; i < 8; i++) { e += r[i]; r[i] = (u32)e; e >>= 32; }
}

// ---------------------------------------------------------------- V3: 9x29-bit limbs, carry-free Comba
struct fe9 { u32 v[9]; };
#define M29 0x1FFFFFFFu
__device__ __forceinline__ void to9(fe9 &r, const fe8 &a) {
  // limb k = bits [29k, 29k+29)
#pragma unroll
  for (int k = 0; k < 9; k++) {
    int bit = 29 * k, w = bit >> 5, s = bit & 31;
    u64 x = a.v[w];
    if (w + 1 < 8) x |= (u64)a.v[w + 1] << 32;
    r.v[k] = (u32)(x >> s) & M29;
  }
}
__device__ __forceinline__ void norm9(fe9 &a);
__device__ __forceinline__ void from9(fe8 &r, fe9 a) {
  // a must be fully carried (limbs < 2^29, value < 2^256)
#pragma unroll
  for (int w = 0; w < 8; w++) {
    int bit = 32 * w, k = bit / 29, s = bit - 29 * k;
    u64 x = (u64)a.v[k] >> s;
    x |= (u64)a.v[k + 1] << (29 - s);
    if (k + 2 < 9) x |= (u64)a.v[k + 2] << (58 - s);
    r.v[w] = (u32)x;
  }
}
__device__ __forceinline__ void fold9(u32 r[9], const u64 acc[17]) {
  u32 t[18]; u64 c = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) { c += acc[k]; t[k] = (u32)c & M29; c >>= 29; }
  t[17] = (u32)c;
  u64 u[10];
#pragma unroll
  for (int k = 0; k < 9; k++) u[k] = (u64)t[9 + k] * 31264u + t[k];
#pragma unroll
  for (int k = 1; k < 10; k++) { u64 sh = (u64)t[8 + k] << 8; if (k < 9) u[k] += sh; else u[9] = sh; }
  u64 H = (u[8] >> 24) + (u[9] << 5);
  u[8] &= 0xFFFFFFu;
  u[0] += H * 977u;
  u[1] += H << 3;
  c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) { c += u[k]; r[k] = (u32)c & M29; c >>= 29; }
  r[8] = (u32)(c + u[8]);
}
__device__ __forceinline__ void mul9(fe9 &r, const fe9 &a, const fe9 &b) {
  u64 acc[17];
#pragma unroll
  for (int k = 0; k < 17; k++) {
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { int j = k - i; if (j < 0 || j > 8) continue; s += (u64)a.v[i] * b.v[j]; }
    acc[k] = s;
  }
  fold9(r.v, acc);
}
__device__ __forceinline__ void sqr9(fe9 &r, const fe9 &a) {
  u32 d[9];
#pragma unroll
  for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
  u64 acc[17];
#pragma unroll
  for (int k = 0; k < 17; k++) {
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { int j = k - i; if (j < 0 || j > 8 || i > j) continue; s += (i == j) ? (u64)a.v[i] * a.v[j] : (u64)d[i] * a.v[j]; }
    acc[k] = s;
  }
  fold9(r.v, acc);
}
// full canonical reduction to [0, p) for output
__device__ __forceinline__ void canon9(fe9 &a) {
  // two rounds of top fold + carry, then conditional subtract p
#pragma unroll
  for (int round = 0; round < 2; round++) {
    u32 h = a.v[8] >> 24; a.v[8] &= 0xFFFFFFu;
    u64 c = (u64)a.v[0] + (u64)h * 977u; a.v[0] = (u32)c & M29; c >>= 29;
    c += (u64)a.v[1] + ((u64)h << 3); a.v[1] = (u32)c & M29; c >>= 29;
#pragma unroll
    for (int k = 2; k < 8; k++) { c += a.v[k]; a.v[k] = (u32)c & M29; c >>= 29; }
    a.v[8] += (u32)c;
  }
  // now value < 2^256 (+tiny impossible); subtract p if >= p:  a >= p  <=>  a + (2^32+977) >= 2^256
  u32 t[9]; u64 c = (u64)a.v[0] + 977u; t[0] = (u32)c & M29; c >>= 29;
  c += (u64)a.v[1] + 8u; t[1] = (u32)c & M29; c >>= 29;
#pragma unroll
  for (int k = 2; k < 8; k++) { c += a.v[k]; t[k] = (u32)c & M29; c >>= 29; }
  c += a.v[8]; t[8] = (u32)c;
  if (t[8] >> 24) {
#pragma unroll
    for (int k = 0; k < 8; k++) a.v[k] = t[k];
    a.v[8] = t[8] & 0xFFFFFFu;
  }
}

template <int V>
__global__ void __launch_bounds__(256) kmul(fe8* out, const fe8* a, const fe8* b, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  fe8 x = a[i], y = b[i];
  if (V == 8 || V == 9) {
    bpmi::fe X, Y;
    bpmi::fe_from_words(X, x.v); bpmi::fe_from_words(Y, y.v);
    for (int it = 0; it < iters; it++) { if (V == 8) bpmi::fe_mul(X, X, Y); else bpmi::fe_sqr(X, X); }
    bpmi::fe_canon(X, X); bpmi::fe_to_words(x.v, X);
  } else if (V == 3 || V == 7) {
    fe9 X, Y; to9(X, x); to9(Y, y);
    for (int it = 0; it < iters; it++) { if (V == 3) mul9(X, X, Y); else sqr9(X, X); }
    canon9(X); from9(x, X);
  } else {
    for (int it = 0; it < iters; it++) {
      u32 t[16];
      if (V == 0) mulwide0(t, x.v, y.v); else mulwide4(t, x.v, y.v);
      reduce8(x.v, t);
    }
    // canonicalise: x < 2^256; subtract p if x >= p
    u64 c = (u64)x.v[0] + 977u; u32 t[8]; t[0] = (u32)c; c >>= 32;
    c += (u64)x.v[1] + 1u; t[1] = (u32)c; c >>= 32;
    for (int k = 2; k < 8; k++) { c += x.v[k]; t[k] = (u32)c; c >>= 32; }
    if (c) for (int k = 0; k < 8; k++) x.v[k] = t[k];
  }
  out[i] = x;
}

// ---------------------------------------------------------------- raw instruction throughput
template <int OP>
__global__ void __launch_bounds__(256) kraw(u64* out, int iters) {
  u64 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  u32 x = threadIdx.x * 2654435761u + 12345u, y = x ^ 0x9e3779b9u;
  double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7, dx = 1.0000001, dy = 0.5;
  for (int it = 0; it < iters; it++) {
#define REP8(S) S(a0) S(a1) S(a2) S(a3) S(a4) S(a5) S(a6) S(a7)
    if (OP == 0) {
#define S(r) asm volatile("v_mad_u64_u32 %0, s[4:5], %1, %2, %0" : "+v"(r) : "v"(x), "v"(y) : "s4", "s5");
      REP8(S) REP8(S)
#undef S
    } else if (OP == 1) {
#define S(r) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r) : "v"(a0));
      REP8(S) REP8(S)
#undef S
    } else if (OP == 2) {
#define S(r) { u32 lo = (u32)r; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); r = lo; }
      REP8(S) REP8(S)
#undef S
    } else if (OP == 3) {
#define S(r) { u32 lo = (u32)r; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); r = lo; }
      REP8(S) REP8(S)
#undef S
    } else if (OP == 4) {
#define S(r) { u32 lo = (u32)r; asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(x)); r = lo; }
      REP8(S) REP8(S)
#undef S
    } else if (OP == 5) {
#define S(r) { u32 lo = (u32)r; asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(lo) : "v"(x), "v"(y)); r = lo; }
      REP8(S) REP8(S)
#undef S
    } else if (OP == 6) {
#define S(r) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r) : "v"(dx), "v"(dy));
      S(d0) S(d1) S(d2) S(d3) S(d4) S(d5) S(d6) S(d7) S(d0) S(d1) S(d2) S(d3) S(d4) S(d5) S(d6) S(d7)
#undef S
    } else if (OP == 7) {  // add_co + addc pair through vcc with the required wait states
#define S(r) { u32 lo = (u32)r, hi = (u32)(r >> 32); asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x) : "vcc"); r = ((u64)hi << 32) | lo; }
      REP8(S) REP8(S)
#undef S
    } else if (OP == 8) {
#define S(r) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(r));
      REP8(S) REP8(S)
#undef S
    }
  }
  u64 dsum = (u64)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ dsum;
}

// ---------------------------------------------------------------- host reference
static const u64 HP[4] = {0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL};
static void h_mulmod(u64 r[4], const u64 a[4], const u64 b[4]) {
  u64 t[8] = {0};
  for (int i = 0; i < 4; i++) { u128 c = 0; for (int j = 0; j < 4; j++) { c += (u128)a[i] * b[j] + t[i + j]; t[i + j] = (u64)c; c >>= 64; } t[i + 4] = (u64)c; }
  for (int round = 0; round < 3; round++) {
    u64 m[8] = {0}; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)t[4 + i] * 0x1000003D1ULL + t[i]; m[i] = (u64)c; c >>= 64; }
    m[4] = (u64)c; memcpy(t, m, sizeof(m));
  }
  // t < 2^256 + small, t[4] == 0 now
  for (;;) {
    int ge = 1; for (int i = 3; i >= 0; i--) { if (t[i] != HP[i]) { ge = t[i] > HP[i]; break; } }
    if (!ge) break;
    u64 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - HP[i] - br; t[i] = (u64)d; br = (u64)(d >> 64) & 1; }
  }
  memcpy(r, t, 32);
}

int main() {
  const int nthreads = 256 * 2048, iters = 200, viters = 37;
  std::vector<fe8> ha(nthreads), hb(nthreads), ho(nthreads);
  srand(1234);
  for (int i = 0; i < nthreads; i++) for (int k = 0; k < 8; k++) { ha[i].v[k] = ((u32)rand() << 16) ^ rand(); hb[i].v[k] = ((u32)rand() << 16) ^ rand(); }
  // edge inputs
  for (int k = 0; k < 8; k++) { ha[0].v[k] = 0xFFFFFFFFu; hb[0].v[k] = 0xFFFFFFFFu; ha[1].v[k] = 0; hb[1].v[k] = 0xFFFFFFFFu; }
  memcpy(ha[2].v, HP, 32); memcpy(hb[2].v, HP, 32); ha[2].v[0] -= 1; hb[2].v[0] -= 1;  // p-1
  fe8 *da, *db, *dout; u64* draw;
  CK(hipMalloc(&da, nthreads * 32)); CK(hipMalloc(&db, nthreads * 32)); CK(hipMalloc(&dout, nthreads * 32)); CK(hipMalloc(&draw, nthreads * 8));
  CK(hipMemcpy(da, ha.data(), nthreads * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), nthreads * 32, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  // expected after viters: x_{k+1} = x_k * y  (V0,3,4) or x^2 (V7)
  const int nchk = 4096;
  auto check = [&](int V, const char* name) {
    int bad = 0;
    for (int i = 0; i < nchk; i++) {
      u64 x[4], y[4]; memcpy(x, ha[i].v, 32); memcpy(y, hb[i].v, 32);
      for (int it = 0; it < viters; it++) { if (V == 7 || V == 9) h_mulmod(x, x, x); else h_mulmod(x, x, y); }
      if (memcmp(x, ho[i].v, 32) != 0) { if (bad < 3) printf("  MISMATCH %s idx %d\n", name, i); bad++; }
    }
    printf("check %-28s %s (%d bad of %d)\n", name, bad ? "FAIL" : "ok", bad, nchk);
  };
#define RUNMUL(V, name) { hipLaunchKernelGGL(kmul<V>, dim3(nthreads / 256), dim3(256), 0, 0, dout, da, db, viters); CK(hipDeviceSynchronize()); \
    CK(hipMemcpy(ho.data(), dout, nchk * 32, hipMemcpyDeviceToHost)); check(V, name); \
    float best = 1e9; for (int rep = 0; rep < 5; rep++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(kmul<V>, dim3(nthreads / 256), dim3(256), 0, 0, dout, da, db, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
    printf("time  %-28s %.3f ms  -> %.1f G modmul/s\n", name, best, (double)nthreads * iters / best / 1e6); }
  RUNMUL(0, "V0 8x32 compiler opscan")
  RUNMUL(4, "V4 8x32 asm mad+addc")
  RUNMUL(3, "V3 9x29 carry-free mul")
  RUNMUL(7, "V7 9x29 carry-free sqr")
  RUNMUL(8, "V8 field.hpp fe_mul (fused fold)")
  RUNMUL(9, "V9 field.hpp fe_sqr (fused fold)")
#define RUNRAW(OP, name, per) { float best = 1e9; for (int rep = 0; rep < 3; rep++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(kraw<OP>, dim3(nthreads / 256), dim3(256), 0, 0, draw, 2000); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; } \
    double ops = (double)nthreads * 2000 * 16 * per; printf("raw   %-28s %.3f ms -> %.2f T lane-ops/s  (%.2f cycles/wave-instr/SIMD @2.4GHz)\n", name, best, ops / best / 1e9, 2.4e9 * 256 * 4 * 64 / (ops / best * 1e3)); }
  RUNRAW(0, "v_mad_u64_u32", 1)
  RUNRAW(1, "v_lshl_add_u64", 1)
  RUNRAW(2, "v_mul_lo_u32", 1)
  RUNRAW(3, "v_mul_hi_u32", 1)
  RUNRAW(4, "v_add_u32", 1)
  RUNRAW(5, "v_mad_u32_u24", 1)
  RUNRAW(6, "v_fma_f64", 1)
  RUNRAW(7, "add_co+nop+addc (pair)", 1)
  RUNRAW(8, "v_lshrrev_b64", 1)
  return 0;
}
