// How much slower does a dependent chain of field multiplications run when a wave has its SIMD to itself?  One wave per
// block, grid = w x 1024 blocks (w waves per SIMD when the dispatcher spreads them evenly), every lane squares-and-multiplies
// K times with the product's own fe_mul / fe_sqr (csrc/field.hpp); `pair` runs TWO independent chains per lane (the instruction-
// level parallelism a single wave can offer).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lone_wave_microbench.hip -o tools/bin/lone_wave_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../python-bulletproofs_amd/csrc/curve.hpp"
using namespace bpmi;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k_chain(u32 *io, int K) {
  const u32 t = blockIdx.x * 64 + threadIdx.x;
  fe a, b;
  for (int k = 0; k < 9; k++) { a.v[k] = io[t * 18 + k] & M29; b.v[k] = io[t * 18 + 9 + k] & M29; }
  for (int i = 0; i < K; i++) { fe_mul(a, a, b); fe_sqr(b, a); }
  for (int k = 0; k < 9; k++) io[t * 18 + k] = a.v[k] ^ b.v[k];
}
__global__ void __launch_bounds__(64) k_chain_pair(u32 *io, int K) {
  const u32 t = blockIdx.x * 64 + threadIdx.x;
  fe a, b, c, d;
  for (int k = 0; k < 9; k++) { a.v[k] = io[t * 18 + k] & M29; b.v[k] = io[t * 18 + 9 + k] & M29; c.v[k] = b.v[k] ^ 5; d.v[k] = a.v[k] ^ 3; }
  for (int i = 0; i < K; i++) { fe_mul(a, a, b); fe_mul(c, c, d); fe_sqr(b, a); fe_sqr(d, c); }
  for (int k = 0; k < 9; k++) io[t * 18 + k] = a.v[k] ^ b.v[k] ^ c.v[k] ^ d.v[k];
}
int main() {
  u32 *d;
  const int maxblocks = 8 * 1024;
  CK(hipMalloc(&d, 72ull * 64 * maxblocks));
  CK(hipMemset(d, 0x5A, 72ull * 64 * maxblocks));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int K = 2000;
  for (int pair = 0; pair < 2; pair++)
    for (int w = 1; w <= 8; w *= 2) {
      float ms = 0;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        if (pair) hipLaunchKernelGGL(k_chain_pair, dim3(w * 1024), dim3(64), 0, 0, d, K);
        else hipLaunchKernelGGL(k_chain, dim3(w * 1024), dim3(64), 0, 0, d, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double ops = (pair ? 4.0 : 2.0) * K;
      printf("%s  %d wave(s) per SIMD: %8.3f ms  -> %7.1f ns per field multiplication per wave, %6.1f G mul/s chip-wide\n", pair ? "two chains per lane" : "one chain per lane ",
             w, ms, ms * 1e6 / ops, ops * w * 1024 * 64 / ms / 1e6);
    }
  return 0;
}
