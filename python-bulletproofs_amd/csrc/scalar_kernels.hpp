// scalar_kernels.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).
// Bulk mod-q kernels (/root/reference/src/utils/utils.py:134-137, src/innerproduct/inner_product_prover.py:109-110).
#pragma once

// ------------------------------------------------------------------------------------
// scalar kernels
// ------------------------------------------------------------------------------------
// partial[b] = sum over the block's stride of a_i * b_i ; then k_sc_dot_final sums partials
__global__ void __launch_bounds__(256) k_sc_dot(const u32 *__restrict__ a, const u32 *__restrict__ b, u32 n, u32 *__restrict__ partial) {
  __shared__ u32 sh[256 * 8];
  sc acc;
#pragma unroll
  for (int k = 0; k < 8; k++) acc.v[k] = 0;
  const u32 stride = gridDim.x * blockDim.x;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    sc x, y, t;
    load_words8(x.v, a + 8ull * i);
    load_words8(y.v, b + 8ull * i);
    sc_mul(t, x, y);
    sc_add(acc, acc, t);
  }
  for (u32 d = 128; d > 0; d >>= 1) {
#pragma unroll
    for (int k = 0; k < 8; k++) sh[threadIdx.x * 8 + k] = acc.v[k];
    __syncthreads();
    if (threadIdx.x < d) {
      sc o;
#pragma unroll
      for (int k = 0; k < 8; k++) o.v[k] = sh[(threadIdx.x + d) * 8 + k];
      sc_add(acc, acc, o);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) store_words8(partial + 8ull * blockIdx.x, acc.v);
}
__global__ void __launch_bounds__(256) k_sc_sum(const u32 *__restrict__ partial, u32 n, u32 *__restrict__ out) {
  __shared__ u32 sh[256 * 8];
  sc acc;
#pragma unroll
  for (int k = 0; k < 8; k++) acc.v[k] = 0;
  for (u32 i = threadIdx.x; i < n; i += 256u) { sc x; load_words8(x.v, partial + 8ull * i); sc_add(acc, acc, x); }
  for (u32 d = 128; d > 0; d >>= 1) {
#pragma unroll
    for (int k = 0; k < 8; k++) sh[threadIdx.x * 8 + k] = acc.v[k];
    __syncthreads();
    if (threadIdx.x < d) {
      sc o;
#pragma unroll
      for (int k = 0; k < 8; k++) o.v[k] = sh[(threadIdx.x + d) * 8 + k];
      sc_add(acc, acc, o);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) store_words8(out, acc.v);
}
// out[i] = x * lo[i] + y * hi[i]
__global__ void __launch_bounds__(256) k_sc_fold(const u32 *lo, const u32 *hi, Sc2 xy, u32 n, u32 *out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sc X, Y, a, b, t, s;
#pragma unroll
  for (int k = 0; k < 8; k++) { X.v[k] = xy.k1[k]; Y.v[k] = xy.k2[k]; }
  load_words8(a.v, lo + 8ull * i);
  load_words8(b.v, hi + 8ull * i);
  sc_mul(t, X, a);
  sc_mul(s, Y, b);
  sc_add(t, t, s);
  store_words8(out + 8ull * i, t.v);
}
