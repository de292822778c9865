// scalar_kernels.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).
// Bulk mod-q kernels (/root/reference/src/utils/utils.py:134-137, src/innerproduct/inner_product_prover.py:109-110).
#pragma once

// ------------------------------------------------------------------------------------
// scalar kernels
// ------------------------------------------------------------------------------------
// One or two inner products per launch (blockIdx.y = job: the IPA's cl and cr of a round): partial[job][b] = sum over the
// block's stride of a_i * b_i ; then k_sc_sum adds the partials of every job.
#define SC_DOT_MAX_BLOCKS 1024
struct DotJobs { const u32 *a[2], *b[2]; u32 *out[2]; };
__global__ void __launch_bounds__(256) k_sc_dot(DotJobs jobs, u32 n, u32 *__restrict__ partial_all) {
  __shared__ u32 sh[256 * 8];
  const u32 *__restrict__ a = jobs.a[blockIdx.y], *__restrict__ b = jobs.b[blockIdx.y];
  u32 *partial = partial_all + 8u * SC_DOT_MAX_BLOCKS * blockIdx.y;
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
  // a single block per job IS the sum (short vectors: the late rounds of the inner-product argument save a launch each)
  if (threadIdx.x == 0) store_words8(gridDim.x == 1 ? jobs.out[blockIdx.y] : partial + 8ull * blockIdx.x, acc.v);
}
__global__ void __launch_bounds__(256) k_sc_sum(const u32 *__restrict__ partial_all, u32 n, DotJobs jobs) {
  __shared__ u32 sh[256 * 8];
  const u32 *partial = partial_all + 8u * SC_DOT_MAX_BLOCKS * blockIdx.x;
  u32 *out = jobs.out[blockIdx.x];
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
// out[i] = x * lo[i] + y * hi[i]; one or two folds per launch (blockIdx.y = job: the IPA's a and b vectors)
struct FoldJobs { const u32 *lo[2], *hi[2]; u32 *out[2]; Sc2 xy[2]; };
__global__ void __launch_bounds__(256) k_sc_fold(FoldJobs jobs, u32 n) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 *lo = jobs.lo[blockIdx.y], *hi = jobs.hi[blockIdx.y];
  u32 *out = jobs.out[blockIdx.y];
  sc X, Y, a, b, t, s;
#pragma unroll
  for (int k = 0; k < 8; k++) { X.v[k] = blockIdx.y ? jobs.xy[1].k1[k] : jobs.xy[0].k1[k]; Y.v[k] = blockIdx.y ? jobs.xy[1].k2[k] : jobs.xy[0].k2[k]; }
  load_words8(a.v, lo + 8ull * i);
  load_words8(b.v, hi + 8ull * i);
  sc_mul(t, X, a);
  sc_mul(s, Y, b);
  sc_add(t, t, s);
  store_words8(out + 8ull * i, t.v);
}

// ---- a late round of the inner-product argument between two challenges, ONE launch (round 4) ---------------------------------
// For short vectors (the rounds behind the product fold, or a short argument from the start: M <= 4096 base points per side) the
// four launches between a challenge and the next pair of MSMs -- a' / b' (k_sc_fold), the coefficient tables (k_ipa_coef_update),
// c_L / c_R of the NEXT round (k_sc_dot), the expanded scalars of its L and R (k_ipa_expand) -- are one block of 1 024 threads
// with three block barriers.  Same arithmetic, statement for statement, as the four kernels.  MEASURED AND OFF (option
// "ipa_small_step"): the launches it saves cost ~10 us of a round, the single block's 8 192 expansions on ONE CU cost 60 us more
// than the four kernels spread over the chip (profiles/r04_C3_small_step_ab.txt: 25.1-25.3 ms per 2^20 proof against 24.5).
struct IpaStep {
  u32 *a, *b;                       // n = 2 np elements each, folded in place to np
  const u32 *cg, *ch;               // coefficient tables in (K entries) ...
  u32 *cg2, *ch2;                   // ... and out (2 K)
  Sc2 x_xinv;
  u32 np, K, M;                     // M = np * 2 K: the unfolded bases per side
  const u32 *hscale;                // or null
  u32 *cl, *cr;                     // out: <a'_lo, b'_hi>, <a'_hi, b'_lo>   (np >= 2)
  u32 *eg[2], *eh[2];               // out: the scalars of the next L (0) and R (1) over the M bases
};
__global__ void __launch_bounds__(1024) k_ipa_small_step(IpaStep p) {
  __shared__ u32 sh[1024 * 8];
  const u32 tid = threadIdx.x;
  sc X, XI;
#pragma unroll
  for (int k = 0; k < 8; k++) { X.v[k] = p.x_xinv.k1[k]; XI.v[k] = p.x_xinv.k2[k]; }
  // a' = x a_lo + x^-1 a_hi ; b' = x^-1 b_lo + x b_hi          (inner_product_prover.py:109-110)
  for (u32 i = tid; i < 2u * p.np; i += 1024u) {
    const bool isb = i >= p.np;
    const u32 j = isb ? i - p.np : i;
    u32 *v = isb ? p.b : p.a;
    sc lo, hi, t, s2;
    load_words8(lo.v, v + 8ull * j);
    load_words8(hi.v, v + 8ull * (j + p.np));
    sc_mul(t, isb ? XI : X, lo);
    sc_mul(s2, isb ? X : XI, hi);
    sc_add(t, t, s2);
    store_words8(v + 8ull * j, t.v);
  }
  // g' = x^-1 g_lo + x g_hi ; h' = x h_lo + x^-1 h_hi  (:107-108), deferred: the coefficient tables double
  for (u32 j = tid; j < 2u * p.K; j += 1024u) {
    sc c, r;
    load_words8(c.v, p.cg + 8ull * (j >> 1));
    sc_mul(r, c, (j & 1u) ? X : XI);
    store_words8(p.cg2 + 8ull * j, r.v);
    load_words8(c.v, p.ch + 8ull * (j >> 1));
    sc_mul(r, c, (j & 1u) ? XI : X);
    store_words8(p.ch2 + 8ull * j, r.v);
  }
  __syncthreads();
  if (p.np < 2u) return;                                  // the argument is down to one element: no next round
  const u32 half = p.np >> 1;
  // c_L = <a'_lo, b'_hi>, c_R = <a'_hi, b'_lo>  (:96-97): threads [0, 512) the first, [512, 1024) the second
  {
    const u32 job = tid >> 9, l = tid & 511u;
    sc acc;
#pragma unroll
    for (int k = 0; k < 8; k++) acc.v[k] = 0;
    for (u32 i = l; i < half; i += 512u) {
      sc x, y, t;
      load_words8(x.v, p.a + 8ull * (job ? half + i : i));
      load_words8(y.v, p.b + 8ull * (job ? i : half + i));
      sc_mul(t, x, y);
      sc_add(acc, acc, t);
    }
    for (u32 d = 256; d > 0; d >>= 1) {
#pragma unroll
      for (int k = 0; k < 8; k++) sh[tid * 8 + k] = acc.v[k];
      __syncthreads();
      if (l < d) {
        sc o;
#pragma unroll
        for (int k = 0; k < 8; k++) o.v[k] = sh[(tid + d) * 8 + k];
        sc_add(acc, acc, o);
      }
      __syncthreads();
    }
    if (l == 0) store_words8(job ? p.cr : p.cl, acc.v);
  }
  // the scalars of the next L and R over the unfolded bases (k_ipa_expand with the new tables and the new length)
  u32 logm = 0;
  while ((1u << logm) < p.np) logm++;
  const u32 m = p.np;
  for (u32 idx = tid; idx < 2u * p.M; idx += 1024u) {
    const u32 right = idx >= p.M ? 1u : 0u, k = right ? idx - p.M : idx;
    const u32 i = k & (m - 1u), t = k >> logm;
    const bool hi = i >= half;
    sc z;
#pragma unroll
    for (int q = 0; q < 8; q++) z.v[q] = 0;
    sc rg = z, rh = z;
    if (hi != (right != 0u)) {
      sc av, c;
      load_words8(av.v, p.a + 8ull * (right ? half + i : i - half));
      load_words8(c.v, p.cg2 + 8ull * t);
      sc_mul(rg, av, c);
    }
    if (hi == (right != 0u)) {
      sc bv, c;
      load_words8(bv.v, p.b + 8ull * (right ? i - half : half + i));
      load_words8(c.v, p.ch2 + 8ull * t);
      sc_mul(rh, bv, c);
      if (p.hscale) {
        load_words8(c.v, p.hscale + 8ull * k);
        sc_mul(rh, rh, c);
      }
    }
    store_words8(p.eg[right] + 8ull * k, rg.v);
    store_words8(p.eh[right] + 8ull * k, rh.v);
  }
}

// The verifier's s-vector with the proof's final scalars folded in (Verifier2.get_ss,
// /root/reference/src/innerproduct/inner_product_verifier.py:91-102, and the `a * s_i`, `b / s_i` lists of :131-133):
//   s_i = prod_j x_j^(+1 if bit (k-1-j) of i is set else -1),  sa[i] = a s_i,  sb[i] = b s_i^-1 (c_i)
// with c_i an optional per-generator scale (hsp_i = y^-i hs_i folded into the scalars).  The index splits as
// i = hi 2^kl + lo: a first launch fills the two half tables (<= 2^ceil(k/2) entries, <= k/2 + 1 multiplications each),
// the second multiplies them -- 2 (3 with c_i) multiplications per element instead of 2k, and no host loop over n.
// xt: k pairs (x_j, x_j^-1), 16 words each.  tab: [0, 2^kl): (s_lo, s_lo^-1) interleaved; then [.., + 2^kh): (a s_hi, b s_hi^-1).
__global__ void __launch_bounds__(256) k_sc_svector_tables(const u32 *__restrict__ xt, u32 k, u32 kl, Sc2 ab, u32 *__restrict__ tab) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 nlo = 1u << kl, nhi = 1u << (k - kl);
  if (t >= nlo + nhi) return;
  const bool is_hi = t >= nlo;
  const u32 idx = is_hi ? t - nlo : t;
  sc f, g;                        // f -> s, g -> s^-1
#pragma unroll
  for (int q = 0; q < 8; q++) { f.v[q] = is_hi ? ab.k1[q] : (q == 0 ? 1u : 0u); g.v[q] = is_hi ? ab.k2[q] : (q == 0 ? 1u : 0u); }
  // the low half covers index bits [0, kl) = challenges j in [k - kl, k); the high half bits [kl, k) = j in [0, k - kl)
  const u32 j0 = is_hi ? 0u : k - kl, j1 = is_hi ? k - kl : k;
  const u32 width = j1 - j0;
  for (u32 j = j0; j < j1; j++) {
    const u32 bit = (idx >> (width - 1u - (j - j0))) & 1u;        // MSB first inside the half
    sc x, xi;
    load_words8(x.v, xt + 16ull * j);
    load_words8(xi.v, xt + 16ull * j + 8);
    sc_mul(f, f, bit ? x : xi);
    sc_mul(g, g, bit ? xi : x);
  }
  store_words8(tab + 16ull * t, f.v);
  store_words8(tab + 16ull * t + 8, g.v);
}
__global__ void __launch_bounds__(256) k_sc_svector(const u32 *__restrict__ tab, u32 k, u32 kl, const u32 *__restrict__ scale, u32 n,
                                                    u32 *__restrict__ sa, u32 *__restrict__ sb) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 nlo = 1u << kl;
  const u32 lo = i & (nlo - 1u), hi = i >> kl;
  sc fl, gl, fh, gh, r;
  load_words8(fl.v, tab + 16ull * lo);
  load_words8(gl.v, tab + 16ull * lo + 8);
  load_words8(fh.v, tab + 16ull * (nlo + hi));
  load_words8(gh.v, tab + 16ull * (nlo + hi) + 8);
  sc_mul(r, fl, fh);
  store_words8(sa + 8ull * i, r.v);
  sc_mul(r, gl, gh);
  if (scale) {
    sc c;
    load_words8(c.v, scale + 8ull * i);
    sc_mul(r, r, c);
  }
  store_words8(sb + 8ull * i, r.v);
}

// Self-test hook: the DEVICE bodies of the multiplication family (csrc/field_gen.hpp) on raw limbs, so that a
// test can feed lazy magnitudes and compare the limbs with the host build of the same header (tests/test_gpu_field.py).
// op 0 mul(a,b), 1 sqr(a), 2 mul_add(a,b,c), 3 sqr_add(a,c), 4 mul2(a,b,c,d), 5 carry(a), 6 canon(a); 10..15: mod-q limb arithmetic (below)
__global__ void __launch_bounds__(256) k_debug_fe_op(int op, const u32 *__restrict__ a, const u32 *__restrict__ b, const u32 *__restrict__ c,
                                                     const u32 *__restrict__ d, u32 n, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe A, B, C, D, r;
#pragma unroll
  for (int k = 0; k < 9; k++) { A.v[k] = a[9ull * i + k]; B.v[k] = b[9ull * i + k]; C.v[k] = c[9ull * i + k]; D.v[k] = d[9ull * i + k]; }
  switch (op) {
    case 0: fe_mul(r, A, B); break;
    case 1: fe_sqr(r, A); break;
    case 2: fe_mul_add(r, A, B, C); break;
    case 3: fe_sqr_add(r, A, C); break;
    case 4: fe_mul2(r, A, B, C, D); break;
    case 5: fe_carry(r, A); break;
    case 6: fe_canon(r, A); break;
    default: fe_set_zero(r);
  }
  if (op >= 10) {
    // arithmetic mod q on 9 x 29-bit limbs (scalar.hpp "sq"; sq_mul is the generated chain of scalar_gen.hpp): 10 mul, 11 add,
    // 12 sub, 13 neg(a) -> raw limbs; 14: canonical words of a (8 words + 0); 15: inverse of the canonical value of a
    sq X, Y, R;
#pragma unroll
    for (int k = 0; k < 9; k++) { X.v[k] = A.v[k]; Y.v[k] = B.v[k]; R.v[k] = 0; }
    sc cv;
    switch (op) {
      case 10: sq_mul(R, X, Y); break;
      case 11: sq_add(R, X, Y); break;
      case 12: sq_sub(R, X, Y); break;
      case 13: sq_neg(R, X); break;
      case 14: sq_to_sc(cv, X); for (int k = 0; k < 8; k++) R.v[k] = cv.v[k]; break;
      case 15: { sc in; sq_to_sc(in, X); sc_inv(cv, in); for (int k = 0; k < 8; k++) R.v[k] = cv.v[k]; break; }
      default: break;
    }
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = R.v[k];
  }
#pragma unroll
  for (int k = 0; k < 9; k++) out[9ull * i + k] = r.v[k];
}
