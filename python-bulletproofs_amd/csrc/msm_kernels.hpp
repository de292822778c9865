// msm_kernels.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).
// Kernels of the bucket-method MSM (replaces Pippenger.multiexp, /root/reference/src/pippenger/pippenger.py:22-94).
#pragma once

// ------------------------------------------------------------------------------------
// MSM kernels
// ------------------------------------------------------------------------------------
struct MsmGeom {
  u32 n;       // pairs (with GLV: virtual pairs = 2 x the caller's)
  u32 c;       // window bits
  u32 W;       // windows handled by this launch sequence: [w0, w0 + W) of the recoding
  u32 w0;      // first window (> 0 when one MSM is split into window groups on two lanes)
  u32 B;       // buckets per window = 2^(c-1)
  u32 G;       // W * B
  u32 L;       // entries per thread in k_accum_l0
  u32 nv;      // partial sums per window that the bucket reduction hands to the tail (1 or 4)
  u32 prio;    // mask of the stages (PRIO_*) whose kernels raise their waves' issue priority (see raise_priority)
  u32 fuse;    // 1: k_accum_l0 folds the partial records of a wave's 64 chunks itself (two records per WAVE go to k_segscan, not two per thread)
  u32 top2;    // Wb, the number of WIDE windows (round 5): the last Wb of the W windows have c + 1 bits and 2B buckets each, chosen so that
               // (W - Wb) c + Wb (c + 1) = 256 -- the windows cover exactly the 256 bit positions, the top bit of a folded scalar (< 2^255)
               // is 0, so the last window's digit never exceeds its 2^c and there is NO carry window and no short one (a window of a few
               // bits is one partition of n entries for the sort and a handful of giant buckets for the accumulation).  c = 15: 16 + 1
               // windows; c = 13: 10 + 9; c = 12: 17 + 4.  Keys of window w start at (w + max(0, w - (W - Wb))) B;  G = (W + Wb) B.  0: uniform
  u32 inblock; // 1 (n <= 2^17): k_fine_sort_part sorts a partition of ANY size itself (a heavy one without the LDS staging buffer);
               // the two tile kernels for heavy partitions are not launched
};
// Experiment (option "priority", default off; profiles/r03_wave_priority_ab.txt).  The stages around the accumulation are chains
// of dependent work with few waves; beside the OTHER lane's accumulation (three busy waves on every SIMD) they stretch three- to
// five-fold (k_digit_final: 66 us alone, 320 us beside k_accum_l0).  Raising their waves' issue priority with s_setprio was
// expected to bring the lane behind them to its own accumulation sooner; measured, two MSMs in flight get 2.6 % (2^20) to
// 6 % (2^16) SLOWER -- the accumulation's waves lose more than the chains gain.
// Round 6: re-measured on the multi-round accumulation (chunks of ~30 entries: the accumulation's wave slots turn over while it runs, so the
// other lane's kernels are RESIDENT beside it and what they lack is issue slots, not occupancy): the finish of the reduction took 276 + 154 us
// beside an accumulation instead of 18 + 21 alone, and with the priority raised two MSMs in flight gain 2 % on top of the short chunks' 5.6 %
// (profiles/r06_wave_priority_and_chunk_ab.txt).  MsmGeom.prio is a mask of the stages that raise it:
#define PRIO_SORT 1u          // recoding, partition, level B of the sort
#define PRIO_SCAN 2u          // segmented scan over the partial records
#define PRIO_SUMS 4u          // stage 1 of the bucket reduction (throughput-bound: 2^20 general additions)
#define PRIO_FINISH 8u        // the reduction's finish on quads of lanes (latency-bound chains)
__device__ __forceinline__ void raise_priority(u32 on) { if (on) __builtin_amdgcn_s_setprio(3); }
// mixed window widths (MsmGeom.top2 = Wb): is window w one of the wide ones, and the first key / partition of window w in units of B / (B >> 8)
__device__ __forceinline__ u32 geom_wide(const MsmGeom &g, u32 w) { return (g.top2 && w + g.top2 >= g.W) ? 1u : 0u; }
__device__ __forceinline__ u32 geom_slot(const MsmGeom &g, u32 w) { return (g.top2 && w + g.top2 > g.W) ? 2u * w + g.top2 - g.W : w; }   // w + max(0, w - (W - Wb))

// Signed-digit recoding of scalar i: calls f(w, b, sign) for every window, b = |digit|
// in [0, B] (0 = nothing to add), sign = 1 when the NEGATED point is added.
//   s > (q-1)/2  ->  use q - s on the negated point: halves the digit range, keeps
//   s < 2^255 so W*c >= 256 never overflows, and turns the range-proof scalar q-1
//   (aR, rangeproof_prover.py:43-45) into the single digit -1.
//   for_each_digit_raw reports the digit's own sign and returns whether the scalar was negated
//   (final sign = digit sign ^ negated); a digit of magnitude B is never negative.
// magnitude (below 2^255) and sign of the scalar behind digit source i: scalar i reduced mod q and folded to s or q - s, or with
// GLV the 128-bit magnitude and sign of virtual scalar i
__device__ __forceinline__ bool load_digit_source(sc &s, const Segs &segs, u32 i) {
  if (segs.glv_sub) {
    const uint4 t = *reinterpret_cast<const uint4 *>(segs.glv_sub + 4ull * i);
    s.v[0] = t.x; s.v[1] = t.y; s.v[2] = t.z; s.v[3] = t.w;
    s.v[4] = s.v[5] = s.v[6] = s.v[7] = 0;
    return segs.glv_neg[i] != 0;
  }
  load_words8(s.v, seg_scalar(segs, i));
  sc_reduce_once(s);
  const bool neg = sc_is_high(s);
  if (neg) sc_neg(s, s);
  return neg;
}
template <typename F>
__device__ __forceinline__ bool for_each_digit_raw(const Segs &segs, const MsmGeom &g, u32 i, F f) {
  sc s;
  const bool neg = load_digit_source(s, segs, i);
  u32 carry = 0;
  const u32 wend = g.w0 + g.W;
  for (u32 w = 0; w < wend; w++) {
    const u32 wd = g.c + geom_wide(g, w);               // (mixed widths only with w0 = 0)
    const u32 t = (s.v[0] & ((1u << wd) - 1u)) + carry;
#pragma unroll
    for (int k = 0; k < 7; k++) s.v[k] = (u32)((((u64)s.v[k + 1] << 32) | s.v[k]) >> wd);
    s.v[7] >>= wd;
    u32 b, sign;
    if (t > (1u << (wd - 1u))) { b = (1u << wd) - t; sign = 1; carry = 1; }
    else { b = t; sign = 0; carry = 0; }              // (mixed widths: the last window's t is at most its 2^(wd - 1): never a carry out)
    if (w >= g.w0) f(w - g.w0, b, b ? sign : 0u);
  }
  return neg;
}
template <typename F>
__device__ __forceinline__ void for_each_digit(const Segs &segs, const MsmGeom &g, u32 i, F f) {
  sc s;
  const bool neg = load_digit_source(s, segs, i);
  u32 carry = 0;
  const u32 wend = g.w0 + g.W;
  for (u32 w = 0; w < wend; w++) {
    const u32 wd = g.c + geom_wide(g, w);
    const u32 t = (s.v[0] & ((1u << wd) - 1u)) + carry;
    // shift the 256-bit register right by the window's width (static register indexing)
#pragma unroll
    for (int k = 0; k < 7; k++) s.v[k] = (u32)((((u64)s.v[k + 1] << 32) | s.v[k]) >> wd);
    s.v[7] >>= wd;
    u32 b, sign;
    if (t > (1u << (wd - 1u))) { b = (1u << wd) - t; sign = 1; carry = 1; }
    else { b = t; sign = 0; carry = 0; }
    if (w >= g.w0) f(w - g.w0, b, b ? (sign ^ (u32)neg) : 0u);      // lower windows only feed the carry
  }
}

// GLV preparation, one thread per pair: scalar i (mod q) -> |k1|, |k2| and signs (virtual scalars 2i, 2i + 1); x_i -> beta x_i.
__global__ void __launch_bounds__(256) k_glv_prepare(Segs segs, u32 n, u32 *__restrict__ sub, unsigned char *__restrict__ neg, u32 *__restrict__ bx) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sc k;
  load_words8(k.v, seg_scalar(segs, i));
  sc_reduce_once(k);
  u32 k1[4], k2[4];
  bool n1, n2;
  glv_split(k1, n1, k2, n2, k);
  uint4 *o = reinterpret_cast<uint4 *>(sub + 8ull * i);
  o[0] = make_uint4(k1[0], k1[1], k1[2], k1[3]);
  o[1] = make_uint4(k2[0], k2[1], k2[2], k2[3]);
  neg[2ull * i] = n1 ? 1 : 0;
  neg[2ull * i + 1] = n2 ? 1 : 0;
  u32 w[8];
  load_words8(w, seg_point(segs, i));
  fe x, r, c;
  fe_from_words(x, w);
  fe_mul_beta(r, x);
  fe_canon(c, r);
  fe_to_words(w, c);
  store_words8(bx + 8ull * i, w);
}

// ================= sort path 1 (small n, c < 10): global-atomic counting sort ===========
// dig[w * n + i] = |d| | (sign << 31); histogram with one atomic per lane, or one per wave
// when the whole wave agrees (degenerate inputs)
__global__ void __launch_bounds__(256) k_digits_hist(Segs segs, MsmGeom g, u32 *__restrict__ dig, u32 *__restrict__ hist) {
  raise_priority(g.prio & PRIO_SORT);
  const u32 stride = gridDim.x * blockDim.x;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += stride) {
    for_each_digit(segs, g, i, [&](u32 w, u32 b, u32 sign) {
      dig[(u64)w * g.n + i] = b | (sign << 31);
      const u32 key = b ? (w * g.B + b - 1u) : 0xFFFFFFFFu;
      const unsigned long long act = __ballot(1);
      const u32 first = __builtin_amdgcn_readfirstlane(key);
      const unsigned long long same = __ballot(key == first);
      if (same == act) {
        if (first != 0xFFFFFFFFu) {
          const u32 lane_rank = __builtin_amdgcn_mbcnt_hi((u32)(act >> 32), __builtin_amdgcn_mbcnt_lo((u32)act, 0));
          if (lane_rank == 0) atomicAdd(&hist[first], (u32)__popcll(act));
        }
      } else if (b) {
        atomicAdd(&hist[key], 1u);
      }
    });
  }
}

// ================= sort path 2 (c >= 10): two-level LDS partition sort ==================
// Bucket key k = b - 1 (c-1 bits) = hi * 256 + lo.  Level A partitions all W*n digits by
// (window, hi) with LDS histograms -- global atomics only to reserve one range per
// (tile, partition); level B gives every partition to one block, which counting-sorts it
// by lo entirely in LDS.  No per-element global atomic anywhere.
#define PART_MAX 2048          // W * (B / 256) <= 2048 for every c in [10, 16]
#define COARSE_HIST_WORDS (PART_MAX + 192)     // the partition counts, the any_heavy flag (+0), the tickets of the last-block fusions (+1, +2) and of
                                               // k_digit_final_spread (+8 .. +8 + 4 W), padded to whole 256-byte lines
#define FINE_CAP 12288          // entries of a partition that level B sorts in one block's LDS
// The recoded digits are kept, 16 bits each, window-major: dig16[w * n + i] = (b - 1) | digit sign << 15,
// DIG_NONE for b = 0 (a negative digit has b <= 2^(c-1) - 1, so that code is free), and one byte
// per scalar says whether it was negated.  Level A then reads one window of one tile as a
// contiguous 2-byte stream instead of recoding the scalars.
#define DIG_NONE 0xFFFFu
// Round 5 (`scan` != nullptr): the block that flushes its histogram LAST also runs the exclusive scan of the <= 2048 partition counts
// (k_coarse_scan's work: coarse_off[0..P], coarse_cursor = copy, *offG = total) -- one launch fewer on the path of every MSM of the
// bucket pipeline.  The ticket word is zeroed with the histogram by the MSM's memset.
struct CoarseScanOut { u32 *coarse_off, *coarse_cursor, *offG, *ticket; };
__global__ void __launch_bounds__(1024) k_coarse_hist(Segs segs, MsmGeom g, u32 P, u32 *coarse_hist, unsigned short *__restrict__ dig16,
                                                     unsigned char *__restrict__ negs, CoarseScanOut scan) {
  raise_priority(g.prio & PRIO_SORT);
  __shared__ u32 lh[PART_MAX];
  __shared__ u32 s_last;
  for (u32 p = threadIdx.x; p < P; p += blockDim.x) lh[p] = 0;
  __syncthreads();
  const u32 Bc = g.B >> 8;
  const u32 stride = gridDim.x * blockDim.x;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += stride) {
    const bool neg = for_each_digit_raw(segs, g, i, [&](u32 w, u32 b, u32 dsign) {
      dig16[(u64)w * g.n + i] = (unsigned short)(b ? ((b - 1u) | (dsign << 15)) : DIG_NONE);
      if (b) atomicAdd(&lh[geom_slot(g, w) * Bc + ((b - 1u) >> 8)], 1u);
    });
    negs[i] = neg ? 1 : 0;
  }
  __syncthreads();
  for (u32 p = threadIdx.x; p < P; p += blockDim.x) { const u32 v = lh[p]; if (v) atomicAdd(&coarse_hist[p], v); }
  if (!scan.ticket) return;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(scan.ticket, 1u) == gridDim.x - 1u) ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // every block's counts are in: exclusive scan, `per` consecutive partitions per thread (8 at 256 threads), the threads' sums in lh
  const u32 tid = threadIdx.x, nt = blockDim.x, per = (P + nt - 1u) / nt;
  u32 loc[8], sum = 0;
#pragma unroll
  for (u32 k = 0; k < 8u; k++) {
    const u32 idx = tid * per + k;
    loc[k] = (k < per && idx < P) ? __atomic_load_n(&coarse_hist[idx], __ATOMIC_RELAXED) : 0u;
    sum += loc[k];
  }
  lh[tid] = sum;
  __syncthreads();
  for (u32 d = 1; d < nt; d <<= 1) {
    const u32 v = tid >= d ? lh[tid - d] : 0u;
    __syncthreads();
    lh[tid] += v;
    __syncthreads();
  }
  u32 excl = lh[tid] - sum;
#pragma unroll
  for (u32 k = 0; k < 8u; k++) {
    const u32 idx = tid * per + k;
    if (k < per && idx < P) { scan.coarse_off[idx] = excl; scan.coarse_cursor[idx] = excl; excl += loc[k]; }
  }
  if (tid == nt - 1u) { scan.coarse_off[P] = lh[tid]; *scan.offG = lh[tid]; }
}
// exclusive scan of the <= PART_MAX partition counts in one block:
// coarse_off[0..P] (coarse_off[P] = total), coarse_cursor = copy, off[G] = total
__global__ void __launch_bounds__(1024) k_coarse_scan(const u32 *__restrict__ coarse_hist, u32 P, u32 *__restrict__ coarse_off,
                                                      u32 *__restrict__ coarse_cursor, u32 *__restrict__ offG) {
  __shared__ u32 sh[1024];
  const u32 tid = threadIdx.x;
  const u32 a = (2u * tid < P) ? coarse_hist[2u * tid] : 0u, b = (2u * tid + 1u < P) ? coarse_hist[2u * tid + 1u] : 0u;
  sh[tid] = a + b;
  __syncthreads();
  for (u32 d = 1; d < 1024u; d <<= 1) {
    const u32 v = tid >= d ? sh[tid - d] : 0u;
    __syncthreads();
    sh[tid] += v;
    __syncthreads();
  }
  const u32 excl = sh[tid] - (a + b);
  if (2u * tid < P) { coarse_off[2u * tid] = excl; coarse_cursor[2u * tid] = excl; }
  if (2u * tid + 1u < P) { coarse_off[2u * tid + 1u] = excl + a; coarse_cursor[2u * tid + 1u] = excl + a; }
  if (tid == 1023u) { coarse_off[P] = sh[1023]; *offG = sh[1023]; }
}
// Level A.  part[pos] = lo << 24 | sign << 23 | i   (n <= 2^23), grouped by partition.
// One block = one window of one tile of TS scalars: count the window's B/256 partitions in LDS,
// reserve one range per partition (the only global atomics), rank the entries into an LDS
// staging buffer in partition order and write every partition's run with consecutive lanes.
// With TS = 16384 a run is ~128 entries = 512 contiguous bytes; the earlier version (all W
// windows of a 4096-scalar tile per block, direct stores) produced 128-byte runs written by
// scattered lanes and cost 323 MB of HBM writes for 67 MB of entries (profiles/, PMC).
// Also prepares level B's path for HEAVY partitions (more than FINE_CAP entries; only skewed
// digit distributions have them): zeroes their rows of the fine histogram and raises
// *any_heavy, which the two k_fine_*_heavy kernels test before doing anything.
#define PT_MAX 16384
__global__ void __launch_bounds__(1024) k_partition(MsmGeom g, u32 P, u32 TS, const u32 *__restrict__ coarse_off, u32 *__restrict__ coarse_cursor,
                                                    const unsigned short *__restrict__ dig16, const unsigned char *__restrict__ negs,
                                                    u32 *__restrict__ part, u32 *__restrict__ fine_hist, u32 *__restrict__ any_heavy) {
  raise_priority(g.prio & PRIO_SORT);
  __shared__ u32 cnt[128], excl[128], delta[128];
  __shared__ u32 s_out[PT_MAX];
  const u32 w = blockIdx.y, tid = threadIdx.x;
  const u32 Bc0 = g.B >> 8, Bc = Bc0 << geom_wide(g, w), pbase = geom_slot(g, w) * Bc0;       // partitions of this window (a wide one owns 2B buckets; <= 128) and the first of them
  const u32 lin = blockIdx.y * gridDim.x + blockIdx.x, nblk = gridDim.x * gridDim.y;
  if (!g.inblock) for (u32 p = lin; p < P; p += nblk) {
    if (coarse_off[p + 1] - coarse_off[p] > FINE_CAP) {        // block-uniform
      if (tid < 256u) fine_hist[p * 256u + tid] = 0;
      if (tid == 0) *any_heavy = 1u;
    }
  }
  const u32 i0 = blockIdx.x * TS;
  const u32 i1 = (i0 + TS < g.n) ? i0 + TS : g.n;
  const unsigned short *dw = dig16 + (u64)w * g.n;
  if (tid < 128u) cnt[tid] = 0;
  __syncthreads();
  for (u32 i = i0 + tid; i < i1; i += blockDim.x) {
    const u32 code = dw[i];
    if (code != DIG_NONE) atomicAdd(&cnt[(code & 0x7FFFu) >> 8], 1u);
  }
  __syncthreads();
  // exclusive prefix over the (<= 128) partition counts of this window
  u32 c = 0;
  if (tid < 128u) { c = tid < Bc ? cnt[tid] : 0u; excl[tid] = c; }
  __syncthreads();
  for (u32 d = 1; d < 128u; d <<= 1) {
    u32 v = 0;
    if (tid < 128u && tid >= d) v = excl[tid - d];
    __syncthreads();
    if (tid < 128u) excl[tid] += v;
    __syncthreads();
  }
  if (tid < 128u) {
    const u32 start = excl[tid] - c;
    excl[tid] = start;
    cnt[tid] = start;                                          // LDS write cursor of the partition
    if (c) delta[tid] = atomicAdd(&coarse_cursor[pbase + tid], c) - start;
  }
  __syncthreads();
  for (u32 i = i0 + tid; i < i1; i += blockDim.x) {
    const u32 code = dw[i];
    if (code != DIG_NONE) {
      const u32 k = code & 0x7FFFu;
      s_out[atomicAdd(&cnt[k >> 8], 1u)] = ((k & 255u) << 24) | (((code >> 15) ^ negs[i]) << 23) | i;
    }
  }
  __syncthreads();
  // partition p's run: staged at [excl[p], cnt[p]), goes to part[delta[p] + j]
  const u32 wave = tid >> 6, lane = tid & 63u, nwaves = blockDim.x >> 6;
  for (u32 p = wave; p < Bc; p += nwaves) {
    const u32 lo = excl[p], hi = cnt[p], dl = delta[p];
    for (u32 j = lo + lane; j < hi; j += 64u) part[dl + j] = s_out[j];
  }
}
// chunk_key[t] = bucket that contains sorted position t * L (for every chunk start inside [lo, hi))
// Called by EVERY lane of a wave (lanes without a bucket pass lo == hi).  A bucket with few chunk starts is filled by its own
// lane; a long one (skewed digits, the short top window of c = 12, 14, 15: thousands of chunk starts) by the whole wave,
// instead of thousands of dependent stores from one lane (0.24 ms at n = 2^16, c = 15).
#define CHUNK_FILL_OWN 16u
__device__ __forceinline__ void fill_chunk_keys(u32 *__restrict__ chunk_key, u32 L, u32 key, u32 lo, u32 hi) {
  const u32 t0 = (lo + L - 1u) / L;
  const u32 t1 = hi > lo ? (u32)(((u64)hi + L - 1u) / L) : t0;         // chunk starts t0 .. t1 - 1 lie in [lo, hi)
  const bool big = t1 - t0 > CHUNK_FILL_OWN;
  if (!big) for (u32 t = t0; t < t1; t++) chunk_key[t] = key;
  unsigned long long m = __ballot(big);
  const u32 lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  while (m) {
    const int src = __ffsll((long long)m) - 1;
    m &= m - 1ull;
    const u32 k = (u32)__shfl((int)key, src, 64), a = (u32)__shfl((int)t0, src, 64), b = (u32)__shfl((int)t1, src, 64);
    for (u32 t = a + lane; t < b; t += 64u) chunk_key[t] = k;
  }
}
// Level B, one block per partition: the partition's entries are counting-sorted by the low 8
// key bits.  Its output range [base, base + m) is known from level A, so no global atomic and
// no global histogram are needed: the bucket offsets off[] (and the chunk keys) come out of
// the block's own 256-bin prefix sum, and the sorted entries are staged in LDS and written
// as ONE contiguous, fully coalesced range (scattered 4-byte stores were the cost of a
// tile-based version: 0.125 ms of its 0.15 ms at n = 2^20).  For a partition larger than the
// staging buffer the block only turns the counts of k_fine_hist_heavy into offsets.
#define FINE_THREADS 512
// bins[bin]++ and the old value.  A partition that outgrew the staging buffer is a few buckets with thousands of entries each (equal
// scalars, bit vectors): the lanes of a wave that name the same bin share ONE atomic -- up to four such groups per call, whoever is
// left takes its own (tens of thousands of increments of one LDS word otherwise)
__device__ __forceinline__ u32 lds_take_slot(u32 *bins, u32 bin) {
  const u32 lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  u32 pos = 0;
  bool done = false;
  for (int it = 0; it < 4; it++) {
    const unsigned long long rem = __ballot(!done);
    if (rem == 0ull) break;
    const u32 first = (u32)__shfl((int)bin, __ffsll((long long)rem) - 1, 64);
    const bool match = !done && bin == first;
    const unsigned long long m = __ballot(match);
    if (match) {
      const u32 rank = (u32)__popcll(m & ((1ull << lane) - 1ull));
      u32 basepos = 0;
      if (rank == 0) basepos = atomicAdd(&bins[first], (u32)__popcll(m));
      basepos = (u32)__shfl((int)basepos, __ffsll((long long)m) - 1, 64);
      pos = basepos + rank;
      done = true;
    }
  }
  if (!done) pos = atomicAdd(&bins[bin], 1u);
  return pos;
}
__global__ void __launch_bounds__(FINE_THREADS) k_fine_sort_part(MsmGeom g, const u32 *__restrict__ coarse_off, const u32 *__restrict__ part,
                                                                  const u32 *__restrict__ fine_hist, u32 *__restrict__ off, u32 *__restrict__ cursor,
                                                                  u32 *__restrict__ chunk_key, u32 *__restrict__ sidx, u32 *__restrict__ buckets) {
  raise_priority(g.prio & PRIO_SORT);
  __shared__ u32 bins[256];
  __shared__ u32 s_out[FINE_CAP];
  const u32 p = blockIdx.x, tid = threadIdx.x;
  const u32 base = coarse_off[p], m = coarse_off[p + 1] - base;
  const bool big = m > FINE_CAP;                 // does not fit the staging buffer (skewed digits only)
  const bool heavy = big && !g.inblock;          // counted by k_fine_hist_heavy, scattered by k_fine_scatter_heavy
  if (tid < 256u) bins[tid] = heavy ? fine_hist[p * 256u + tid] : 0u;
  __syncthreads();
  if (!heavy) {
    if (big) for (u32 i = tid; i < m; i += FINE_THREADS) (void)lds_take_slot(bins, part[base + i] >> 24);
    else for (u32 i = tid; i < m; i += FINE_THREADS) atomicAdd(&bins[part[base + i] >> 24], 1u);
    __syncthreads();
  }
  // exclusive prefix over the 256 bins (threads 0..255; Hillis-Steele in place)
  u32 cnt = 0;
  if (tid < 256u) cnt = bins[tid];
  for (u32 d = 1; d < 256u; d <<= 1) {
    u32 v = 0;
    if (tid < 256u && tid >= d) v = bins[tid - d];
    __syncthreads();
    if (tid < 256u) bins[tid] += v;
    __syncthreads();
  }
  if (tid < 256u) {
    const u32 start = bins[tid] - cnt;
    const u32 key = p * 256u + tid;
    off[key] = base + start;
    fill_chunk_keys(chunk_key, g.L, key, base + start, base + start + cnt);
    if (heavy) cursor[key] = base + start;
    if (cnt == 0) {          // nobody will write this bucket: make it the identity here (no memset of all buckets)
      xyzz inf;
      xyzz_set_inf(inf);
      xyzz_store_g(buckets + (u64)key * XYZZ_WORDS, inf);
    }
  }
  if (heavy) return;
  __syncthreads();
  if (tid < 256u) bins[tid] -= cnt;            // bins = tile-local write cursor of every bucket
  __syncthreads();
  if (big) {
    // inblock (n <= 2^17: a partition has at most n entries): this block scatters the partition straight to memory -- two passes of
    // at most 256 entries per thread instead of two more launches on the path of EVERY mid-sized MSM
    for (u32 i = tid; i < m; i += FINE_THREADS) {
      const u32 e = part[base + i];
      sidx[base + lds_take_slot(bins, e >> 24)] = (e & 0x7FFFFFu) | ((e & 0x800000u) << 8);
    }
    return;
  }
  for (u32 i = tid; i < m; i += FINE_THREADS) {
    const u32 e = part[base + i];
    s_out[atomicAdd(&bins[e >> 24], 1u)] = (e & 0x7FFFFFu) | ((e & 0x800000u) << 8);
  }
  __syncthreads();
  for (u32 i = tid; i < m; i += FINE_THREADS) sidx[base + i] = s_out[i];
}
// Heavy partitions (skewed digit distributions, e.g. half of all scalars in {0, 1}) are spread
// over fixed-size TILES of the partitioned array instead, so a partition -- or a single bucket
// -- of any size is shared by many blocks:
//   k_fine_hist_heavy     per tile: LDS histogram over fine buckets -> global fine histogram
//   (k_fine_sort_part)    -> off[], cursor[], chunk keys of the heavy partitions' buckets
//   k_fine_scatter_heavy  per tile: LDS histogram again, reserve one range per touched bucket,
//                         scatter with LDS cursors
// Both return at once when no partition is heavy.  The LDS table covers FINE_BINS consecutive
// buckets from the tile's first one; entries beyond it use a global atomic directly.
#define FINE_TILE 4096
#define FINE_BINS 4096
struct FineTile {
  u32 j0, j1;        // positions covered
  u32 p_first;       // partition of position j0
};
__device__ __forceinline__ FineTile fine_tile_setup(const u32 *__restrict__ coarse_off, u32 P, u32 E, u32 *s_off) {
  for (u32 i = threadIdx.x; i <= P; i += 256u) s_off[i] = coarse_off[i];
  __syncthreads();
  FineTile t;
  t.j0 = blockIdx.x * FINE_TILE;
  t.j1 = (t.j0 + FINE_TILE < E) ? t.j0 + FINE_TILE : E;
  // largest p with s_off[p] <= j0 and s_off[p+1] > j0 (binary search, same in every thread)
  u32 lo = 0, hi = P;
  while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (s_off[mid] <= t.j0) lo = mid; else hi = mid; }
  t.p_first = lo;
  return t;
}
// the partition that holds position j (> the current one's range): bisection over the offsets in LDS -- a walk costs one
// step per EMPTY partition, and a range proof's commitment scalars (one giant partition, then 2000 nearly empty ones) made
// every thread of every tile walk them all (0.2 ms of the sort at n = 2^16)
__device__ __forceinline__ u32 fine_partition_of(const u32 *s_off, u32 P, u32 pcur, u32 j) {
  u32 lo = pcur + 1u, hi = P - 1u;                  // smallest p > pcur with s_off[p + 1] > j   (j < s_off[P])
  while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (s_off[mid + 1] > j) hi = mid; else lo = mid + 1u; }
  return lo;
}
__global__ void __launch_bounds__(256) k_fine_hist_heavy(MsmGeom g, u32 P, const u32 *__restrict__ coarse_off, const u32 *__restrict__ part,
                                                         const u32 *__restrict__ offE, const u32 *__restrict__ any_heavy, u32 *__restrict__ fine_hist) {
  raise_priority(g.prio & PRIO_SORT);
  __shared__ u32 s_off[PART_MAX + 1];
  __shared__ u32 bins[FINE_BINS];
  if (!*any_heavy) return;
  const u32 E = offE[0];
  if (blockIdx.x * FINE_TILE >= E) return;
  const FineTile t = fine_tile_setup(coarse_off, P, E, s_off);
  for (u32 i = threadIdx.x; i < FINE_BINS; i += 256u) bins[i] = 0;
  __syncthreads();
  const u32 g_first = t.p_first * 256u;
  u32 pcur = t.p_first, bound = s_off[pcur + 1];
  bool heavy = bound - s_off[pcur] > FINE_CAP;
  for (u32 j = t.j0 + threadIdx.x; j < t.j1; j += 256u) {
    if (j >= bound) { pcur = fine_partition_of(s_off, P, pcur, j); bound = s_off[pcur + 1]; heavy = bound - s_off[pcur] > FINE_CAP; }
    if (heavy) {
      const u32 key = pcur * 256u + (part[j] >> 24);
      const u32 rel = key - g_first;
      if (rel < FINE_BINS) atomicAdd(&bins[rel], 1u); else atomicAdd(&fine_hist[key], 1u);
    }
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < FINE_BINS; i += 256u) { const u32 v = bins[i]; if (v) atomicAdd(&fine_hist[g_first + i], v); }
}
__global__ void __launch_bounds__(256) k_fine_scatter_heavy(MsmGeom g, u32 P, const u32 *__restrict__ coarse_off, const u32 *__restrict__ part,
                                                            const u32 *__restrict__ offE, const u32 *__restrict__ any_heavy, u32 *__restrict__ cursor,
                                                            u32 *__restrict__ sidx) {
  raise_priority(g.prio & PRIO_SORT);
  __shared__ u32 s_off[PART_MAX + 1];
  __shared__ u32 bins[FINE_BINS];
  if (!*any_heavy) return;
  const u32 E = offE[0];
  if (blockIdx.x * FINE_TILE >= E) return;
  const FineTile t = fine_tile_setup(coarse_off, P, E, s_off);
  for (u32 i = threadIdx.x; i < FINE_BINS; i += 256u) bins[i] = 0;
  __syncthreads();
  const u32 g_first = t.p_first * 256u;
  u32 pcur = t.p_first, bound = s_off[pcur + 1];
  bool heavy = bound - s_off[pcur] > FINE_CAP;
  for (u32 j = t.j0 + threadIdx.x; j < t.j1; j += 256u) {
    if (j >= bound) { pcur = fine_partition_of(s_off, P, pcur, j); bound = s_off[pcur + 1]; heavy = bound - s_off[pcur] > FINE_CAP; }
    if (heavy) {
      const u32 rel = pcur * 256u + (part[j] >> 24) - g_first;
      if (rel < FINE_BINS) atomicAdd(&bins[rel], 1u);
    }
  }
  __syncthreads();
  // reserve this tile's range in every touched bucket: count -> base position
  for (u32 i = threadIdx.x; i < FINE_BINS; i += 256u) { const u32 v = bins[i]; if (v) bins[i] = atomicAdd(&cursor[g_first + i], v); }
  __syncthreads();
  pcur = t.p_first; bound = s_off[pcur + 1];
  heavy = bound - s_off[pcur] > FINE_CAP;
  for (u32 j = t.j0 + threadIdx.x; j < t.j1; j += 256u) {
    if (j >= bound) { pcur = fine_partition_of(s_off, P, pcur, j); bound = s_off[pcur + 1]; heavy = bound - s_off[pcur] > FINE_CAP; }
    if (heavy) {
      const u32 e = part[j];
      const u32 key = pcur * 256u + (e >> 24);
      const u32 rel = key - g_first;
      const u32 pos = (rel < FINE_BINS) ? atomicAdd(&bins[rel], 1u) : atomicAdd(&cursor[key], 1u);
      sidx[pos] = (e & 0x7FFFFFu) | ((e & 0x800000u) << 8);
    }
  }
}
// path 1 equivalent of the chunk-key fill: one thread per bucket
__global__ void __launch_bounds__(256) k_chunk_keys(MsmGeom g, const u32 *__restrict__ off, u32 *__restrict__ chunk_key) {
  raise_priority(g.prio & PRIO_SORT);
  const u32 key = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = key < g.G;
  const u32 lo = valid ? off[key] : 0u, hi = valid ? off[key + 1] : 0u;
  fill_chunk_keys(chunk_key, g.L, key, lo, hi);
}

// ---- exclusive scan of hist[0..G) -> off[0..G], cursor[0..G) = off ------------------
#define SCAN_PER_THREAD 16
#define SCAN_TILE (256 * SCAN_PER_THREAD)
__global__ void __launch_bounds__(256) k_scan_partials(const u32 *__restrict__ hist, u32 G, u32 *__restrict__ bsum) {
  __shared__ u32 red[256];
  const u32 base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; k++) { const u32 j = base + k; if (j < G) s += hist[j]; }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if (threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}
// single block: exclusive scan of bsum[0..nb) in place, total -> off[G]
__global__ void __launch_bounds__(1024) k_scan_top(u32 *__restrict__ bsum, u32 nb, u32 *__restrict__ off, u32 G) {
  __shared__ u32 sh[1024];
  u32 running = 0;
  for (u32 base = 0; base < nb; base += 1024) {
    const u32 j = base + threadIdx.x;
    const u32 v = j < nb ? bsum[j] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      u32 t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (j < nb) bsum[j] = running + sh[threadIdx.x] - v;
    const u32 tot = sh[1023];
    __syncthreads();
    running += tot;
  }
  if (threadIdx.x == 0) off[G] = running;
}
__global__ void __launch_bounds__(256) k_scan_final(const u32 *__restrict__ hist, u32 G, const u32 *__restrict__ bsum,
                                                    u32 *__restrict__ off, u32 *__restrict__ cursor) {
  __shared__ u32 sh[256];
  const u32 base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  u32 v[SCAN_PER_THREAD];
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; k++) { const u32 j = base + k; v[k] = j < G ? hist[j] : 0; s += v[k]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    u32 t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  u32 run = bsum[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; k++) {
    const u32 j = base + k;
    if (j < G) { off[j] = run; cursor[j] = run; }
    run += v[k];
  }
}

// ---- counting-sort scatter (path 1) ---------------------------------------------------
__global__ void __launch_bounds__(256) k_scatter(MsmGeom g, const u32 *__restrict__ dig, u32 *__restrict__ cursor,
                                                 u32 *__restrict__ sidx) {
  raise_priority(g.prio & PRIO_SORT);
  const u32 stride = gridDim.x * blockDim.x;
  for (u32 w = 0; w < g.W; w++) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += stride) {
      const u32 d = dig[(u64)w * g.n + i];
      const u32 b = d & 0x7FFFFFFFu;
      const u32 key = b ? (w * g.B + b - 1u) : 0xFFFFFFFFu;
      const unsigned long long act = __ballot(1);
      const u32 first = __builtin_amdgcn_readfirstlane(key);
      const unsigned long long same = __ballot(key == first);
      u32 pos = 0;
      if (same == act) {
        if (first != 0xFFFFFFFFu) {
          const u32 lane_rank = __builtin_amdgcn_mbcnt_hi((u32)(act >> 32), __builtin_amdgcn_mbcnt_lo((u32)act, 0));
          u32 basepos = 0;
          if (lane_rank == 0) basepos = atomicAdd(&cursor[first], (u32)__popcll(act));
          basepos = __builtin_amdgcn_readfirstlane(basepos);
          pos = basepos + lane_rank;
        }
      } else if (b) {
        pos = atomicAdd(&cursor[key], 1u);
      }
      if (b) sidx[pos] = i | (d & 0x80000000u);
    }
  }
}

// ---- level 0: every thread adds exactly L sorted entries --------------------------------
__device__ __forceinline__ void xyzz_shfl_up(xyzz &r, const xyzz &a, int d) {
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.X.v[i] = (u32)__shfl_up((int)a.X.v[i], d, 64);
    r.Y.v[i] = (u32)__shfl_up((int)a.Y.v[i], d, 64);
    r.ZZ.v[i] = (u32)__shfl_up((int)a.ZZ.v[i], d, 64);
    r.ZZZ.v[i] = (u32)__shfl_up((int)a.ZZZ.v[i], d, 64);
  }
}
// FUSE (round 4): the first level of the segmented scan happens HERE, inside the wave.  A chunk leaves a head run (the part of
// its first bucket that lies in the chunk) and a tail run (of its last bucket); the tail of lane i and the head of lane i + 1
// are almost always the same bucket.  Without FUSE both go to memory as 148-byte records -- 2 per thread, 390 000 at n = 2^20 --
// and k_segscan's first level (1 525 blocks, every wave paying a general addition) adds them up: 52 us of a 1.04 ms step, and at
// n = 2^16 0.035 of 0.36 ms.  With FUSE the head waits in LDS, and after the loop the wave runs a segmented scan over its 64
// lanes with shuffles: tails that cover a whole chunk ("single": the chunk is one run) chain through log-many steps (none for
// uniform digits), then every head takes the tail in front of it -- ONE general addition per thread --, completed runs go
// straight to their buckets and only the wave's first and last run become records: 2 per WAVE (6 100 at 2^20; k_segscan is two
// tiny launches).  Heavy buckets keep their log depth: 6 steps in the wave, then k_segscan over waves.
template <bool GLV, bool FUSE> __global__ void __launch_bounds__(256) k_accum_l0(Segs segs, MsmGeom g, const u32 *__restrict__ off,
                                                  const u32 *__restrict__ chunk_key, const u32 *__restrict__ sidx,
                                                  u32 *__restrict__ buckets, u32 *__restrict__ rec_key, u32 *__restrict__ rec_pt) {
  __shared__ u32 s_head[FUSE ? 256 * LDS_STRIDE : 1];
  const u32 E = off[g.G];
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 start = t * g.L;
  const bool valid = start < E;
  if (!FUSE) { if (!valid) return; }
  else if (__ballot(valid) == 0ull) return;          // (valid lanes are a prefix of the wave)
  const u32 end = (u32)((start + g.L < E) ? start + g.L : E);
  xyzz acc;
  xyzz_set_inf(acc);
  u32 cur = valid ? chunk_key[t] : 0u;              // bucket containing position `start`
  const u32 hk = cur;                  // ... = the key of the chunk's head run
  bool first = true;
  if (valid) {
  u32 boundary = off[cur + 1];         // first position after that bucket's run
  u32 boundary2 = off[cur + 2 < g.G ? cur + 2 : g.G];       // ... and after the next bucket's: on its way before a flush needs it
  // software pipeline: the point of entry j+1 and the index of entry j+2 are in flight while entry j is added (the address of
  // a point depends on its index: with the index only one entry ahead every iteration began with a full load latency)
  u32 e_next = sidx[start];
  u32 e_next2 = (start + 1 < end) ? sidx[start + 1] : 0u;
  u32 w_next[16];
  load_entry_point<GLV>(w_next, segs, e_next & 0x7FFFFFFFu);
  for (u32 j = (u32)start; j < end; j++) {
    const u32 e = e_next;
    affine P;
    affine_from_words(P, w_next);
    // the limbs exist HERE, before the next loads overwrite the words (otherwise the conversion sinks below them and the 16
    // words are copied aside first)
    asm volatile("" : "+v"(P.x.v[0]), "+v"(P.x.v[1]), "+v"(P.x.v[2]), "+v"(P.x.v[3]), "+v"(P.x.v[4]), "+v"(P.x.v[5]), "+v"(P.x.v[6]),
                 "+v"(P.x.v[7]), "+v"(P.x.v[8]), "+v"(P.y.v[0]), "+v"(P.y.v[1]), "+v"(P.y.v[2]), "+v"(P.y.v[3]), "+v"(P.y.v[4]),
                 "+v"(P.y.v[5]), "+v"(P.y.v[6]), "+v"(P.y.v[7]), "+v"(P.y.v[8]));
    if (j + 1 < end) {
      e_next = e_next2;
      if (j + 2 < end) e_next2 = sidx[j + 2];
      load_entry_point<GLV>(w_next, segs, e_next & 0x7FFFFFFFu);
    }
    if (j == boundary) {               // the run of `cur` ended: flush, move to the next non-empty bucket
      if (first) {
        if (FUSE) xyzz_store(s_head + threadIdx.x * LDS_STRIDE, acc);
        else { rec_key[2 * t] = cur; xyzz_store_g(rec_pt + (2 * t) * XYZZ_WORDS, acc); }
        first = false;
      }
      else xyzz_store_g(buckets + (u64)cur * XYZZ_WORDS, acc);
      xyzz_set_inf(acc);
      cur++; boundary = boundary2;
      if (boundary == j) {
        // empty buckets ahead.  Uniform digits leave one or two (2 entries per bucket on average at n = 2^16: a walk of a load
        // or two); the scalars of a range proof (bits, and one blinding factor per window) leave THOUSANDS between two entries,
        // and a walk over those is one dependent load each (2 ms per MSM at n = 16 385, c = 13): after eight steps, bisect
        // off[] for the bucket of sorted position j.  (Bisecting at once costs the common case 19 loads: 0.09 -> 0.14 ms at 2^16.)
        u32 steps = 0;
        do { cur++; boundary = off[cur + 1]; steps++; } while (boundary == j && steps < 8u);
        if (boundary == j) {
          u32 lo = cur + 1u, hi = g.G - 1u;                   // smallest key > cur with off[key + 1] > j
          while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (off[mid + 1] > j) hi = mid; else lo = mid + 1u; }
          cur = lo; boundary = off[cur + 1];
        }
      }
      boundary2 = off[cur + 2 < g.G ? cur + 2 : g.G];
    }
    xyzz_madd_signed(acc, P, (e >> 31) != 0);
  }
  }
  if (!FUSE) {
    if (first) {
      rec_key[2 * t] = cur; xyzz_store_g(rec_pt + (2 * t) * XYZZ_WORDS, acc);
      xyzz_set_inf(acc);
    }
    rec_key[2 * t + 1] = cur;
    xyzz_store_g(rec_pt + (2 * t + 1) * XYZZ_WORDS, acc);
    return;
  }
  // ---- the wave's segmented scan.  Lane i holds: tail run S = acc with key tk = cur; head run (key hk) in LDS unless the chunk
  // is one single run (`single`: then S IS the chunk and hk == tk).  link: the run in front of this chunk continues into it.
  const u32 lane = threadIdx.x & 63u;
  const u32 gw = (u32)(t >> 6);                                  // global wave index: records 2 gw (head run of the wave), 2 gw + 1 (tail run)
  const bool single = first;
  const u32 tk = cur;
  const u32 prev_tk = (u32)__shfl_up((int)tk, 1, 64);
  const bool link = valid && lane > 0u && prev_tk == hk;
  bool F = !(valid && single && link);                          // true: this lane's S starts a segment
  for (u32 d = 1; d < 64u; d <<= 1) {
    const bool take = !F;                                        // (implies lane >= d: lane 0 always starts a segment)
    if (__ballot(take) == 0ull) break;
    xyzz o;
    xyzz_shfl_up(o, acc, (int)d);
    const bool oF = __shfl_up((int)F, (int)d, 64) != 0;
    if (take) { xyzz_add(acc, o, acc); F = oF; }
  }
  const u32 first_key = (u32)__shfl((int)hk, 0, 64);
  const bool next_valid = __shfl_down((int)valid, 1, 64) != 0, next_link = __shfl_down((int)link, 1, 64) != 0;
  const bool last = valid && (lane == 63u || !next_valid);
  // tails first (then S is dead and the heads have the registers): S_i is complete when the next chunk does not continue it (or
  // there is no next chunk in the wave); the copy for the lane behind is taken before
  xyzz Sp;
  xyzz_shfl_up(Sp, acc, 1);
  const bool s_done = valid && (last || !next_link);
  const bool s_is_head = tk == first_key;                         // the wave's first run reaches to here (every chunk so far single)
  {
    // one store site, per-lane destination (see k_segscan about merged stores in divergent flow)
    u32 *dst = s_is_head ? rec_pt + (u64)(2u * gw) * XYZZ_WORDS : (last ? rec_pt + (u64)(2u * gw + 1u) * XYZZ_WORDS : buckets + (u64)tk * XYZZ_WORDS);
    if (s_done) xyzz_store_g(dst, acc);
  }
  if (last) {
    rec_key[2u * gw + 1u] = tk;
    if (s_is_head) {                                             // the whole wave is one run: empty tail record
      xyzz_set_inf(acc);
      xyzz_store_g(rec_pt + (u64)(2u * gw + 1u) * XYZZ_WORDS, acc);
    }
  }
  // heads: H_i += S_(i-1) where the run continues and ends inside chunk i; a head with the wave's first key is the wave's head record
  const bool has_head = valid && !single;
  if (__ballot(has_head) != 0ull) {
    if (has_head) xyzz_load(acc, s_head + threadIdx.x * LDS_STRIDE); else xyzz_set_inf(acc);
    if (__ballot(has_head && link) != 0ull) { if (has_head && link) xyzz_add(acc, Sp, acc); }
    u32 *dst = (hk == first_key) ? rec_pt + (u64)(2u * gw) * XYZZ_WORDS : buckets + (u64)hk * XYZZ_WORDS;
    if (has_head) xyzz_store_g(dst, acc);
  }
  if (lane == 0u) rec_key[2u * gw] = first_key;
}

// number of records entering segscan level `level` (1-based); 0 when that level has nothing to do
__device__ __forceinline__ u32 records_at_level(u32 E, u32 L, u32 fuse, int level, bool &is_final) {
  is_final = false;
  if (E == 0) return 0;
  const u32 chunks = (E + L - 1) / L;
  u32 R = 2u * (fuse ? (chunks + 63u) / 64u : chunks);           // two records per wave of k_accum_l0<.., true>, per thread otherwise
  for (int l = 1; l < level; l++) {
    const u32 nb = (R + 255u) / 256u;
    if (nb <= 1) return 0;          // the previous level was already final
    R = 2u * nb;
  }
  is_final = ((R + 255u) / 256u) <= 1;
  return R;
}

// ---- levels >= 1: block-wide segmented scan over partial records -------------------------
// `ticket` (round 5; nullptr = off): when this level leaves at most 256 records (nb <= 128 blocks), the block that finishes LAST
// runs the final level itself instead of a second launch of one block (8 us on the path of every MSM up to ~2^21 pairs: the fused
// wave scan of k_accum_l0 leaves two records per wave, so level 1 has <= 128 blocks up to 16 384 waves).  The ticket word is
// zeroed by the MSM's memset and left at nb.
__global__ void __launch_bounds__(256) k_segscan(MsmGeom g, const u32 *__restrict__ off, int level,
                                                 const u32 *in_key, const u32 *in_pt,
                                                 u32 *out_key, u32 *out_pt, u32 *__restrict__ buckets, u32 *ticket) {
  raise_priority(g.prio & PRIO_SCAN);
  __shared__ u32 s_key[256];
  __shared__ u32 s_val[256 * LDS_STRIDE];
  __shared__ u32 s_last;
  bool is_final;
  u32 R = records_at_level(off[g.G], g.L, g.fuse, level, is_final);
  u32 nb = (R + 255u) / 256u;
  if (blockIdx.x >= nb) return;
  const u32 tid = threadIdx.x;
  u32 blk = blockIdx.x;
  for (;;) {
    const u32 j = blk * 256u + tid;
    const bool valid = j < R;
    const u32 key = valid ? in_key[j] : 0xFFFFFFFFu;
    xyzz val;
    if (valid) xyzz_load_g(val, in_pt + (u64)j * XYZZ_WORDS); else xyzz_set_inf(val);
    s_key[tid] = key;
    __syncthreads();
    for (u32 d = 1; d < 256; d <<= 1) {
      // the records are in bucket order, so equal keys are contiguous: when no thread of the block
      // finds its key at distance d, none will at 2d, 4d, ... (uniform scalars stop after d = 1 or 2)
      const bool act = valid && tid >= d && s_key[tid - d] == key;
      if (!__syncthreads_or(act)) break;
      xyzz_store(s_val + tid * LDS_STRIDE, val);
      __syncthreads();
      if (act) {
        xyzz other;
        xyzz_load(other, s_val + (tid - d) * LDS_STRIDE);
        xyzz_add(val, other, val);
      }
      __syncthreads();
    }
    const u32 last_idx = (R - blk * 256u >= 256u) ? 255u : (R - blk * 256u - 1u);
    const bool run_end = valid && ((tid == last_idx) || (s_key[tid + (tid < 255u ? 1u : 0u)] != key));
    if (run_end) {
      const u32 first_key = s_key[0], last_key = s_key[last_idx];
      // One store site with a per-thread destination.  (A three-way if/else over
      // buckets / head record / tail record made hipcc 7.2 merge the stores behind
      // scalar base-pointer selects in divergent flow, and the multi-block case faulted
      // on gfx950; tests/test_gpu_msm.py::test_msm_multiblock_segscan pins this.)
      const bool to_bucket = is_final || (key != first_key && key != last_key);
      const bool is_head = !to_bucket && (key == first_key);
      const u32 slot = 2u * blk + (is_head ? 0u : 1u);
      u32 *dst = to_bucket ? buckets + (u64)key * XYZZ_WORDS : out_pt + (u64)slot * XYZZ_WORDS;
      if (!to_bucket) out_key[slot] = key;
      xyzz_store_g(dst, val);
      if (is_head && first_key == last_key) {       // the block is one single run: empty tail record
        xyzz inf;
        xyzz_set_inf(inf);
        out_key[slot + 1u] = key;
        xyzz_store_g(out_pt + (u64)(slot + 1u) * XYZZ_WORDS, inf);
      }
    }
    if (is_final || !ticket || nb > 128u) return;            // (block-uniform)
    // the last block to get here owns the final level: its records are this level's output, complete and visible
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(ticket, 1u) == nb - 1u) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    R = 2u * nb; nb = 1; blk = 0; is_final = true;
    in_key = out_key; in_pt = out_pt;
  }
}

// block-wide tree sum of one XYZZ value per thread (256 threads); result valid in thread 0
__device__ __forceinline__ void block_tree_sum(xyzz &val, u32 *s_val) {
  const u32 tid = threadIdx.x;
  for (u32 d = blockDim.x >> 1; d > 0; d >>= 1) {
    xyzz_store(s_val + tid * LDS_STRIDE, val);
    __syncthreads();
    if (tid < d) {
      xyzz other;
      xyzz_load(other, s_val + (tid + d) * LDS_STRIDE);
      xyzz_add(val, val, other);
    }
    __syncthreads();
  }
}

// ---- bucket reduction by two-position digit sums, sub-wave groups, cross-lane trees ----------
// For an array X[1..N] and a split s:  sum_b b X[b] = sum_lo lo D0[lo] + 2^s sum_hi hi D1[hi]
//   D0[lo] = sum of X[hi 2^s + lo] over hi      (lo = 1 .. 2^s - 1)
//   D1[hi] = sum of X[hi 2^s + lo] over lo      (hi = 1 .. N >> s)
// i.e. TWO additions per element; applying it twice takes a window's 2^(c-1) buckets to four
// arrays of <= 16 sums, which k_digit_final (the second application and the finish in one launch) turns into E[a][0..3].  Every sum belongs to a group of
// 2^gl_log lanes of one wave: each lane adds its share serially (about 8 elements), then a
// butterfly of __shfl_xor exchanges folds the group -- no LDS, no block barrier.
struct DigitJob {
  u32 in_off, in_stride;     // array a starts at record a * in_stride + in_off of X
  u32 N, s, type;            // entries, split bits, 0 = D0 (by lo) / 1 = D1 (by hi)
  u32 glanes, gpw;           // lanes per sum (1 .. 64, any value) and sums per wave = 64 / glanes: a sum never straddles two waves
  u32 nsums;                 // sums per array: 2^s - 1 (type 0) or N >> s (type 1)
  u32 out_off, out_stride;   // sum idx (1-based) of array a -> record a * out_stride + out_off + idx - 1 of D
  u32 blk0;                  // first block of this job
  u32 cnt;                   // arrays (windows) of this job
};
struct DigitJobs { DigitJob j[4]; u32 njobs, prio; };

__device__ __forceinline__ void xyzz_shfl_xor(xyzz &r, const xyzz &a, int mask) {
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.X.v[i] = (u32)__shfl_xor((int)a.X.v[i], mask, 64);
    r.Y.v[i] = (u32)__shfl_xor((int)a.Y.v[i], mask, 64);
    r.ZZ.v[i] = (u32)__shfl_xor((int)a.ZZ.v[i], mask, 64);
    r.ZZZ.v[i] = (u32)__shfl_xor((int)a.ZZZ.v[i], mask, 64);
  }
}
__device__ __forceinline__ void xyzz_shfl_down(xyzz &r, const xyzz &a, int d) {        // across the wave (lanes past the end read their own)
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.X.v[i] = (u32)__shfl_down((int)a.X.v[i], d, 64);
    r.Y.v[i] = (u32)__shfl_down((int)a.Y.v[i], d, 64);
    r.ZZ.v[i] = (u32)__shfl_down((int)a.ZZ.v[i], d, 64);
    r.ZZZ.v[i] = (u32)__shfl_down((int)a.ZZZ.v[i], d, 64);
  }
}
__device__ __forceinline__ void xyzz_shfl_down16(xyzz &r, const xyzz &a, int d) {      // within groups of 16 lanes
#pragma unroll
  for (int i = 0; i < 9; i++) {
    r.X.v[i] = (u32)__shfl_down((int)a.X.v[i], d, 16);
    r.Y.v[i] = (u32)__shfl_down((int)a.Y.v[i], d, 16);
    r.ZZ.v[i] = (u32)__shfl_down((int)a.ZZ.v[i], d, 16);
    r.ZZZ.v[i] = (u32)__shfl_down((int)a.ZZZ.v[i], d, 16);
  }
}

__global__ void __launch_bounds__(256) k_digit_sums(const u32 *__restrict__ X, u32 *__restrict__ D, DigitJobs jobs) {
  raise_priority(jobs.prio);
  u32 ji = 0;
  for (u32 k = 1; k < jobs.njobs; k++) if (blockIdx.x >= jobs.j[k].blk0) ji = k;
  const DigitJob J = jobs.j[ji];
  const u32 t = (blockIdx.x - J.blk0) * blockDim.x + threadIdx.x;
  const u32 GL = J.glanes, lane = t & 63u;
  const u32 grp = lane / GL, l = lane - grp * GL;
  const u32 sum_id = (t >> 6) * J.gpw + grp;
  const bool active = grp < J.gpw && sum_id < J.cnt * J.nsums;
  const u32 a = sum_id / J.nsums, idx = sum_id % J.nsums + 1u;
  xyzz acc;
  xyzz_set_inf(acc);
  const u32 *base = X + ((u64)a * J.in_stride + J.in_off) * XYZZ_WORDS;
  if (active) {
    if (J.type == 0) {
      for (u32 hi = l; ((hi << J.s) | idx) <= J.N; hi += GL) {
        xyzz x;
        xyzz_load_g(x, base + (u64)(((hi << J.s) | idx) - 1u) * XYZZ_WORDS);
        xyzz_add(acc, acc, x);
      }
    } else {
      for (u32 lo = l; lo < (1u << J.s) && ((idx << J.s) | lo) <= J.N; lo += GL) {
        xyzz x;
        xyzz_load_g(x, base + (u64)(((idx << J.s) | lo) - 1u) * XYZZ_WORDS);
        xyzz_add(acc, acc, x);
      }
    }
  }
  // tree over the group's lanes: the first step folds the lanes above the largest power of two below GL, the rest halve
  // the group's tree.  Powers of two: a butterfly of __shfl_xor (DPP for the short distances), every lane adding.  Other sizes: lanes l < d
  // take lane l + d, first the ones above the largest power of two below GL -- every lane still runs the addition (against the identity
  // where it has no partner): the addition under a partial EXEC mask measured 11 us slower per launch at c = 16 than the butterfly, this
  // form 4 (gpurun_out/r05_exp_tree.txt -> profiles/r05_reduction_tree_forms.txt)
  if ((GL & (GL - 1u)) == 0u) {
    for (u32 m = 1; m < GL; m <<= 1) {
      xyzz other;
      xyzz_shfl_xor(other, acc, (int)m);
      xyzz_add(acc, acc, other);
    }
  } else {
    u32 d = 1;
    while ((d << 1) < GL) d <<= 1;
    for (u32 width = GL; d > 0 && width > 1; width = d, d >>= 1) {
      xyzz other;
      xyzz_shfl_down(other, acc, (int)d);
      if (!(l < d && l + d < width)) xyzz_set_inf(other);
      xyzz_add(acc, acc, other);
    }
  }
  if (active && l == 0) xyzz_store_g(D + ((u64)a * J.out_stride + J.out_off + idx - 1u) * XYZZ_WORDS, acc);
}
// Stage 2 and the finish in ONE launch: block (array r of window a) = 16 groups of 16 lanes; group j adds up the (<= 16)
// elements of sum j + 1 of the array's digit job with a butterfly, parks it in LDS, and the first 16 lanes turn the (<= 16)
// sums into sum_d d * X[d]: inclusive suffix scan (4 steps), then the sum of all suffixes (4 steps).  E[a][r] out.  Saves a launch and the trip of 64 records per window
// through HBM on a path that is nothing but latency.
// (mixed widths: the windows from `top_w` on have 2B buckets and their own job set `jtop`, their arrays numbered from 0; top_w = ~0: none)
__global__ void __launch_bounds__(256) k_digit_final(const u32 *__restrict__ X, u32 *__restrict__ Eout, DigitJobs jobs, DigitJobs jtop, u32 top_w) {
  raise_priority(jobs.prio);
  __shared__ u32 s_val[16 * LDS_STRIDE];
  const u32 r = blockIdx.x & 3u, tid = threadIdx.x;
  const bool is_top = (blockIdx.x >> 2) >= top_w;
  const u32 a = is_top ? (blockIdx.x >> 2) - top_w : (blockIdx.x >> 2);
  const DigitJob J = is_top ? jtop.j[r] : jobs.j[r];
  const u32 idx = (tid >> 4) + 1u, l = tid & 15u;
  xyzz acc;
  xyzz_set_inf(acc);
  if (idx <= J.nsums) {
    const u32 *base = X + ((u64)a * J.in_stride + J.in_off) * XYZZ_WORDS;
    if (J.type == 0) {
      for (u32 hi = l; ((hi << J.s) | idx) <= J.N; hi += 16u) {
        xyzz x;
        xyzz_load_g(x, base + (u64)(((hi << J.s) | idx) - 1u) * XYZZ_WORDS);
        xyzz_add(acc, acc, x);
      }
    } else {
      for (u32 lo = l; lo < (1u << J.s) && ((idx << J.s) | lo) <= J.N; lo += 16u) {
        xyzz x;
        xyzz_load_g(x, base + (u64)(((idx << J.s) | lo) - 1u) * XYZZ_WORDS);
        xyzz_add(acc, acc, x);
      }
    }
  }
  for (u32 m = 1; m < 16u; m <<= 1) {
    xyzz other;
    xyzz_shfl_xor(other, acc, (int)m);
    xyzz_add(acc, acc, other);
  }
  if (l == 0) xyzz_store(s_val + (idx - 1u) * LDS_STRIDE, acc);
  __syncthreads();
  if (tid >= 64u) return;
  xyzz val;
  xyzz_set_inf(val);
  if (tid < J.nsums) xyzz_load(val, s_val + tid * LDS_STRIDE);            // lane d - 1 holds X[d]; lanes 16 .. 63 idle along
  for (u32 d = 1; d < 16; d <<= 1) {
    xyzz other;
    xyzz_shfl_down16(other, val, (int)d);
    if (l + d < 16u) xyzz_add(val, val, other);
  }
  for (u32 m = 1; m < 16; m <<= 1) {
    xyzz other;
    xyzz_shfl_xor(other, val, (int)m);
    xyzz_add(val, val, other);
  }
  if (tid == 0) xyzz_store_g(Eout + (u64)blockIdx.x * XYZZ_WORDS, val);
}

// ---- a point addition on FOUR lanes -------------------------------------------------------------------------------
// The scan / reduction stages are chains of dependent general additions run by a handful of waves: what they cost is the
// LENGTH of one addition (14 multiplications one after the other, ~1 950 instructions, 5.5 us), not its work.  Here a point
// lives in a quad of lanes, lane q holding coordinate q (0: X, 1: Y, 2: ZZ, 3: ZZZ) -- exactly the four 9-word groups of the
// 144-byte record -- and the 14 multiplications run as FOUR levels of four (operands exchanged inside the quad with DPP):
//   level 1   U1 = X1 ZZ2 | S1 = Y1 ZZZ2 | U2 = X2 ZZ1 | S2 = Y2 ZZZ1        (own a times the b of lane q ^ 2)
//             P = U2 - U1 (lanes 0, 2), R = S2 - S1 (lanes 1, 3), carried: the zero tests of the exceptional cases
//   level 2   PP = P^2 | RR = R^2 | zz = ZZ1 ZZ2 | zzz = ZZZ1 ZZZ2
//   level 3   PPP = P PP | Q = U1 PP | ZZ3 = zz PP | -
//             X3 = RR - PPP - 2Q (lane 1)
//   level 4   S1 PPP | R (Q - X3) | - | ZZZ3 = zzz PPP ;   Y3 = R (Q - X3) - S1 PPP (lane 1)
// ~950 instructions per lane instead of ~1 950.  Same formulas, same field routines, the same projective result as xyzz_add.
// Identity operands and P = -Q are selects; P = Q (a doubling) gathers the point into every lane and runs xyzz_dbl there.
#define QP_SWAP2 0x4E        // quad_perm [2, 3, 0, 1]
#define QP_B0 0x00           // every lane reads lane 0 of its quad
#define QP_B1 0x55
#define QP_B2 0xAA
#define QP_B3 0xFF
// (The exchanged value is pinned in its own register: left to itself the compiler folds the DPP move into the instruction that
// uses it, and for a subtraction with the DPP operand on the right it emitted the operands REVERSED -- r4[0] - r4[q] instead of
// r4[q] - r4[0] in the last line of quad_add; tests/test_gpu_field.py::test_four_lane_point_addition... caught it.)
template <int CTRL> __device__ __forceinline__ void fe_quad(fe &r, const fe &s) {
#pragma unroll
  for (int k = 0; k < 9; k++) {
    u32 v = (u32)__builtin_amdgcn_update_dpp(0, (int)s.v[k], CTRL, 0xF, 0xF, false);
    asm volatile("" : "+v"(v));
    r.v[k] = v;
  }
}
template <int CTRL> __device__ __forceinline__ u32 u32_quad(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ void fe_select(fe &r, bool c, const fe &x, const fe &y) {      // c ? x : y
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = c ? x.v[k] : y.v[k];
}
// a <- a + b; a, b: this lane's coordinate of the two points, q = lane & 3.  Every lane of the quad must call it.
__device__ __forceinline__ void quad_add(fe &a, const fe &b, u32 q) {
  const bool lo = q < 2u;
  const u32 infA = u32_quad<QP_B2>(fe_is_zero_tight(a) ? 1u : 0u), infB = u32_quad<QP_B2>(fe_is_zero_tight(b) ? 1u : 0u);
  fe bx, r1, o1, t, u;
  fe_quad<QP_SWAP2>(bx, b);
  fe_mul(r1, a, bx);                                   // U1 | S1 | U2 | S2
  fe_quad<QP_SWAP2>(o1, r1);
  fe_select(u, lo, o1, r1);                            // U2 | S2 | U2 | S2
  fe_select(t, lo, r1, o1);                            // U1 | S1 | U1 | S1
  fe_sub(t, u, t);
  fe_carry(t, t);                                      // P | R | P | R, tight
  const u32 z = fe_is_zero_tight(t) ? 1u : 0u;
  const u32 pz = u32_quad<QP_B0>(z), rz = u32_quad<QP_B1>(z);
  fe opa, opb, r2, r3, r4, ppb, tmp;
  fe_select(opa, lo, t, a);
  fe_select(opb, lo, t, b);
  fe_mul(r2, opa, opb);                                // PP | RR | zz | zzz
  fe_quad<QP_B0>(ppb, r2);
  fe_quad<QP_B0>(tmp, r1);                             // U1
  fe_select(opa, q == 0u, t, r2);
  fe_select(opa, q == 1u, tmp, opa);                   // P | U1 | zz | zzz
  fe_mul(r3, opa, ppb);                                // PPP | Q | ZZ3 | (unused)
  fe pppb, x3, qx;
  fe_quad<QP_B0>(pppb, r3);
  fe_sub(x3, r2, pppb);                                // lane 1: RR - PPP + 2p
  fe_add(tmp, r3, r3);
  fe_sub_m2(x3, x3, tmp);                              // - 2Q + 4p          (limbs < 2^32: 1 + 2 + 3 magnitudes)
  fe_carry(x3, x3);                                    // lane 1: X3, tight
  fe_sub(qx, r3, x3);                                  // lane 1: Q - X3 + 2p, magnitude 3
  fe_quad<QP_B1>(tmp, r1);                             // S1
  fe_select(opa, q == 0u, tmp, r2);                    // S1 | . | . | zzz
  fe_select(opa, q == 1u, t, opa);                     // S1 | R | zz | zzz
  fe_select(opb, q == 1u, qx, pppb);                   // PPP | Q - X3 | PPP | PPP      (lane 0's own r3 IS pppb)
  fe_mul(r4, opa, opb);                                // S1 PPP | R (Q - X3) | (unused) | ZZZ3
  fe y3;
  fe_quad<QP_B0>(tmp, r4);
  fe_sub(y3, r4, tmp);
  fe_carry(y3, y3);                                    // lane 1: Y3, tight
  fe x3b, res;
  fe_quad<QP_B1>(x3b, x3);
  fe_select(res, q == 0u, x3b, y3);
  fe_select(res, q == 2u, r3, res);
  fe_select(res, q == 3u, r4, res);
  // exceptional cases (uniform inside the quad)
  const bool dbl = !infA && !infB && pz && rz;
  if (__ballot(dbl)) {                                 // some quad of the wave doubles: every lane gathers its point and doubles it
    xyzz A;
    fe_quad<QP_B0>(A.X, a); fe_quad<QP_B1>(A.Y, a); fe_quad<QP_B2>(A.ZZ, a); fe_quad<QP_B3>(A.ZZZ, a);
    if (dbl) {
      xyzz D;
      xyzz_dbl(D, A);
      fe_select(res, q == 0u, D.X, D.Y);
      fe_select(res, q == 2u, D.ZZ, res);
      fe_select(res, q == 3u, D.ZZZ, res);
    }
  }
  fe zero;
  fe_set_zero(zero);
  fe_select(res, pz && !rz, zero, res);                // P = -Q: the identity (the all-zero record)
  fe_select(res, infB != 0u, a, res);
  fe_select(res, infA != 0u, b, res);
  a = res;
}

// k_digit_final on quads: block (array r of window a) = 16 waves; wave j adds up the (<= 16) elements of sum j + 1, one element
// per QUAD of lanes, with a butterfly of quad additions, parks it in LDS, and the first wave turns the (<= 16) sums into
// sum_d d * X[d] (suffix scan over quads, then the sum of all suffixes).  The same 12 dependent additions, each 4 levels deep.
__global__ void __launch_bounds__(1024) k_digit_final_quad(const u32 *__restrict__ X, u32 *__restrict__ Eout, DigitJobs jobs, DigitJobs jtop, u32 top_w) {
  __shared__ u32 s_val[16 * XYZZ_WORDS];
  const u32 r = blockIdx.x & 3u, tid = threadIdx.x;
  const bool is_top = (blockIdx.x >> 2) >= top_w;
  const u32 a_idx = is_top ? (blockIdx.x >> 2) - top_w : (blockIdx.x >> 2);
  const DigitJob J = is_top ? jtop.j[r] : jobs.j[r];
  const u32 wave = tid >> 6, lane = tid & 63u, e = lane >> 2, q = lane & 3u;
  const u32 idx = wave + 1u;
  fe a;
  fe_set_zero(a);
  if (idx <= J.nsums) {
    const u32 *base = X + ((u64)a_idx * J.in_stride + J.in_off) * XYZZ_WORDS;
    u32 rec = 0;                                            // 1-based record of this quad's element, 0 = none
    if (J.type == 0) { const u32 b = (e << J.s) | idx; if (b <= J.N) rec = b; }
    else { const u32 b = (idx << J.s) | e; if (e < (1u << J.s) && b <= J.N) rec = b; }
    if (rec) {
      const u32 *p = base + (u64)(rec - 1u) * XYZZ_WORDS + q * 9u;
#pragma unroll
      for (int k = 0; k < 9; k++) a.v[k] = p[k];
    }
  }
#pragma unroll 1
  for (u32 m = 4; m < 64u; m <<= 1) {
    fe b;
#pragma unroll
    for (int k = 0; k < 9; k++) b.v[k] = (u32)__shfl_xor((int)a.v[k], (int)m, 64);
    quad_add(a, b, q);
  }
  if (e == 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) s_val[wave * XYZZ_WORDS + q * 9u + k] = a.v[k];
  }
  __syncthreads();
  if (wave) return;
  fe_set_zero(a);
  if (e < J.nsums) {                                        // quad d holds X[d + 1]
#pragma unroll
    for (int k = 0; k < 9; k++) a.v[k] = s_val[e * XYZZ_WORDS + q * 9u + k];
  }
#pragma unroll 1
  for (u32 d = 1; d < 16u; d <<= 1) {                       // inclusive suffix scan over the 16 quads
    fe b;
#pragma unroll
    for (int k = 0; k < 9; k++) { const u32 v = (u32)__shfl_down((int)a.v[k], (int)(4u * d), 64); b.v[k] = (e + d < 16u) ? v : 0u; }
    quad_add(a, b, q);
  }
#pragma unroll 1
  for (u32 m = 4; m < 64u; m <<= 1) {                       // the sum of all suffixes
    fe b;
#pragma unroll
    for (int k = 0; k < 9; k++) b.v[k] = (u32)__shfl_xor((int)a.v[k], (int)m, 64);
    quad_add(a, b, q);
  }
  if (e == 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) Eout[(u64)blockIdx.x * XYZZ_WORDS + q * 9u + k] = a.v[k];
  }
}

// k_digit_final_quad with its first phase spread over the chip (round 5).  In the kernel above the 16 waves of a block share ONE CU -- four
// per SIMD, each running the same 4 dependent quad additions -- while 68 blocks leave three quarters of the CUs idle: the phase
// costs 4 waves x 4 additions of issue time (~30 of the kernel's 49 us) on a path that is nothing but latency.  Here every sum is a
// wave of its own (<= 1 088 waves: about one per SIMD) that parks its result in `F` (16 records per array), and the last 8 additions
// over an array's <= 16 sums run in a wave per array.  MODE 1 + MODE 2: two launches (18 + 21 us, the default); MODE 0: one launch,
// the wave that draws the array's last ticket runs the finish -- measured 60 us: the device-scope fences around the ticket cost more
// than a launch (as in k_coarse_hist's last-block scan).  `ticket` (MODE 0): one word per array, zero on entry, left zero.
template <int MODE> __global__ void __launch_bounds__(256) k_digit_final_spread(const u32 *__restrict__ X, u32 *F, u32 *ticket, u32 *__restrict__ Eout, DigitJobs jobs,
                                                                               DigitJobs jtop, u32 top_w) {
  raise_priority(jobs.prio);
  const u32 gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);            // blocks of one wave or of four
  const u32 arr = MODE == 2 ? gw : gw >> 4, wave = MODE == 2 ? 0u : gw & 15u, r = arr & 3u;
  const bool is_top = (arr >> 2) >= top_w;
  const u32 a_idx = is_top ? (arr >> 2) - top_w : (arr >> 2);
  const DigitJob J = is_top ? jtop.j[r] : jobs.j[r];
  const u32 lane = threadIdx.x & 63u, e = lane >> 2, q = lane & 3u;
  const u32 idx = wave + 1u;
  fe a;
  if (MODE != 2 && idx <= J.nsums) {                        // (wave-uniform)
    fe_set_zero(a);
    const u32 *base = X + ((u64)a_idx * J.in_stride + J.in_off) * XYZZ_WORDS;
    u32 rec = 0;                                            // 1-based record of this quad's element, 0 = none
    if (J.type == 0) { const u32 b = (e << J.s) | idx; if (b <= J.N) rec = b; }
    else { const u32 b = (idx << J.s) | e; if (e < (1u << J.s) && b <= J.N) rec = b; }
    if (rec) {
      const u32 *p = base + (u64)(rec - 1u) * XYZZ_WORDS + q * 9u;
#pragma unroll
      for (int k = 0; k < 9; k++) a.v[k] = p[k];
    }
#pragma unroll 1
    for (u32 m = 4; m < 64u; m <<= 1) {
      fe b;
#pragma unroll
      for (int k = 0; k < 9; k++) b.v[k] = (u32)__shfl_xor((int)a.v[k], (int)m, 64);
      quad_add(a, b, q);
    }
    if (e == 0) {
#pragma unroll
      for (int k = 0; k < 9; k++) F[((u64)arr * 16u + wave) * XYZZ_WORDS + q * 9u + k] = a.v[k];
    }
  }
  if (MODE == 1) return;
  if (MODE == 0) {
    __threadfence();
    u32 t = 0;
    if (lane == 0) t = atomicAdd(ticket + arr, 1u);
    t = (u32)__shfl((int)t, 0, 64);
    if (t != 15u) return;
    __threadfence();                                        // the other 15 waves' sums are complete and visible
  }
  fe_set_zero(a);
  if (e < J.nsums) {                                        // quad d holds X[d + 1]
#pragma unroll
    for (int k = 0; k < 9; k++) a.v[k] = F[((u64)arr * 16u + e) * XYZZ_WORDS + q * 9u + k];
  }
#pragma unroll 1
  for (u32 d = 1; d < 16u; d <<= 1) {                       // inclusive suffix scan over the 16 quads
    fe b;
#pragma unroll
    for (int k = 0; k < 9; k++) { const u32 v = (u32)__shfl_down((int)a.v[k], (int)(4u * d), 64); b.v[k] = (e + d < 16u) ? v : 0u; }
    quad_add(a, b, q);
  }
#pragma unroll 1
  for (u32 m = 4; m < 64u; m <<= 1) {                       // the sum of all suffixes
    fe b;
#pragma unroll
    for (int k = 0; k < 9; k++) b.v[k] = (u32)__shfl_xor((int)a.v[k], (int)m, 64);
    quad_add(a, b, q);
  }
  if (e == 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) Eout[(u64)arr * XYZZ_WORDS + q * 9u + k] = a.v[k];
  }
  if (MODE == 0 && lane == 0) ticket[arr] = 0u;
}

// self-test hook (bpmi_debug_quad_add): out[i] = a[i] + b[i] for XYZZ records, one quad per pair
__global__ void __launch_bounds__(256) k_debug_quad_add(const u32 *__restrict__ ra, const u32 *__restrict__ rb, u32 n, u32 *__restrict__ out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  u32 i = t >> 2;
  const u32 q = t & 3u;
  const bool live = i < n;
  if (!live) i = n - 1u;                        // whole quads stay active for the lane exchanges
  fe a, b;
#pragma unroll
  for (int k = 0; k < 9; k++) { a.v[k] = ra[(u64)i * XYZZ_WORDS + q * 9u + k]; b.v[k] = rb[(u64)i * XYZZ_WORDS + q * 9u + k]; }
  quad_add(a, b, q);
  if (live) {
#pragma unroll
    for (int k = 0; k < 9; k++) out[(u64)i * XYZZ_WORDS + q * 9u + k] = a.v[k];
  }
}

// ---- bucket reduction for small windows (B <= 256): one block of B threads per window
// computes sum_b b * B[w][b] directly as sum_j Suffix_j (inclusive suffix scan + tree sum,
// 2 log2(B) dependent additions) -- shorter than the digit-sum stages when B is small.
// Eout[w] then has nv = 1.
__global__ void __launch_bounds__(256) k_window_weighted_small(MsmGeom g, const u32 *__restrict__ buckets, u32 *__restrict__ Eout) {
  raise_priority(g.prio & PRIO_FINISH);
  __shared__ u32 s_val[256 * LDS_STRIDE];
  const u32 tid = threadIdx.x, w = blockIdx.x;
  xyzz val;
  if (tid < g.B) xyzz_load_g(val, buckets + ((u64)w * g.B + tid) * XYZZ_WORDS);      // bucket b = tid + 1
  else xyzz_set_inf(val);
  for (u32 d = 1; d < g.B; d <<= 1) {
    xyzz_store(s_val + tid * LDS_STRIDE, val);
    __syncthreads();
    if (tid + d < g.B) {
      xyzz other;
      xyzz_load(other, s_val + (tid + d) * LDS_STRIDE);
      xyzz_add(val, val, other);
    }
    __syncthreads();
  }
  block_tree_sum(val, s_val);
  if (tid == 0) xyzz_store_g(Eout + (u64)w * XYZZ_WORDS, val);
}

// the same on quads of lanes (quad_add): 4 B threads, bucket b = (tid >> 2) + 1, lane tid & 3 holds one coordinate
__global__ void __launch_bounds__(1024) k_window_weighted_small_quad(MsmGeom g, const u32 *__restrict__ buckets, u32 *__restrict__ Eout) {
  __shared__ u32 s_val[256 * XYZZ_WORDS];
  const u32 tid = threadIdx.x, w = blockIdx.x, rec = tid >> 2, q = tid & 3u;
  fe a;
  fe_set_zero(a);
  if (rec < g.B) {
    const u32 *p = buckets + ((u64)w * g.B + rec) * XYZZ_WORDS + q * 9u;
#pragma unroll
    for (int k = 0; k < 9; k++) a.v[k] = p[k];
  }
  u32 *mine = s_val + rec * XYZZ_WORDS + q * 9u;
#pragma unroll 1
  for (u32 d = 1; d < g.B; d <<= 1) {                       // inclusive suffix scan
#pragma unroll
    for (int k = 0; k < 9; k++) mine[k] = a.v[k];
    __syncthreads();
    fe b;
#pragma unroll
    for (int k = 0; k < 9; k++) b.v[k] = (rec + d < g.B) ? mine[d * XYZZ_WORDS + k] : 0u;
    __syncthreads();
    quad_add(a, b, q);
  }
#pragma unroll 1
  for (u32 d = (blockDim.x >> 3); d > 0; d >>= 1) {         // sum of all suffixes: records [0, 2d) -> [0, d)
#pragma unroll
    for (int k = 0; k < 9; k++) mine[k] = a.v[k];
    __syncthreads();
    fe b;
#pragma unroll
    for (int k = 0; k < 9; k++) b.v[k] = (rec < d) ? mine[d * XYZZ_WORDS + k] : 0u;
    __syncthreads();
    quad_add(a, b, q);
  }
  if (rec == 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) Eout[(u64)w * XYZZ_WORDS + q * 9u + k] = a.v[k];
  }
}

// ---- small MSMs (n <= a few thousand): latency, not throughput, is what counts ------------
// One dependent point addition costs 5-8 us when a wave issues alone, so the bucket pipeline
// (sort, L sequential additions per thread, segmented scan, bucket reduction: >= 60
// dependent additions and a dozen launches) bottoms out near 0.5 ms however small n is.
// Here window w (8 bits, signed digits) is ONE block: every thread multiplies its point by
// the digit |d| <= 128 (at most 7 doublings + 7 additions, Jacobian), and the block adds the
// terms up with shuffle butterflies: depth ~ 12 + log2(n), one or two launches, E[w] to the same tail.
#define SMALL_C 8
// grid = (W, S): block (w, s) covers the points i = s * blockDim + tid (+ k * S * blockDim) and
// writes its partial sum to out[w * S + s]; k_small_combine adds the S partials of a window.
__device__ __forceinline__ void msm_small_block(const Segs &segs, const MsmGeom &g, u32 *__restrict__ out, u32 *s_val, u32 S) {
  const u32 w = blockIdx.x, tid = threadIdx.x;
  xyzz acc;
  xyzz_set_inf(acc);
  for (u32 i = blockIdx.y * blockDim.x + tid; i < g.n; i += S * blockDim.x) {
    u32 b = 0, sign = 0;
    for_each_digit(segs, g, i, [&](u32 ww, u32 bb, u32 sg) { if (ww == w) { b = bb; sign = sg; } });
    affine P;
    load_affine(P, seg_point(segs, i));
    if (b == 0 || affine_is_inf(P)) continue;
    jac q;
    q.X = P.x; q.Y = P.y; fe_set_one(q.Z);
    int top = 31 - __clz((int)b);
    for (int bit = top - 1; bit >= 0; bit--) {
      jac_dbl(q, q);
      if ((b >> bit) & 1u) jac_madd(q, P.x, P.y);
    }
    xyzz t;
    t.X = q.X;
    if (sign) { fe ny; fe_neg(ny, q.Y); fe_carry(t.Y, ny); } else t.Y = q.Y;
    fe_sqr(t.ZZ, q.Z);
    fe_mul(t.ZZZ, t.ZZ, q.Z);
    xyzz_add(acc, acc, t);
  }
  for (u32 m = 1; m < 64u; m <<= 1) {
    xyzz other;
    xyzz_shfl_xor(other, acc, (int)m);
    xyzz_add(acc, acc, other);
  }
  const u32 wave = tid >> 6, nwaves = blockDim.x >> 6;
  if ((tid & 63u) == 0) xyzz_store(s_val + wave * LDS_STRIDE, acc);
  __syncthreads();
  if (wave == 0) {
    if (tid < nwaves) xyzz_load(acc, s_val + tid * LDS_STRIDE); else xyzz_set_inf(acc);
    for (u32 m = 1; m < 4u; m <<= 1) {
      xyzz other;
      xyzz_shfl_xor(other, acc, (int)m);
      xyzz_add(acc, acc, other);
    }
    if (tid == 0) xyzz_store_g(out + ((u64)w * S + blockIdx.y) * XYZZ_WORDS, acc);
  }
}
__global__ void __launch_bounds__(256) k_msm_small(Segs segs, MsmGeom g, u32 *__restrict__ out) {
  __shared__ u32 s_val[4 * LDS_STRIDE];
  msm_small_block(segs, g, out, s_val, gridDim.y);
}
// two independent small MSMs in ONE launch (blockIdx.z = job): the L and R of a late inner-product round, the A / S and T1 / T2
// pairs of a 64-bit range proof -- on this path a launch and a stream hand-over are a tenth of the MSM
struct SmallPair { Segs segs[2]; MsmGeom g[2]; u32 *out[2]; u32 S[2]; };
__global__ void __launch_bounds__(256) k_msm_small_pair(SmallPair p) {
  __shared__ u32 s_val[4 * LDS_STRIDE];
  const u32 job = blockIdx.z;
  if (blockIdx.y >= p.S[job]) return;                     // (the grid is sized for the larger job)
  msm_small_block(p.segs[job], p.g[job], p.out[job], s_val, p.S[job]);
}
// E[w] = sum of the S (<= 64) partials of window w: one wave per window
__global__ void __launch_bounds__(64) k_small_combine(const u32 *__restrict__ part, u32 S, u32 *__restrict__ E) {
  const u32 w = blockIdx.x, l = threadIdx.x;
  xyzz acc;
  if (l < S) xyzz_load_g(acc, part + ((u64)w * S + l) * XYZZ_WORDS); else xyzz_set_inf(acc);
  for (u32 m = 1; m < S; m <<= 1) {
    xyzz other;
    xyzz_shfl_xor(other, acc, (int)m);
    xyzz_add(acc, acc, other);
  }
  if (l == 0) xyzz_store_g(E + (u64)w * XYZZ_WORDS, acc);
}

struct CombinePair { const u32 *part[2]; u32 S[2]; u32 *E[2]; };
__global__ void __launch_bounds__(64) k_small_combine_pair(CombinePair c) {
  const u32 w = blockIdx.x, l = threadIdx.x, job = blockIdx.y, S = c.S[job];
  if (S <= 1) return;                                      // that job's k_msm_small blocks wrote E themselves
  xyzz acc;
  if (l < S) xyzz_load_g(acc, c.part[job] + ((u64)w * S + l) * XYZZ_WORDS); else xyzz_set_inf(acc);
  for (u32 m = 1; m < S; m <<= 1) {
    xyzz other;
    xyzz_shfl_xor(other, acc, (int)m);
    xyzz_add(acc, acc, other);
  }
  if (l == 0) xyzz_store_g(c.E[job] + (u64)w * XYZZ_WORDS, acc);
}

// ---- mid-size MSMs in ONE launch: a whole window's bucket method in one block (round 4) -----------------------------------------
// Between the one-launch kernel above (work ~300 point operations per pair: every window multiplies every point by its digit
// from scratch -- fine for hundreds of pairs, 275 us for the two 4 097-pair MSMs of a late inner-product round) and the bucket
// pipeline (34 operations per pair, but a dozen dependent launches: 0.26-0.3 ms however small the input) sits this kernel: block
// (w, job) runs the bucket method for window w of one MSM ENTIRELY in LDS --
//   1. every thread recodes its scalars and keeps window w's signed 7-bit digit (u16 per pair in LDS), counting the 64 buckets;
//   2. the bucket offsets, and lanes handed out in proportion to the bucket sizes (q entries per lane, q = E / (512 - 64): a
//      bucket of ANY size is shared evenly -- all pairs in one bucket, the bit vectors of a range proof, cost what uniform digits cost);
//   3. a counting sort of the (pair, sign) entries by bucket; every lane adds its share of one bucket (strided within the
//      bucket, the next point in flight while one is added);
//   4. a block-wide segmented scan over the lanes (keys = buckets, already in order; stops when no lane finds its key at the
//      current distance: two or three steps for uniform digits) leaves the bucket sums in LDS;
//   5. sum_b b X[b] on quads of lanes (suffix scan + tree, 12 dependent four-lane additions), one XYZZ record out.
// 37 windows of 7 bits; the two MSMs of a pair are blockIdx.y.  ~0.1 ms for a pair of 4 097-pair MSMs.
#define MID_C 7
#define MID_B 64                 // 2^(MID_C - 1)
#define MID_THREADS 512
#define MID_NMAX 8448            // pairs per MSM: the digit and entry arrays live in LDS (139 KB of the 160 KB a gfx950 CU has: this kernel does not build for earlier CDNA parts)
struct MidPair { Segs segs[2]; MsmGeom g[2]; u32 *E[2]; };
__global__ void __launch_bounds__(MID_THREADS) k_msm_mid(MidPair p) {
  __shared__ u32 s_cnt[MID_B + 2], s_off[MID_B + 2], s_cur[MID_B + 2], s_lane0[MID_B + 2];
  __shared__ unsigned short s_dig[MID_NMAX];
  __shared__ u32 s_ent[MID_NMAX];
  __shared__ u32 s_key[MID_THREADS];
  __shared__ u32 s_val[MID_THREADS * LDS_STRIDE];
  __shared__ u32 s_bkt[MID_B * XYZZ_WORDS];
  const u32 job = blockIdx.y, w = blockIdx.x, tid = threadIdx.x;
  const MsmGeom g = p.g[job];
  const u32 n = g.n;
  // round 5: a window's pairs may be split over gridDim.z blocks (`parts`): block z runs the whole method on pairs [i0, i1) and leaves its
  // own window sum -- sum_b b X[b] is linear in the buckets -- in E[w][z]; the host tail adds the parts of a window at the same bit offset.
  // 74 blocks of a pair of 8 193-pair MSMs used 74 of 256 CUs with 19 additions per lane; three parts: 222 blocks, 7 per lane
  const u32 parts = gridDim.z, part = blockIdx.z;
  const u32 i0 = (u32)((u64)n * part / parts), i1 = (u32)((u64)n * (part + 1u) / parts);
  if (tid < MID_B + 2) s_cnt[tid] = 0;
  for (u32 i = tid; i < MID_B * XYZZ_WORDS; i += MID_THREADS) s_bkt[i] = 0;          // empty buckets are the identity
  __syncthreads();
  // 1. digits of window w.  The carry chain of the signed recoding (for_each_digit: 37 dependent steps per scalar, and every one
  // of the 37 blocks of an MSM would walk it for every scalar) is replaced by its closed form: with K = 64 sum_{j < 36} 2^(7 j),
  // the 7-bit field j of s + K is d_j + 64 for digits d_j in [-64, 63] that represent the same s (the top field, bits 252 .. 255,
  // stays as it is: s < 2^255 and K < 2^252, so it is at most 9) -- one 256-bit addition and a bit-field extraction per scalar.
  for (u32 i = i0 + tid; i < i1; i += MID_THREADS) {
    sc v;
    const bool neg = load_digit_source(v, p.segs[job], i);       // s or q - s (< 2^255), and whether the point is negated
    {
      // K = 0x0408102040810204081020408102040810204081020408102040810204081020 40 (bit 6 + 7 j set, j < 36)
      u64 cy = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        u32 kw = 0;
#pragma unroll
        for (int j = 0; j < 36; j++) { const int bit = 6 + 7 * j; if ((bit >> 5) == k) kw |= 1u << (bit & 31); }
        cy += (u64)v.v[k] + kw;
        v.v[k] = (u32)cy;
        cy >>= 32;
      }
    }
    const u32 pos = MID_C * w, wi = pos >> 5, sh = pos & 31u;
    u32 lo_w = 0, hi_w = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { if ((u32)k == wi) lo_w = v.v[k]; if ((u32)k == wi + 1u) hi_w = v.v[k]; }
    const u32 field = (u32)((((u64)hi_w << 32) | lo_w) >> sh) & ((1u << MID_C) - 1u);
    int d = (int)field - (w + 1u < g.W ? (int)MID_B : 0);
    const u32 b = (u32)(d < 0 ? -d : d);
    const u32 sign = (d < 0 ? 1u : 0u) ^ (neg ? 1u : 0u);
    s_dig[i - i0] = (unsigned short)(b | ((b ? sign : 0u) << 15));
    if (b) atomicAdd(&s_cnt[b], 1u);
  }
  __syncthreads();
  // 2. offsets and lanes (65 values: one thread)
  if (tid == 0) {
    u32 E = 0;
    for (u32 b = 1; b <= MID_B; b++) { s_off[b] = E; s_cur[b] = E; E += s_cnt[b]; }
    s_off[MID_B + 1] = E;
    const u32 q = (E + (MID_THREADS - MID_B) - 1) / (MID_THREADS - MID_B);                 // entries per lane (0 when E = 0)
    u32 L = 0;
    for (u32 b = 1; b <= MID_B; b++) { s_lane0[b] = L; L += q ? (s_cnt[b] + q - 1) / q : 0u; }
    s_lane0[MID_B + 1] = L;                                                            // <= E / q + 64 <= 512
  }
  __syncthreads();
  // 3. counting sort, then every lane adds its share of its bucket
  for (u32 i = i0 + tid; i < i1; i += MID_THREADS) {
    const u32 d = s_dig[i - i0], b = d & 0x7FFFu;
    if (b) s_ent[atomicAdd(&s_cur[b], 1u)] = i | ((d >> 15) << 31);
  }
  __syncthreads();
  const u32 TL = s_lane0[MID_B + 1];
  u32 key = 0xFFFFFFFFu;
  xyzz acc;
  xyzz_set_inf(acc);
  if (tid < TL) {
    u32 lo = 1, hi = MID_B;                                     // largest b with s_lane0[b] <= tid (buckets without lanes share their successor's start)
    while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (s_lane0[mid] <= tid) lo = mid; else hi = mid - 1; }
    // several empty buckets may start at the same lane: the one that owns it is the last of them (the only one with lanes)
    const u32 b = lo;
    key = b;
    const u32 lanes = s_lane0[b + 1] - s_lane0[b], j = tid - s_lane0[b];
    const u32 beg = s_off[b], end = s_off[b + 1];
    u32 pos = beg + j;
    u32 w_next[16];
    u32 e_next = 0;
    if (pos < end) { e_next = s_ent[pos]; load_entry_point<false>(w_next, p.segs[job], e_next & 0x7FFFFFFFu); }
    while (pos < end) {
      const u32 e = e_next;
      affine P;
      affine_from_words(P, w_next);
      pos += lanes;
      if (pos < end) { e_next = s_ent[pos]; load_entry_point<false>(w_next, p.segs[job], e_next & 0x7FFFFFFFu); }
      xyzz_madd_signed(acc, P, (e >> 31) != 0);
    }
  }
  // 4. segmented scan over the lanes (keys ascending); the last lane of a bucket leaves its sum in LDS
  s_key[tid] = key;
  __syncthreads();
  for (u32 d = 1; d < MID_THREADS; d <<= 1) {
    const bool act = tid >= d && key != 0xFFFFFFFFu && s_key[tid - d] == key;
    if (!__syncthreads_or(act)) break;
    xyzz_store(s_val + tid * LDS_STRIDE, acc);
    __syncthreads();
    if (act) {
      xyzz other;
      xyzz_load(other, s_val + (tid - d) * LDS_STRIDE);
      xyzz_add(acc, other, acc);
    }
    __syncthreads();
  }
  if (key != 0xFFFFFFFFu && (tid + 1 == MID_THREADS || s_key[tid + 1] != key)) xyzz_store(s_bkt + (key - 1u) * XYZZ_WORDS, acc);
  __syncthreads();
  // 5. sum_b b X[b] over the 64 buckets: bucket b = rec + 1 on the quad of lanes 4 rec .. 4 rec + 3 -- the first four waves; the
  // other four only keep the block's barriers company (`live` is wave-uniform: a barrier that part of the block never reaches is
  // undefined behaviour in the programming model, whatever today's hardware makes of ended waves)
  const bool live = tid < 4 * MID_B;
  const u32 rec = tid >> 2, q4 = tid & 3u;
  fe a;
  fe_set_zero(a);
  if (live) {
#pragma unroll
    for (int k = 0; k < 9; k++) a.v[k] = s_bkt[rec * XYZZ_WORDS + q4 * 9u + k];
  }
  u32 *mine = s_val + (live ? rec : 0u) * XYZZ_WORDS + q4 * 9u;                // (s_val is free again)
#pragma unroll 1
  for (u32 d = 1; d < MID_B; d <<= 1) {                           // inclusive suffix scan
    if (live) {
#pragma unroll
      for (int k = 0; k < 9; k++) mine[k] = a.v[k];
    }
    __syncthreads();
    fe b;
    fe_set_zero(b);
    if (live) {
#pragma unroll
      for (int k = 0; k < 9; k++) b.v[k] = (rec + d < MID_B) ? mine[d * XYZZ_WORDS + k] : 0u;
    }
    __syncthreads();
    if (live) quad_add(a, b, q4);
  }
#pragma unroll 1
  for (u32 d = MID_B >> 1; d > 0; d >>= 1) {                      // sum of all suffixes: records [0, 2d) -> [0, d)
    if (live) {
#pragma unroll
      for (int k = 0; k < 9; k++) mine[k] = a.v[k];
    }
    __syncthreads();
    fe b;
    fe_set_zero(b);
    if (live) {
#pragma unroll
      for (int k = 0; k < 9; k++) b.v[k] = (rec < d) ? mine[d * XYZZ_WORDS + k] : 0u;
    }
    __syncthreads();
    if (live) quad_add(a, b, q4);
  }
  if (!live) return;
  if (rec == 0) {
#pragma unroll
    for (int k = 0; k < 9; k++) p.E[job][((u64)w * parts + part) * XYZZ_WORDS + q4 * 9u + k] = a.v[k];
  }
}

// ---- tail: result = sum_w 2^(c w) sum_v 2^(off[v]) E[w][v], to canonical affine -------------
BPMI_HD void msm_tail_combine(u32 out_words[16], const u32 *E, u32 W, u32 c, const TailOffs &to) {
  // ONE Horner chain over bit positions: E[w][v] carries weight 2^(start of window w + off[v]), so walking from the top bit down
  // costs one doubling per bit position in total.  A flat loop over positions with the (<= 4) offsets in scalars: no indexed
  // private array, the accumulator stays in registers.  (to.top = Wb: the last Wb windows are c + 1 bits wide and were split at
  // their own bit offsets top_off.)
  const u32 o0 = to.off[0], o1 = to.nv > 1 ? to.off[1] : 0xFFFFFFFFu, o2 = to.nv > 2 ? to.off[2] : 0xFFFFFFFFu,
            o3 = to.nv > 3 ? to.off[3] : 0xFFFFFFFFu;
  xyzz acc;
  xyzz_set_inf(acc);
  u32 w = W, r = 0;                                   // position = start of window w + r
  for (u32 pos = W * c + to.top; pos-- > 0;) {
    if (r == 0) { w--; r = c + ((to.top && w + to.top >= W) ? 1u : 0u); }
    r--;
    xyzz_dbl(acc, acc);
    const bool tw = to.top && w + to.top >= W;        // a wide window
    const u32 v = tw ? (r == to.top_off[0] ? 0u : r == to.top_off[1] ? 1u : r == to.top_off[2] ? 2u : r == to.top_off[3] ? 3u : 4u)
                     : (r == o0 ? 0u : r == o1 ? 1u : r == o2 ? 2u : r == o3 ? 3u : 4u);
    if (v < 4u) {
      xyzz e;
      xyzz_load(e, E + ((u64)w * to.nv + v) * XYZZ_WORDS);
      xyzz_add(acc, acc, e);
    }
  }
  affine r_aff;
  xyzz_to_affine(r_aff, acc);
  affine_to_words(out_words, r_aff);
}
__global__ void __launch_bounds__(64) k_tail(const u32 *__restrict__ E, u32 W, u32 c, TailOffs to, u32 *__restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    u32 w16[16];
    msm_tail_combine(w16, E, W, c, to);
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = w16[i];
  }
}

#if defined(BPMI_ISA_PROBE)
// tools/isa_counts.py: the main path of the mixed addition on its own (no loads from the point array, no exceptional
// cases), so that its instructions can be counted in the ISA.  Not part of the product build.
__global__ void __launch_bounds__(256) k_isa_probe_madd(u32 *__restrict__ accs, const u32 *__restrict__ xy) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  xyzz acc;
  xyzz_load(acc, accs + 36ull * t);
  fe x2, y2, P, R, nY;
#pragma unroll
  for (int k = 0; k < 9; k++) { x2.v[k] = xy[18ull * t + k]; y2.v[k] = xy[18ull * t + 9 + k]; }
  asm volatile("; BPMI_MARK madd_main_begin");
  xyzz_madd_pr(P, R, nY, acc, x2, y2);
  xyzz_madd_finish(acc, P, R, nY);
  asm volatile("; BPMI_MARK madd_main_end");
  xyzz_store(accs + 36ull * t, acc);
}
#endif
