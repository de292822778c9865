// bpmi.hip -- libbpmi.so: MSM + inner-product-argument engine for MI355X (gfx950).
// C-ABI in include/bpmi.h; design notes in DESIGN.md.
//
// MSM pipeline (replaces Pippenger.multiexp, /root/reference/src/pippenger/pippenger.py:22-94,
// by the signed-digit bucket method; the result -- a canonical affine point -- is
// schedule independent, so it is bit-identical to the reference's subset-table schedule):
//   sort            (window, bucket) counting sort of (point index, sign):
//                   n >= 2^13: k_coarse_hist / k_partition / k_fine_hist / k_fine_scatter --
//                   a two-level radix partition whose histograms and ranks live in LDS
//                   (global atomics only reserve one range per tile and bin);
//                   smaller n: k_digits_hist / k_scatter with global atomics
//   k_scan_*        exclusive scans of the histograms -> run offsets
//   k_accum_l0      every thread adds exactly L sorted entries (perfectly balanced for
//                   ANY digit distribution); runs that end inside a chunk go to their
//                   bucket, the first/last run of a chunk become partial records
//   k_segscan       block-wide segmented scan over partial records, 256 -> 2 per block,
//                   repeated until one block is left
//   k_bucket_digit_sums / k_weighted31   sum_b b*B[w][b] as base-32 digit sums + a
//                   31-term suffix scan per (window, digit position)
//   tail            O(256) sequential doublings: window combine + to-affine
//                   (device kernel or host thread, same formulas)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bpmi.h"
#include "curve.hpp"
#include "field.hpp"
#include "scalar.hpp"

using namespace bpmi;

#define BPMI_VERSION 100

// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
enum Stage {
  ST_DIGITS = 0, ST_SCAN, ST_SCATTER, ST_ACCUM, ST_SEGSCAN, ST_BREDUCE, ST_TAIL,
  ST_MULBATCH, ST_LINCOMB2, ST_SCDOT, ST_SCFOLD, ST_MISC
};
static const char *STAGE_NAMES[BPMI_NSTAGES] = {
  "msm_digits_hist", "msm_scan", "msm_scatter", "msm_accumulate", "msm_segscan", "msm_bucket_reduce",
  "msm_tail", "ec_mul_batch", "ec_lincomb2", "sc_dot", "sc_fold", "misc"
};

struct EvPair { int stage; hipEvent_t a, b; };

struct bpmi_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  // workspace (grown on demand, never shrunk)
  void *ws = nullptr; size_t ws_bytes = 0;
  void *pin = nullptr; size_t pin_bytes = 0;       // pinned host staging
  // second MSM lane: an independent stream + workspace, so two independent MSMs (the L and
  // R of an IPA round) overlap -- the latency-bound stages of one hide under the
  // throughput-bound stages of the other
  hipStream_t stream1 = nullptr;
  void *ws1 = nullptr; size_t ws1_bytes = 0;
  void *pin1 = nullptr; size_t pin1_bytes = 0;
  hipEvent_t ev_fork = nullptr;
  struct PendingMsm { bool active = false; u32 W = 0, nv = 0, c = 0; int tail = 2; u32 *E = nullptr, *out = nullptr; } pend[2];
  void *stage_in = nullptr; size_t stage_in_bytes = 0;  // device staging for host-pointer entry points
  // options
  int opt_c = 0;        // window bits, 0 = auto
  int opt_tail = 0;     // 0 auto, 1 device, 2 host
  int opt_chunk = 0;    // entries per thread in k_accum_l0, 0 = auto
  int64_t opt_ipa_big = 0;   // base length from which the IPA folds generators 16-way (0 = default 2^18)
  // profiling
  bool prof = false;
  std::vector<EvPair> evs;
  double prof_ms[BPMI_NSTAGES] = {0};
  uint64_t prof_calls[BPMI_NSTAGES] = {0};
};

static std::string g_create_err;
static std::mutex g_mu;

static int fail(bpmi_ctx *ctx, int code, const std::string &msg) {
  if (ctx) ctx->err = msg;
  else { std::lock_guard<std::mutex> lk(g_mu); g_create_err = msg; }
  return code;
}
#define HIPCHK(ctx, call)                                                                   \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(ctx, e_ == hipErrorOutOfMemory ? BPMI_E_NOMEM : BPMI_E_HIP,               \
                  std::string(#call) + ": " + hipGetErrorString(e_));                       \
  } while (0)

static bool g_debug_sync = getenv("BPMI_DEBUG_SYNC") != nullptr;
static void debug_sync(bpmi_ctx *ctx, const char *what, hipStream_t stream = nullptr) {
  if (!g_debug_sync) return;
  fprintf(stderr, "[bpmi] sync after %s ... ", what); fflush(stderr);
  hipError_t e = hipStreamSynchronize(stream ? stream : ctx->stream);
  fprintf(stderr, "%s\n", hipGetErrorString(e)); fflush(stderr);
}
struct StageTimer {
  bpmi_ctx *ctx; int stage; hipStream_t stream; hipEvent_t a = nullptr, b = nullptr;
  StageTimer(bpmi_ctx *c, int s, hipStream_t st = nullptr) : ctx(c), stage(s), stream(st ? st : c->stream) {
    if (ctx->prof) {
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
      (void)hipEventRecord(a, stream);
    }
  }
  ~StageTimer() {
    if (ctx->prof && a) { (void)hipEventRecord(b, stream); ctx->evs.push_back({stage, a, b}); }
  }
};

static int ensure_ws(bpmi_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return BPMI_OK;
  if (ctx->ws) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
  size_t want = bytes + bytes / 8;
  HIPCHK(ctx, hipMalloc(&ctx->ws, want));
  ctx->ws_bytes = want;
  return BPMI_OK;
}
static int ensure_lane(bpmi_ctx *ctx, int lane) {
  if (lane == 0 || ctx->stream1) return BPMI_OK;
  HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream1, hipStreamNonBlocking));
  HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  return BPMI_OK;
}
static hipStream_t lane_stream(bpmi_ctx *ctx, int lane) { return lane ? ctx->stream1 : ctx->stream; }
static int ensure_ws_lane(bpmi_ctx *ctx, int lane, size_t bytes) {
  if (lane == 0) return ensure_ws(ctx, bytes);
  if (bytes <= ctx->ws1_bytes) return BPMI_OK;
  if (ctx->ws1) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream1)); HIPCHK(ctx, hipFree(ctx->ws1)); ctx->ws1 = nullptr; ctx->ws1_bytes = 0; }
  const size_t want = bytes + bytes / 8;
  HIPCHK(ctx, hipMalloc(&ctx->ws1, want));
  ctx->ws1_bytes = want;
  return BPMI_OK;
}
static int ensure_pin_lane(bpmi_ctx *ctx, int lane, size_t bytes);
static int ensure_stage_in(bpmi_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->stage_in_bytes) return BPMI_OK;
  if (ctx->stage_in) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->stage_in)); ctx->stage_in = nullptr; ctx->stage_in_bytes = 0; }
  HIPCHK(ctx, hipMalloc(&ctx->stage_in, bytes));
  ctx->stage_in_bytes = bytes;
  return BPMI_OK;
}
static int ensure_pin(bpmi_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->pin_bytes) return BPMI_OK;
  if (ctx->pin) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipHostFree(ctx->pin)); ctx->pin = nullptr; ctx->pin_bytes = 0; }
  HIPCHK(ctx, hipHostMalloc(&ctx->pin, bytes, hipHostMallocDefault));
  ctx->pin_bytes = bytes;
  return BPMI_OK;
}

static int ensure_pin_lane(bpmi_ctx *ctx, int lane, size_t bytes) {
  if (lane == 0) return ensure_pin(ctx, bytes);
  if (bytes <= ctx->pin1_bytes) return BPMI_OK;
  if (ctx->pin1) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream1)); HIPCHK(ctx, hipHostFree(ctx->pin1)); ctx->pin1 = nullptr; ctx->pin1_bytes = 0; }
  HIPCHK(ctx, hipHostMalloc(&ctx->pin1, bytes, hipHostMallocDefault));
  ctx->pin1_bytes = bytes;
  return BPMI_OK;
}

// ------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------
#define XYZZ_WORDS 36
#define LDS_STRIDE 37   // odd stride: conflict-free ds_read/ds_write of 36-word records

// up to three (points, scalars) segments presented as one logical array, so that
// e.g. L = <a_lo, g_hi> + <b_hi, h_lo> + cl*u is ONE MSM without any gather/concat
// (the reference concatenates Python lists: src/utils/commitments.py:13)
struct Segs {
  const u32 *pts[3];
  const u32 *sc[3];
  u32 n[3];
  u32 total;
};

__device__ __forceinline__ void load_words16(u32 w[16], const u32 *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 4; i++) { uint4 t = q[i]; w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w; }
}
__device__ __forceinline__ void load_words8(u32 w[8], const u32 *p) {
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 2; i++) { uint4 t = q[i]; w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w; }
}
__device__ __forceinline__ void store_words16(u32 *p, const u32 w[16]) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 4; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ void store_words8(u32 *p, const u32 w[8]) {
  uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 2; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
__device__ __forceinline__ const u32 *seg_point(const Segs &s, u32 i) {
  if (i < s.n[0]) return s.pts[0] + 16ull * i;
  i -= s.n[0];
  if (i < s.n[1]) return s.pts[1] + 16ull * i;
  i -= s.n[1];
  return s.pts[2] + 16ull * i;
}
__device__ __forceinline__ const u32 *seg_scalar(const Segs &s, u32 i) {
  if (i < s.n[0]) return s.sc[0] + 8ull * i;
  i -= s.n[0];
  if (i < s.n[1]) return s.sc[1] + 8ull * i;
  i -= s.n[1];
  return s.sc[2] + 8ull * i;
}
__device__ __forceinline__ void load_affine(affine &P, const u32 *p) {
  u32 w[16];
  load_words16(w, p);
  affine_from_words(P, w);
}
__device__ __forceinline__ void xyzz_load_g(xyzz &a, const u32 *p) {   // 144 B, 16-B aligned
  u32 w[XYZZ_WORDS];
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 9; i++) { uint4 t = q[i]; w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w; }
  xyzz_load(a, w);
}
__device__ __forceinline__ void xyzz_store_g(u32 *p, const xyzz &a) {
  u32 w[XYZZ_WORDS];
  xyzz_store(w, a);
  uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ------------------------------------------------------------------------------------
// MSM kernels
// ------------------------------------------------------------------------------------
struct MsmGeom {
  u32 n;       // pairs
  u32 c;       // window bits
  u32 W;       // windows
  u32 B;       // buckets per window = 2^(c-1)
  u32 G;       // W * B
  u32 L;       // entries per thread in k_accum_l0
  u32 nv;      // base-32 digit positions of a bucket index (ceil(c / 5))
};

// Signed-digit recoding of scalar i: calls f(w, b, sign) for every window, b = |digit|
// in [0, B] (0 = nothing to add), sign = 1 when the NEGATED point is added.
//   s > (q-1)/2  ->  use q - s on the negated point: halves the digit range, keeps
//   s < 2^255 so W*c >= 256 never overflows, and turns the range-proof scalar q-1
//   (aR, rangeproof_prover.py:43-45) into the single digit -1.
template <typename F>
__device__ __forceinline__ void for_each_digit(const Segs &segs, const MsmGeom &g, u32 i, F f) {
  sc s;
  load_words8(s.v, seg_scalar(segs, i));
  const bool neg = sc_is_high(s);
  if (neg) sc_neg(s, s);
  u32 carry = 0;
  const u32 mask = (1u << g.c) - 1u;
  for (u32 w = 0; w < g.W; w++) {
    const u32 t = (s.v[0] & mask) + carry;
    // shift the 256-bit register right by c (static register indexing)
#pragma unroll
    for (int k = 0; k < 7; k++) s.v[k] = (u32)((((u64)s.v[k + 1] << 32) | s.v[k]) >> g.c);
    s.v[7] >>= g.c;
    u32 b, sign;
    if (t > g.B) { b = (1u << g.c) - t; sign = 1; carry = 1; }
    else { b = t; sign = 0; carry = 0; }
    f(w, b, b ? (sign ^ (u32)neg) : 0u);
  }
}

// ================= sort path 1 (small n, c < 10): global-atomic counting sort ===========
// dig[w * n + i] = |d| | (sign << 31); histogram with one atomic per lane, or one per wave
// when the whole wave agrees (degenerate inputs)
__global__ void __launch_bounds__(256) k_digits_hist(Segs segs, MsmGeom g, u32 *__restrict__ dig, u32 *__restrict__ hist) {
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
#define TILE_SCALARS 4096      // scalars per block-iteration in level A
__global__ void __launch_bounds__(256) k_coarse_hist(Segs segs, MsmGeom g, u32 P, u32 *__restrict__ coarse_hist) {
  __shared__ u32 lh[PART_MAX];
  for (u32 p = threadIdx.x; p < P; p += 256u) lh[p] = 0;
  __syncthreads();
  const u32 Bc = g.B >> 8;
  const u32 stride = gridDim.x * blockDim.x;
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += stride) {
    for_each_digit(segs, g, i, [&](u32 w, u32 b, u32) {
      if (b) atomicAdd(&lh[w * Bc + ((b - 1u) >> 8)], 1u);
    });
  }
  __syncthreads();
  for (u32 p = threadIdx.x; p < P; p += 256u) { const u32 v = lh[p]; if (v) atomicAdd(&coarse_hist[p], v); }
}
// part[pos] = lo << 24 | sign << 23 | i   (n <= 2^23), grouped by partition
__global__ void __launch_bounds__(256) k_partition(Segs segs, MsmGeom g, u32 P, u32 *__restrict__ coarse_cursor, u32 *__restrict__ part) {
  __shared__ u32 lh[PART_MAX];
  const u32 Bc = g.B >> 8;
  const u32 ntiles = (g.n + TILE_SCALARS - 1) / TILE_SCALARS;
  for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (u32 p = threadIdx.x; p < P; p += 256u) lh[p] = 0;
    __syncthreads();
    const u32 i0 = tile * TILE_SCALARS;
    const u32 i1 = (i0 + TILE_SCALARS < g.n) ? i0 + TILE_SCALARS : g.n;
    for (u32 i = i0 + threadIdx.x; i < i1; i += 256u) {
      for_each_digit(segs, g, i, [&](u32 w, u32 b, u32) {
        if (b) atomicAdd(&lh[w * Bc + ((b - 1u) >> 8)], 1u);
      });
    }
    __syncthreads();
    // reserve this tile's range in every partition: count -> base position
    for (u32 p = threadIdx.x; p < P; p += 256u) { const u32 v = lh[p]; if (v) lh[p] = atomicAdd(&coarse_cursor[p], v); }
    __syncthreads();
    for (u32 i = i0 + threadIdx.x; i < i1; i += 256u) {
      for_each_digit(segs, g, i, [&](u32 w, u32 b, u32 sign) {
        if (b) {
          const u32 k = b - 1u;
          const u32 pos = atomicAdd(&lh[w * Bc + (k >> 8)], 1u);
          part[pos] = ((k & 255u) << 24) | (sign << 23) | i;
        }
      });
    }
    __syncthreads();
  }
}
// chunk_key[t] = bucket that contains sorted position t * L (for every chunk start inside [lo, hi))
__device__ __forceinline__ void fill_chunk_keys(u32 *__restrict__ chunk_key, u32 L, u32 key, u32 lo, u32 hi) {
  for (u32 t = (lo + L - 1u) / L; (u64)t * L < hi; t++) chunk_key[t] = key;
}
// Level B works on fixed-size TILES of the partitioned array (not one block per
// partition), so a partition -- or a single bucket -- of any size is spread over many
// blocks: balanced for every digit distribution (e.g. a top window that holds only the
// recoding carry puts n/2 entries into one bucket).
//   k_fine_hist     per tile: LDS histogram over fine buckets -> global fine histogram
//   (k_scan_*)      -> off[], cursor[]
//   k_fine_scatter  per tile: LDS histogram again, reserve one range per touched bucket,
//                   scatter with LDS cursors
// The LDS table covers FINE_BINS consecutive buckets starting at the tile's first one
// (16 partitions); entries beyond it (only when many tiny partitions share a tile) use a
// global atomic directly.
#define FINE_TILE 4096
#define FINE_BINS 4096
struct FineTile {
  u32 j0, j1;        // positions covered
  u32 p_first;       // partition of position j0
};
__device__ __forceinline__ FineTile fine_tile_setup(const u32 *__restrict__ coarse_off, u32 P, u32 E, u32 *s_off) {
  // s_off[0..P] = coarse_off (LDS copy for the partition walk)
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
__global__ void __launch_bounds__(256) k_fine_hist(MsmGeom g, u32 P, const u32 *__restrict__ coarse_off, const u32 *__restrict__ part,
                                                   const u32 *__restrict__ offE, u32 *__restrict__ fine_hist) {
  __shared__ u32 s_off[PART_MAX + 1];
  __shared__ u32 bins[FINE_BINS];
  const u32 E = offE[0];
  if (blockIdx.x * FINE_TILE >= E) return;
  const FineTile t = fine_tile_setup(coarse_off, P, E, s_off);
  for (u32 i = threadIdx.x; i < FINE_BINS; i += 256u) bins[i] = 0;
  __syncthreads();
  const u32 g_first = t.p_first * 256u;
  u32 pcur = t.p_first;
  for (u32 j = t.j0 + threadIdx.x; j < t.j1; j += 256u) {
    while (j >= s_off[pcur + 1]) pcur++;
    const u32 key = pcur * 256u + (part[j] >> 24);
    const u32 rel = key - g_first;
    if (rel < FINE_BINS) atomicAdd(&bins[rel], 1u); else atomicAdd(&fine_hist[key], 1u);
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < FINE_BINS; i += 256u) { const u32 v = bins[i]; if (v) atomicAdd(&fine_hist[g_first + i], v); }
}
__global__ void __launch_bounds__(256) k_fine_scatter(MsmGeom g, u32 P, const u32 *__restrict__ coarse_off, const u32 *__restrict__ part,
                                                      const u32 *__restrict__ offE, u32 *__restrict__ cursor, u32 *__restrict__ sidx) {
  __shared__ u32 s_off[PART_MAX + 1];
  __shared__ u32 bins[FINE_BINS];
  const u32 E = offE[0];
  if (blockIdx.x * FINE_TILE >= E) return;
  const FineTile t = fine_tile_setup(coarse_off, P, E, s_off);
  for (u32 i = threadIdx.x; i < FINE_BINS; i += 256u) bins[i] = 0;
  __syncthreads();
  const u32 g_first = t.p_first * 256u;
  u32 pcur = t.p_first;
  for (u32 j = t.j0 + threadIdx.x; j < t.j1; j += 256u) {
    while (j >= s_off[pcur + 1]) pcur++;
    const u32 rel = pcur * 256u + (part[j] >> 24) - g_first;
    if (rel < FINE_BINS) atomicAdd(&bins[rel], 1u);
  }
  __syncthreads();
  // reserve this tile's range in every touched bucket: count -> base position
  for (u32 i = threadIdx.x; i < FINE_BINS; i += 256u) { const u32 v = bins[i]; if (v) bins[i] = atomicAdd(&cursor[g_first + i], v); }
  __syncthreads();
  pcur = t.p_first;
  for (u32 j = t.j0 + threadIdx.x; j < t.j1; j += 256u) {
    while (j >= s_off[pcur + 1]) pcur++;
    const u32 e = part[j];
    const u32 key = pcur * 256u + (e >> 24);
    const u32 rel = key - g_first;
    const u32 pos = (rel < FINE_BINS) ? atomicAdd(&bins[rel], 1u) : atomicAdd(&cursor[key], 1u);
    sidx[pos] = (e & 0x7FFFFFu) | ((e & 0x800000u) << 8);
  }
}
// path 1 equivalent of the chunk-key fill: one thread per bucket
__global__ void __launch_bounds__(256) k_chunk_keys(MsmGeom g, const u32 *__restrict__ off, u32 *__restrict__ chunk_key) {
  const u32 key = blockIdx.x * blockDim.x + threadIdx.x;
  if (key >= g.G) return;
  fill_chunk_keys(chunk_key, g.L, key, off[key], off[key + 1]);
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
__global__ void __launch_bounds__(256) k_accum_l0(Segs segs, MsmGeom g, const u32 *__restrict__ off,
                                                  const u32 *__restrict__ chunk_key, const u32 *__restrict__ sidx,
                                                  u32 *__restrict__ buckets, u32 *__restrict__ rec_key, u32 *__restrict__ rec_pt) {
  const u32 E = off[g.G];
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 start = t * g.L;
  if (start >= E) return;
  const u32 end = (u32)((start + g.L < E) ? start + g.L : E);
  xyzz acc;
  xyzz_set_inf(acc);
  u32 cur = chunk_key[t];              // bucket containing position `start`
  u32 boundary = off[cur + 1];         // first position after that bucket's run
  bool first = true;
  // software pipeline: the (index, point) of entry j+1 is in flight while entry j is added
  u32 e_next = sidx[start];
  u32 w_next[16];
  load_words16(w_next, seg_point(segs, e_next & 0x7FFFFFFFu));
  for (u32 j = (u32)start; j < end; j++) {
    const u32 e = e_next;
    affine P;
    affine_from_words(P, w_next);
    if (j + 1 < end) {
      e_next = sidx[j + 1];
      load_words16(w_next, seg_point(segs, e_next & 0x7FFFFFFFu));
    }
    if (j == boundary) {               // the run of `cur` ended: flush, move to the next non-empty bucket
      if (first) { rec_key[2 * t] = cur; xyzz_store_g(rec_pt + (2 * t) * XYZZ_WORDS, acc); first = false; }
      else xyzz_store_g(buckets + (u64)cur * XYZZ_WORDS, acc);
      xyzz_set_inf(acc);
      do { cur++; boundary = off[cur + 1]; } while (boundary == j);
    }
    xyzz_madd_signed(acc, P, (e >> 31) != 0);
  }
  if (first) {
    rec_key[2 * t] = cur; xyzz_store_g(rec_pt + (2 * t) * XYZZ_WORDS, acc);
    xyzz_set_inf(acc);
  }
  rec_key[2 * t + 1] = cur;
  xyzz_store_g(rec_pt + (2 * t + 1) * XYZZ_WORDS, acc);
}

// number of records entering segscan level `level` (1-based); 0 when that level has nothing to do
__device__ __forceinline__ u32 records_at_level(u32 E, u32 L, int level, bool &is_final) {
  is_final = false;
  if (E == 0) return 0;
  u32 R = 2u * ((E + L - 1) / L);
  for (int l = 1; l < level; l++) {
    const u32 nb = (R + 255u) / 256u;
    if (nb <= 1) return 0;          // the previous level was already final
    R = 2u * nb;
  }
  is_final = ((R + 255u) / 256u) <= 1;
  return R;
}

// ---- levels >= 1: block-wide segmented scan over partial records -------------------------
__global__ void __launch_bounds__(256) k_segscan(MsmGeom g, const u32 *__restrict__ off, int level,
                                                 const u32 *__restrict__ in_key, const u32 *__restrict__ in_pt,
                                                 u32 *__restrict__ out_key, u32 *__restrict__ out_pt, u32 *__restrict__ buckets) {
  __shared__ u32 s_key[256];
  __shared__ u32 s_val[256 * LDS_STRIDE];
  bool is_final;
  const u32 R = records_at_level(off[g.G], g.L, level, is_final);
  const u32 nb = (R + 255u) / 256u;
  if (blockIdx.x >= nb) return;
  const u32 tid = threadIdx.x;
  const u32 j = blockIdx.x * 256u + tid;
  const bool valid = j < R;
  const u32 key = valid ? in_key[j] : 0xFFFFFFFFu;
  xyzz val;
  if (valid) xyzz_load_g(val, in_pt + (u64)j * XYZZ_WORDS); else xyzz_set_inf(val);
  s_key[tid] = key;
  __syncthreads();
  for (u32 d = 1; d < 256; d <<= 1) {
    xyzz_store(s_val + tid * LDS_STRIDE, val);
    __syncthreads();
    if (valid && tid >= d && s_key[tid - d] == key) {
      xyzz other;
      xyzz_load(other, s_val + (tid - d) * LDS_STRIDE);
      xyzz_add(val, other, val);
    }
    __syncthreads();
  }
  if (!valid) return;
  const u32 last_idx = (R - blockIdx.x * 256u >= 256u) ? 255u : (R - blockIdx.x * 256u - 1u);
  const bool run_end = (tid == last_idx) || (s_key[tid + 1] != key);
  if (!run_end) return;
  const u32 first_key = s_key[0], last_key = s_key[last_idx];
  // One store site with a per-thread destination.  (A three-way if/else over
  // buckets / head record / tail record made hipcc 7.2 merge the stores behind
  // scalar base-pointer selects in divergent flow, and the multi-block case faulted
  // on gfx950; tests/test_gpu_msm.py::test_msm_multiblock_segscan pins this.)
  const bool to_bucket = is_final || (key != first_key && key != last_key);
  const bool is_head = !to_bucket && (key == first_key);
  const u32 slot = 2u * blockIdx.x + (is_head ? 0u : 1u);
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

// ---- bucket reduction, step 1: D[w][v][d] = sum of buckets b in [1,B] whose base-32 digit v is d
// grid = W * nv * 31 blocks of 256
__global__ void __launch_bounds__(256, 4) k_bucket_digit_sums(MsmGeom g, const u32 *__restrict__ buckets, u32 *__restrict__ D) {
  __shared__ u32 s_val[256 * LDS_STRIDE];
  const u32 blk = blockIdx.x;
  const u32 d = blk % 31u + 1u;
  const u32 v = (blk / 31u) % g.nv;
  const u32 w = blk / (31u * g.nv);
  const u32 sh = 5u * v;
  xyzz acc;
  xyzz_set_inf(acc);
  // element e -> b = (hi << (sh+5)) | (d << sh) | lo,  lo = low `sh` bits of e, hi = e >> sh
  // valid (hi, lo): all lo for hi < hi_max, and lo <= B - base for hi == hi_max
  const u32 hi_max = g.B >> (sh + 5u);
  const u64 base_last = ((u64)hi_max << (sh + 5u)) | ((u64)d << sh);
  u32 last_cnt = 0;
  if (base_last <= g.B) { const u64 r = (u64)g.B - base_last + 1u; last_cnt = r < (1ull << sh) ? (u32)r : (1u << sh); }
  const u32 ecount = (hi_max << sh) + last_cnt;
  for (u32 e = threadIdx.x; e < ecount; e += 256u) {
    const u32 lo = e & ((1u << sh) - 1u), hi = e >> sh;
    const u64 b = ((u64)hi << (sh + 5u)) | ((u64)d << sh) | lo;
    {
      xyzz x;
      xyzz_load_g(x, buckets + ((u64)w * g.B + (b - 1u)) * XYZZ_WORDS);
      xyzz_add(acc, acc, x);
    }
  }
  block_tree_sum(acc, s_val);
  if (threadIdx.x == 0) xyzz_store_g(D + (u64)blk * XYZZ_WORDS, acc);
}
// ---- step 2: E[w][v] = sum_{d=1..31} d * D[w][v][d]  (suffix scan + sum over 32 lanes)
// grid = W * nv blocks of 64
__global__ void __launch_bounds__(64) k_weighted31(const u32 *__restrict__ D, u32 *__restrict__ Eout) {
  __shared__ u32 s_val[64 * LDS_STRIDE];
  const u32 tid = threadIdx.x;
  xyzz val;
  if (tid < 31u) xyzz_load_g(val, D + ((u64)blockIdx.x * 31u + tid) * XYZZ_WORDS); else xyzz_set_inf(val);
  // inclusive suffix scan: val[l] = sum_{j >= l} D[j]
  for (u32 d = 1; d < 32; d <<= 1) {
    xyzz_store(s_val + tid * LDS_STRIDE, val);
    __syncthreads();
    if (tid + d < 31u) {
      xyzz other;
      xyzz_load(other, s_val + (tid + d) * LDS_STRIDE);
      xyzz_add(val, val, other);
    }
    __syncthreads();
  }
  block_tree_sum(val, s_val);
  if (tid == 0) xyzz_store_g(Eout + (u64)blockIdx.x * XYZZ_WORDS, val);
}

// ---- tail: result = sum_w 2^(c w) sum_v 32^v E[w][v], to canonical affine -----------------
BPMI_HD void msm_tail_combine(u32 out_words[16], const u32 *E, u32 W, u32 nv, u32 c) {
  // ONE Horner chain over bit positions: E[w][v] carries weight 2^(c*w + 5*v), so walking
  // t from the top bit down costs c*W doublings in total (not c*W + 5*nv*W)
  xyzz acc;
  xyzz_set_inf(acc);
  for (int w = (int)W - 1; w >= 0; w--) {
    int prev = (int)c;                              // bit offset (within the window) already reached
    for (int v = (int)nv - 1; v >= 0; v--) {
      for (int k = prev; k > 5 * v; k--) xyzz_dbl(acc, acc);
      prev = 5 * v;
      xyzz e;
      xyzz_load(e, E + ((u64)w * nv + v) * XYZZ_WORDS);
      xyzz_add(acc, acc, e);
    }
  }
  affine r;
  xyzz_to_affine(r, acc);
  affine_to_words(out_words, r);
}
__global__ void k_tail(const u32 *__restrict__ E, u32 W, u32 nv, u32 c, u32 *__restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    u32 w16[16];
    msm_tail_combine(w16, E, W, nv, c);
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = w16[i];
  }
}

// ------------------------------------------------------------------------------------
// batched point kernels
// ------------------------------------------------------------------------------------
// out[i] = k_i * P_i   (left-to-right double-and-add in Jacobian coordinates; the scalars
// differ per lane, so lanes diverge on the addition only)
__global__ void __launch_bounds__(256, 3) k_ec_mul_batch(const u32 *__restrict__ pts, const u32 *__restrict__ scs, u32 n, u32 *__restrict__ out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine P;
  load_affine(P, pts + 16ull * i);
  sc s;
  load_words8(s.v, scs + 8ull * i);
  const bool neg = sc_is_high(s);
  if (neg) sc_neg(s, s);
  jac acc;
  jac_set_inf(acc);
  const bool pinf = affine_is_inf(P);
  for (int word = 7; word >= 0; word--) {
    // static word selection keeps s.v[] in registers
    u32 wv = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) if (k == word) wv = s.v[k];
    for (int bit = 31; bit >= 0; bit--) {
      jac_dbl(acc, acc);
      if (((wv >> bit) & 1u) && !pinf) jac_madd_signed(acc, P.x, P.y, neg);
    }
  }
  affine r;
  jac_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16(out + 16ull * i, w16);
}

struct Sc2 { u32 k1[8]; u32 k2[8]; };

// Non-adjacent forms of the two shared scalars, computed once on the host: bit i of nz*
// says digit i is non-zero, bit i of sg* says it is -1.  257 positions each.
struct NafPair { u32 nz1[9], sg1[9], nz2[9], sg2[9]; int top; };
static void host_naf(const uint8_t k32[32], u32 nz[9], u32 sg[9], int &top) {
  u32 w[9];
  memcpy(w, k32, 32);
  w[8] = 0;
  for (int i = 0; i < 9; i++) nz[i] = sg[i] = 0;
  for (int pos = 0; pos < 257; pos++) {
    if (w[0] & 1u) {
      const bool minus = (w[0] & 3u) == 3u;          // k mod 4 == 3 -> digit -1, k += 1
      nz[pos >> 5] |= 1u << (pos & 31);
      if (minus) {
        sg[pos >> 5] |= 1u << (pos & 31);
        for (int i = 0; i < 9; i++) { if (++w[i] != 0) break; }
      } else {
        w[0] &= ~1u;
      }
      if (pos > top) top = pos;
    }
    for (int i = 0; i < 8; i++) w[i] = (w[i] >> 1) | (w[i + 1] << 31);
    w[8] >>= 1;
  }
}

// out[i] = k1 * P1_i + k2 * P2_i with k1, k2 shared by all i (the generator fold,
// inner_product_prover.py:107-108).  Jacobian ladder driven by the NAF digits: every
// branch is on a kernel argument, hence wave-uniform; only mixed additions; the two input
// points of each thread are parked in LDS ([word][thread], conflict-free) to keep the
// register count at the ladder's working set.
struct LincombJob { const u32 *p1, *p2; u32 *out; u32 n; };
// Two independent jobs share one launch (threads [0, A.n) do job A, the next B.n do job
// B): a ladder thread is ~2 ms of serial issue, so per-launch latency, not throughput,
// bounds the small rounds of the IPA -- g and h are therefore folded together.
__global__ void __launch_bounds__(256, 3) k_ec_lincomb2(LincombJob ja, NafPair nfa, LincombJob jb, NafPair nfb) {
  __shared__ u32 s_pts[36 * 256];
  const u32 tid = threadIdx.x;
  u32 i = blockIdx.x * blockDim.x + tid;
  const bool second = i >= ja.n;          // may differ inside one wave only at the seam
  if (second) i -= ja.n;
  const u32 n = second ? jb.n : ja.n;
  if (i >= n) return;
  const u32 *p1 = second ? jb.p1 : ja.p1;
  const u32 *p2 = second ? jb.p2 : ja.p2;
  u32 *out = second ? jb.out : ja.out;
  bool inf1, inf2;
  {
    affine A;
    load_affine(A, p1 + 16ull * i);
    inf1 = affine_is_inf(A);
#pragma unroll
    for (int k = 0; k < 9; k++) { s_pts[k * 256 + tid] = A.x.v[k]; s_pts[(9 + k) * 256 + tid] = A.y.v[k]; }
    load_affine(A, p2 + 16ull * i);
    inf2 = affine_is_inf(A);
#pragma unroll
    for (int k = 0; k < 9; k++) { s_pts[(18 + k) * 256 + tid] = A.x.v[k]; s_pts[(27 + k) * 256 + tid] = A.y.v[k]; }
  }
  jac acc;
  jac_set_inf(acc);
  const int top = nfa.top > nfb.top ? nfa.top : nfb.top;
  for (int pos = top; pos >= 0; pos--) {
    jac_dbl(acc, acc);
    const u32 m = 1u << (pos & 31);
    const int wd = pos >> 5;
    // one inlined copy of the addition serves both points (the loop is kept rolled)
#pragma unroll 1
    for (int which = 0; which < 2; which++) {
      const u32 nzw = second ? (which ? nfb.nz2[wd] : nfb.nz1[wd]) : (which ? nfa.nz2[wd] : nfa.nz1[wd]);
      const u32 sgw = second ? (which ? nfb.sg2[wd] : nfb.sg1[wd]) : (which ? nfa.sg2[wd] : nfa.sg1[wd]);
      const bool isinf = which ? inf2 : inf1;
      if ((nzw & m) && !isinf) {
        fe x, y;
        const u32 base = which ? 18u * 256u : 0u;
#pragma unroll
        for (int k = 0; k < 9; k++) { x.v[k] = s_pts[base + k * 256 + tid]; y.v[k] = s_pts[base + (9 + k) * 256 + tid]; }
        jac_madd_signed(acc, x, y, (sgw & m) != 0);
      }
    }
  }
  affine r;
  jac_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16(out + 16ull * i, w16);
}

// ---- deferred generator folding (IPA) ----------------------------------------------------
// After d deferred folds the logical generator i (i < m) is sum_t coef[t] * G[i + t*m],
// t < 2^d, over the UNFOLDED base array G of length M = m << d; the newest fold is the
// least significant bit of t:  coef'[2t + s] = coef[t] * (s ? hi_factor : lo_factor).
__global__ void __launch_bounds__(256) k_ipa_coef_update(const u32 *cg, const u32 *ch, Sc2 x_xinv, u32 K, u32 *cg2, u32 *ch2) {
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;     // new index in [0, 2K)
  if (j >= 2u * K) return;
  sc X, XI, c, r;
#pragma unroll
  for (int k = 0; k < 8; k++) { X.v[k] = x_xinv.k1[k]; XI.v[k] = x_xinv.k2[k]; }
  // g' = x^-1 g_lo + x g_hi ;  h' = x h_lo + x^-1 h_hi   (inner_product_prover.py:107-108)
  load_words8(c.v, cg + 8ull * (j >> 1));
  sc_mul(r, c, (j & 1u) ? X : XI);
  store_words8(cg2 + 8ull * j, r.v);
  load_words8(c.v, ch + 8ull * (j >> 1));
  sc_mul(r, c, (j & 1u) ? XI : X);
  store_words8(ch2 + 8ull * j, r.v);
}
// scalars of the L (right = 0) or R (right = 1) MSM over the unfolded bases:
//   L = <a_lo, g_hi> + <b_hi, h_lo>,  R = <a_hi, g_lo> + <b_lo, h_hi>   (:98-99)
__global__ void __launch_bounds__(256) k_ipa_expand(const u32 *__restrict__ a, const u32 *__restrict__ b, const u32 *__restrict__ cg,
                                                    const u32 *__restrict__ ch, u32 M, u32 logm, int right,
                                                    u32 *__restrict__ eg, u32 *__restrict__ eh) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  const u32 m = 1u << logm, half = m >> 1;
  const u32 i = k & (m - 1u), t = k >> logm;
  const bool hi = i >= half;
  sc z;
#pragma unroll
  for (int q = 0; q < 8; q++) z.v[q] = 0;
  sc rg = z, rh = z;
  // g side uses the g-half OPPOSITE to the a-half: L pairs a_lo with g_hi
  if (hi != (right != 0)) {
    sc av, c;
    load_words8(av.v, a + 8ull * (right ? half + i : i - half));
    load_words8(c.v, cg + 8ull * t);
    sc_mul(rg, av, c);
  }
  if (hi == (right != 0)) {
    sc bv, c;
    load_words8(bv.v, b + 8ull * (right ? i - half : half + i));
    load_words8(c.v, ch + 8ull * t);
    sc_mul(rh, bv, c);
  }
  store_words8(eg + 8ull * k, rg.v);
  store_words8(eh + 8ull * k, rh.v);
}
// materialise 2^d-way folded generators: out[i] = sum_t coef[t] * G[i + t*m], i < m, as an
// interleaved NAF ladder (shared scalars -> wave-uniform branches); two jobs (g and h) per launch
#define MULTIFOLD_MAXK 16
struct NafK { u32 nz[MULTIFOLD_MAXK][9]; u32 sg[MULTIFOLD_MAXK][9]; int top; };
struct MultifoldJob { const u32 *base; u32 *out; };
__global__ void __launch_bounds__(256, 3) k_ec_multifold(MultifoldJob ja, MultifoldJob jb, const NafK *__restrict__ nfa, const NafK *__restrict__ nfb,
                                                         u32 m, u32 K) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = i >= m;
  if (second) i -= m;
  if (i >= m) return;
  const u32 *base = second ? jb.base : ja.base;
  u32 *out = second ? jb.out : ja.out;
  const NafK *nf = second ? nfb : nfa;
  jac acc;
  jac_set_inf(acc);
  const int top = nf->top;
  for (int pos = top; pos >= 0; pos--) {
    jac_dbl(acc, acc);
    const u32 msk = 1u << (pos & 31);
    const int wd = pos >> 5;
#pragma unroll 1
    for (u32 t = 0; t < K; t++) {
      if (nf->nz[t][wd] & msk) {
        affine P;
        load_affine(P, base + 16ull * ((u64)i + (u64)t * m));
        if (!affine_is_inf(P)) jac_madd_signed(acc, P.x, P.y, (nf->sg[t][wd] & msk) != 0);
      }
    }
  }
  affine r;
  jac_to_affine(r, acc);
  u32 w16[16];
  affine_to_words(w16, r);
  store_words16(out + 16ull * i, w16);
}

// out = sum of n affine points (one block)
__global__ void __launch_bounds__(256) k_ec_sum(const u32 *__restrict__ pts, u32 n, u32 *__restrict__ out) {
  __shared__ u32 s_val[256 * LDS_STRIDE];
  xyzz acc;
  xyzz_set_inf(acc);
  for (u32 i = threadIdx.x; i < n; i += 256u) {
    affine P;
    load_affine(P, pts + 16ull * i);
    xyzz_madd_signed(acc, P, false);
  }
  block_tree_sum(acc, s_val);
  if (threadIdx.x == 0) {
    affine r;
    xyzz_to_affine(r, acc);
    u32 w16[16];
    affine_to_words(w16, r);
#pragma unroll
    for (int k = 0; k < 16; k++) out[k] = w16[k];
  }
}

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

// ------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Window bits from the tools/tune_msm.py sweeps on MI355X (profiles/r01_tune_msm.txt).
// Besides the usual bucket-count trade-off, windows whose TOP window holds only a few
// bits (255 mod c small: c = 15, 14, 12, 11) concentrate a whole window's digits in a
// handful of buckets, so c in {8, 13, 16} (top window 7, 8, 15 bits) are preferred.
static u32 pick_window_bits(const bpmi_ctx *ctx, uint64_t n) {
  if (ctx->opt_c >= 2 && ctx->opt_c <= 16) return (u32)ctx->opt_c;
  if (n >= (1u << 18)) return 16;
  if (n >= (1u << 16)) return 13;
  if (n >= (1u << 10)) return 8;
  u32 lg = 0;
  while ((1ull << (lg + 1)) <= n) lg++;
  const int c = (int)lg - 2;
  return (u32)(c < 4 ? 4 : c);
}

struct MsmWs {
  u32 *dig, *hist, *off, *cursor, *bsum, *sidx, *buckets, *chunk_key, *coarse_hist, *coarse_off, *coarse_cursor;
  u32 P;          // partitions of sort path 2 (0 = path 1)
  u32 *rec_key[2], *rec_pt[2];
  u32 *D, *E, *out;
  size_t total;
  u32 nscan_blocks, rec0_max;
};
static void msm_layout(const MsmGeom &g, MsmWs &w, char *base) {
  size_t o = 0;
  auto take = [&](size_t bytes) { char *p = base ? base + o : nullptr; o += align_up(bytes, 256); return (u32 *)p; };
  const size_t nW = (size_t)g.n * g.W;
  w.nscan_blocks = (u32)((g.G + SCAN_TILE - 1) / SCAN_TILE);
  w.rec0_max = (u32)(2 * ((nW + g.L - 1) / g.L));
  const u32 rec1_max = 2 * ((w.rec0_max + 255) / 256);
  // sort path 2 (LDS partition sort) when the bucket key has more than 8 bits and the
  // packed entry (8-bit lo | sign | 23-bit index) fits; path 1 (global atomics) otherwise
  w.P = (g.c >= 10 && g.n <= (1u << 23)) ? g.W * (g.B >> 8) : 0;
  w.hist = take(4ull * g.G);                 // path 1 only
  w.off = take(4ull * (g.G + 1));
  w.cursor = take(4ull * g.G);               // path 1 only
  w.bsum = take(4ull * (w.nscan_blocks + 1));
  w.coarse_hist = take(4ull * (PART_MAX + 1));
  w.coarse_off = take(4ull * (PART_MAX + 1));
  w.coarse_cursor = take(4ull * (PART_MAX + 1));
  w.dig = take(4ull * nW);                   // path 1: digits; path 2: partitioned entries
  w.sidx = take(4ull * nW);
  w.chunk_key = take(4ull * (w.rec0_max / 2 + 1));
  w.buckets = take(4ull * XYZZ_WORDS * g.G);
  w.rec_key[0] = take(4ull * w.rec0_max);
  w.rec_pt[0] = take(4ull * XYZZ_WORDS * w.rec0_max);
  w.rec_key[1] = take(4ull * rec1_max);
  w.rec_pt[1] = take(4ull * XYZZ_WORDS * rec1_max);
  w.D = take(4ull * XYZZ_WORDS * g.W * g.nv * 31);
  w.E = take(4ull * XYZZ_WORDS * g.W * g.nv);
  w.out = take(64);
  w.total = o;
}

// Enqueue every GPU stage of one MSM on `lane` (0 = the ctx stream, 1 = the second lane),
// including the device->pinned-host copy the tail needs; returns without synchronising.
static int msm_enqueue(bpmi_ctx *ctx, int lane, const Segs &segs) {
  const uint64_t n = segs.total;
  bpmi_ctx::PendingMsm &pd = ctx->pend[lane];
  pd.active = false;
  if (n == 0) return BPMI_OK;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  MsmGeom g;
  g.n = (u32)n;
  g.c = pick_window_bits(ctx, n);
  g.W = 255u / g.c + 1u;
  g.B = 1u << (g.c - 1);
  g.G = g.W * g.B;
  g.L = ctx->opt_chunk > 0 ? (u32)ctx->opt_chunk : (n >= (1u << 19) ? 64u : (n >= (1u << 16) ? 32u : 16u));   // tools/tune_msm.py sweeps
  g.nv = (g.c + 4u) / 5u;
  MsmWs w;
  msm_layout(g, w, nullptr);
  int rc = ensure_lane(ctx, lane);
  if (rc) return rc;
  rc = ensure_ws_lane(ctx, lane, w.total);
  if (rc) return rc;
  msm_layout(g, w, (char *)(lane ? ctx->ws1 : ctx->ws));
  hipStream_t st = lane_stream(ctx, lane);
  const u32 nblk_n = (u32)std::min<uint64_t>((n + 255) / 256, 8192);
  {
    StageTimer t(ctx, ST_MISC, st);
    if (w.P) HIPCHK(ctx, hipMemsetAsync(w.coarse_hist, 0, 4ull * (PART_MAX + 1), st));
    else HIPCHK(ctx, hipMemsetAsync(w.hist, 0, 4ull * g.G, st));
    HIPCHK(ctx, hipMemsetAsync(w.buckets, 0, 4ull * XYZZ_WORDS * g.G, st));
  }
  debug_sync(ctx, "ST_MISC", st);
  if (w.P) {
    {
      StageTimer t(ctx, ST_DIGITS, st);
      hipLaunchKernelGGL(k_coarse_hist, dim3(std::min<u32>(nblk_n, 512)), dim3(256), 0, st, segs, g, w.P, w.coarse_hist);
    }
    debug_sync(ctx, "ST_DIGITS", st);
    {
      StageTimer t(ctx, ST_SCAN, st);
      // exclusive scan of <= 2048 partition counts; total -> coarse_off[P] and off[G]
      hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, st, w.coarse_hist, w.P, w.bsum);
      hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, w.bsum, 1u, w.off, g.G);
      hipLaunchKernelGGL(k_scan_final, dim3(1), dim3(256), 0, st, w.coarse_hist, w.P, w.bsum, w.coarse_off, w.coarse_cursor);
      HIPCHK(ctx, hipMemcpyAsync(w.coarse_off + w.P, w.off + g.G, 4, hipMemcpyDeviceToDevice, st));
    }
    debug_sync(ctx, "ST_SCAN", st);
    {
      StageTimer t(ctx, ST_SCATTER, st);
      const u32 ntiles = (g.n + TILE_SCALARS - 1) / TILE_SCALARS;
      hipLaunchKernelGGL(k_partition, dim3(std::min<u32>(ntiles, 2048)), dim3(256), 0, st, segs, g, w.P, w.coarse_cursor, w.dig);
      // level B over fixed-size tiles of the partitioned array (grid sized for the maximum E)
      const u32 nft = (u32)(((size_t)g.n * g.W + FINE_TILE - 1) / FINE_TILE);
      HIPCHK(ctx, hipMemsetAsync(w.hist, 0, 4ull * g.G, st));
      hipLaunchKernelGGL(k_fine_hist, dim3(nft), dim3(256), 0, st, g, w.P, w.coarse_off, w.dig, w.coarse_off + w.P, w.hist);
      hipLaunchKernelGGL(k_scan_partials, dim3(w.nscan_blocks), dim3(256), 0, st, w.hist, g.G, w.bsum);
      hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, w.bsum, w.nscan_blocks, w.off, g.G);
      hipLaunchKernelGGL(k_scan_final, dim3(w.nscan_blocks), dim3(256), 0, st, w.hist, g.G, w.bsum, w.off, w.cursor);
      hipLaunchKernelGGL(k_fine_scatter, dim3(nft), dim3(256), 0, st, g, w.P, w.coarse_off, w.dig, w.coarse_off + w.P, w.cursor, w.sidx);
      hipLaunchKernelGGL(k_chunk_keys, dim3((g.G + 255) / 256), dim3(256), 0, st, g, w.off, w.chunk_key);
    }
    debug_sync(ctx, "ST_SCATTER", st);
  } else {
    {
      StageTimer t(ctx, ST_DIGITS, st);
      hipLaunchKernelGGL(k_digits_hist, dim3(nblk_n), dim3(256), 0, st, segs, g, w.dig, w.hist);
    }
    debug_sync(ctx, "ST_DIGITS", st);
    {
      StageTimer t(ctx, ST_SCAN, st);
      hipLaunchKernelGGL(k_scan_partials, dim3(w.nscan_blocks), dim3(256), 0, st, w.hist, g.G, w.bsum);
      hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, st, w.bsum, w.nscan_blocks, w.off, g.G);
      hipLaunchKernelGGL(k_scan_final, dim3(w.nscan_blocks), dim3(256), 0, st, w.hist, g.G, w.bsum, w.off, w.cursor);
    }
    debug_sync(ctx, "ST_SCAN", st);
    {
      StageTimer t(ctx, ST_SCATTER, st);
      hipLaunchKernelGGL(k_scatter, dim3(nblk_n), dim3(256), 0, st, g, w.dig, w.cursor, w.sidx);
      hipLaunchKernelGGL(k_chunk_keys, dim3((g.G + 255) / 256), dim3(256), 0, st, g, w.off, w.chunk_key);
    }
    debug_sync(ctx, "ST_SCATTER", st);
  }
  {
    StageTimer t(ctx, ST_ACCUM, st);
    const u32 nthreads = w.rec0_max / 2;
    hipLaunchKernelGGL(k_accum_l0, dim3((nthreads + 255) / 256), dim3(256), 0, st, segs, g, w.off, w.chunk_key, w.sidx,
                       w.buckets, w.rec_key[0], w.rec_pt[0]);
  }
  debug_sync(ctx, "ST_ACCUM", st);
  {
    StageTimer t(ctx, ST_SEGSCAN, st);
    u32 R = w.rec0_max;
    int level = 1, src = 0;
    for (;;) {
      const u32 nb = (R + 255) / 256;
      hipLaunchKernelGGL(k_segscan, dim3(nb), dim3(256), 0, st, g, w.off, level, w.rec_key[src], w.rec_pt[src],
                         w.rec_key[src ^ 1], w.rec_pt[src ^ 1], w.buckets);
      if (g_debug_sync) { fprintf(stderr, "[bpmi] segscan level %d nb %u R %u\n", level, nb, R); debug_sync(ctx, "segscan level", st); }
      if (nb <= 1) break;
      R = 2 * nb;
      // ping-pong: level 1 reads buffer 0 (large) and writes buffer 1; later levels are
      // small enough for either buffer (rec1_max >= every later level)
      src ^= 1;
      level++;
    }
  }
  debug_sync(ctx, "ST_SEGSCAN", st);
  {
    StageTimer t(ctx, ST_BREDUCE, st);
    hipLaunchKernelGGL(k_bucket_digit_sums, dim3(g.W * g.nv * 31), dim3(256), 0, st, g, w.buckets, w.D);
    hipLaunchKernelGGL(k_weighted31, dim3(g.W * g.nv), dim3(64), 0, st, w.D, w.E);
  }
  debug_sync(ctx, "ST_BREDUCE", st);
  {
    StageTimer t(ctx, ST_TAIL, st);
    const int tail = ctx->opt_tail ? ctx->opt_tail : 2;
    if (tail == 1) {
      hipLaunchKernelGGL(k_tail, dim3(1), dim3(64), 0, st, w.E, g.W, g.nv, g.c, w.out);
      rc = ensure_pin_lane(ctx, lane, 4096);
      if (rc) return rc;
      HIPCHK(ctx, hipMemcpyAsync(lane ? ctx->pin1 : ctx->pin, w.out, 64, hipMemcpyDeviceToHost, st));
    } else {
      const size_t eb = 4ull * XYZZ_WORDS * g.W * g.nv;
      rc = ensure_pin_lane(ctx, lane, eb);
      if (rc) return rc;
      HIPCHK(ctx, hipMemcpyAsync(lane ? ctx->pin1 : ctx->pin, w.E, eb, hipMemcpyDeviceToHost, st));
    }
    pd.active = true; pd.W = g.W; pd.nv = g.nv; pd.c = g.c; pd.tail = tail;
  }
  HIPCHK(ctx, hipGetLastError());
  return BPMI_OK;
}
// Wait for the lane and run the host part of the tail; out = the MSM result.
static int msm_finish(bpmi_ctx *ctx, int lane, uint8_t out[64]) {
  bpmi_ctx::PendingMsm &pd = ctx->pend[lane];
  if (!pd.active) { memset(out, 0, 64); return BPMI_OK; }      // n == 0
  hipStream_t st = lane_stream(ctx, lane);
  HIPCHK(ctx, hipStreamSynchronize(st));
  const void *pin = lane ? ctx->pin1 : ctx->pin;
  if (pd.tail == 1) {
    memcpy(out, pin, 64);
  } else {
    u32 w16[16];
    msm_tail_combine(w16, (const u32 *)pin, pd.W, pd.nv, pd.c);
    memcpy(out, w16, 64);
  }
  pd.active = false;
  debug_sync(ctx, "ST_TAIL", st);
  return BPMI_OK;
}
static int msm_run(bpmi_ctx *ctx, const Segs &segs, uint8_t out[64]) {
  if (segs.total > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  int rc = msm_enqueue(ctx, 0, segs);
  if (rc) return rc;
  return msm_finish(ctx, 0, out);
}
// two independent MSMs, overlapped on the two lanes; everything already enqueued on the
// ctx stream (the producers of the scalars) is ordered before both
static int msm_run_pair(bpmi_ctx *ctx, const Segs &s0, uint8_t out0[64], const Segs &s1, uint8_t out1[64]) {
  if (s0.total > BPMI_MAX_N || s1.total > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  int rc = ensure_lane(ctx, 1);
  if (rc) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream1, ctx->ev_fork, 0));
  rc = msm_enqueue(ctx, 0, s0);
  if (rc) return rc;
  rc = msm_enqueue(ctx, 1, s1);
  if (rc) return rc;
  rc = msm_finish(ctx, 0, out0);
  const int rc1 = msm_finish(ctx, 1, out1);
  return rc ? rc : rc1;
}

// second-level segscan buffer sizing relies on this: every level after the first has
// at most rec1_max records (R shrinks monotonically)

// ------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------
extern "C" {

int bpmi_version(void) { return BPMI_VERSION; }

int bpmi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

bpmi_ctx *bpmi_ctx_create(int device, void *stream) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) { fail(nullptr, BPMI_E_NODEVICE, "no HIP device visible (libbpmi has no CPU fallback)"); return nullptr; }
  if (device < 0 || device >= n) { fail(nullptr, BPMI_E_ARG, "device index out of range"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { fail(nullptr, BPMI_E_HIP, "hipSetDevice failed"); return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail(nullptr, BPMI_E_HIP, "hipGetDeviceProperties failed"); return nullptr; }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fail(nullptr, BPMI_E_NODEVICE, std::string("device is ") + prop.gcnArchName + ", libbpmi is built for gfx950 only");
    return nullptr;
  }
  bpmi_ctx *ctx = new bpmi_ctx();
  ctx->device = device;
  if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
  else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx; fail(nullptr, BPMI_E_HIP, "hipStreamCreate failed"); return nullptr;
    }
    ctx->own_stream = true;
  }
  return ctx;
}

void bpmi_ctx_destroy(bpmi_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &e : ctx->evs) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->stage_in) (void)hipFree(ctx->stage_in);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  if (ctx->stream1) { (void)hipStreamSynchronize(ctx->stream1); (void)hipStreamDestroy(ctx->stream1); }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ws1) (void)hipFree(ctx->ws1);
  if (ctx->pin1) (void)hipHostFree(ctx->pin1);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char *bpmi_last_error(const bpmi_ctx *ctx) {
  if (ctx) return ctx->err.c_str();
  return g_create_err.c_str();
}

int bpmi_sync(bpmi_ctx *ctx) {
  if (!ctx) return BPMI_E_ARG;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

int bpmi_set_option(bpmi_ctx *ctx, const char *name, int64_t value) {
  if (!ctx || !name) return BPMI_E_ARG;
  if (!strcmp(name, "window_bits")) { if (value != 0 && (value < 2 || value > 16)) return fail(ctx, BPMI_E_ARG, "window_bits must be 0 or 2..16"); ctx->opt_c = (int)value; return BPMI_OK; }
  if (!strcmp(name, "tail")) { if (value < 0 || value > 2) return fail(ctx, BPMI_E_ARG, "tail must be 0, 1 or 2"); ctx->opt_tail = (int)value; return BPMI_OK; }
  if (!strcmp(name, "chunk")) { if (value < 0 || value > 4096) return fail(ctx, BPMI_E_ARG, "chunk must be 0..4096"); ctx->opt_chunk = (int)value; return BPMI_OK; }
  if (!strcmp(name, "ipa_big_m")) { if (value < 0 || (value & (value - 1))) return fail(ctx, BPMI_E_ARG, "ipa_big_m must be 0 or a power of two"); ctx->opt_ipa_big = value; return BPMI_OK; }
  return fail(ctx, BPMI_E_ARG, std::string("unknown option ") + name);
}

int bpmi_malloc(bpmi_ctx *ctx, size_t bytes, void **dptr) {
  if (!ctx || !dptr) return BPMI_E_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc(dptr, bytes ? bytes : 16));
  return BPMI_OK;
}
int bpmi_free(bpmi_ctx *ctx, void *dptr) {
  if (!ctx) return BPMI_E_ARG;
  if (!dptr) return BPMI_OK;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipFree(dptr));
  return BPMI_OK;
}
int bpmi_upload(bpmi_ctx *ctx, void *dptr, const void *host, size_t bytes) {
  if (!ctx || (bytes && (!dptr || !host))) return BPMI_E_ARG;
  if (!bytes) return BPMI_OK;
  HIPCHK(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}
int bpmi_download(bpmi_ctx *ctx, void *host, const void *dptr, size_t bytes) {
  if (!ctx || (bytes && (!dptr || !host))) return BPMI_E_ARG;
  if (!bytes) return BPMI_OK;
  HIPCHK(ctx, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

// ---- MSM ------------------------------------------------------------------------------
int bpmi_msm_dev(bpmi_ctx *ctx, const void *d_pts, const void *d_scalars, uint64_t n, uint8_t out[64]) {
  if (!ctx || !out || (n && (!d_pts || !d_scalars))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Segs s = {};
  s.pts[0] = (const u32 *)d_pts; s.sc[0] = (const u32 *)d_scalars; s.n[0] = (u32)n; s.total = (u32)n;
  return msm_run(ctx, s, out);
}
int bpmi_msm(bpmi_ctx *ctx, const uint8_t *pts, const uint8_t *scalars, uint64_t n, uint8_t out[64]) {
  if (!ctx || !out || (n && (!pts || !scalars))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) { memset(out, 0, 64); return BPMI_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage_in(ctx, 96 * n + 512);
  if (rc) return rc;
  char *dp = (char *)ctx->stage_in;
  char *ds = dp + align_up(64 * n, 256);
  HIPCHK(ctx, hipMemcpyAsync(dp, pts, 64 * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ds, scalars, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  return bpmi_msm_dev(ctx, dp, ds, n, out);
}

// ---- batched point ops --------------------------------------------------------------------
int bpmi_ec_mul_batch_dev(bpmi_ctx *ctx, const void *d_pts, const void *d_scalars, uint64_t n, void *d_out) {
  if (!ctx || (n && (!d_pts || !d_scalars || !d_out))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  {
    StageTimer t(ctx, ST_MULBATCH);
    hipLaunchKernelGGL(k_ec_mul_batch, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const u32 *)d_pts,
                       (const u32 *)d_scalars, (u32)n, (u32 *)d_out);
  }
  HIPCHK(ctx, hipGetLastError());
  return BPMI_OK;
}
int bpmi_ec_mul_batch(bpmi_ctx *ctx, const uint8_t *pts, const uint8_t *scalars, uint64_t n, uint8_t *out) {
  if (!ctx || (n && (!pts || !scalars || !out))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage_in(ctx, 160 * n + 1024);
  if (rc) return rc;
  char *dp = (char *)ctx->stage_in, *ds = dp + align_up(64 * n, 256), *dout = ds + align_up(32 * n, 256);
  HIPCHK(ctx, hipMemcpyAsync(dp, pts, 64 * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ds, scalars, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  rc = bpmi_ec_mul_batch_dev(ctx, dp, ds, n, dout);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, 64 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}
static void make_naf_pair(NafPair &nf, const uint8_t k1[32], const uint8_t k2[32]) {
  nf.top = -1;
  host_naf(k1, nf.nz1, nf.sg1, nf.top);
  host_naf(k2, nf.nz2, nf.sg2, nf.top);
}
// out_a[i] = ka1 * a1[i] + ka2 * a2[i] (i < na)  and  out_b[i] = kb1 * b1[i] + kb2 * b2[i] (i < nb), one launch
static int lincomb2_pair_dev(bpmi_ctx *ctx, const void *a1, const void *a2, const uint8_t ka1[32], const uint8_t ka2[32], uint64_t na, void *out_a,
                             const void *b1, const void *b2, const uint8_t kb1[32], const uint8_t kb2[32], uint64_t nb, void *out_b) {
  LincombJob ja = {(const u32 *)a1, (const u32 *)a2, (u32 *)out_a, (u32)na};
  LincombJob jb = {(const u32 *)b1, (const u32 *)b2, (u32 *)out_b, (u32)nb};
  NafPair nfa, nfb;
  make_naf_pair(nfa, ka1, ka2);
  if (nb) make_naf_pair(nfb, kb1, kb2); else { memset(&nfb, 0, sizeof(nfb)); nfb.top = -1; }
  const uint64_t total = na + nb;
  {
    StageTimer t(ctx, ST_LINCOMB2);
    hipLaunchKernelGGL(k_ec_lincomb2, dim3((u32)((total + 255) / 256)), dim3(256), 0, ctx->stream, ja, nfa, jb, nfb);
  }
  HIPCHK(ctx, hipGetLastError());
  return BPMI_OK;
}
int bpmi_ec_lincomb2_batch_dev(bpmi_ctx *ctx, const void *d_p1, const void *d_p2, const uint8_t k1[32], const uint8_t k2[32],
                               uint64_t n, void *d_out) {
  if (!ctx || !k1 || !k2 || (n && (!d_p1 || !d_p2 || !d_out))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return lincomb2_pair_dev(ctx, d_p1, d_p2, k1, k2, n, d_out, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
}
int bpmi_ec_lincomb2_batch(bpmi_ctx *ctx, const uint8_t *p1, const uint8_t *p2, const uint8_t k1[32], const uint8_t k2[32],
                           uint64_t n, uint8_t *out) {
  if (!ctx || !k1 || !k2 || (n && (!p1 || !p2 || !out))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage_in(ctx, 192 * n + 1024);
  if (rc) return rc;
  char *d1 = (char *)ctx->stage_in, *d2 = d1 + align_up(64 * n, 256), *dout = d2 + align_up(64 * n, 256);
  HIPCHK(ctx, hipMemcpyAsync(d1, p1, 64 * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(d2, p2, 64 * n, hipMemcpyHostToDevice, ctx->stream));
  rc = bpmi_ec_lincomb2_batch_dev(ctx, d1, d2, k1, k2, n, dout);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, 64 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}
int bpmi_ec_sum(bpmi_ctx *ctx, const uint8_t *pts, uint64_t n, uint8_t out[64]) {
  if (!ctx || !out || (n && !pts)) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) { memset(out, 0, 64); return BPMI_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage_in(ctx, 64 * n + 512);
  if (rc) return rc;
  char *dp = (char *)ctx->stage_in, *dout = dp + align_up(64 * n, 256);
  HIPCHK(ctx, hipMemcpyAsync(dp, pts, 64 * n, hipMemcpyHostToDevice, ctx->stream));
  {
    StageTimer t(ctx, ST_MISC);
    hipLaunchKernelGGL(k_ec_sum, dim3(1), dim3(256), 0, ctx->stream, (const u32 *)dp, (u32)n, (u32 *)dout);
  }
  HIPCHK(ctx, hipMemcpyAsync(out, dout, 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

// ---- scalar ops -------------------------------------------------------------------------------
static int sc_dot_dev_to(bpmi_ctx *ctx, const void *d_a, const void *d_b, uint64_t n, u32 *d_out, u32 *d_partial) {
  const u32 nb = (u32)std::min<uint64_t>((n + 255) / 256, 1024);
  StageTimer t(ctx, ST_SCDOT);
  hipLaunchKernelGGL(k_sc_dot, dim3(nb), dim3(256), 0, ctx->stream, (const u32 *)d_a, (const u32 *)d_b, (u32)n, d_partial);
  hipLaunchKernelGGL(k_sc_sum, dim3(1), dim3(256), 0, ctx->stream, d_partial, nb, d_out);
  return BPMI_OK;
}
int bpmi_sc_dot_dev(bpmi_ctx *ctx, const void *d_a, const void *d_b, uint64_t n, uint8_t out[32]) {
  if (!ctx || !out || (n && (!d_a || !d_b))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) { memset(out, 0, 32); return BPMI_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_ws(ctx, 32 * 1024 + 256);
  if (rc) return rc;
  u32 *partial = (u32 *)ctx->ws, *dout = partial + 8 * 1024;
  sc_dot_dev_to(ctx, d_a, d_b, n, dout, partial);
  rc = ensure_pin(ctx, 4096);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(ctx->pin, dout, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(out, ctx->pin, 32);
  return BPMI_OK;
}
int bpmi_sc_dot(bpmi_ctx *ctx, const uint8_t *a, const uint8_t *b, uint64_t n, uint8_t out[32]) {
  if (!ctx || !out || (n && (!a || !b))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) { memset(out, 0, 32); return BPMI_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage_in(ctx, 64 * n + 512);
  if (rc) return rc;
  char *da = (char *)ctx->stage_in, *db = da + align_up(32 * n, 256);
  HIPCHK(ctx, hipMemcpyAsync(da, a, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(db, b, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  return bpmi_sc_dot_dev(ctx, da, db, n, out);
}
int bpmi_sc_fold_dev(bpmi_ctx *ctx, const void *d_lo, const void *d_hi, const uint8_t x[32], const uint8_t y[32], uint64_t n, void *d_out) {
  if (!ctx || !x || !y || (n && (!d_lo || !d_hi || !d_out))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Sc2 xy;
  memcpy(xy.k1, x, 32); memcpy(xy.k2, y, 32);
  {
    StageTimer t(ctx, ST_SCFOLD);
    hipLaunchKernelGGL(k_sc_fold, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const u32 *)d_lo, (const u32 *)d_hi, xy, (u32)n, (u32 *)d_out);
  }
  HIPCHK(ctx, hipGetLastError());
  return BPMI_OK;
}
int bpmi_sc_fold(bpmi_ctx *ctx, const uint8_t *lo, const uint8_t *hi, const uint8_t x[32], const uint8_t y[32], uint64_t n, uint8_t *out) {
  if (!ctx || !x || !y || (n && (!lo || !hi || !out))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage_in(ctx, 96 * n + 1024);
  if (rc) return rc;
  char *dl = (char *)ctx->stage_in, *dh = dl + align_up(32 * n, 256), *dout = dh + align_up(32 * n, 256);
  HIPCHK(ctx, hipMemcpyAsync(dl, lo, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(dh, hi, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  rc = bpmi_sc_fold_dev(ctx, dl, dh, x, y, n, dout);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

// ---- IPA prover state ---------------------------------------------------------------------------
}  // extern "C"

struct bpmi_ipa {
  bpmi_ctx *ctx;
  uint64_t n0, n;       // initial and current LOGICAL length m
  uint64_t M;           // length of the (unfolded) base arrays g, h;  M = n << d
  u32 d;                // deferred folds
  uint64_t big_m;       // materialisation threshold (ctx option ipa_big_m)
  u32 *g, *h;           // device bases (M points each)
  u32 *g2, *h2;         // device, materialisation targets (M/16 points each)
  u32 *a, *b;           // device, folded in place every round
  u32 *eg, *eh, *eg2, *eh2;   // device, expanded scalars for the deferred L and R MSMs (M each)
  u32 *cg[2], *ch[2];   // device coefficient tables (ping-pong), 2^d entries in cg[cur]
  int cur;
  u32 *u;               // device, 64 B point
  u32 *cl, *cr;         // device, 32 B scalars (stay on device between dot and MSM)
  u32 *partial;         // device scratch for dots
  NafK *nafk[2];        // device, NAF tables for the multifold kernel
  void *block;          // one allocation
  std::vector<sc> hcg, hch;   // host copies of the coefficient tables while 2^d <= 16
  bool lr_done;
};

// deferral policy: bases of 2^18 points or more are folded 16-way at once (an MSM over the
// unfolded bases costs ~2.6 ms per L/R at 2^20 against ~50 ms for the pairwise ladder fold,
// while a K-term ladder stays throughput-bound only for many outputs); smaller bases are
// never folded -- the prover needs L and R, not the folded generators, and a ladder launch
// is ~2 ms of pure latency.
#define IPA_BIG_M_DEFAULT (1u << 18)
#define IPA_BIG_D 4

extern "C" {

static int ipa_alloc(bpmi_ctx *ctx, uint64_t n, bpmi_ipa **out) {
  bpmi_ipa *st = new bpmi_ipa();
  st->ctx = ctx; st->n0 = st->n = st->M = n; st->d = 0; st->cur = 0; st->lr_done = false;
  st->big_m = ctx->opt_ipa_big > 0 ? (uint64_t)ctx->opt_ipa_big : IPA_BIG_M_DEFAULT;
  if (st->big_m < 32) st->big_m = 32;
  const size_t pts = align_up(64 * n, 256), scs = align_up(32 * n, 256);
  const size_t pts2 = align_up(64 * (n / 16 + 1), 256), coef = align_up(32 * n, 256);
  const size_t bytes = pts * 2 + pts2 * 2 + scs * 6 + coef * 4 + 256 * 3 + 32 * 1024 + 2 * align_up(sizeof(NafK), 256);
  hipError_t e = hipMalloc(&st->block, bytes);
  if (e != hipSuccess) { delete st; return fail(ctx, BPMI_E_NOMEM, std::string("hipMalloc(ipa state): ") + hipGetErrorString(e)); }
  char *p = (char *)st->block;
  st->g = (u32 *)p; p += pts;
  st->h = (u32 *)p; p += pts;
  st->g2 = (u32 *)p; p += pts2;
  st->h2 = (u32 *)p; p += pts2;
  st->a = (u32 *)p; p += scs;
  st->b = (u32 *)p; p += scs;
  st->eg = (u32 *)p; p += scs;
  st->eh = (u32 *)p; p += scs;
  st->eg2 = (u32 *)p; p += scs;
  st->eh2 = (u32 *)p; p += scs;
  for (int k = 0; k < 2; k++) { st->cg[k] = (u32 *)p; p += coef; st->ch[k] = (u32 *)p; p += coef; }
  st->u = (u32 *)p; p += 256;
  st->cl = (u32 *)p; p += 256;
  st->cr = (u32 *)p; p += 256;
  st->partial = (u32 *)p; p += 32 * 1024;
  for (int k = 0; k < 2; k++) { st->nafk[k] = (NafK *)p; p += align_up(sizeof(NafK), 256); }
  // coefficient tables start as [1]
  uint8_t one[32] = {1};
  e = hipMemcpyAsync(st->cg[0], one, 32, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(st->ch[0], one, 32, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { (void)hipFree(st->block); delete st; return fail(ctx, BPMI_E_HIP, std::string("ipa_alloc: ") + hipGetErrorString(e)); }
  sc o; memset(&o, 0, sizeof(o)); o.v[0] = 1;
  st->hcg.assign(1, o); st->hch.assign(1, o);
  *out = st;
  return BPMI_OK;
}
static bool is_pow2(uint64_t n) { return n && !(n & (n - 1)); }

int bpmi_ipa_create_dev(bpmi_ctx *ctx, const void *d_g, const void *d_h, const void *d_a, const void *d_b, uint64_t n,
                        const uint8_t u[64], bpmi_ipa **out) {
  if (!ctx || !d_g || !d_h || !d_a || !d_b || !u || !out) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (!is_pow2(n) || n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n must be a power of two <= BPMI_MAX_N");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  bpmi_ipa *st = nullptr;
  int rc = ipa_alloc(ctx, n, &st);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  hipError_t e = hipMemcpyAsync(st->g, d_g, 64 * n, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(st->h, d_h, 64 * n, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(st->a, d_a, 32 * n, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(st->b, d_b, 32 * n, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(st->u, u, 64, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { (void)hipFree(st->block); delete st; return fail(ctx, BPMI_E_HIP, std::string("ipa_create copy: ") + hipGetErrorString(e)); }
  *out = st;
  return BPMI_OK;
}
int bpmi_ipa_create(bpmi_ctx *ctx, const uint8_t *g, const uint8_t *h, const uint8_t *a, const uint8_t *b, uint64_t n,
                    const uint8_t u[64], bpmi_ipa **out) {
  if (!ctx || !g || !h || !a || !b || !u || !out) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (!is_pow2(n) || n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n must be a power of two <= BPMI_MAX_N");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  bpmi_ipa *st = nullptr;
  int rc = ipa_alloc(ctx, n, &st);
  if (rc) return rc;
  hipStream_t s = ctx->stream;
  hipError_t e = hipMemcpyAsync(st->g, g, 64 * n, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(st->h, h, 64 * n, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(st->a, a, 32 * n, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(st->b, b, 32 * n, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(st->u, u, 64, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { (void)hipFree(st->block); delete st; return fail(ctx, BPMI_E_HIP, std::string("ipa_create copy: ") + hipGetErrorString(e)); }
  *out = st;
  return BPMI_OK;
}
uint64_t bpmi_ipa_len(const bpmi_ipa *st) { return st ? st->n : 0; }

int bpmi_ipa_round_LR(bpmi_ipa *st, uint8_t L[64], uint8_t R[64]) {
  if (!st || !L || !R) return BPMI_E_ARG;
  bpmi_ctx *ctx = st->ctx;
  if (st->n < 2) return fail(ctx, BPMI_E_STATE, "ipa already reduced to length 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t np = st->n / 2;
  u32 *a_lo = st->a, *a_hi = st->a + 8 * np, *b_lo = st->b, *b_hi = st->b + 8 * np;
  // cl = <a_lo, b_hi>, cr = <a_hi, b_lo>  (inner_product_prover.py:96-97), kept on the device
  sc_dot_dev_to(ctx, a_lo, b_hi, np, st->cl, st->partial);
  sc_dot_dev_to(ctx, a_hi, b_lo, np, st->cr, st->partial);
  int rc;
  if (st->d == 0) {
    // bases are the current generators: L = <a_lo, g_hi> + <b_hi, h_lo> + cl*u  (:98) as ONE
    // three-segment MSM, R = <a_hi, g_lo> + <b_lo, h_hi> + cr*u  (:99)
    u32 *g_lo = st->g, *g_hi = st->g + 16 * np, *h_lo = st->h, *h_hi = st->h + 16 * np;
    Segs sL = {};
    sL.pts[0] = g_hi; sL.sc[0] = a_lo; sL.n[0] = (u32)np;
    sL.pts[1] = h_lo; sL.sc[1] = b_hi; sL.n[1] = (u32)np;
    sL.pts[2] = st->u; sL.sc[2] = st->cl; sL.n[2] = 1;
    sL.total = (u32)(2 * np + 1);
    Segs sR = {};
    sR.pts[0] = g_lo; sR.sc[0] = a_hi; sR.n[0] = (u32)np;
    sR.pts[1] = h_hi; sR.sc[1] = b_lo; sR.n[1] = (u32)np;
    sR.pts[2] = st->u; sR.sc[2] = st->cr; sR.n[2] = 1;
    sR.total = (u32)(2 * np + 1);
    rc = msm_run_pair(ctx, sL, L, sR, R);
    if (rc) return rc;
  } else {
    // deferred: MSM over the UNFOLDED bases with the fold coefficients multiplied into the
    // scalars (half of them are zero and drop out in the digit pass)
    u32 logm = 0;
    while ((1ull << logm) < st->n) logm++;
    Segs sg[2];
    for (int right = 0; right < 2; right++) {
      u32 *eg = right ? st->eg2 : st->eg, *eh = right ? st->eh2 : st->eh;
      {
        StageTimer t(ctx, ST_SCFOLD);
        hipLaunchKernelGGL(k_ipa_expand, dim3((u32)((st->M + 255) / 256)), dim3(256), 0, ctx->stream, st->a, st->b,
                           st->cg[st->cur], st->ch[st->cur], (u32)st->M, logm, right, eg, eh);
      }
      sg[right] = Segs{};
      sg[right].pts[0] = st->g; sg[right].sc[0] = eg; sg[right].n[0] = (u32)st->M;
      sg[right].pts[1] = st->h; sg[right].sc[1] = eh; sg[right].n[1] = (u32)st->M;
      sg[right].pts[2] = st->u; sg[right].sc[2] = right ? st->cr : st->cl; sg[right].n[2] = 1;
      sg[right].total = (u32)(2 * st->M + 1);
    }
    rc = msm_run_pair(ctx, sg[0], L, sg[1], R);
    if (rc) return rc;
  }
  st->lr_done = true;
  return BPMI_OK;
}

static void host_sc_from(sc &r, const uint8_t b[32]) { memcpy(r.v, b, 32); }

int bpmi_ipa_fold(bpmi_ipa *st, const uint8_t x[32], const uint8_t xinv[32]) {
  if (!st || !x || !xinv) return BPMI_E_ARG;
  bpmi_ctx *ctx = st->ctx;
  if (st->n < 2) return fail(ctx, BPMI_E_STATE, "ipa already reduced to length 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t np = st->n / 2;
  int rc;
  // a' = x a_lo + x^-1 a_hi ; b' = x^-1 b_lo + x b_hi  (:109-110)
  rc = bpmi_sc_fold_dev(ctx, st->a, st->a + 8 * np, x, xinv, np, st->a);
  if (rc) return rc;
  rc = bpmi_sc_fold_dev(ctx, st->b, st->b + 8 * np, xinv, x, np, st->b);
  if (rc) return rc;
  // g' = x^-1 g_lo + x g_hi ; h' = x h_lo + x^-1 h_hi  (:107-108): deferred -- only the
  // coefficient tables double
  const u32 K = 1u << st->d;
  Sc2 xs;
  memcpy(xs.k1, x, 32); memcpy(xs.k2, xinv, 32);
  {
    StageTimer t(ctx, ST_SCFOLD);
    hipLaunchKernelGGL(k_ipa_coef_update, dim3((2 * K + 255) / 256), dim3(256), 0, ctx->stream, st->cg[st->cur], st->ch[st->cur],
                       xs, K, st->cg[st->cur ^ 1], st->ch[st->cur ^ 1]);
  }
  st->cur ^= 1;
  const bool track_host = (2 * K <= MULTIFOLD_MAXK) && st->M >= st->big_m;
  if (track_host) {
    sc X, XI;
    host_sc_from(X, x); host_sc_from(XI, xinv);
    std::vector<sc> ng(2 * K), nh(2 * K);
    for (u32 j = 0; j < 2 * K; j++) {
      sc_mul(ng[j], st->hcg[j >> 1], (j & 1u) ? X : XI);
      sc_mul(nh[j], st->hch[j >> 1], (j & 1u) ? XI : X);
    }
    st->hcg.swap(ng); st->hch.swap(nh);
  }
  st->d += 1;
  st->n = np;
  st->lr_done = false;
  if (st->M >= st->big_m && st->d == IPA_BIG_D && st->n > 1) {
    // materialise the 16-way folded generators: out[i] = sum_t coef[t] * base[i + t*m]
    const u32 K2 = 1u << st->d;
    NafK ha, hb;
    memset(&ha, 0, sizeof(ha)); memset(&hb, 0, sizeof(hb));
    ha.top = hb.top = -1;
    for (u32 t = 0; t < K2; t++) {
      host_naf((const uint8_t *)st->hcg[t].v, ha.nz[t], ha.sg[t], ha.top);
      host_naf((const uint8_t *)st->hch[t].v, hb.nz[t], hb.sg[t], hb.top);
    }
    HIPCHK(ctx, hipMemcpyAsync(st->nafk[0], &ha, sizeof(NafK), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(st->nafk[1], &hb, sizeof(NafK), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // ha / hb are stack objects
    MultifoldJob ja = {st->g, st->g2}, jb = {st->h, st->h2};
    {
      StageTimer t(ctx, ST_LINCOMB2);
      hipLaunchKernelGGL(k_ec_multifold, dim3((u32)((2 * st->n + 255) / 256)), dim3(256), 0, ctx->stream, ja, jb, st->nafk[0], st->nafk[1],
                         (u32)st->n, K2);
    }
    HIPCHK(ctx, hipGetLastError());
    // the folded generators become the new bases
    std::swap(st->g, st->g2);
    std::swap(st->h, st->h2);
    st->M = st->n;
    st->d = 0;
    uint8_t one[32] = {1};
    HIPCHK(ctx, hipMemcpyAsync(st->cg[st->cur], one, 32, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(st->ch[st->cur], one, 32, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    sc o; memset(&o, 0, sizeof(o)); o.v[0] = 1;
    st->hcg.assign(1, o); st->hch.assign(1, o);
  }
  return BPMI_OK;
}

int bpmi_ipa_finish(bpmi_ipa *st, uint8_t a[32], uint8_t b[32]) {
  if (!st || !a || !b) return BPMI_E_ARG;
  bpmi_ctx *ctx = st->ctx;
  if (st->n != 1) return fail(ctx, BPMI_E_STATE, "ipa not yet reduced to length 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(a, st->a, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(b, st->b, 32, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

void bpmi_ipa_destroy(bpmi_ipa *st) {
  if (!st) return;
  (void)hipSetDevice(st->ctx->device);
  (void)hipStreamSynchronize(st->ctx->stream);
  (void)hipFree(st->block);
  delete st;
}

// ---- profiling -----------------------------------------------------------------------------------
static void prof_drain(bpmi_ctx *ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &e : ctx->evs) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { ctx->prof_ms[e.stage] += ms; ctx->prof_calls[e.stage]++; }
    (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
  }
  ctx->evs.clear();
}
int bpmi_profile(bpmi_ctx *ctx, int enable) {
  if (!ctx) return BPMI_E_ARG;
  prof_drain(ctx);
  ctx->prof = enable != 0;
  return BPMI_OK;
}
int bpmi_profile_reset(bpmi_ctx *ctx) {
  if (!ctx) return BPMI_E_ARG;
  prof_drain(ctx);
  for (int i = 0; i < BPMI_NSTAGES; i++) { ctx->prof_ms[i] = 0; ctx->prof_calls[i] = 0; }
  return BPMI_OK;
}
int bpmi_profile_read(bpmi_ctx *ctx, double ms[BPMI_NSTAGES], uint64_t calls[BPMI_NSTAGES]) {
  if (!ctx || !ms || !calls) return BPMI_E_ARG;
  prof_drain(ctx);
  for (int i = 0; i < BPMI_NSTAGES; i++) { ms[i] = ctx->prof_ms[i]; calls[i] = ctx->prof_calls[i]; }
  return BPMI_OK;
}
const char *bpmi_profile_stage_name(int stage) {
  if (stage < 0 || stage >= BPMI_NSTAGES) return "";
  return STAGE_NAMES[stage];
}

}  // extern "C"
