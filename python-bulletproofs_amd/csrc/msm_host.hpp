// msm_host.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).
// Host orchestration of one MSM: geometry, workspace layout, enqueue / finish on a lane.
#pragma once

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
//   [w0, w0 + wcount): the windows this call handles (wcount = 0: all of them)
static int msm_enqueue(bpmi_ctx *ctx, int lane, const Segs &segs, u32 w0 = 0, u32 wcount = 0) {
  const uint64_t n = segs.total;
  bpmi_ctx::PendingMsm &pd = ctx->pend[lane];
  pd.active = false;
  if (n == 0) return BPMI_OK;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  MsmGeom g;
  g.n = (u32)n;
  g.c = pick_window_bits(ctx, n);
  g.W = wcount ? wcount : 255u / g.c + 1u;
  g.w0 = w0;
  g.B = 1u << (g.c - 1);
  g.G = g.W * g.B;
  g.L = ctx->opt_chunk > 0 ? (u32)ctx->opt_chunk : (n >= (1u << 19) ? 64u : (n >= (1u << 16) ? 32u : 16u));   // tools/tune_msm.py sweeps
  g.nv = (g.B <= 256u) ? 1u : (g.c + 4u) / 5u;     // small windows: direct weighted sum, one value per window
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
      hipLaunchKernelGGL(k_partition, dim3(std::min<u32>(ntiles, 2048)), dim3(1024), 0, st, segs, g, w.P, w.coarse_cursor, w.dig);
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
    if (g.B <= 256u) {
      hipLaunchKernelGGL(k_window_weighted_small, dim3(g.W), dim3(g.B < 64u ? 64u : g.B), 0, st, g, w.buckets, w.E);
    } else {
      hipLaunchKernelGGL(k_bucket_digit_sums, dim3(g.W * g.nv * 31), dim3(256), 0, st, g, w.buckets, w.D);   // 128 / 64 threads measured slower
      hipLaunchKernelGGL(k_weighted31, dim3(g.W * g.nv), dim3(64), 0, st, w.D, w.E);
    }
  }
  debug_sync(ctx, "ST_BREDUCE", st);
  {
    StageTimer t(ctx, ST_TAIL, st);
    const int tail = wcount ? 2 : (ctx->opt_tail ? ctx->opt_tail : 2);      // window groups are combined on the host
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
    bpmi_host::tail_combine(out, (const u32 *)pin, pd.W, pd.nv, pd.c);     // host_tail.hpp
  }
  pd.active = false;
  debug_sync(ctx, "ST_TAIL", st);
  return BPMI_OK;
}
// One MSM as two window groups, one per lane (option "split").  Measured on MI355X
// (tools/try_split.py): the lanes' kernels barely overlap -- every stage is bound by the
// same ALUs or by LDS atomics in blocks that cannot co-reside with the accumulation's --
// so the gain is the kernels' tails only (+5 % at n = 2^20, -8 % at 2^19); off by default.
static int msm_run_split(bpmi_ctx *ctx, const Segs &segs, uint8_t out[64]) {
  const u32 c = pick_window_bits(ctx, segs.total), W = 255u / c + 1u, Wa = W / 2;
  int rc = ensure_lane(ctx, 1);
  if (rc) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream1, ctx->ev_fork, 0));
  rc = msm_enqueue(ctx, 0, segs, 0, Wa);
  if (rc) return rc;
  rc = msm_enqueue(ctx, 1, segs, Wa, W - Wa);
  if (rc) return rc;
  bpmi_ctx::PendingMsm &p0 = ctx->pend[0], &p1 = ctx->pend[1];
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream1));
  const size_t b0 = 4ull * XYZZ_WORDS * p0.W * p0.nv, b1 = 4ull * XYZZ_WORDS * p1.W * p1.nv;
  std::vector<u32> E((b0 + b1) / 4);
  memcpy(E.data(), ctx->pin, b0);
  memcpy((char *)E.data() + b0, ctx->pin1, b1);
  bpmi_host::tail_combine(out, E.data(), W, p0.nv, c);
  p0.active = p1.active = false;
  return BPMI_OK;
}
static int msm_run(bpmi_ctx *ctx, const Segs &segs, uint8_t out[64]) {
  if (segs.total > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (ctx->opt_split == 1 && segs.total >= 2) return msm_run_split(ctx, segs, out);
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
