// msm_host.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).
// Host orchestration of one MSM: geometry, workspace layout, enqueue / finish on a lane.
#pragma once

// ------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Window bits from the tools/tune_msm.py sweeps on MI355X (profiles/r01_tune_msm_after_sort_and_reduce_rewrites.txt).
// Besides the usual bucket-count trade-off, windows whose TOP window holds only a few
// bits (255 mod c small: c = 15, 14, 12, 11) concentrate a whole window's digits in a
// handful of buckets, so c in {8, 16} (top window 7, 15 bits) are preferred.
// the 12-bit mixed-width pipeline's lower end (when the one-block-per-window kernel is switched off; with it -- three blocks per window --
// that kernel keeps its whole range for single MSMs too: 8 192 pairs 0.184 ms against 0.207, profiles/r05_mid_kernel_parts_ab.txt)
#define MID_SINGLE_MAX_MIXED 5632u
static u32 pick_window_bits(const bpmi_ctx *ctx, uint64_t n) {
  if (ctx->opt_c >= 2 && ctx->opt_c <= 16) return (u32)ctx->opt_c;
  // Round 5, with mixed window widths (MsmGeom.top2: no carry window and no short top window at ANY width; tools/r05_exp_mixed.sh,
  // profiles/r05_mixed_window_widths_sweep.txt): 12 bits (17 + 4 windows, 51 k buckets) from the end of the one-block kernel's range (8 448)
  // to 19 000 pairs, 13 bits (10 + 9 windows, 115 k buckets) to 185 000 -- 2^16: 0.272 ms against 0.299 for c = 15 and 0.311 for c = 16 --,
  // 16 bits beyond (14 / 15 tie with it around 2 x 10^5 and lose above).  Without them: the table of the first half of the round
  // (c = 15 with its one wide window from 15 360 to 2^17, 12 from 10 240, 8 below; profiles/r05_window_table_sweep.txt)
  if (ctx->opt_mixed && ctx->opt_top2) {
    if (n >= 185000u) return 16;
    if (n >= 19000u) return 13;
    if (n >= MID_SINGLE_MAX_MIXED) return 12;
  } else {
    if (n >= (1u << 17)) return 16;
    if (n >= 15360u) return ctx->opt_top2 ? 15 : 16;
    if (n >= 10240u) return 12;
  }
  if (n >= (1u << 10)) return 8;
  u32 lg = 0;
  while ((1ull << (lg + 1)) <= n) lg++;
  const int c = (int)lg - 2;
  return (u32)(c < 4 ? 4 : c);
}

// tools/try_small.py (profiles/r04_small_msm_vs_bucket_pipeline.txt): the one-launch kernel wins up to 2^12 pairs (0.21 ms against
// 0.26) and loses at 2^13 (0.33 against 0.29); the threshold sits just above 4 097 = the L / R of an inner-product round over
// 4 096 generators (a 64-bit x 32 aggregated range proof; every late round of a larger proof after its product fold)
#define SMALL_N_DEFAULT 4608
// a PAIR of MSMs (bpmi_msm2, the L / R of an inner-product round) whose larger one has this many pairs or more, up to MID_NMAX, is ONE
// launch of k_msm_mid (option "mid_min": 0 default, -1 never)
#define MID_MIN_DEFAULT 1536
#define MID_SINGLE_MIN_DEFAULT 2560      // one MSM at a time (option "mid_single_min": 0 default, -1 never)

// blocks per window of k_msm_mid (option "mid_parts": 0 = this rule, 1 .. 4 forced): from 3 000 pairs three -- a part more costs every window
// one more addition in the host tail (~13 us per result), a third of the pairs less per block saves 18 us at 2 049 pairs, 37 at 4 097, 70 at 8 193
static u32 mid_parts(const bpmi_ctx *ctx, uint64_t n) {
  if (ctx->opt_mid_parts >= 1 && ctx->opt_mid_parts <= 4) return (u32)ctx->opt_mid_parts;
  return n >= 3000u ? 3u : 1u;
}
struct MsmWs {
  u32 *glv_sub, *glv_bx;      // GLV: 2n x 16 B magnitudes, n x 32 B beta x
  unsigned char *glv_neg;     // GLV: 2n sign bytes
  u32 *dig, *hist, *off, *cursor, *bsum, *sidx, *buckets, *chunk_key, *coarse_hist, *coarse_off, *coarse_cursor;
  unsigned short *dig16;      // path 2: recoded digits, window-major
  unsigned char *negs;        // path 2: 1 = the scalar was replaced by q - s
  u32 P;          // partitions of sort path 2 (0 = path 1)
  u32 *rec_key[2], *rec_pt[2];
  u32 *D, *E, *F, *out;
  size_t total;
  u32 nscan_blocks, rec0_max, nchunks;
};
static void msm_layout(const MsmGeom &g, MsmWs &w, char *base, bool glv = false) {
  size_t o = 0;
  auto take = [&](size_t bytes) { char *p = base ? base + o : nullptr; o += align_up(bytes, 256); return (u32 *)p; };
  w.glv_sub = take(glv ? 16ull * g.n : 0);               // g.n = virtual pairs
  w.glv_bx = take(glv ? 16ull * g.n : 0);
  w.glv_neg = (unsigned char *)take(glv ? g.n : 0);
  const size_t nW = (size_t)g.n * g.W;
  w.nscan_blocks = (u32)((g.G + SCAN_TILE - 1) / SCAN_TILE);
  w.nchunks = (u32)((nW + g.L - 1) / g.L);                  // threads of k_accum_l0
  w.rec0_max = 2u * (g.fuse ? (w.nchunks + 63u) / 64u : w.nchunks);
  const u32 rec1_max = 2 * ((w.rec0_max + 255) / 256);
  // sort path 2 (LDS partition sort) when the bucket key has more than 8 bits and the
  // packed entry (8-bit lo | sign | 23-bit index) fits; path 1 (global atomics) otherwise
  w.P = (g.c >= 10 && g.n <= (1u << 23)) ? (g.G >> 8) : 0;
  w.hist = take(4ull * g.G);                 // path 1 only
  w.off = take(4ull * (g.G + 1));
  w.cursor = take(4ull * g.G);               // path 1 only
  w.bsum = take(4ull * (w.nscan_blocks + 1));
  w.coarse_hist = take(4ull * COARSE_HIST_WORDS);
  w.coarse_off = take(4ull * (PART_MAX + 1));
  w.coarse_cursor = take(4ull * (PART_MAX + 1));
  w.dig = take(4ull * nW);                   // path 1: digits; path 2: partitioned entries
  w.sidx = take(4ull * nW);
  w.dig16 = (unsigned short *)take(w.P ? 2ull * nW : 0);
  w.negs = (unsigned char *)take(w.P ? g.n : 0);
  w.chunk_key = take(4ull * (w.nchunks + 1));
  w.buckets = take(4ull * XYZZ_WORDS * g.G);
  w.rec_key[0] = take(4ull * w.rec0_max);
  w.rec_pt[0] = take(4ull * XYZZ_WORDS * w.rec0_max);
  w.rec_key[1] = take(4ull * rec1_max);
  w.rec_pt[1] = take(4ull * XYZZ_WORDS * rec1_max);
  w.D = take(4ull * XYZZ_WORDS * g.W * (g.B > 256u ? (1u << ((g.c + 1u) / 2u)) + (1u << (g.c / 2u)) : 1u));   // stage-1 digit sums
  w.E = take(4ull * XYZZ_WORDS * g.W * 4);
  w.F = take(4ull * XYZZ_WORDS * g.W * 64);       // k_digit_final_spread: 16 sums per (window, array)
  w.out = take(64);
  w.total = o;
}

static u32 msb_index(u32 v) { u32 k = 0; while ((2u << k) <= v) k++; return k; }     // floor(log2 v), v >= 1
// lanes per sum for `epl` elements per lane: a sum lives in ONE wave, so the group is widened to the largest size that keeps the same
// number of sums per wave (12 lanes -> 5 sums per wave; 13 .. 16 lanes -> 4)
static void digit_group(DigitJob &j, u32 elements, u32 epl) {
  u32 lanes = (elements + epl - 1u) / epl;
  if (lanes > 64u) lanes = 64u;
  if (lanes < 1u) lanes = 1u;
  j.gpw = 64u / lanes;
  j.glanes = 64u / j.gpw;
  if (j.glanes > elements) j.glanes = elements ? elements : 1u;
}
static u32 digit_job_waves(const DigitJobs &J, u32 k) {
  const uint64_t sums = (uint64_t)J.j[k].cnt * J.j[k].nsums;
  return (u32)((sums + J.j[k].gpw - 1u) / J.j[k].gpw);
}
static u32 digit_job_blocks(const DigitJobs &J, u32 k) { return (digit_job_waves(J, k) + 3u) / 4u; }
// the two jobs (by lo, by hi) that split every array [in_off .. in_off + N) of `cnt` arrays at bit s;
// results at out_off (2^s - 1 sums) and behind them (N >> s sums)
static DigitJobs digit_jobs2(u32 cnt, u32 in_off, u32 in_stride, u32 N, u32 s, u32 out_off, u32 out_stride, u32 epl) {
  DigitJobs J;
  memset(&J, 0, sizeof(J));
  J.njobs = 2;
  for (u32 type = 0; type < 2; type++) {
    DigitJob &j = J.j[type];
    j.cnt = cnt;
    j.in_off = in_off; j.in_stride = in_stride; j.N = N; j.s = s; j.type = type;
    j.nsums = type ? (N >> s) : ((1u << s) - 1u);
    digit_group(j, type ? (1u << s) : ((N - 1u) >> s) + 1u, epl);
    j.out_off = out_off + (type ? (1u << s) - 1u : 0u); j.out_stride = out_stride;
  }
  J.j[0].blk0 = 0;
  J.j[1].blk0 = digit_job_blocks(J, 0);
  return J;
}
static DigitJobs digit_jobs_concat(const DigitJobs &a, const DigitJobs &b) {
  DigitJobs J = a;
  u32 blk = a.j[1].blk0 + digit_job_blocks(a, 1);
  for (u32 k = 0; k < 2; k++) { J.j[2 + k] = b.j[k]; J.j[2 + k].blk0 = blk; blk += digit_job_blocks(b, k); }
  J.njobs = 4;
  return J;
}

// ---- replay of an MSM's launch sequence as a HIP graph (option "graphs") -------------------------------------------------------
// An MSM is a dozen to 17 launches; the host pays ~9 us for each, and for every MSM below ~2^18 pairs the GPU finishes the sort's
// short kernels faster than the host can queue the next one (round 4, profiles/r04_C3_kernel_timeline_mid_round.txt: 160 us between
// the first kernels of the two lanes of one inner-product round).  Callers repeat the SAME sequence -- same input arrays, same
// size, same workspace: the rounds of an inner-product argument, a verifier's batches, a benchmark loop -- so the sequence is
// captured once per (lane, slot, inputs, geometry, options) and launched as one graph afterwards.  Only launches are captured: the
// slot's completion event is recorded behind the graph, allocation happens before the capture, a reallocation of the workspace or
// of the slot's pinned buffer drops the cache.  Not used with stage timers, debug syncs or the chained asynchronous pipeline.
struct MsmGraphKey {
  int lane, slot, glv, small, opt_quad, opt_tail, opt_epl, opt_hist_threads, opt_hist_blocks;
  u32 w0, wcount;
  Segs segs;
  MsmGeom g;
  const void *ws, *pin;
};
struct MsmGraphEntry {
  MsmGraphKey key;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  u32 W = 0, nv = 0, c = 0; int tail = 2; TailOffs to;
};
struct MsmGraphCache { std::vector<MsmGraphEntry> entries; };
static void msm_graphs_clear(bpmi_ctx *ctx) {
  if (!ctx->graphs || ctx->graphs->entries.empty()) return;
  // a replayed graph may still be running on any lane: nothing is destroyed under it
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream1) (void)hipStreamSynchronize(ctx->stream1);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  for (auto &e : ctx->graphs->entries) { if (e.exec) (void)hipGraphExecDestroy(e.exec); if (e.graph) (void)hipGraphDestroy(e.graph); }
  ctx->graphs->entries.clear();
}
// ends a capture that an error path would otherwise leave open on the stream
struct CaptureGuard {
  hipStream_t st; bool open = false;
  explicit CaptureGuard(hipStream_t s) : st(s) {}
  ~CaptureGuard() { if (open) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(st, &g); if (g) (void)hipGraphDestroy(g); (void)hipGetLastError(); } }
};

// The geometry of one MSM of n pairs under the ctx's options: which kernel family (mid: the one-block-per-window kernel, small: the
// one-launch kernel, neither: the bucket pipeline), window bits, windows, buckets, chunk length.  A function of (options, n, w0,
// wcount, ctx->chain_accum) only: msm_enqueue computes it, bpmi_msm_geometry reports it (bench.py's multiply-add count).
static void msm_pick_geometry(const bpmi_ctx *ctx, uint64_t n, u32 w0, u32 wcount, MsmGeom &g, bool &mid, bool &small, bool &glv) {
  g = MsmGeom{};
  g.n = (u32)n;
  const uint64_t small_max = ctx->opt_small < 0 ? 0 : (ctx->opt_small ? (uint64_t)ctx->opt_small : SMALL_N_DEFAULT);
  // the one-block-per-window bucket kernel (k_msm_mid) between the small-MSM kernel and the pipeline
  // (measured, profiles/r04_mid_kernel_latency.txt: one MSM at a time it wins from ~2 500 pairs -- 0.18 ms against 0.21 at 3 000, 0.25
  // against 0.29 at 8 193 --, a PAIR in one launch from ~1 500 pairs each: 0.29 ms against 0.44 for two lanes of the pipeline at 4 097)
  const uint64_t mid_single = ctx->opt_mid_single > 0 ? (uint64_t)ctx->opt_mid_single : MID_SINGLE_MIN_DEFAULT;
  // (with one block per window the 12-bit mixed-width pipeline passes it at ~5 600 pairs: mid_parts = 1 keeps that bound)
  const uint64_t mid_single_max = (ctx->opt_mixed && ctx->opt_top2 && ctx->opt_mid_single == 0 && ctx->opt_mid_parts == 1) ? MID_SINGLE_MAX_MIXED - 1u : MID_NMAX;
  mid = ctx->opt_mid_single >= 0 && n >= mid_single && n <= mid_single_max && ctx->opt_c == 0 && wcount == 0 && ctx->opt_glv <= 0;
  small = !mid && n <= small_max && ctx->opt_c == 0 && wcount == 0;
  // GLV (option "glv" = 1; OFF by default): 2n virtual pairs with 128-bit scalars (+ 1 bit of signed-digit carry) instead of n
  // with 255-bit ones: as many bucket additions, half the windows.  Measured (profiles/r03_glv_msm_on_off.txt) it LOSES at every
  // size from 2^15: the bucket reduction is bound by the depth of its addition chains, not by the number of windows (0.16 ms
  // with 9 windows as with 16); a fifth of the 128-bit magnitudes exceed 2^127, so the signed recoding carries into a ninth
  // window whose entries all land in ONE bucket (the sort's heavy-partition path: 0.09 -> 0.33 ms at 2^20, segmented scan
  // 0.065 -> 0.14); and an entry's x and y come from two arrays (two 32-byte requests instead of one 64-byte one: accumulate
  // 0.82 -> 1.02 ms).  Kept behind the option, with its tests, as the record of the experiment.  The sorted entry packs a
  // 23-bit index, so 2n must fit it.
  glv = ctx->opt_glv > 0 && !small && wcount == 0 && n >= 2 && 2 * n <= (1ull << 23);
  if (glv) g.n = (u32)(2 * n);
  g.c = mid ? MID_C : (small ? SMALL_C : pick_window_bits(ctx, n));
  // Mixed window widths (round 5; MsmGeom.top2 = Wb): W = 256 / c windows of which the last Wb = 256 - W c are c + 1 bits wide with 2B
  // buckets, so the windows cover the 256 bit positions exactly -- no carry window, no short top window.  c = 15: 16 + 1 (the
  // "unsigned last window" of the first half of the round is this case), c = 14: 14 + 4, c = 13: 10 + 9, c = 12: 17 + 4, c = 11: 20 + 3,
  // c = 10: 19 + 6; c = 16 is uniform by itself (16 windows of 16 bits).  LDS-sort path only (c >= 10), never for window groups / GLV.
  g.top2 = 0;
  if (g.c >= 10u && g.c <= 15u && !glv && !wcount && !mid && !small && ctx->opt_top2 && (g.c == 15u || ctx->opt_mixed)) g.top2 = 256u - (256u / g.c) * g.c;
  g.W = wcount ? wcount : (glv ? 128u / g.c + 1u : (g.top2 ? 256u / g.c : 255u / g.c + 1u));
  g.w0 = w0;
  g.B = 1u << (g.c - 1);
  g.G = (g.W + g.top2) * g.B;
  // up to 2^17 pairs a partition of the sort (<= n entries: one window's) is sorted by ONE block whatever its size (k_fine_sort_part)
  // (c = 16 only above 2^15: the short top window of c = 12 .. 14 is ONE partition of n entries by construction, and one block's two passes
  // over 2^17 entries are 0.15 ms -- measured, profiles/r05_mid_size_ab.txt)
  g.inblock = (ctx->opt_inblock && (n <= (1u << 15) || ((g.c == 16u || g.top2) && n <= (1u << 17)))) ? 1u : 0u;
  // tools/tune_msm.py sweeps; on the two-lane pipeline 86 entries per thread fill the 3 waves per SIMD exactly once at n = 2^20
  // (profiles/r02_chunk_sweep_two_lanes.txt).  Round 3, at steady clocks (profiles/r03_chunk_sweep_steady_clocks.txt,
  // r03_chunk_length_vs_kernel_events.txt): L = 128 -- one round of TWO waves per SIMD, room for a 144-VGPR wave of the other lane's
  // segmented scan / bucket reduction -- measures 0.99-1.01 ms per step against 1.04 under bench.py, but ONLY there: the HIP events
  // bench.py records around this kernel change the interleaving of the two lanes, and without them (every other caller) L = 128
  // costs 1.19 ms against 1.04.  86 stays; anything between the quantisation points is far worse (L = 120: 1.21).
  // Round 6 (profiles/r06_wave_priority_and_chunk_ab.txt): where another MSM's kernels run BESIDE this accumulation (the chained pipeline
  // of bpmi_msm_dev_enqueue and of the slices of a large MSM; a synchronous pair from 2^19 pairs) the chunk is the length that makes the
  // accumulation `rounds` rounds of three waves per SIMD, ceil(W n / (64 x 3072 x rounds)), never under 20 entries -- at n = 2^20: 86 for one
  // round (rounds 2 .. 5's choice: every wave slot of the chip taken once, for the whole kernel), 29 for three (the default now).  With one
  // round the other lane's sort and reduction find NO wave slot until the accumulation ends (k_digit_sums 750 us instead of 93, 0.3 ms in
  // every 1.9 without an accumulation running); with three the slots turn over every 0.26 ms, the other lane's kernels become resident
  // beside the accumulation, and -- now that they are resident -- raising their waves' issue priority (option "priority", on by default
  // from this round) lets their dependent chains run at their own speed: 1.043 -> 0.985 ms per step from the chunks alone, -> 0.963 / 0.915
  // (two boxes) with the priority.  Option "rounds" (0 = 3).
  const u32 rounds = ctx->opt_rounds > 0 ? (u32)ctx->opt_rounds : 3u;
  const u32 L_lanes = (u32)std::max<uint64_t>(20, ((uint64_t)g.W * g.n + 64ull * 3072 * rounds - 1) / (64ull * 3072 * rounds));
  // One MSM at a time (and the pairs of the IPA): the accumulation as ONE round of three waves per SIMD (3072 waves) from the size
  // where that leaves chunks of 20 entries, one round of two below (a chunk is a chain of dependent additions and every chunk
  // costs a pair of partial records), at most 64.  Powers of two missed the quantisation points:
  // 311 427 pairs at L = 32 are 2 433 waves -- a third round for a fifth of the chip, 0.635 ms against 0.590 at L = 26
  // (profiles/r03_chunk_sweep_wave_quantisation.txt).
  u32 L_one = 64u;
  {
    const uint64_t e_max = (uint64_t)g.W * n;
    auto chunks_for = [&](uint64_t waves) { return (u32)((e_max * 1000 + 64 * waves * 1005 - 1) / (64 * waves * 1005)); };   // 0.5 % over is no extra round
    const u32 l3 = chunks_for(3072), l2 = chunks_for(2048), l1 = chunks_for(1024);
    // Round 5 (profiles/r05_chunk_length_mid_sizes.txt): below ~50 000 pairs the old floor of 8 entries per chunk left 500 .. 1 600 waves
    // -- whatever the count, a SIMD with two waves sets the time -- and the wave counts just under 2 048 win at every size measured
    // (32 768 pairs: L = 5, 1 741 waves, 0.275 ms against 0.293 at L = 8 with 1 088); under three entries per chunk one wave per SIMD
    L_one = l3 >= 20u ? l3 : (l2 >= 3u ? l2 : (l1 < 2u ? 2u : l1));
    if (L_one > 64u) L_one = 64u;
  }
  g.L = ctx->opt_chunk > 0 ? (u32)ctx->opt_chunk : ((n >= (1u << 19) && (ctx->chain_accum || ctx->beside)) ? L_lanes : L_one);
  g.nv = (g.B <= 256u) ? 1u : 4u;                  // partial sums per window handed to the tail
  g.prio = ctx->opt_prio == 1 ? 15u : (ctx->opt_prio > 1 ? (u32)(ctx->opt_prio & 15) : 0u);      // (1 = every stage; 16 + mask = those stages)
  g.fuse = ctx->opt_fuse ? 1u : 0u;
}
// Enqueue every GPU stage of one MSM on `lane` (0 = the ctx stream, 1 = the second lane: stream +
// workspace), including the device->pinned-host copy the tail needs, into pending slot `slot`
// (pinned buffer + completion event); returns without synchronising.
//   [w0, w0 + wcount): the windows this call handles (wcount = 0: all of them)
//   phase: 0 = everything; 1 = the sort only (recoding .. sorted entries), nothing becomes pending; 2 = the rest of an MSM whose
//   sort a phase-1 call with the SAME arguments queued on this lane (the geometry is a function of the arguments: it is simply
//   computed again).  A synchronous pair queues both sorts before either accumulation (msm_run_pair).  MSMs on the one-launch
//   kernels or on GLV scalars have no separate sort: phase 1 does nothing for them and phase 2 everything.
static int msm_enqueue(bpmi_ctx *ctx, int lane, int slot, const Segs &segs_in, u32 w0 = 0, u32 wcount = 0, int phase = 0) {
  Segs segs = segs_in;
  const uint64_t n = segs.total;
  bpmi_ctx::PendingMsm &pd = ctx->pend[slot];
  if (pd.active) return fail(ctx, BPMI_E_STATE, "an MSM is still pending in this slot (bpmi_msm_finish it first)");
  pd.active = false;
  if (n == 0) return BPMI_OK;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  MsmGeom g{};
  bool mid, small, glv;
  msm_pick_geometry(ctx, n, w0, wcount, g, mid, small, glv);
  MsmWs w;
  msm_layout(g, w, nullptr, glv);
  int rc = ensure_lane(ctx, lane);
  if (rc) return rc;
  rc = ensure_ws_lane(ctx, lane, w.total);
  if (rc) return rc;
  msm_layout(g, w, (char *)lane_ws(ctx, lane), glv);
  hipStream_t st = lane_stream(ctx, lane);
  // the slot's pinned buffer before anything is queued (a capture must not allocate)
  rc = ensure_pin_slot(ctx, slot, std::max<size_t>(4096, 4ull * XYZZ_WORDS * g.W * 4));
  if (rc) return rc;
  // Round 4: the kernels that produce an MSM's last device-side values (the window sums, or the point of the device tail) write
  // them straight into the slot's page-locked host buffer -- it is mapped into the device's address space -- instead of into the
  // workspace with a copy behind: one kernel boundary less on a path where every boundary is ~8 us (option "direct_result" = 0: the copy)
  u32 *const E_dst = ctx->opt_direct ? (u32 *)pd.pin : w.E;
  CaptureGuard cap(st);
  MsmGraphKey key;
  const bool use_graph = phase == 0 && ctx->opt_graph && !ctx->prof && !g_debug_sync && !ctx->chain_accum;
  const bool has_sort_phase = !mid && !small && !glv;
  if (phase == 1 && !has_sort_phase) return BPMI_OK;
  const bool skip_sort = phase == 2 && has_sort_phase;
  // queued behind the graph or behind the launches: the completion event, the slot's bookkeeping
  auto commit = [&](u32 W_, u32 nv_, u32 c_, int tail_, const TailOffs &to_) -> int {
    if (cap.open) {
      MsmGraphEntry e;
      memcpy(&e.key, &key, sizeof(key)); e.W = W_; e.nv = nv_; e.c = c_; e.tail = tail_; e.to = to_;
      cap.open = false;
      HIPCHK(ctx, hipStreamEndCapture(st, &e.graph));
      hipError_t ie = hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0);
      if (ie != hipSuccess) { (void)hipGraphDestroy(e.graph); return fail(ctx, BPMI_E_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie)); }
      if (!ctx->graphs) ctx->graphs = new MsmGraphCache();
      if (ctx->graphs->entries.size() >= 48) msm_graphs_clear(ctx);
      ctx->graphs->entries.push_back(e);
      HIPCHK(ctx, hipGraphLaunch(e.exec, st));
    }
    HIPCHK(ctx, hipEventRecord(pd.done, st));
    pd.active = true; pd.W = W_; pd.nv = nv_; pd.c = c_; pd.tail = tail_; pd.to = to_;
    HIPCHK(ctx, hipGetLastError());
    return BPMI_OK;
  };
  if (use_graph) {
    memset(&key, 0, sizeof(key));
    key.lane = lane; key.slot = slot; key.glv = glv; key.small = small ? 1 : (mid ? 2 : 0); key.opt_quad = ctx->opt_quad; key.opt_tail = ctx->opt_tail; key.opt_epl = ctx->opt_epl;
    key.opt_hist_threads = ctx->opt_hist_threads; key.opt_hist_blocks = ctx->opt_hist_blocks; key.w0 = w0; key.wcount = wcount;
    memcpy(&key.segs, &segs_in, sizeof(Segs)); memcpy(&key.g, &g, sizeof(MsmGeom));
    key.ws = lane_ws(ctx, lane); key.pin = pd.pin;
    if (ctx->graphs)
      for (auto &e : ctx->graphs->entries)
        if (!memcmp(&e.key, &key, sizeof(key))) {
          HIPCHK(ctx, hipGraphLaunch(e.exec, st));
          return commit(e.W, e.nv, e.c, e.tail, e.to);
        }
    HIPCHK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    cap.open = true;
  }
  if (glv) {
    StageTimer t(ctx, ST_DIGITS, st);
    hipLaunchKernelGGL(k_glv_prepare, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, segs, (u32)n, w.glv_sub, w.glv_neg, w.glv_bx);
    segs.glv_sub = w.glv_sub; segs.glv_neg = w.glv_neg; segs.glv_bx = w.glv_bx;
  }
  TailOffs to;
  memset(&to, 0, sizeof(to));
  to.nv = 1;
  if (mid) {
    const u32 parts = mid_parts(ctx, n);
    g.nv = parts;
    to.nv = parts;                                   // (every part of a window at bit offset 0: the host tail adds them)
    {
      StageTimer t(ctx, ST_ACCUM, st);
      MidPair mp;
      memset(&mp, 0, sizeof(mp));
      mp.segs[0] = segs; mp.g[0] = g; mp.E[0] = E_dst;
      hipLaunchKernelGGL(k_msm_mid, dim3(g.W, 1, parts), dim3(MID_THREADS), 0, st, mp);
    }
    debug_sync(ctx, "k_msm_mid", st);
    if (!ctx->opt_direct) HIPCHK(ctx, hipMemcpyAsync(pd.pin, w.E, 4ull * XYZZ_WORDS * g.W * parts, hipMemcpyDeviceToHost, st));
    return commit(g.W, parts, g.c, 2, to);
  }
  if (small) {
    g.nv = 1;
    {
      StageTimer t(ctx, ST_ACCUM, st);
      const u32 threads = (u32)std::min<uint64_t>(256, (n + 63) / 64 * 64);
      const u32 S = (u32)std::min<uint64_t>(64, (n + 255) / 256);
      hipLaunchKernelGGL(k_msm_small, dim3(g.W, S), dim3(threads), 0, st, segs, g, S > 1 ? w.buckets : E_dst);
      if (S > 1) hipLaunchKernelGGL(k_small_combine, dim3(g.W), dim3(64), 0, st, w.buckets, S, E_dst);
    }
    debug_sync(ctx, "k_msm_small", st);
    const size_t eb = 4ull * XYZZ_WORDS * g.W;
    if (!ctx->opt_direct) HIPCHK(ctx, hipMemcpyAsync(pd.pin, w.E, eb, hipMemcpyDeviceToHost, st));
    return commit(g.W, 1, g.c, 2, to);
  }
  const u32 nblk_n = (u32)std::min<uint64_t>(((uint64_t)g.n + 255) / 256, 8192);
  if (!skip_sort) {
  {
    StageTimer t(ctx, ST_MISC, st);
    if (w.P) HIPCHK(ctx, hipMemsetAsync(w.coarse_hist, 0, 4ull * COARSE_HIST_WORDS, st));      // (a multiple of 256 bytes: ONE fill kernel, not an aligned part and a tail)
    else {
      HIPCHK(ctx, hipMemsetAsync(w.hist, 0, 4ull * g.G, st));
      HIPCHK(ctx, hipMemsetAsync(w.buckets, 0, 4ull * XYZZ_WORDS * g.G, st));      // path 2: k_fine_sort_part clears the empty buckets
    }
  }
  debug_sync(ctx, "ST_MISC", st);
  if (w.P) {
    {
      StageTimer t(ctx, ST_DIGITS, st);
      // few blocks (every block ends with a flush of its 2048-bin LDS histogram), many threads: a thread recodes its scalars one after
      // the other and every one starts with a load, so the waves per SIMD are what hides that latency (512 blocks of 256 threads: 38 us at
      // n = 2^20, 256 blocks of 1024: 28.5; options "hist_threads" / "hist_blocks", tools/hist_sweep.py)
      const u32 ht = ctx->opt_hist_threads > 0 ? (u32)ctx->opt_hist_threads : 1024u;
      const u32 hb = (u32)std::min<uint64_t>(((uint64_t)g.n + ht - 1) / ht, ctx->opt_hist_blocks > 0 ? (u32)ctx->opt_hist_blocks : 256u);
      // (round 5: the last block to flush runs the scan of the partition counts itself; ticket word behind the any_heavy flag and the
      // segmented scan's ticket, zeroed by the memset above)
      CoarseScanOut so = {nullptr, nullptr, nullptr, nullptr};
      if (ctx->opt_histscan) { so.coarse_off = w.coarse_off; so.coarse_cursor = w.coarse_cursor; so.offG = w.off + g.G; so.ticket = w.coarse_hist + PART_MAX + 2; }
      hipLaunchKernelGGL(k_coarse_hist, dim3(hb), dim3(ht), 0, st, segs, g, w.P, w.coarse_hist, w.dig16, w.negs, so);
    }
    debug_sync(ctx, "ST_DIGITS", st);
    if (!ctx->opt_histscan) {
      StageTimer t(ctx, ST_SCAN, st);
      // exclusive scan of <= 2048 partition counts; total -> coarse_off[P] and off[G]
      hipLaunchKernelGGL(k_coarse_scan, dim3(1), dim3(1024), 0, st, w.coarse_hist, w.P, w.coarse_off, w.coarse_cursor, w.off + g.G);
    }
    debug_sync(ctx, "ST_SCAN", st);
    {
      StageTimer t(ctx, ST_SCATTER, st);
      const u32 TS = g.n >= (1u << 19) ? PT_MAX : 4096u;          // scalars per level-A tile: long runs once there are enough tiles
      hipLaunchKernelGGL(k_partition, dim3((g.n + TS - 1) / TS, g.W), dim3(1024), 0, st, g, w.P, TS, w.coarse_off, w.coarse_cursor, w.dig16, w.negs, w.dig,
                         w.hist, w.coarse_hist + PART_MAX);
      // level B: one block per partition; writes off[0..G), the chunk keys and the sorted entries
      // (heavy partitions: counted and scattered by the tile kernels, which return at once when there are none)
      const u32 nft = (u32)(((size_t)g.n * g.W + FINE_TILE - 1) / FINE_TILE);
      const u32 *any_heavy = w.coarse_hist + PART_MAX;
      if (!g.inblock) hipLaunchKernelGGL(k_fine_hist_heavy, dim3(nft), dim3(256), 0, st, g, w.P, w.coarse_off, w.dig, w.coarse_off + w.P, any_heavy, w.hist);
      hipLaunchKernelGGL(k_fine_sort_part, dim3(w.P), dim3(FINE_THREADS), 0, st, g, w.coarse_off, w.dig, w.hist, w.off, w.cursor, w.chunk_key, w.sidx, w.buckets);
      if (!g.inblock) hipLaunchKernelGGL(k_fine_scatter_heavy, dim3(nft), dim3(256), 0, st, g, w.P, w.coarse_off, w.dig, w.coarse_off + w.P, any_heavy, w.cursor, w.sidx);
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
  }      // !skip_sort
  if (phase == 1) { HIPCHK(ctx, hipGetLastError()); return BPMI_OK; }
  // Round 6 experiment (option "accum_stream"): on the chained pipeline every accumulation runs on ONE stream of its own, created with the
  // LOWEST queue priority, between two events of its lane -- the accumulations are in order by construction, and the lanes' sort and
  // reduction kernels sit on queues the dispatcher prefers whenever a wave slot frees up (with "chunk" below the one-round length the
  // accumulation's slots turn over while it runs).  profiles/r06_accum_stream_and_chunk_ab.txt
  const bool own_acc = ctx->chain_accum && ctx->opt_accum_stream && ctx->stream_acc;
  hipStream_t st_acc = own_acc ? ctx->stream_acc : st;
  if (own_acc) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_sorted[lane], st));
    HIPCHK(ctx, hipStreamWaitEvent(st_acc, ctx->ev_sorted[lane], 0));
  } else if (ctx->chain_accum && !ctx->chain_free && ctx->accum_chain_lane >= 0 && ctx->accum_chain_lane != lane)
    HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_accum[ctx->accum_chain_lane], 0));
  {
    StageTimer t(ctx, ST_ACCUM, st_acc);
    const u32 nthreads = w.nchunks;
    auto kern = glv ? (g.fuse ? k_accum_l0<true, true> : k_accum_l0<true, false>) : (g.fuse ? k_accum_l0<false, true> : k_accum_l0<false, false>);
    hipLaunchKernelGGL(kern, dim3((nthreads + 255) / 256), dim3(256), 0, st_acc, segs, g, w.off, w.chunk_key, w.sidx, w.buckets, w.rec_key[0], w.rec_pt[0]);
  }
  if (own_acc) {
    HIPCHK(ctx, hipEventRecord(ctx->ev_accum[lane], st_acc));
    HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_accum[lane], 0));
  } else if (ctx->chain_accum) { HIPCHK(ctx, hipEventRecord(ctx->ev_accum[lane], st)); ctx->accum_chain_lane = lane; }
  debug_sync(ctx, "ST_ACCUM", st);
  {
    StageTimer t(ctx, ST_SEGSCAN, st);
    u32 R = w.rec0_max;
    int level = 1, src = 0;
    // the ticket word of the last-block-done fusion (zeroed by the memset above; sort path 2 only, first level only)
    u32 *ticket = (w.P && ctx->opt_segfuse) ? w.coarse_hist + PART_MAX + 1 : nullptr;
    for (;;) {
      const u32 nb = (R + 255) / 256;
      if (nb > 128u || nb <= 1u) ticket = nullptr;      // (the kernel's own block count is at most this one: fewer entries than the bound)
      hipLaunchKernelGGL(k_segscan, dim3(nb), dim3(256), 0, st, g, w.off, level, w.rec_key[src], w.rec_pt[src],
                         w.rec_key[src ^ 1], w.rec_pt[src ^ 1], w.buckets, ticket);
      if (g_debug_sync) { fprintf(stderr, "[bpmi] segscan level %d nb %u R %u\n", level, nb, R); debug_sync(ctx, "segscan level", st); }
      if (nb <= 1 || ticket) break;
      R = 2 * nb;
      // ping-pong: level 1 reads buffer 0 (large) and writes buffer 1; later levels are
      // small enough for either buffer (rec1_max >= every later level)
      src ^= 1;
      level++;
    }
  }
  debug_sync(ctx, "ST_SEGSCAN", st);
  const int tail_pre = wcount ? 2 : (ctx->opt_tail ? ctx->opt_tail : 2);
  u32 *const E_red = (tail_pre == 1) ? w.E : E_dst;             // the device tail reads the window sums where they are
  {
    StageTimer t(ctx, ST_BREDUCE, st);
    if (g.B <= 256u) {
      if (ctx->opt_quad) hipLaunchKernelGGL(k_window_weighted_small_quad, dim3(g.W), dim3(g.B < 16u ? 64u : 4u * g.B), 0, st, g, w.buckets, E_red);
      else hipLaunchKernelGGL(k_window_weighted_small, dim3(g.W), dim3(g.B < 64u ? 64u : g.B), 0, st, g, w.buckets, E_red);
    } else {
      // bucket index b in [1, B], B = 2^(c-1):  b = hi 2^s0 + lo, then each digit again in two
      const u32 s0 = g.c / 2u, N0 = (1u << s0) - 1u, N1 = g.B >> s0;          // stage-1 arrays: D0[1..N0], D1[1..N1]
      const u32 t0 = (s0 + 1u) / 2u, t1 = (msb_index(N1) + 1u) / 2u;          // stage-2 splits
      const u32 stride1 = N0 + N1;
      // stage 1: as many elements per lane as keep about one wave on every SIMD (16 at c = 16 with all 16 windows: measured
      // best of 4 / 8 / 12 / 16 there; fewer buckets -- smaller c, a window group of a split MSM -- get shorter chains
      // instead of idle SIMDs); stage 2 + the finish are pure latency: one element per lane, 16-lane butterflies, one launch
      const u32 Wr = g.W - g.top2;                       // windows with B buckets
      // the wide windows (mixed widths, g.top2 of them): arrays of 2B buckets behind the others, split like windows of c + 1 bits (their D sums
      // fit the per-window slot of w.D: 2^((c+1)/2) + 2^(c/2) records)
      const u32 Bt = 2u * g.B, s0t = (g.c + 1u) / 2u, N0t = (1u << s0t) - 1u, N1t = Bt >> s0t;
      const u32 t0t = (s0t + 1u) / 2u, t1t = (msb_index(N1t) + 1u) / 2u;
      const u32 d_top = Wr * stride1, stride1t = N0t + N1t;      // first D record of the wide windows, and their records per window
      auto stage1 = [&](u32 epl, DigitJobs &j) -> u32 {  // the jobs for `epl` elements per lane; returns their waves
        j = digit_jobs2(Wr, 0, g.B, g.B, s0, 0, stride1, epl);
        u32 waves = digit_job_waves(j, 0) + digit_job_waves(j, 1);
        if (g.top2) {
          DigitJobs jt = digit_jobs2(g.top2, Wr * g.B, Bt, Bt, s0t, d_top, stride1t, epl);
          waves += digit_job_waves(jt, 0) + digit_job_waves(jt, 1);
          j = digit_jobs_concat(j, jt);
        }
        return waves;
      };
      DigitJobs j1;
      if (ctx->opt_epl > 0) stage1((u32)ctx->opt_epl, j1);
      else if (ctx->opt_reduce_fit) {
        // the fewest elements per lane whose waves fit the chip's 1 024 SIMDs at one each: a wave alone on its SIMD already runs at
        // 86 % of the multiply-add pipe, so a SIMD with two takes twice as long (c = 15 with 16-lane sums: 1 148 waves, 124 SIMDs doubled,
        // no faster than c = 16; with 12-lane sums, five to a wave: 931)
        u32 epl = 2;
        while (epl < 64u && stage1(epl, j1) > 1024u) epl++;
      } else {
        u32 epl = (u32)(((uint64_t)g.G) >> 15);
        stage1(epl < 2u ? 2u : (epl > 16u ? 16u : epl), j1);
      }
      // stage 2: D0 -> (D00, D01), D1 -> (D10, D11), each <= 16 sums of <= 16 elements, and E[a][r] = sum_d d * D..[d]
      DigitJobs ja = digit_jobs2(Wr, 0, stride1, N0, t0, 0, 64, 1);
      DigitJobs jb = digit_jobs2(Wr, N0, stride1, N1, t1, 0, 64, 1);
      DigitJobs j2 = digit_jobs_concat(ja, jb), j2top;
      memset(&j2top, 0, sizeof(j2top));
      to.nv = 4; to.off[0] = 0; to.off[1] = t0; to.off[2] = s0; to.off[3] = s0 + t1;
      if (g.top2) {
        DigitJobs jta = digit_jobs2(g.top2, d_top, stride1t, N0t, t0t, 0, 64, 1);
        DigitJobs jtb = digit_jobs2(g.top2, d_top + N0t, stride1t, N1t, t1t, 0, 64, 1);
        j2top = digit_jobs_concat(jta, jtb);
        to.top = g.top2; to.top_off[0] = 0; to.top_off[1] = t0t; to.top_off[2] = s0t; to.top_off[3] = s0t + t1t;
      }
      j1.prio = g.prio & PRIO_SUMS;
      hipLaunchKernelGGL(k_digit_sums, dim3(j1.j[j1.njobs - 1].blk0 + digit_job_blocks(j1, j1.njobs - 1)), dim3(256), 0, st, w.buckets, w.D, j1);
      j2.prio = j2top.prio = g.prio & PRIO_FINISH;
      const u32 top_w = g.top2 ? Wr : 0xFFFFFFFFu;
      // (the tickets of the spread finish live behind the sort's partition counts: zeroed by this MSM's memset when the LDS sort runs)
      if (ctx->opt_quad && ctx->opt_final_spread == 1 && w.P)
        hipLaunchKernelGGL(k_digit_final_spread<0>, dim3(g.W * 64u), dim3(64), 0, st, w.D, w.F, w.coarse_hist + PART_MAX + 8, E_red, j2, j2top, top_w);
      else if (ctx->opt_quad && ctx->opt_final_spread >= 2) {
        if (ctx->opt_final_spread == 2) hipLaunchKernelGGL(k_digit_final_spread<1>, dim3(g.W * 64u), dim3(64), 0, st, w.D, w.F, nullptr, E_red, j2, j2top, top_w);
        else hipLaunchKernelGGL(k_digit_final_spread<1>, dim3(g.W * 16u), dim3(256), 0, st, w.D, w.F, nullptr, E_red, j2, j2top, top_w);
        hipLaunchKernelGGL(k_digit_final_spread<2>, dim3(g.W * 4u), dim3(64), 0, st, w.D, w.F, nullptr, E_red, j2, j2top, top_w);
      }
      else if (ctx->opt_quad) hipLaunchKernelGGL(k_digit_final_quad, dim3(g.W * 4u), dim3(1024), 0, st, w.D, E_red, j2, j2top, top_w);
      else hipLaunchKernelGGL(k_digit_final, dim3(g.W * 4u), dim3(256), 0, st, w.D, E_red, j2, j2top, top_w);
    }
  }
  debug_sync(ctx, "ST_BREDUCE", st);
  int tail_mode = 2;
  {
    StageTimer t(ctx, ST_TAIL, st);
    tail_mode = wcount ? 2 : (ctx->opt_tail ? ctx->opt_tail : 2);      // window groups are combined on the host
    if (tail_mode == 1) {
      hipLaunchKernelGGL(k_tail, dim3(1), dim3(64), 0, st, w.E, g.W, g.c, to, ctx->opt_direct ? (u32 *)pd.pin : w.out);
      if (!ctx->opt_direct) HIPCHK(ctx, hipMemcpyAsync(pd.pin, w.out, 64, hipMemcpyDeviceToHost, st));
    } else if (!ctx->opt_direct) {
      const size_t eb = 4ull * XYZZ_WORDS * g.W * g.nv;
      HIPCHK(ctx, hipMemcpyAsync(pd.pin, w.E, eb, hipMemcpyDeviceToHost, st));
    }
  }
  return commit(g.W, g.nv, g.c, tail_mode, to);
}
// error path of a caller that has MSMs of ITS OWN queued in pending slots (bit s of `mine` = slot s was enqueued by this call):
// wait for both lanes, release those slots and no others -- a slot that holds a caller's asynchronous MSM (bpmi_msm_dev_enqueue)
// keeps it, so a later bpmi_msm_finish still returns that MSM's result and never the identity of an emptied slot
static void msm_abandon_pending(bpmi_ctx *ctx, unsigned mine) {
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream1) (void)hipStreamSynchronize(ctx->stream1);
  if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
  for (int s = 0; s < BPMI_LANES; s++) if ((mine >> s) & 1u) ctx->pend[s].active = false;
}
// Wait for the slot's MSM (its completion event: work enqueued behind it keeps running) and run the
// host part of the tail; out = the MSM result.
static int msm_finish(bpmi_ctx *ctx, int slot, uint8_t out[64]) {
  bpmi_ctx::PendingMsm &pd = ctx->pend[slot];
  if (!pd.active) { memset(out, 0, 64); return BPMI_OK; }      // n == 0
  pd.active = false;
  HIPCHK(ctx, wait_event(ctx, pd.done));
  const void *pin = pd.pin;
  if (pd.tail == 1) {
    memcpy(out, pin, 64);
  } else {
    bpmi_host::tail_combine(out, (const u32 *)pin, pd.W, pd.c, pd.to);     // host_tail.hpp
  }
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
  rc = msm_enqueue(ctx, 0, 0, segs, 0, Wa);
  if (rc) return rc;
  rc = msm_enqueue(ctx, 1, 1, segs, Wa, W - Wa);
  if (rc) { msm_abandon_pending(ctx, 1u); return rc; }
  bpmi_ctx::PendingMsm &p0 = ctx->pend[0], &p1 = ctx->pend[1];
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream1));
  const size_t b0 = 4ull * XYZZ_WORDS * p0.W * p0.nv, b1 = 4ull * XYZZ_WORDS * p1.W * p1.nv;
  std::vector<u32> E((b0 + b1) / 4);
  memcpy(E.data(), p0.pin, b0);
  memcpy((char *)E.data() + b0, p1.pin, b1);
  bpmi_host::tail_combine(out, E.data(), W, c, p0.to);
  p0.active = p1.active = false;
  return BPMI_OK;
}
// ---- large inputs as slices of the size where the engine peaks (round 6) -------------------------------------------------------
// The reference's multiexp takes any N (/root/reference/src/pippenger/pippenger.py:22-61) and its verifier calls it with 2n + 1 pairs
// (/root/reference/src/innerproduct/inner_product_verifier.py:134-139: 2^21 + 1 at config C3's size).  One MSM of more than ~2^20 pairs
// runs BELOW the 2^20 rate here (7.75-8.3 x 10^8 pairs/s at 2^21 .. 2^24 against 1.0 x 10^9, profiles/r03_msm_big_n.txt): its 64-byte
// gathers, once per window, no longer fit the Infinity Cache, and one MSM at a time leaves the chip to the sort and to the bucket
// reduction for 0.3 ms per MSM.  So an input of slice_min (1.625 slice_n) pairs or more is cut into K = ceil(total / (slice_n 17/16)) equal slices,
// which run as the two-deep pipeline of bench.py's headline (lanes 0 / 1, the accumulations chained): the sort and the reduction of one
// slice beside the accumulation of the other, the host tail of slice k under the kernels of slice k + 1.  The slices' affine results
// are added on the host (XYZZ, one inversion).  Options "slice_n" (0 = 2^20; -1 = never slice below the sort's 2^23 limit) and
// "slice_min" (0 = default).  Inputs with half-block selection (the IPA's deferred folds: msm_run_pair) are never sliced.
#define SLICE_N_DEFAULT (1u << 20)
#define SLICE_N_LIMIT (1u << 23)          // the packed sort entry holds a 23-bit pair index
static bool segs_dense(const Segs &s) { return s.hlog[0] >= 32u && s.hlog[1] >= 32u && s.hlog[2] >= 32u && !s.glv_sub; }
// the logical pairs [lo, lo + cnt) of a dense `s`
static Segs segs_slice(const Segs &s, uint64_t lo, uint64_t cnt) {
  Segs r = segs_init();
  u32 k = 0;
  uint64_t base = 0;
  for (int i = 0; i < 3; i++) {
    const uint64_t a = std::max<uint64_t>(lo, base), b = std::min<uint64_t>(lo + cnt, base + s.n[i]);
    if (b > a) { r.pts[k] = s.pts[i] + 16ull * (a - base); r.sc[k] = s.sc[i] + 8ull * (a - base); r.n[k] = (u32)(b - a); k++; }
    base += s.n[i];
  }
  r.total = (u32)cnt;
  return r;
}
static uint64_t msm_slice_count(const bpmi_ctx *ctx, const Segs &segs) {
  // (forced window bits, window groups, half-block selections: ONE MSM whatever its size -- beyond 2^23 pairs on the global-atomic sort)
  if (!segs_dense(segs) || ctx->opt_c || ctx->opt_split) return 1;
  const uint64_t slice_n = ctx->opt_slice_n < 0 ? SLICE_N_LIMIT : std::min<uint64_t>(ctx->opt_slice_n ? (uint64_t)ctx->opt_slice_n : SLICE_N_DEFAULT, SLICE_N_LIMIT);
  const uint64_t slice_min = ctx->opt_slice_n < 0 ? SLICE_N_LIMIT + 1 : (ctx->opt_slice_min ? (uint64_t)ctx->opt_slice_min : slice_n + slice_n / 2 + slice_n / 8);      // (measured crossover of one MSM against two slices: ~1.65 x 2^20 pairs)
  if (segs.total < slice_min && segs.total <= SLICE_N_LIMIT) return 1;
  const uint64_t cap = std::min<uint64_t>(slice_n + slice_n / 16, SLICE_N_LIMIT);       // a slice may be a sixteenth over (2^21 + 1 pairs: two slices, not three)
  return std::max<uint64_t>(2, (segs.total + cap - 1) / cap);
}
static int msm_finish_pair(bpmi_ctx *ctx, uint8_t out0[64], uint8_t out1[64]);
static int msm_run_sliced(bpmi_ctx *ctx, const Segs &segs, uint64_t K, uint8_t out[64]) {
  for (int s = 0; s < 2; s++) if (ctx->pend[s].active || ctx->pend[s].async) return fail(ctx, BPMI_E_STATE, "an MSM is still pending in this slot (bpmi_msm_finish it first)");
  int rc = ensure_lane(ctx, 1);
  if (rc) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream1, ctx->ev_fork, 0));
  const uint64_t total = segs.total, per = (total + K - 1) / K;
  bpmi_host::pt acc;
  bpmi_host::pt_set_inf(acc);
  // an error in the middle leaves the OTHER lane's slice queued: drain both lanes and release the slots of THIS call, or every later
  // MSM on this ctx would fail with "still pending"
  unsigned mine = 0;
  const bool chain = per >= (1u << 19);              // (the accumulations chained as in bpmi_msm_dev_enqueue: each has the chip)
  auto leave = [&](int code) { ctx->chain_accum = false; ctx->accum_chain_lane = -1; if (code) msm_abandon_pending(ctx, mine); return code; };
  auto take = [&](int slot) -> int {
    uint8_t part[64];
    mine &= ~(1u << slot);
    const int r = msm_finish(ctx, slot, part);
    if (r) return r;
    bpmi_host::pt p;
    bpmi_host::pt_from_affine(p, part);
    bpmi_host::pt_add(acc, acc, p);
    return BPMI_OK;
  };
  ctx->accum_chain_lane = -1;
  for (uint64_t k = 0; k < K; k++) {
    const int lane = (int)(k & 1);
    if (k >= 2) { rc = take(lane); if (rc) return leave(rc); }
    const uint64_t lo = k * per, cnt = std::min<uint64_t>(per, total - lo);
    ctx->chain_accum = chain;
    rc = msm_enqueue(ctx, lane, lane, segs_slice(segs, lo, cnt));
    ctx->chain_accum = false;
    if (rc) return leave(rc);
    mine |= 1u << lane;
  }
  {
    // the last two slices: their host tails side by side (the second on the ctx's helper thread, msm_finish_pair), oldest first in the sum
    uint8_t part[2][64];
    mine = 0;
    rc = msm_finish_pair(ctx, part[0], part[1]);
    if (rc) return leave(rc);
    for (uint64_t k = K - 2; k < K; k++) {
      bpmi_host::pt p;
      bpmi_host::pt_from_affine(p, part[k & 1]);
      bpmi_host::pt_add(acc, acc, p);
    }
  }
  bpmi_host::pt_to_affine(out, acc);
  return leave(BPMI_OK);
}
static int msm_run(bpmi_ctx *ctx, const Segs &segs, uint8_t out[64]) {
  if (segs.total > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  {
    const uint64_t K = msm_slice_count(ctx, segs);
    if (K > 1) return msm_run_sliced(ctx, segs, K, out);
  }
  if (ctx->opt_split == 1 && segs.total >= 2) return msm_run_split(ctx, segs, out);
  int rc = msm_enqueue(ctx, 0, 0, segs);
  if (rc) return rc;
  return msm_finish(ctx, 0, out);
}
// finish both slots of a synchronous pair; the second tail on the helper thread (it waits for its own event there)
static int msm_finish_pair(bpmi_ctx *ctx, uint8_t out0[64], uint8_t out1[64]) {
  if (!ctx->opt_tail_thread || !ctx->pend[0].active || !ctx->pend[1].active) {
    const int rc = msm_finish(ctx, 0, out0), rc1 = msm_finish(ctx, 1, out1);
    return rc ? rc : rc1;
  }
  if (!ctx->helper) ctx->helper = new HostHelper();
  bpmi_ctx::PendingMsm &pd = ctx->pend[1];
  hipError_t e1 = hipSuccess;
  const int dev = ctx->device;
  ctx->helper->submit([&pd, &e1, out1, dev, ctx] {           // (no ctx->err from this thread: the code travels back in e1)
    (void)hipSetDevice(dev);
    e1 = wait_event(ctx, pd.done);
    if (e1 != hipSuccess) return;
    if (pd.tail == 1) memcpy(out1, pd.pin, 64);
    else bpmi_host::tail_combine(out1, (const u32 *)pd.pin, pd.W, pd.c, pd.to);
  });
  const int rc = msm_finish(ctx, 0, out0);
  ctx->helper->wait();
  pd.active = false;
  if (rc) return rc;
  HIPCHK(ctx, e1);
  return BPMI_OK;
}
// Two SMALL MSMs (both on the one-launch kernel's path) as one launch sequence on the ctx stream: one k_msm_small_pair, one
// combine, the two copies; pending slots 0 and 1 as for a pair on two lanes.  (Two lanes cost a fork event, a second queue's
// doorbell and a second wait: ~40 us of a 0.3 ms round of the inner-product argument.)
static int msm_enqueue_small_pair(bpmi_ctx *ctx, const Segs &s0, const Segs &s1, bool mid) {
  bpmi_ctx::PendingMsm &p0 = ctx->pend[0], &p1 = ctx->pend[1];
  if (p0.active || p1.active) return fail(ctx, BPMI_E_STATE, "an MSM is still pending in this slot (bpmi_msm_finish it first)");
  SmallPair sp;
  CombinePair cp;
  MidPair mp;
  memset(&mp, 0, sizeof(mp));
  const Segs *ss[2] = {&s0, &s1};
  size_t off[2], total = 0;
  MsmWs w[2];
  MsmGeom gg[2];
  const u32 c = mid ? MID_C : SMALL_C;
  for (int j = 0; j < 2; j++) {
    MsmGeom &g = gg[j];
    memset(&g, 0, sizeof(g));
    g.n = ss[j]->total; g.c = c; g.W = 255u / g.c + 1u; g.w0 = 0; g.B = 1u << (g.c - 1); g.G = g.W * g.B; g.L = 8; g.nv = 1;
    msm_layout(g, w[j], nullptr);
    off[j] = total;
    total += align_up(w[j].total, 256);
  }
  int rc = ensure_ws(ctx, total);
  if (rc) return rc;
  const u32 parts = mid ? mid_parts(ctx, std::max(s0.total, s1.total)) : 1u;
  const size_t eb = 4ull * XYZZ_WORDS * gg[0].W * parts;
  for (int j = 0; j < 2; j++) { rc = ensure_pin_slot(ctx, j, eb); if (rc) return rc; }
  u32 Smax = 1, threads = 64;
  for (int j = 0; j < 2; j++) {
    msm_layout(gg[j], w[j], (char *)ctx->ws + off[j]);
    const uint64_t n = ss[j]->total;
    const u32 S = (u32)std::min<uint64_t>(64, (n + 255) / 256);
    u32 *const E_dst = ctx->opt_direct ? (u32 *)ctx->pend[j].pin : w[j].E;        // (see msm_enqueue: straight into the slot's host buffer)
    sp.segs[j] = *ss[j]; sp.g[j] = gg[j]; sp.S[j] = S; sp.out[j] = S > 1 ? w[j].buckets : E_dst;
    cp.part[j] = w[j].buckets; cp.S[j] = S; cp.E[j] = E_dst;
    mp.segs[j] = *ss[j]; mp.g[j] = gg[j]; mp.E[j] = E_dst;
    Smax = std::max(Smax, S);
    threads = std::max(threads, (u32)std::min<uint64_t>(256, (n + 63) / 64 * 64));
  }
  hipStream_t st = ctx->stream;
  {
    StageTimer t(ctx, ST_ACCUM, st);
    if (mid) hipLaunchKernelGGL(k_msm_mid, dim3(gg[0].W, 2, parts), dim3(MID_THREADS), 0, st, mp);
    else {
      hipLaunchKernelGGL(k_msm_small_pair, dim3(gg[0].W, Smax, 2), dim3(threads), 0, st, sp);
      if (Smax > 1) hipLaunchKernelGGL(k_small_combine_pair, dim3(gg[0].W, 2), dim3(64), 0, st, cp);
    }
  }
  debug_sync(ctx, "k_msm_small_pair / k_msm_mid", st);
  TailOffs to;
  memset(&to, 0, sizeof(to));
  to.nv = parts;
  // a failure from here on leaves none of THIS call's slots pending (both were free on entry: nobody else's is touched)
  unsigned mine = 0;
  auto queue_results = [&]() -> int {
    for (int j = 0; j < 2; j++) {
      bpmi_ctx::PendingMsm &pd = ctx->pend[j];
      if (!ctx->opt_direct) HIPCHK(ctx, hipMemcpyAsync(pd.pin, w[j].E, eb, hipMemcpyDeviceToHost, st));
      HIPCHK(ctx, hipEventRecord(pd.done, st));
      pd.active = true; pd.W = gg[j].W; pd.nv = parts; pd.c = c; pd.tail = 2; pd.to = to;
      mine |= 1u << j;
    }
    HIPCHK(ctx, hipGetLastError());
    return BPMI_OK;
  };
  rc = queue_results();
  if (rc && mine) msm_abandon_pending(ctx, mine);
  return rc;
}
// two independent MSMs, overlapped on the two lanes; everything already enqueued on the
// ctx stream (the producers of the scalars) is ordered before both
static int msm_run_pair(bpmi_ctx *ctx, const Segs &s0, uint8_t out0[64], const Segs &s1, uint8_t out1[64]) {
  if (s0.total > BPMI_MAX_N || s1.total > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  {
    const uint64_t small_max = ctx->opt_small < 0 ? 0 : (ctx->opt_small ? (uint64_t)ctx->opt_small : SMALL_N_DEFAULT);
    const uint64_t mid_min = ctx->opt_mid_min > 0 ? (uint64_t)ctx->opt_mid_min : MID_MIN_DEFAULT;
    const uint64_t big = std::max<uint64_t>(s0.total, s1.total);
    const bool mid = ctx->opt_mid_min >= 0 && big >= mid_min && big <= MID_NMAX && ctx->opt_glv <= 0;
    if (ctx->opt_pair1 && ctx->opt_c == 0 && s0.total && s1.total && (mid || big <= small_max)) {
      int rc = msm_enqueue_small_pair(ctx, s0, s1, mid);           // (cleans up after itself)
      if (rc) return rc;
      return msm_finish_pair(ctx, out0, out1);
    }
  }
  int rc = ensure_lane(ctx, 1);
  if (rc) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream1, ctx->ev_fork, 0));
  // (chaining the pair's accumulate kernels with the event of the asynchronous pipeline was measured on the IPA's 2^20-sized
  // rounds: 2.43 ms per round against 2.35 -- with only two MSMs there is no steady state to pipeline)
  const bool big = s0.total >= (1u << 19) && s1.total >= (1u << 19);
  // Round 6 experiment (option "pair_sched", off): a pair of LARGE MSMs as  sort 0 | sort 1 -> accumulation 0 -> accumulation 1 (reduction 0
  // beside it) -> reduction 1.  Queued one whole MSM after the other, the second MSM's sort meets the first one's accumulation and crawls
  // beside it (k_partition -- 1 024-thread blocks with 67 KB of LDS -- 1 025 us instead of 60, the accumulation beside it 1 126 instead of
  // 780: profiles/r06_C3_big_round_timeline.txt); with the schedule every kernel runs at its own speed (same file, second half) and the
  // round takes exactly as long: 2.41-2.57 ms either way, 23.9 / 25.6 against 24.4 / 24.6 ms per proof (profiles/r06_C3_pair_sched_ab.txt).
  // A pair has no steady state: two sorts (0.3 ms) + two accumulations (0.78 + 0.95 with the first reduction beside the second) + the last
  // reduction + the host tail IS the round, however it is interleaved -- what round 4 found for "pair_phases" and round 3 for "pair_chain".
  const bool sched = ctx->opt_pair_sched && big && !ctx->opt_graph;
  const bool chain = sched || (ctx->opt_pair_chain && big);
  if (chain) { ctx->chain_accum = true; ctx->accum_chain_lane = -1; }
  ctx->beside = ctx->opt_pair_rounds != 1 && big;      // (msm_pick_geometry: multi-round chunks; the same geometry in both phases)
  auto leave = [&](int code) { ctx->beside = false; if (chain) { ctx->chain_accum = false; ctx->accum_chain_lane = -1; } return code; };
  // Round 4 experiment (option "pair_phases", off): both sorts first, then both accumulations, nothing else ordered.  Measured: NO difference
  // (2.336 / 2.330 against 2.343 / 2.319 ms for a round of the 2^20-element prover, profiles/r04_C3_pair_phases_ab.txt)
  const bool phases = sched || (ctx->opt_pair_phases && !chain && !ctx->opt_graph && s0.total >= (1u << 15) && s1.total >= (1u << 15));
  if (phases) {
    rc = msm_enqueue(ctx, 0, 0, s0, 0, 0, 1);
    if (rc == BPMI_OK) rc = msm_enqueue(ctx, 1, 1, s1, 0, 0, 1);
    if (rc) { msm_abandon_pending(ctx, 0u); return leave(rc); }        // (nothing pending yet: only drains the lanes)
    if (sched) {                                         // lane 0's accumulation behind lane 1's sort as well
      hipError_t e = hipEventRecord(ctx->ev_join, ctx->stream1);
      if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
      if (e != hipSuccess) { msm_abandon_pending(ctx, 0u); return leave(fail(ctx, BPMI_E_HIP, std::string("msm_run_pair: ") + hipGetErrorString(e))); }
    }
  }
  rc = msm_enqueue(ctx, 0, 0, s0, 0, 0, phases ? 2 : 0);
  if (rc == BPMI_OK) {
    rc = msm_enqueue(ctx, 1, 1, s1, 0, 0, phases ? 2 : 0);
    if (rc) msm_abandon_pending(ctx, 1u);
  }
  leave(0);
  if (rc) return rc;
  return msm_finish_pair(ctx, out0, out1);
}

// second-level segscan buffer sizing relies on this: every level after the first has
// at most rec1_max records (R shrinks monotonically)
