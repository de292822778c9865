// context.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).
// Engine context: streams, workspaces, error reporting, per-stage HIP-event timers.
#pragma once

// ------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------
enum Stage {
  ST_DIGITS = 0, ST_SCAN, ST_SCATTER, ST_ACCUM, ST_SEGSCAN, ST_BREDUCE, ST_TAIL,
  ST_MULBATCH, ST_LINCOMB2, ST_SCDOT, ST_SCFOLD, ST_MISC, ST_RPPREP, ST_DECOMP, ST_RPELEM
};
static const char *STAGE_NAMES[BPMI_NSTAGES] = {
  "msm_digits_hist", "msm_scan", "msm_scatter", "msm_accumulate", "msm_segscan", "msm_bucket_reduce",
  "msm_tail", "ec_mul_batch", "ec_lincomb2", "sc_dot", "sc_fold", "misc", "rp_prepare", "ec_decompress", "rp_elements"
};

struct EvPair { int stage; hipEvent_t a, b; };
#define BPMI_LANES 3          // streams / workspaces / pending-MSM slots of a ctx

struct HostHelper;
struct MsmGraphCache;
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
  // third lane: only the asynchronous MSM pipeline uses it (slot 2 of bpmi_msm_dev_enqueue with option async_lanes): with
  // three MSMs in flight the sort of MSM k + 1 is on the GPU while MSM k accumulates and MSM k - 1 is being reduced
  hipStream_t stream2 = nullptr;
  void *ws2 = nullptr; size_t ws2_bytes = 0;
  hipEvent_t ev_slice[4] = {nullptr, nullptr, nullptr, nullptr};      // batch preparation: upload slice c has arrived
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;      // fork: lane 1 may start; join: lane 1's work is in (batch preparation)
  // software pipeline of the asynchronous MSM pair on two lanes: the accumulate kernel of an MSM waits for the
  // accumulate kernel of the MSM enqueued before it (on the other lane), so the throughput-bound stage always has the
  // whole GPU while the other lane's latency-bound tail (segmented scan, bucket reduction) and next sort run beside it
  hipEvent_t ev_accum[BPMI_LANES] = {nullptr, nullptr, nullptr};
  int accum_chain_lane = -1;     // lane whose ev_accum is the newest, -1: none pending
  bool chain_accum = false;      // set by bpmi_msm_dev_enqueue around msm_enqueue
  void *up_ring = nullptr;       // page-locked staging ring of h2d()
  size_t up_cursor = 0;
  hipEvent_t up_ev = nullptr;
  bool up_pending = false;
  // One in-flight MSM: its own pinned host buffer (what the tail reads) and its own completion event, so
  // that finishing it never waits for work enqueued behind it (bpmi_msm_dev_enqueue / bpmi_msm_finish
  // keep two MSMs in flight on ONE stream: the host tail of MSM k overlaps the kernels of MSM k + 1).
  struct PendingMsm {
    bool active = false; u32 W = 0, nv = 0, c = 0; int tail = 2; TailOffs to;
    void *pin = nullptr; size_t pin_bytes = 0; hipEvent_t done = nullptr; bool async = false, async_empty = false;
  } pend[BPMI_LANES];
  void *stage_in = nullptr; size_t stage_in_bytes = 0;  // device staging for host-pointer entry points
  // options
  int opt_c = 0;        // window bits, 0 = auto
  int opt_tail = 0;     // 0 auto, 1 device, 2 host
  int opt_prio = 1;     // MSM: the stages around the accumulation raise their waves' issue priority (s_setprio): 0 none, 1 all (default from round 6: beside a multi-round
                        // accumulation it is a gain, profiles/r06_wave_priority_and_chunk_ab.txt; round 3 measured a loss beside the one-round kernel), 16 + mask = those PRIO_* stages
  int opt_hist_threads = 0, opt_hist_blocks = 0;     // k_coarse_hist launch shape (0 = default)
  int opt_quad = 1;     // bucket reduction's finish with four-lane point additions (k_digit_final_quad); 0 = one lane per point
  int opt_mulb = 1;     // bpmi_ec_mul_batch: 1 = GLV + fixed signed windows over affine odd multiples (n >= MULB_MIN_N), 0 = the bit-serial ladder
  int opt_chunk = 0;    // entries per thread in k_accum_l0, 0 = auto
  int opt_small = 0;    // largest n handled by the one-launch small-MSM kernel (0 = default, -1 = never)
  int opt_fold_wnaf = 2;     // the IPA's 16-way generator fold: 2 = width-4 NAF of the coefficients' GLV halves over affine tables of odd multiples, 1 = of the whole coefficients, 0 = plain NAF ladder
  int opt_rp_only_role = -1; // profiling only: run one role of the batch preparation kernel (the call then reports proof 0 as bad)
  int opt_glv = 0;           // MSM on GLV-split scalars (an experiment that lost, profiles/r03_glv_msm_on_off.txt): 0 / -1 = never (default), 1 = whenever the bucket pipeline runs
  int opt_rp_prio = 1;       // batch preparation: its kernels (expander, roles, elements) raise their waves' issue priority: 0 never, 1 on wire formats 1 and 2 (square roots run beside them), 2 always
  int opt_rp_slices = 0;     // batch preparation: uploads of a batch of >= 4096 proofs (1 .. 4); 0 = 4 for formats 1 and 2 (a slice's points are decoded beside the next upload), 1 for format 3
  int opt_rp_overlap = 1;    // batch preparation: point decoding on the second lane beside the preparation kernels (0: behind them; measurements)
  int opt_rp_rows = 0;       // batch preparation: proofs per launch (0 = as many as fit ~256 MB of contribution cells)
  int opt_rp_lanes = 0;      // batch preparation kernel: proofs per wave (0 = chosen from the batch size)
  void *rp_buf = nullptr; size_t rp_buf_bytes = 0;   // batch preparation: per-proof contributions to the shared generators
  int opt_epl = 0;           // bucket reduction stage 1: elements per lane (0 = default 16)
  int opt_tail_thread = 1;   // a synchronous PAIR of MSMs: the host tail of the second one runs on the ctx's helper thread beside the first one's (0: one after the other)
  int opt_pair_chain = 0;    // a synchronous pair of LARGE MSMs: 1 = their accumulate kernels chained as in the asynchronous pipeline (A/B; round 3 measured it slower)
  int opt_mid_min = 0;       // a pair of MSMs runs as one launch of k_msm_mid from this many pairs in the larger one (0 = default 1536, -1 = never)
  int opt_mid_single = 0;    // a single MSM runs on k_msm_mid from this many pairs (0 = default 2560, -1 = never)
  int opt_pair1 = 1;         // a pair of SMALL MSMs (bpmi_msm2, the L / R of an inner-product round) as one launch sequence on one stream (0: two lanes)
  int opt_fuse = 1;          // k_accum_l0 folds a wave's partial records itself (0: two records per thread, the round-3 path; A/B and tests)
  int opt_spin_wait = 0;     // polls of an event / stream before sleeping in the runtime (see wait_event; measured: no gain, off)
  int opt_async_lanes = 0;   // 1: slot 1 of the asynchronous MSM pair runs on the second lane
  bool async_lane1_ordered = false, async_lane2_ordered = false;
  int opt_split = 0;    // 1: one MSM as two window groups, one per lane (measured: +5 % at 2^20, -8 % at 2^19; off)
  int64_t opt_ipa_big = 0;   // base length from which the IPA folds generators 16-way (0 = default 2^18)
  int opt_ipa_step = 0;      // short inner-product vectors: fold + coefficient tables + the next round's dots and scalars in ONE launch (k_ipa_small_step).
                             // Measured (profiles/r04_C3_small_step_ab.txt): the one block takes 60 us where the four launches it replaces take 25 + gaps:
                             // 25.1-25.3 ms per proof against 24.5.  OFF; kept with its tests (tools/fuzz_ops.py draws it)
  int opt_fold_shared = 1;   // the product fold of a state without per-generator scales: shared GLV halves, two terms per thread (0: per-lane products)
  int64_t opt_ipa_small = 0; // logical length at which smaller bases are folded through products (0 = default 4096, 1 = never)
  void *fold_tab = nullptr; size_t fold_tab_bytes = 0;     // tables + scratch of the width-4 NAF generator fold, allocated at the first fold, kept
  int opt_ipa_fixed = 0;     // 1: the generator arrays of bpmi_ipa_create_dev are deployment constants: the fold's tables of their odd multiples are kept between proofs
  const void *fold_key_g = nullptr, *fold_key_h = nullptr; uint64_t fold_key_n = 0;     // whose tables fold_tab holds (nullptr: nobody's)
  // profiling
  bool prof = false;
  int prof_only = -1;      // >= 0: time only this stage (every event record costs a ~10 us bubble between kernels)
  std::vector<EvPair> evs;
  std::vector<hipEvent_t> ev_pool;      // recycled timing events (creating one costs more than recording it)
  HostHelper *helper = nullptr;
  MsmGraphCache *graphs = nullptr;      // captured launch sequences of repeated MSMs (msm_host.hpp)
  int opt_direct = 1;                   // the last kernel of an MSM writes its result into the slot's page-locked host buffer (0: workspace + copy)
  int opt_pair_phases = 0;              // 1: a synchronous pair of MSMs queues both sorts before either accumulation (measured neutral: profiles/r04_C3_pair_phases_ab.txt)
  int opt_graph = 0;                    // 1: replay an MSM's launch sequence as a HIP graph when the same call comes again
  // round 5 (the mid-size floor; every one on by default, 0 = the round-4 path for A/B runs and tests)
  int opt_mid_parts = 0;                // k_msm_mid: blocks per window (0: three from 3 000 pairs, else one; 1 .. 4 forced)
  int opt_mixed = 1;                    // window bits 10 .. 14 as mixed widths c / c + 1 covering 256 bits exactly (15 always does, under opt_top2)
  int opt_top2 = 1;                     // c = 15: 17 windows, the last one unsigned with 2B buckets (0: 18 windows, the last one a carry window)
  int opt_reduce_fit = 1;               // stage 1 of the bucket reduction: elements per lane chosen so that its waves fit the SIMDs at one each
  int opt_final_spread = 3;             // the bucket reduction's finish: 0 one 16-wave block per array, 1 one-wave blocks + tickets, 2 / 3 two launches (include/bpmi.h)
  int opt_inblock = 1;                  // n <= 2^17: the sort's level B handles partitions of any size itself, the two heavy-tile launches are skipped
  int opt_prover_tw = 0;                // bpmi_rp_prover_create: window bits of the fixed-base tables (0 = default 12; 4 .. 13)
  int opt_validate = 1;                 // on-curve check of the points a caller hands in: 0 never, 1 the host-pointer entry points (default), 2 the synchronous _dev ones too
  u32 *vflag = nullptr, *vflag_dev = nullptr;      // the check's verdict (smallest bad index, ~0 = none): device word, and the page-locked word it is copied to
  int opt_histscan = 0;                 // 1: the scan of the sort's partition counts runs in the block of k_coarse_hist that flushes last.  LOST (profiles/r05_last_block_fusions_ab.txt):
                                        // the device-scope fence every block needs writes its XCD's L2 back behind 33 MB of digit codes -- +60 us at 2^20, +16 us at 2^16.  Off; kept with its tests
  int opt_segfuse = 0;                  // 1: the segmented scan's last level runs in the block that finishes the level before it last.  No gain one MSM at a time, and the fence costs
                                        // two MSMs in flight 3 % (the other lane's dirty bucket lines are written back with it).  Off; kept with its tests
  // round 6
  int opt_slice_n = 0;                  // an MSM of more than slice_min pairs runs as slices of about this many, two in flight (0 = 2^20, -1 = only beyond the sort's 2^23 limit; msm_host.hpp)
  int opt_slice_min = 0;                // ... the size from which it does (0 = default: 1.25 x slice_n)
  int opt_pair_sched = 0;               // 1: a synchronous pair of large MSMs as both sorts, then the accumulations one after the other (msm_run_pair).  Measured neutral
                                        // (profiles/r06_C3_pair_sched_ab.txt): off
  int opt_prover_wire = 2;              // bpmi_rp_prove_batch: the wire format of the proofs it returns, 2 or 3 (3: with the points' y coordinates, rp_wire_v2_host.hpp)
  int opt_prover_split = 0;             // bpmi_rp_prove_batch: 1 = a batch of 4 096 proofs or more as two halves on two lanes, N > 1 = from 2 N proofs (rp_prove_host.hpp).
                                        // Measured: 19.7-19.9 ms against 19.4-19.6 for 2^14 proofs (profiles/r06_batch_prover_table_bits.txt): off
  int opt_rounds = 0;                   // rounds of three waves per SIMD of an accumulation that shares the chip with another MSM's kernels (0 = 3; msm_host.hpp)
  int opt_pair_rounds = 0;              // 1: a synchronous pair of large MSMs keeps round 5's one-round chunks (A/B)
  bool beside = false;                  // set by msm_run_pair around its enqueues
  int opt_accum_chain = 1;              // experiment: 0 = the asynchronous pipeline's accumulations are NOT ordered after each other (the lanes run free)
  bool chain_free = false;
  int opt_accum_stream = 0;             // experiment: the chained pipeline's accumulations on one low-priority stream of their own (msm_host.hpp)
  int opt_lane_prio = 0;                // experiment: queue priority of lanes 1 / 2 created AFTER the option is set (0 default, -1 high, 1 low)
  hipStream_t stream_acc = nullptr;
  hipEvent_t ev_sorted[BPMI_LANES] = {nullptr, nullptr, nullptr};
  double prof_ms[BPMI_NSTAGES] = {0};
  uint64_t prof_calls[BPMI_NSTAGES] = {0};
};

// One helper thread per ctx (started at first use) for host work that can run beside the calling thread's: the host tail of the
// second MSM of a synchronous pair (40 us of field arithmetic per MSM; 20 rounds of an inner-product argument pay it twice each).
struct HostHelper {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, quit = false;
  void start() {
    if (th.joinable()) return;
    th = std::thread([this] {
      std::unique_lock<std::mutex> lk(mu);
      for (;;) {
        cv.wait(lk, [this] { return has_job || quit; });
        if (quit) return;
        std::function<void()> j = std::move(job);
        has_job = false;
        lk.unlock();
        j();
        lk.lock();
        done = true;
        cv.notify_all();
      }
    });
  }
  void submit(std::function<void()> j) {
    start();
    std::lock_guard<std::mutex> lk(mu);
    job = std::move(j); has_job = true; done = false;
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return done; });
  }
  ~HostHelper() {
    if (!th.joinable()) return;
    { std::lock_guard<std::mutex> lk(mu); quit = true; cv.notify_all(); }
    th.join();
  }
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
    if (ctx->prof && (ctx->prof_only < 0 || ctx->prof_only == stage)) {
      a = take(); b = take();
      if (!a || !b) { a = b = nullptr; return; }
      (void)hipEventRecord(a, stream);
    }
  }
  hipEvent_t take() {
    if (!ctx->ev_pool.empty()) { hipEvent_t e = ctx->ev_pool.back(); ctx->ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
  }
  ~StageTimer() {
    if (ctx->prof && a) { (void)hipEventRecord(b, stream); ctx->evs.push_back({stage, a, b}); }
  }
};

// Host -> device copies of 4 KB .. 8 MB from PAGEABLE memory go through a page-locked ring of the ctx.  Left to the runtime such
// a copy pins the caller's pages on the fly, and on a busy host that took 9 ms for the 1.5 MB of a bpmi_msm2 call (0.1 ms on a
// quiet one: config C4's prover 31-39 ms instead of 10.5 on one box in four).  Sources that are already page-locked
// (bpmi_host_alloc: the batch verifier's receive buffers) and very large or tiny copies go straight to the runtime.
#define UP_RING_BYTES (32u << 20)
static hipError_t h2d(bpmi_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t st) {
  if (bytes < 4096 || bytes > (8u << 20)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost) return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
  (void)hipGetLastError();                                  // an ordinary host pointer is "invalid value" to the query
  if (!ctx->up_ring) {
    if (hipHostMalloc(&ctx->up_ring, UP_RING_BYTES, hipHostMallocDefault) != hipSuccess) { ctx->up_ring = nullptr; (void)hipGetLastError(); return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st); }
    if (hipEventCreateWithFlags(&ctx->up_ev, hipEventDisableTiming) != hipSuccess) {       // no ring without its event: the runtime's own copy
      (void)hipGetLastError();
      (void)hipHostFree(ctx->up_ring);
      ctx->up_ring = nullptr; ctx->up_ev = nullptr;
      return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    }
  }
  const size_t need = (bytes + 255) & ~(size_t)255;
  if (ctx->up_cursor + need > UP_RING_BYTES) {              // wrap: everything queued from the ring so far must have left it
    if (ctx->up_pending) {                                  // (the event covers the stream of the last copy; the ctx's streams are few)
      hipError_t e = hipEventSynchronize(ctx->up_ev);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e == hipSuccess && ctx->stream1) e = hipStreamSynchronize(ctx->stream1);
      if (e == hipSuccess && ctx->stream2) e = hipStreamSynchronize(ctx->stream2);
      if (e != hipSuccess) return e;
    }
    ctx->up_cursor = 0;
  }
  char *stage = (char *)ctx->up_ring + ctx->up_cursor;
  ctx->up_cursor += need;
  memcpy(stage, src, bytes);
  hipError_t e = hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) { e = hipEventRecord(ctx->up_ev, st); ctx->up_pending = true; }
  return e;
}
// Waits of the latency-critical paths.  Polling the event before sleeping in the runtime (option "spin_wait" = number of polls) was
// tried against the slow boxes of the pool and is OFF: the slowness was the pageable uploads (h2d above), the polls change nothing
// for one caller (C2 0.36 ms, C4 10.3 ms either way) and cost the batch verifier's eight threads 4.5 % of their throughput
// (profiles/r03_spin_wait_ab.txt).
static hipError_t wait_event(const bpmi_ctx *ctx, hipEvent_t ev) {
  for (int spin = 0; spin < ctx->opt_spin_wait; spin++) {           // some tens of milliseconds at most (option "spin_wait", per ctx)
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  return hipEventSynchronize(ev);
}
static hipError_t wait_stream(const bpmi_ctx *ctx, hipStream_t st) {
  for (int spin = 0; spin < ctx->opt_spin_wait; spin++) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
  return hipStreamSynchronize(st);
}
static void msm_graphs_clear(bpmi_ctx *ctx);
static int ensure_ws(bpmi_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return BPMI_OK;
  msm_graphs_clear(ctx);                 // captured sequences hold pointers into the workspace
  if (ctx->ws) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
  size_t want = bytes + bytes / 8;
  HIPCHK(ctx, hipMalloc(&ctx->ws, want));
  ctx->ws_bytes = want;
  return BPMI_OK;
}
static int ensure_lane(bpmi_ctx *ctx, int lane) {
  if (lane == 0) return BPMI_OK;
  if (!ctx->stream1) {
    HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->stream1, hipStreamNonBlocking, ctx->opt_lane_prio));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    for (int k = 0; k < BPMI_LANES; k++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_accum[k], hipEventDisableTiming));
  }
  if (lane == 2 && !ctx->stream2) HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, ctx->opt_lane_prio));
  if (ctx->opt_accum_stream && !ctx->stream_acc) {
    int least = 0, greatest = 0;
    HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->stream_acc, hipStreamNonBlocking, ctx->opt_accum_stream == 2 ? 0 : least));
    for (int k = 0; k < BPMI_LANES; k++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_sorted[k], hipEventDisableTiming));
  }
  return BPMI_OK;
}
static hipStream_t lane_stream(bpmi_ctx *ctx, int lane) { return lane == 0 ? ctx->stream : (lane == 1 ? ctx->stream1 : ctx->stream2); }
static void *lane_ws(bpmi_ctx *ctx, int lane) { return lane == 0 ? ctx->ws : (lane == 1 ? ctx->ws1 : ctx->ws2); }
static int ensure_ws_lane(bpmi_ctx *ctx, int lane, size_t bytes) {
  if (lane == 0) return ensure_ws(ctx, bytes);
  void *&ws = lane == 1 ? ctx->ws1 : ctx->ws2;
  size_t &have = lane == 1 ? ctx->ws1_bytes : ctx->ws2_bytes;
  if (bytes <= have) return BPMI_OK;
  msm_graphs_clear(ctx);
  if (ws) { HIPCHK(ctx, hipStreamSynchronize(lane_stream(ctx, lane))); HIPCHK(ctx, hipFree(ws)); ws = nullptr; have = 0; }
  const size_t want = bytes + bytes / 8;
  HIPCHK(ctx, hipMalloc(&ws, want));
  have = want;
  return BPMI_OK;
}
static int ensure_pin_slot(bpmi_ctx *ctx, int slot, size_t bytes);
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

// pinned buffer + completion event of pending-MSM slot `slot`
static int ensure_pin_slot(bpmi_ctx *ctx, int slot, size_t bytes) {
  bpmi_ctx::PendingMsm &pd = ctx->pend[slot];
  if (!pd.done) HIPCHK(ctx, hipEventCreateWithFlags(&pd.done, hipEventDisableTiming));
  if (bytes <= pd.pin_bytes) return BPMI_OK;
  msm_graphs_clear(ctx);
  if (bytes < 16384) bytes = 16384;
  if (pd.pin) { HIPCHK(ctx, hipEventSynchronize(pd.done)); HIPCHK(ctx, hipHostFree(pd.pin)); pd.pin = nullptr; pd.pin_bytes = 0; }
  HIPCHK(ctx, hipHostMalloc(&pd.pin, bytes, hipHostMallocDefault));
  pd.pin_bytes = bytes;
  return BPMI_OK;
}
