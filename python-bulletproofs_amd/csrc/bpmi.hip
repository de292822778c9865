// bpmi.hip -- libbpmi.so: MSM + inner-product-argument engine for MI355X (gfx950).
// C-ABI in include/bpmi.h; design notes in DESIGN.md.
//
// MSM pipeline (replaces Pippenger.multiexp, /root/reference/src/pippenger/pippenger.py:22-94,
// by the signed-digit bucket method; the result -- a canonical affine point -- is
// schedule independent, so it is bit-identical to the reference's subset-table schedule):
//   sort            (window, bucket) counting sort of (point index, sign):
//                   n >= 2^16 (window bits >= 10): k_coarse_hist / k_partition / k_fine_sort_part --
//                   a two-level radix partition whose histograms and ranks live in LDS
//                   (global atomics only reserve one range per tile and partition; level B
//                   sorts one partition per block and writes one contiguous range);
//                   smaller n: k_digits_hist / k_scatter with global atomics
//   k_scan_*        exclusive scans of the (coarse) histograms -> run offsets
//   k_accum_l0      every thread adds exactly L sorted entries (perfectly balanced for
//                   ANY digit distribution); runs that end inside a chunk go to their
//                   bucket, the first/last run of a chunk become partial records
//   k_segscan       block-wide segmented scan over partial records, 256 -> 2 per block,
//                   repeated until one block is left
//   k_digit_sums / k_digit_final       sum_b b*B[w][b]: the bucket index split in two digits, twice
//                   twice (2 additions per element, sub-wave groups, shuffle butterflies),
//                   then 16-term suffix scans; k_window_weighted_small for <= 256 buckets
//   k_msm_small     n <= 4096: digit multiples per thread + butterfly sums (latency path)
//   tail            O(256) sequential doublings: window combine + to-affine
//                   (host thread on 4x64-bit limbs, host_tail.hpp, or the one-lane device
//                   kernel k_tail; same Horner chain)
// This file: the C-ABI entry points and the IPA state object (deferred generator folding).
// Kernels and host orchestration live in the *.hpp files included below (one translation unit).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/bpmi.h"
#include "curve.hpp"
#include "field.hpp"
#include "scalar.hpp"

using namespace bpmi;

#define BPMI_VERSION 100

#include "context.hpp"
#include "device_util.hpp"
#include "msm_kernels.hpp"
#include "fold_ops_host.hpp"
#include "point_kernels.hpp"
#include "scalar_kernels.hpp"
#include "host_tail.hpp"
#include "msm_host.hpp"
#include "rp_batch_host.hpp"
#include "host_pool.hpp"
#include "rp_algebra_host.hpp"
#include "transcript_host.hpp"
#include "rp_wire_v2_host.hpp"
#include "rp_batch_kernels.hpp"
#include "rp_prove_kernels.hpp"

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
  for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->stage_in) (void)hipFree(ctx->stage_in);
  if (ctx->rp_buf) (void)hipFree(ctx->rp_buf);
  if (ctx->vflag) (void)hipHostFree(ctx->vflag);
  if (ctx->vflag_dev) (void)hipFree(ctx->vflag_dev);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  if (ctx->up_ring) (void)hipHostFree(ctx->up_ring);
  if (ctx->up_ev) (void)hipEventDestroy(ctx->up_ev);
  if (ctx->stream1) { (void)hipStreamSynchronize(ctx->stream1); (void)hipStreamDestroy(ctx->stream1); }
  if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
  if (ctx->stream_acc) { (void)hipStreamSynchronize(ctx->stream_acc); (void)hipStreamDestroy(ctx->stream_acc); }
  for (auto e : ctx->ev_sorted) if (e) (void)hipEventDestroy(e);
  if (ctx->ws2) (void)hipFree(ctx->ws2);
  if (ctx->fold_tab) (void)hipFree(ctx->fold_tab);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  for (auto e : ctx->ev_slice) if (e) (void)hipEventDestroy(e);
  for (auto e : ctx->ev_accum) if (e) (void)hipEventDestroy(e);
  if (ctx->ws1) (void)hipFree(ctx->ws1);
  for (auto &pd : ctx->pend) { if (pd.pin) (void)hipHostFree(pd.pin); if (pd.done) (void)hipEventDestroy(pd.done); }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx->helper;
  msm_graphs_clear(ctx);
  delete ctx->graphs;
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
  if (!strcmp(name, "accum_stream")) { if (value < 0 || value > 2) return fail(ctx, BPMI_E_ARG, "accum_stream must be 0, 1 or 2"); ctx->opt_accum_stream = (int)value; return BPMI_OK; }
  if (!strcmp(name, "lane_priority")) { if (value < -1 || value > 1) return fail(ctx, BPMI_E_ARG, "lane_priority must be -1, 0 or 1"); ctx->opt_lane_prio = (int)value; return BPMI_OK; }
  if (!strcmp(name, "pair_sched")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "pair_sched must be 0 or 1"); ctx->opt_pair_sched = (int)value; return BPMI_OK; }
  if (!strcmp(name, "prover_wire_format")) { if (value != 2 && value != 3) return fail(ctx, BPMI_E_ARG, "prover_wire_format must be 2 or 3"); ctx->opt_prover_wire = (int)value; return BPMI_OK; }
  if (!strcmp(name, "prover_split")) { if (value < 0 || value > (1 << 20)) return fail(ctx, BPMI_E_ARG, "prover_split must be 0, 1 or the smallest half"); ctx->opt_prover_split = (int)value; return BPMI_OK; }
  if (!strcmp(name, "rounds")) { if (value < 0 || value > 16) return fail(ctx, BPMI_E_ARG, "rounds must be 0 .. 16"); ctx->opt_rounds = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "pair_rounds")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "pair_rounds must be 0 or 1"); ctx->opt_pair_rounds = (int)value; return BPMI_OK; }
  if (!strcmp(name, "accum_chain")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "accum_chain must be 0 or 1"); ctx->opt_accum_chain = (int)value; return BPMI_OK; }
  if (!strcmp(name, "slice_n")) { if (value < -1 || value > (1 << 23) || (value > 0 && value < (1 << 16))) return fail(ctx, BPMI_E_ARG, "slice_n must be -1, 0 or 2^16 .. 2^23"); ctx->opt_slice_n = (int)value; return BPMI_OK; }
  if (!strcmp(name, "slice_min")) { if (value < 0 || value > (1 << 23)) return fail(ctx, BPMI_E_ARG, "slice_min must be 0 .. 2^23"); ctx->opt_slice_min = (int)value; return BPMI_OK; }
  if (!strcmp(name, "mid_parts")) { if (value < 0 || value > 4) return fail(ctx, BPMI_E_ARG, "mid_parts must be 0 .. 4"); ctx->opt_mid_parts = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "mixed_windows")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "mixed_windows must be 0 or 1"); ctx->opt_mixed = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "top_window_unsigned")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "top_window_unsigned must be 0 or 1"); ctx->opt_top2 = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "prover_table_bits")) { if (value != 0 && (value < 4 || value > 16)) return fail(ctx, BPMI_E_ARG, "prover_table_bits must be 0 or 4..16"); ctx->opt_prover_tw = (int)value; return BPMI_OK; }
  if (!strcmp(name, "ipa_fixed_generators")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "ipa_fixed_generators must be 0 or 1"); ctx->opt_ipa_fixed = (int)value; ctx->fold_key_g = ctx->fold_key_h = nullptr; return BPMI_OK; }
  if (!strcmp(name, "validate_points")) { if (value < 0 || value > 2) return fail(ctx, BPMI_E_ARG, "validate_points must be 0, 1 or 2"); ctx->opt_validate = (int)value; return BPMI_OK; }
  if (!strcmp(name, "hist_scan_fused")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "hist_scan_fused must be 0 or 1"); ctx->opt_histscan = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "reduce_fit")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "reduce_fit must be 0 or 1"); ctx->opt_reduce_fit = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "final_spread")) { if (value < 0 || value > 3) return fail(ctx, BPMI_E_ARG, "final_spread must be 0 .. 3"); ctx->opt_final_spread = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "sort_inblock")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "sort_inblock must be 0 or 1"); ctx->opt_inblock = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "segscan_fused")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "segscan_fused must be 0 or 1"); ctx->opt_segfuse = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "priority")) { if (value < 0 || value > 31 || (value > 1 && value < 16)) return fail(ctx, BPMI_E_ARG, "priority must be 0, 1 or 16 + a mask of stages"); ctx->opt_prio = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "hist_threads")) { if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(ctx, BPMI_E_ARG, "hist_threads must be 0, 256, 512 or 1024"); ctx->opt_hist_threads = (int)value; return BPMI_OK; }
  if (!strcmp(name, "hist_blocks")) { if (value < 0 || value > 8192) return fail(ctx, BPMI_E_ARG, "hist_blocks must be in [0, 8192]"); ctx->opt_hist_blocks = (int)value; return BPMI_OK; }
  if (!strcmp(name, "direct_result")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "direct_result must be 0 or 1"); ctx->opt_direct = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "graphs")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "graphs must be 0 or 1"); ctx->opt_graph = (int)value; if (!value) msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "tail_thread")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "tail_thread must be 0 or 1"); ctx->opt_tail_thread = (int)value; return BPMI_OK; }
  if (!strcmp(name, "pair_chain")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "pair_chain must be 0 or 1"); ctx->opt_pair_chain = (int)value; return BPMI_OK; }
  if (!strcmp(name, "mid_min")) { if (value < -1 || value > MID_NMAX) return fail(ctx, BPMI_E_ARG, "mid_min must be -1 .. 8448"); ctx->opt_mid_min = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "mid_single_min")) { if (value < -1 || value > MID_NMAX) return fail(ctx, BPMI_E_ARG, "mid_single_min must be -1 .. 8448"); ctx->opt_mid_single = (int)value; msm_graphs_clear(ctx); return BPMI_OK; }
  if (!strcmp(name, "small_pair")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "small_pair must be 0 or 1"); ctx->opt_pair1 = (int)value; return BPMI_OK; }
  if (!strcmp(name, "fused_scan")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "fused_scan must be 0 or 1"); ctx->opt_fuse = (int)value; return BPMI_OK; }
  if (!strcmp(name, "quad_final")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "quad_final must be 0 or 1"); ctx->opt_quad = (int)value; return BPMI_OK; }
  if (!strcmp(name, "spin_wait")) { if (value < 0 || value > 10000000) return fail(ctx, BPMI_E_ARG, "spin_wait must be in [0, 10^7]"); ctx->opt_spin_wait = (int)value; return BPMI_OK; }
  if (!strcmp(name, "mul_batch_glv")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "mul_batch_glv must be 0 or 1"); ctx->opt_mulb = (int)value; return BPMI_OK; }
  if (!strcmp(name, "tail")) { if (value < 0 || value > 2) return fail(ctx, BPMI_E_ARG, "tail must be 0, 1 or 2"); ctx->opt_tail = (int)value; return BPMI_OK; }
  if (!strcmp(name, "small_n")) { if (value < -1 || value > (1 << 16)) return fail(ctx, BPMI_E_ARG, "small_n must be -1 .. 65536"); ctx->opt_small = (int)value; return BPMI_OK; }
  if (!strcmp(name, "split")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "split must be 0 or 1"); ctx->opt_split = (int)value; return BPMI_OK; }
  if (!strcmp(name, "async_lanes")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "async_lanes must be 0 or 1"); ctx->opt_async_lanes = (int)value; return BPMI_OK; }
  if (!strcmp(name, "rp_only_role")) { if (value < -1 || value > 3) return fail(ctx, BPMI_E_ARG, "rp_only_role must be in [-1, 3]"); ctx->opt_rp_only_role = (int)value; return BPMI_OK; }
  if (!strcmp(name, "glv")) { if (value < -1 || value > 1) return fail(ctx, BPMI_E_ARG, "glv must be -1, 0 or 1"); ctx->opt_glv = (int)value; return BPMI_OK; }
  if (!strcmp(name, "rp_priority")) { if (value < 0 || value > 2) return fail(ctx, BPMI_E_ARG, "rp_priority must be 0, 1 or 2"); ctx->opt_rp_prio = (int)value; return BPMI_OK; }
  if (!strcmp(name, "rp_slices")) { if (value < 0 || value > 4) return fail(ctx, BPMI_E_ARG, "rp_slices must be 0 .. 4"); ctx->opt_rp_slices = (int)value; return BPMI_OK; }
  if (!strcmp(name, "rp_overlap")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "rp_overlap must be 0 or 1"); ctx->opt_rp_overlap = (int)value; return BPMI_OK; }
  if (!strcmp(name, "rp_rows")) { if (value < 0) return fail(ctx, BPMI_E_ARG, "rp_rows must be >= 0"); ctx->opt_rp_rows = (int)value; return BPMI_OK; }
  if (!strcmp(name, "rp_lanes")) {
    if (value != 0 && (value < 1 || value > 64 || (value & (value - 1)))) return fail(ctx, BPMI_E_ARG, "rp_lanes must be 0 or a power of two <= 64");
    ctx->opt_rp_lanes = (int)value;
    return BPMI_OK;
  }
  if (!strcmp(name, "pair_phases")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "pair_phases must be 0 or 1"); ctx->opt_pair_phases = (int)value; return BPMI_OK; }
  if (!strcmp(name, "fold_wnaf")) { if (value < 0 || value > 2) return fail(ctx, BPMI_E_ARG, "fold_wnaf must be 0, 1 or 2"); ctx->opt_fold_wnaf = (int)value; return BPMI_OK; }
  if (!strcmp(name, "reduce_epl")) { if (value < 0 || value > 64) return fail(ctx, BPMI_E_ARG, "reduce_epl must be 0..64"); ctx->opt_epl = (int)value; return BPMI_OK; }
  if (!strcmp(name, "chunk")) { if (value < 0 || value > 4096) return fail(ctx, BPMI_E_ARG, "chunk must be 0..4096"); ctx->opt_chunk = (int)value; return BPMI_OK; }
  if (!strcmp(name, "ipa_small_step")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "ipa_small_step must be 0 or 1"); ctx->opt_ipa_step = (int)value; return BPMI_OK; }
  if (!strcmp(name, "fold_shared")) { if (value < 0 || value > 1) return fail(ctx, BPMI_E_ARG, "fold_shared must be 0 or 1"); ctx->opt_fold_shared = (int)value; return BPMI_OK; }
  if (!strcmp(name, "ipa_small_m")) { if (value < 0 || (value & (value - 1))) return fail(ctx, BPMI_E_ARG, "ipa_small_m must be 0 (default), 1 (never) or a power of two"); ctx->opt_ipa_small = value; return BPMI_OK; }
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
  HIPCHK(ctx, h2d(ctx, dptr, host, bytes, ctx->stream));
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

// ---- on-curve check of caller-supplied points (option "validate_points") --------------------------------------------------
// The reference can never hand an off-curve point to multiexp: fastecdsa's Point constructor rejects it (reached from
// /root/reference/src/utils/utils.py:119-131), and ec.py keeps that check for Python callers.  A C caller hands in raw bytes, so
// the entry points that take HOST pointers check them here (level 1, the default; level 2: the synchronous _dev entry points as
// well; 0: never): the identity (64 zero bytes) or x, y < p with y^2 = x^3 + 7.  Large arrays: k_ec_validate queued behind the
// upload, its verdict (the smallest bad index) copied to a page-locked word that is read after the call's own synchronisation --
// no extra wait; the call then fails with BPMI_E_ARG and writes NO result.  A few points: checked on the host.
#define VALIDATE_HOST_MAX 64u
static int validate_begin(bpmi_ctx *ctx, hipStream_t st) {
  if (!ctx->vflag) {
    HIPCHK(ctx, hipHostMalloc((void **)&ctx->vflag, 64, hipHostMallocDefault));
    HIPCHK(ctx, hipMalloc((void **)&ctx->vflag_dev, 256));
  }
  *ctx->vflag = 0xFFFFFFFFu;
  HIPCHK(ctx, hipMemsetAsync(ctx->vflag_dev, 0xFF, 4, st));
  return BPMI_OK;
}
// array `which` (0 .. 3) of the call: d_pts[0 .. n) on the device
static void validate_enqueue(bpmi_ctx *ctx, const void *d_pts, uint64_t n, u32 which, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_ec_validate, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, (const u32 *)d_pts, (u32)n, which << 28, ctx->vflag_dev);
}
static int validate_fetch(bpmi_ctx *ctx, hipStream_t st) {
  HIPCHK(ctx, hipMemcpyAsync(ctx->vflag, ctx->vflag_dev, 4, hipMemcpyDeviceToHost, st));
  return BPMI_OK;
}
// after the stream has been synchronised: names[which] = what the bad array is called in the message
static int validate_end(bpmi_ctx *ctx, const char *fn, const char *const *names) {
  const u32 v = *ctx->vflag;
  if (v == 0xFFFFFFFFu) return BPMI_OK;
  return fail(ctx, BPMI_E_ARG, std::string(fn) + ": " + names[v >> 28] + "[" + std::to_string(v & 0x0FFFFFFFu) + "] is not a point of the curve");
}
static int validate_host(bpmi_ctx *ctx, const uint8_t *pts, uint64_t n, const char *fn, const char *name) {
  for (uint64_t i = 0; i < n; i++) {
    u32 w[16];
    memcpy(w, pts + 64 * i, 64);
    if (!wire_point_valid(w)) return fail(ctx, BPMI_E_ARG, std::string(fn) + ": " + name + "[" + std::to_string(i) + "] is not a point of the curve");
  }
  return BPMI_OK;
}

// ---- MSM ------------------------------------------------------------------------------
static int msm_dev_impl(bpmi_ctx *ctx, const void *d_pts, const void *d_scalars, uint64_t n, uint8_t out[64]);
int bpmi_msm_dev(bpmi_ctx *ctx, const void *d_pts, const void *d_scalars, uint64_t n, uint8_t out[64]) {
  if (!ctx || !out || (n && (!d_pts || !d_scalars))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (ctx->opt_validate < 2 || n == 0) return msm_dev_impl(ctx, d_pts, d_scalars, n, out);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = validate_begin(ctx, ctx->stream);
  if (rc) return rc;
  validate_enqueue(ctx, d_pts, n, 0, ctx->stream);
  rc = validate_fetch(ctx, ctx->stream);
  if (rc) return rc;
  rc = msm_dev_impl(ctx, d_pts, d_scalars, n, out);
  if (rc) return rc;
  static const char *const names[] = {"d_pts"};
  rc = validate_end(ctx, "bpmi_msm_dev", names);
  if (rc) memset(out, 0, 64);
  return rc;
}
static int msm_dev_impl(bpmi_ctx *ctx, const void *d_pts, const void *d_scalars, uint64_t n, uint8_t out[64]) {
  if (!ctx || !out || (n && (!d_pts || !d_scalars))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // (inputs of more than ~1.25 x 2^20 pairs run as slices of about 2^20 pairs, two in flight: msm_run, csrc/msm_host.hpp)
  Segs s = segs_init();
  s.pts[0] = (const u32 *)d_pts; s.sc[0] = (const u32 *)d_scalars; s.n[0] = (u32)n; s.total = (u32)n;
  return msm_run(ctx, s, out);
}
// what an MSM of n pairs runs as under the ctx's current options (no GPU work): bench.py counts its multiply-adds from this
int bpmi_msm_geometry(bpmi_ctx *ctx, uint64_t n, int pipelined, uint32_t geom[8]) {
  if (!ctx || !geom) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  memset(geom, 0, 8 * sizeof(uint32_t));
  if (n == 0) return BPMI_OK;
  Segs s = segs_init();
  s.n[0] = (u32)n; s.total = (u32)n;
  const uint64_t K = pipelined ? 1 : msm_slice_count(ctx, s);
  const uint64_t per = (n + K - 1) / K;
  MsmGeom g;
  bool mid, small, glv;
  const bool saved = ctx->chain_accum;
  ctx->chain_accum = pipelined ? ctx->opt_async_lanes != 0 : (K > 1 && per >= (1u << 19));
  msm_pick_geometry(ctx, per, 0, 0, g, mid, small, glv);
  ctx->chain_accum = saved;
  geom[0] = small ? 1u : (mid ? 2u : 0u); geom[1] = g.c; geom[2] = g.W; geom[3] = g.top2; geom[4] = g.G; geom[5] = g.L; geom[6] = (uint32_t)K; geom[7] = (uint32_t)per;
  return BPMI_OK;
}
// Asynchronous pair: enqueue returns as soon as the MSM's kernels and its device->host copy are queued on
// the ctx stream; finish waits for THAT MSM only and runs the host part of its tail.  Two slots, so the
// host tail (and the caller's own work, e.g. the exchange of partial results) of MSM k overlaps the
// kernels of MSM k + 1.
int bpmi_msm_dev_enqueue(bpmi_ctx *ctx, int slot, const void *d_pts, const void *d_scalars, uint64_t n) {
  if (!ctx || (n && (!d_pts || !d_scalars))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (slot < 0 || slot >= BPMI_LANES) return fail(ctx, BPMI_E_ARG, "slot must be 0, 1 or 2");
  if (n > (1ull << 23)) return fail(ctx, BPMI_E_ARG, "bpmi_msm_dev_enqueue takes at most 2^23 pairs (use bpmi_msm_dev)");
  if (ctx->opt_split) return fail(ctx, BPMI_E_STATE, "option split is not available with the asynchronous entry points");
  if (ctx->pend[slot].async) return fail(ctx, BPMI_E_STATE, "an MSM is still pending in this slot (bpmi_msm_finish it first)");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Segs s = segs_init();
  s.pts[0] = (const u32 *)d_pts; s.sc[0] = (const u32 *)d_scalars; s.n[0] = (u32)n; s.total = (u32)n;
  // option async_lanes: slot 1 runs on the second lane (own stream + workspace), so the latency-bound tail stages of
  // one MSM (segmented scan, bucket reduction: few waves) overlap the throughput stages of the next one
  const int lane = ctx->opt_async_lanes ? slot : 0;
  int rc;
  if (ctx->opt_async_lanes) {
    rc = ensure_lane(ctx, 1);
    if (rc == BPMI_OK) rc = ensure_lane(ctx, 2);               // (both extra lanes exist from the first burst on: see below)
    if (rc) return rc;
  }
  if (ctx->opt_async_lanes && !ctx->async_lane1_ordered) {
    // Order EVERY extra lane after whatever produced the inputs on the ctx stream, once per burst and BEFORE this burst's first MSM
    // is queued (the option's contract: inputs complete before the first enqueue of a burst).  Recorded at a lane's first use
    // instead, the event sits behind slot 0's whole MSM: round 3 found that for lane 1 (a caller that queued a pair and waited for
    // both got them one after the other, profiles/r03_pair_modes.txt), and round 4's lane 2 still did it -- its "three in flight"
    // measurements ran on a partly serialised pipeline (re-measured: profiles/r05_pipeline_depth_ab.txt).
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream1, ctx->ev_fork, 0));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    ctx->async_lane1_ordered = ctx->async_lane2_ordered = true;
  }
  ctx->chain_accum = ctx->opt_async_lanes != 0;
  ctx->chain_free = ctx->opt_accum_chain == 0;
  rc = msm_enqueue(ctx, lane, slot, s);
  ctx->chain_free = false;
  ctx->chain_accum = false;
  if (rc == BPMI_OK) { ctx->pend[slot].async = true; ctx->pend[slot].async_empty = (n == 0); }
  return rc;
}
int bpmi_msm_finish(bpmi_ctx *ctx, int slot, uint8_t out[64]) {
  if (!ctx || !out) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (slot < 0 || slot >= BPMI_LANES) return fail(ctx, BPMI_E_ARG, "slot must be 0, 1 or 2");
  if (!ctx->pend[slot].async) return fail(ctx, BPMI_E_STATE, "no MSM was enqueued in this slot");
  ctx->pend[slot].async = false;
  if (!ctx->pend[0].async && !ctx->pend[1].async && !ctx->pend[2].async) { ctx->async_lane1_ordered = ctx->async_lane2_ordered = false; ctx->accum_chain_lane = -1; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // msm_finish reads an inactive slot as "n was 0" (the identity, which a verifier reads as VALID): only a slot that was enqueued
  // with n = 0 may be inactive here
  if (!ctx->pend[slot].active && !ctx->pend[slot].async_empty) return fail(ctx, BPMI_E_STATE, "the MSM of this slot was abandoned");
  return msm_finish(ctx, slot, out);
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
  HIPCHK(ctx, h2d(ctx, dp, pts, 64 * n, ctx->stream));
  HIPCHK(ctx, h2d(ctx, ds, scalars, 32 * n, ctx->stream));
  const bool check = ctx->opt_validate >= 1;
  if (check) {
    rc = validate_begin(ctx, ctx->stream);
    if (rc) return rc;
    validate_enqueue(ctx, dp, n, 0, ctx->stream);
    rc = validate_fetch(ctx, ctx->stream);
    if (rc) return rc;
  }
  rc = msm_dev_impl(ctx, dp, ds, n, out);
  if (rc || !check) return rc;
  static const char *const names[] = {"pts"};
  rc = validate_end(ctx, "bpmi_msm", names);
  if (rc) memset(out, 0, 64);
  return rc;
}

// two independent MSMs from host buffers, overlapped on the ctx's two lanes (one staging upload,
// one synchronisation): the A / S and T1 / T2 pairs of a range proof, L / R outside the IPA object
int bpmi_msm2(bpmi_ctx *ctx, const uint8_t *pts0, const uint8_t *sc0, uint64_t n0, uint8_t out0[64], const uint8_t *pts1,
              const uint8_t *sc1, uint64_t n1, uint8_t out1[64]) {
  if (!ctx || !out0 || !out1 || (n0 && (!pts0 || !sc0)) || (n1 && (!pts1 || !sc1))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n0 > (1ull << 23) || n1 > (1ull << 23)) return fail(ctx, BPMI_E_ARG, "bpmi_msm2 takes at most 2^23 pairs per MSM");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t o_s0 = align_up(64 * n0, 256), o_p1 = o_s0 + align_up(32 * n0, 256), o_s1 = o_p1 + align_up(64 * n1, 256);
  int rc = ensure_stage_in(ctx, o_s1 + 32 * n1 + 512);
  if (rc) return rc;
  char *d = (char *)ctx->stage_in;
  if (n0) { HIPCHK(ctx, h2d(ctx, d, pts0, 64 * n0, ctx->stream)); HIPCHK(ctx, h2d(ctx, d + o_s0, sc0, 32 * n0, ctx->stream)); }
  if (n1) { HIPCHK(ctx, h2d(ctx, d + o_p1, pts1, 64 * n1, ctx->stream)); HIPCHK(ctx, h2d(ctx, d + o_s1, sc1, 32 * n1, ctx->stream)); }
  Segs a = segs_init(), b = segs_init();
  a.pts[0] = (const u32 *)d; a.sc[0] = (const u32 *)(d + o_s0); a.n[0] = (u32)n0; a.total = (u32)n0;
  b.pts[0] = (const u32 *)(d + o_p1); b.sc[0] = (const u32 *)(d + o_s1); b.n[0] = (u32)n1; b.total = (u32)n1;
  const bool check = ctx->opt_validate >= 1 && (n0 || n1);
  if (check) {
    rc = validate_begin(ctx, ctx->stream);
    if (rc) return rc;
    validate_enqueue(ctx, d, n0, 0, ctx->stream);
    validate_enqueue(ctx, d + o_p1, n1, 1, ctx->stream);
    rc = validate_fetch(ctx, ctx->stream);
    if (rc) return rc;
  }
  rc = msm_run_pair(ctx, a, out0, b, out1);
  if (rc || !check) return rc;
  static const char *const names[] = {"pts0", "pts1"};
  rc = validate_end(ctx, "bpmi_msm2", names);
  if (rc) { memset(out0, 0, 64); memset(out1, 0, 64); }
  return rc;
}

// ---- batched point ops --------------------------------------------------------------------
int bpmi_ec_mul_batch_dev(bpmi_ctx *ctx, const void *d_pts, const void *d_scalars, uint64_t n, void *d_out) {
  if (!ctx || (n && (!d_pts || !d_scalars || !d_out))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (n < (uint64_t)MULB_MIN_N || ctx->opt_mulb == 0) {            // few points: the table kernel's inversion chain is not worth its latency
    StageTimer t(ctx, ST_MULBATCH);
    hipLaunchKernelGGL(k_ec_mul_batch, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const u32 *)d_pts,
                       (const u32 *)d_scalars, (u32)n, (u32 *)d_out);
    HIPCHK(ctx, hipGetLastError());
    return BPMI_OK;
  }
  // slices of 3 x 65 536 points: three waves on every SIMD (the kernels' occupancy) in ONE round, and a bounded workspace
  // (affine 3P, 5P, 7P: 216 B per point; the table kernel's scratch columns: 864 B per point as it is laid out for two arrays)
  const uint64_t slice = 3ull << 16;
  const uint64_t m0 = std::min<uint64_t>(n, slice), nthr0 = (m0 + ODDMUL_PER_THREAD_MULB - 1) / ODDMUL_PER_THREAD_MULB;
  const size_t tab_bytes = align_up(3 * 72 * (size_t)m0, 256), scr_bytes = 144 * 3 * (size_t)ODDMUL_PER_THREAD_MULB * 2 * nthr0;
  int rc = ensure_ws(ctx, tab_bytes + scr_bytes);
  if (rc) return rc;
  u32 *tab = (u32 *)ctx->ws, *scr = (u32 *)((char *)ctx->ws + tab_bytes);
  for (uint64_t base = 0; base < n; base += slice) {
    const u32 m = (u32)std::min<uint64_t>(slice, n - base), nthr = (m + ODDMUL_PER_THREAD_MULB - 1) / ODDMUL_PER_THREAD_MULB;
    const u32 *p = (const u32 *)d_pts + 16 * base, *k = (const u32 *)d_scalars + 8 * base;
    StageTimer t(ctx, ST_MULBATCH);
    hipLaunchKernelGGL(k_ec_odd_multiples<ODDMUL_PER_THREAD_MULB>, dim3((nthr + 255) / 256), dim3(256), 0, ctx->stream, p, (const u32 *)nullptr, m, tab,
                       (u32 *)nullptr, scr);
    hipLaunchKernelGGL(k_ec_mul_batch_glv, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, p, k, m, (const u32 *)tab, (u32 *)d_out + 16 * base);
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
  HIPCHK(ctx, h2d(ctx, dp, pts, 64 * n, ctx->stream));
  HIPCHK(ctx, h2d(ctx, ds, scalars, 32 * n, ctx->stream));
  const bool check = ctx->opt_validate >= 1;
  if (check) {
    rc = validate_begin(ctx, ctx->stream);
    if (rc) return rc;
    validate_enqueue(ctx, dp, n, 0, ctx->stream);
    rc = validate_fetch(ctx, ctx->stream);
    if (rc) return rc;
  }
  rc = bpmi_ec_mul_batch_dev(ctx, dp, ds, n, dout);
  if (rc) return rc;
  if (check) {                                   // the verdict before anything is written to the caller's buffer
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    static const char *const names[] = {"pts"};
    rc = validate_end(ctx, "bpmi_ec_mul_batch", names);
    if (rc) return rc;
  }
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
  HIPCHK(ctx, h2d(ctx, d1, p1, 64 * n, ctx->stream));
  HIPCHK(ctx, h2d(ctx, d2, p2, 64 * n, ctx->stream));
  const bool check = ctx->opt_validate >= 1;
  if (check) {
    rc = validate_begin(ctx, ctx->stream);
    if (rc) return rc;
    validate_enqueue(ctx, d1, n, 0, ctx->stream);
    validate_enqueue(ctx, d2, n, 1, ctx->stream);
    rc = validate_fetch(ctx, ctx->stream);
    if (rc) return rc;
  }
  rc = bpmi_ec_lincomb2_batch_dev(ctx, d1, d2, k1, k2, n, dout);
  if (rc) return rc;
  if (check) {
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    static const char *const names[] = {"p1", "p2"};
    rc = validate_end(ctx, "bpmi_ec_lincomb2_batch", names);
    if (rc) return rc;
  }
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
  HIPCHK(ctx, h2d(ctx, dp, pts, 64 * n, ctx->stream));
  // (the per-GPU partial results a sharded caller folds here are the library's own outputs; a few points: checked on the host)
  const bool check = ctx->opt_validate >= 1;
  if (check && n <= VALIDATE_HOST_MAX) { rc = validate_host(ctx, pts, n, "bpmi_ec_sum", "pts"); if (rc) return rc; }
  else if (check) {
    rc = validate_begin(ctx, ctx->stream);
    if (rc) return rc;
    validate_enqueue(ctx, dp, n, 0, ctx->stream);
    rc = validate_fetch(ctx, ctx->stream);
    if (rc) return rc;
  }
  {
    StageTimer t(ctx, ST_MISC);
    hipLaunchKernelGGL(k_ec_sum, dim3(1), dim3(256), 0, ctx->stream, (const u32 *)dp, (u32)n, (u32 *)dout);
  }
  uint8_t tmp[64];
  HIPCHK(ctx, hipMemcpyAsync(tmp, dout, 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (check && n > VALIDATE_HOST_MAX) {
    static const char *const names[] = {"pts"};
    rc = validate_end(ctx, "bpmi_ec_sum", names);
    if (rc) return rc;
  }
  memcpy(out, tmp, 64);
  return BPMI_OK;
}

int bpmi_ec_sum_dev(bpmi_ctx *ctx, const void *d_pts, uint64_t n, uint8_t out[64]) {
  if (!ctx || !out || (n && !d_pts)) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) { memset(out, 0, 64); return BPMI_OK; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage_in(ctx, 512);
  if (rc) return rc;
  {
    StageTimer t(ctx, ST_MISC);
    hipLaunchKernelGGL(k_ec_sum, dim3(1), dim3(256), 0, ctx->stream, (const u32 *)d_pts, (u32)n, (u32 *)ctx->stage_in);
  }
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->stage_in, 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

// the fold without the wait: the canonical affine sum is written to d_out (64 B of device memory) by a kernel queued on the ctx
// stream; the caller reads it after bpmi_sync / bpmi_download (a pipelined caller overlaps the fold of step j with step j + 1)
int bpmi_ec_sum_dev_enqueue(bpmi_ctx *ctx, const void *d_pts, uint64_t n, void *d_out) {
  if (!ctx || !d_out || (n && !d_pts)) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (n == 0) { HIPCHK(ctx, hipMemsetAsync(d_out, 0, 64, ctx->stream)); return BPMI_OK; }
  {
    StageTimer t(ctx, ST_MISC);
    hipLaunchKernelGGL(k_ec_sum, dim3(1), dim3(256), 0, ctx->stream, (const u32 *)d_pts, (u32)n, (u32 *)d_out);
  }
  HIPCHK(ctx, hipGetLastError());
  return BPMI_OK;
}

int bpmi_ec_decompress_batch(bpmi_ctx *ctx, const uint8_t *comp, uint64_t n, uint8_t *out, uint8_t *ok) {
  if (!ctx || (n && (!comp || !out || !ok))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage_in(ctx, 98 * n + 1024);
  if (rc) return rc;
  char *din = (char *)ctx->stage_in, *dout = din + align_up(33 * n, 256), *dok = dout + align_up(64 * n, 256);
  HIPCHK(ctx, h2d(ctx, din, comp, 33 * n, ctx->stream));
  {
    StageTimer t(ctx, ST_DECOMP);
    hipLaunchKernelGGL(k_ec_decompress, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t *)din, (u32)n,
                       (u32 *)dout, (uint8_t *)dok);
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, dout, 64 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ok, dok, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

// the same with the points left in device memory (d_out: n x 64 B on the ctx's device), e.g. straight into the point array of
// the batch verifier's one MSM: no copy of the decompressed points to the host and back
int bpmi_ec_decompress_batch_dev(bpmi_ctx *ctx, const uint8_t *comp, uint64_t n, void *d_out, uint8_t *ok) {
  if (!ctx || (n && (!comp || !d_out || !ok))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_stage_in(ctx, 34 * n + 1024);
  if (rc) return rc;
  char *din = (char *)ctx->stage_in, *dok = din + align_up(33 * n, 256);
  HIPCHK(ctx, h2d(ctx, din, comp, 33 * n, ctx->stream));
  {
    StageTimer t(ctx, ST_DECOMP);
    hipLaunchKernelGGL(k_ec_decompress, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint8_t *)din, (u32)n, (u32 *)d_out, (uint8_t *)dok);
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(ok, dok, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}
int bpmi_memcpy_dev(bpmi_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
  if (!ctx || (bytes && (!d_dst || !d_src))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (!bytes) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return BPMI_OK;
}
// one MSM over up to three (points, scalars) arrays that live in different device buffers -- no gather / concat
int bpmi_msm_segs_dev(bpmi_ctx *ctx, uint32_t nseg, const void *const *d_pts, const void *const *d_scalars, const uint64_t *n, uint8_t out[64]) {
  if (!ctx || !out || (nseg && (!d_pts || !d_scalars || !n))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (nseg > 3) return fail(ctx, BPMI_E_ARG, "at most three segments");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Segs s = segs_init();
  uint64_t total = 0;
  u32 k = 0;
  for (u32 i = 0; i < nseg; i++) {
    if (!n[i]) continue;
    if (!d_pts[i] || !d_scalars[i]) return fail(ctx, BPMI_E_ARG, "null segment");
    s.pts[k] = (const u32 *)d_pts[i]; s.sc[k] = (const u32 *)d_scalars[i]; s.n[k] = (u32)n[i];
    total += n[i];
    k++;
  }
  if (total > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  s.total = (u32)total;
  return msm_run(ctx, s, out);
}

// ---- scalar ops -------------------------------------------------------------------------------
// njobs (1 or 2) inner products of length n in two launches; d_partial: njobs x SC_DOT_MAX_BLOCKS x 32 B
static int sc_dot_jobs(bpmi_ctx *ctx, const DotJobs &jobs, u32 njobs, uint64_t n, u32 *d_partial) {
  const u32 nb = n <= 4096 ? 1u : (u32)std::min<uint64_t>((n + 255) / 256, SC_DOT_MAX_BLOCKS);
  StageTimer t(ctx, ST_SCDOT);
  hipLaunchKernelGGL(k_sc_dot, dim3(nb, njobs), dim3(256), 0, ctx->stream, jobs, (u32)n, d_partial);
  if (nb > 1) hipLaunchKernelGGL(k_sc_sum, dim3(njobs), dim3(256), 0, ctx->stream, (const u32 *)d_partial, nb, jobs);
  return BPMI_OK;
}
static int sc_dot_dev_to(bpmi_ctx *ctx, const void *d_a, const void *d_b, uint64_t n, u32 *d_out, u32 *d_partial) {
  DotJobs j;
  j.a[0] = j.a[1] = (const u32 *)d_a; j.b[0] = j.b[1] = (const u32 *)d_b; j.out[0] = j.out[1] = d_out;
  return sc_dot_jobs(ctx, j, 1, n, d_partial);
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
  HIPCHK(ctx, h2d(ctx, da, a, 32 * n, ctx->stream));
  HIPCHK(ctx, h2d(ctx, db, b, 32 * n, ctx->stream));
  return bpmi_sc_dot_dev(ctx, da, db, n, out);
}
int bpmi_sc_fold_dev(bpmi_ctx *ctx, const void *d_lo, const void *d_hi, const uint8_t x[32], const uint8_t y[32], uint64_t n, void *d_out) {
  if (!ctx || !x || !y || (n && (!d_lo || !d_hi || !d_out))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n > BPMI_MAX_N) return fail(ctx, BPMI_E_ARG, "n exceeds BPMI_MAX_N");
  if (n == 0) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  FoldJobs fj;
  fj.lo[0] = fj.lo[1] = (const u32 *)d_lo; fj.hi[0] = fj.hi[1] = (const u32 *)d_hi; fj.out[0] = fj.out[1] = (u32 *)d_out;
  memcpy(fj.xy[0].k1, x, 32); memcpy(fj.xy[0].k2, y, 32);
  fj.xy[1] = fj.xy[0];
  {
    StageTimer t(ctx, ST_SCFOLD);
    hipLaunchKernelGGL(k_sc_fold, dim3((u32)((n + 255) / 256), 1), dim3(256), 0, ctx->stream, fj, (u32)n);
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
  HIPCHK(ctx, h2d(ctx, dl, lo, 32 * n, ctx->stream));
  HIPCHK(ctx, h2d(ctx, dh, hi, 32 * n, ctx->stream));
  rc = bpmi_sc_fold_dev(ctx, dl, dh, x, y, n, dout);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(out, dout, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

// ---- IPA verifier: s-vector on the device + ONE multi-segment MSM ---------------------------------------------
// layout in ctx->stage_in: sa (32 n) | sb (32 n) | x table (64 k) | half tables | extra area
struct SvecLayout { u32 *sa, *sb, *xt, *tab; char *extra; u32 k, kl; };
static int svector_layout(bpmi_ctx *ctx, uint64_t n, u32 k, size_t extra_bytes, SvecLayout &L) {
  const u32 kl = k / 2, kh = k - kl;
  const size_t ntab = ((size_t)1 << kl) + ((size_t)1 << kh);
  const size_t o_sb = align_up(32 * n, 256), o_xt = o_sb + align_up(32 * n, 256), o_tab = o_xt + align_up(64 * (size_t)(k ? k : 1), 256),
               o_ex = o_tab + align_up(64 * ntab, 256);
  int rc = ensure_stage_in(ctx, o_ex + extra_bytes + 512);
  if (rc) return rc;
  char *base = (char *)ctx->stage_in;
  L.sa = (u32 *)base; L.sb = (u32 *)(base + o_sb); L.xt = (u32 *)(base + o_xt); L.tab = (u32 *)(base + o_tab); L.extra = base + o_ex;
  L.k = k; L.kl = kl;
  return BPMI_OK;
}
static int svector_launch(bpmi_ctx *ctx, const SvecLayout &L, const void *d_scale, uint64_t n, const uint8_t *xs, const uint8_t *xinvs,
                          const uint8_t a[32], const uint8_t b[32]) {
  const u32 k = L.k;
  const size_t ntab = ((size_t)1 << L.kl) + ((size_t)1 << (k - L.kl));
  std::vector<uint8_t> xt(64 * (size_t)(k ? k : 1));
  for (u32 j = 0; j < k; j++) { memcpy(&xt[64 * j], xs + 32 * j, 32); memcpy(&xt[64 * j + 32], xinvs + 32 * j, 32); }
  HIPCHK(ctx, h2d(ctx, L.xt, xt.data(), xt.size(), ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));          // xt is owned by this frame
  Sc2 ab;
  memcpy(ab.k1, a, 32); memcpy(ab.k2, b, 32);
  {
    StageTimer t(ctx, ST_SCFOLD);
    hipLaunchKernelGGL(k_sc_svector_tables, dim3((u32)((ntab + 255) / 256)), dim3(256), 0, ctx->stream, L.xt, k, L.kl, ab, L.tab);
    hipLaunchKernelGGL(k_sc_svector, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, L.tab, k, L.kl, (const u32 *)d_scale, (u32)n, L.sa, L.sb);
  }
  HIPCHK(ctx, hipGetLastError());
  return BPMI_OK;
}
static bool log2_exact(uint64_t n, u32 &k) { k = 0; while ((1ull << k) < n) k++; return n && (1ull << k) == n; }

int bpmi_sc_svector(bpmi_ctx *ctx, const uint8_t *xs, const uint8_t *xinvs, uint32_t k, const uint8_t a[32], const uint8_t b[32],
                    const uint8_t *scale, uint8_t *sa, uint8_t *sb) {
  if (!ctx || !a || !b || !sa || !sb || (k && (!xs || !xinvs))) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (k > 24) return fail(ctx, BPMI_E_ARG, "k must be <= 24");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t n = 1ull << k;
  SvecLayout L;
  int rc = svector_layout(ctx, n, k, scale ? 32 * n : 0, L);
  if (rc) return rc;
  if (scale) HIPCHK(ctx, h2d(ctx, L.extra, scale, 32 * n, ctx->stream));
  rc = svector_launch(ctx, L, scale ? L.extra : nullptr, n, xs, xinvs, a, b);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpyAsync(sa, L.sa, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(sb, L.sb, 32 * n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

int bpmi_ipa_verify_dev(bpmi_ctx *ctx, const void *d_g, const void *d_h, const void *d_hscale, uint64_t n, const uint8_t *xs, const uint8_t *xinvs,
                        uint32_t k, const uint8_t a[32], const uint8_t b[32], const uint8_t *extra_pts, const uint8_t *extra_scalars,
                        uint64_t n_extra, uint8_t out[64]) {
  if (!ctx || !d_g || !d_h || !a || !b || !out || (k && (!xs || !xinvs)) || (n_extra && (!extra_pts || !extra_scalars)))
    return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  u32 kk;
  if (!log2_exact(n, kk) || kk != k || n > (1ull << 22)) return fail(ctx, BPMI_E_ARG, "n must be 2^k, k <= 22");
  if (n_extra > (1u << 20)) return fail(ctx, BPMI_E_ARG, "too many extra points");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t o_es = align_up(64 * n_extra, 256);
  SvecLayout L;
  int rc = svector_layout(ctx, n, k, o_es + 32 * n_extra, L);
  if (rc) return rc;
  const bool check_host = ctx->opt_validate >= 1 && n_extra && n_extra <= VALIDATE_HOST_MAX;
  const bool check_dev = ctx->opt_validate >= 2 || (ctx->opt_validate >= 1 && n_extra > VALIDATE_HOST_MAX);
  if (check_host) { rc = validate_host(ctx, extra_pts, n_extra, "bpmi_ipa_verify_dev", "extra_pts"); if (rc) return rc; }
  if (check_dev) { rc = validate_begin(ctx, ctx->stream); if (rc) return rc; }
  rc = svector_launch(ctx, L, d_hscale, n, xs, xinvs, a, b);
  if (rc) return rc;
  u32 *d_sa = L.sa, *d_sb = L.sb;
  char *d_ex = L.extra;
  if (n_extra) {
    HIPCHK(ctx, h2d(ctx, d_ex, extra_pts, 64 * n_extra, ctx->stream));
    HIPCHK(ctx, h2d(ctx, d_ex + o_es, extra_scalars, 32 * n_extra, ctx->stream));
  }
  Segs s = segs_init();
  s.pts[0] = (const u32 *)d_g; s.sc[0] = d_sa; s.n[0] = (u32)n;
  s.pts[1] = (const u32 *)d_h; s.sc[1] = d_sb; s.n[1] = (u32)n;
  s.pts[2] = (const u32 *)d_ex; s.sc[2] = (const u32 *)(d_ex + o_es); s.n[2] = (u32)n_extra;
  s.total = (u32)(2 * n + n_extra);
  if (check_dev) {
    if (ctx->opt_validate >= 2) { validate_enqueue(ctx, d_g, n, 0, ctx->stream); validate_enqueue(ctx, d_h, n, 1, ctx->stream); }
    if (n_extra > VALIDATE_HOST_MAX) validate_enqueue(ctx, d_ex, n_extra, 2, ctx->stream);
    rc = validate_fetch(ctx, ctx->stream);
    if (rc) return rc;
  }
  rc = msm_run(ctx, s, out);
  if (rc || !check_dev) return rc;
  static const char *const names[] = {"d_g", "d_h", "extra_pts"};
  rc = validate_end(ctx, "bpmi_ipa_verify_dev", names);
  if (rc) memset(out, 0xFF, 64);                // (never the identity, which a verifier reads as "valid")
  return rc;
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
  u32 *hscale_buf;      // device, room for n0 scalars
  u32 *hscale;          // device, optional per-base scale of the h generators (n0 scalars) or nullptr
  uint64_t small_m;     // logical length at which bases below big_m are folded through products (0: never)
  void *block;          // one allocation
  std::vector<sc> hcg, hch;   // host copies of the coefficient tables while 2^d <= 16
  bool lr_done;
  bool prep_ready;      // the scalars of the next L / R and c_L, c_R are on the device already (k_ipa_small_step did the next round's preparation)
  const void *src_g = nullptr, *src_h = nullptr;      // bpmi_ipa_create_dev: the caller's generator arrays (the key of the kept fold tables, option ipa_fixed_generators)
};

// deferral policy: bases of 2^18 points or more are folded 16-way at once (an MSM over the
// unfolded bases costs ~2.6 ms per L/R at 2^20 against ~50 ms for the pairwise ladder fold,
// while a K-term ladder stays throughput-bound only for many outputs); smaller bases are
// never folded -- the prover needs L and R, not the folded generators, and a ladder launch
// is ~2 ms of pure latency.
#define IPA_BIG_M_DEFAULT (1u << 18)
#define IPA_BIG_D 4
// Round 4: bases BELOW that threshold (the 2^16 points per side a 2^20-element proof is left with after its 16-way fold; the 2^14
// generators of an aggregated 128 x 64-bit range proof) are folded ONCE more, when the logical length reaches 4 096, through
// per-term products (k_ipa_fold_scalars, bpmi_ec_mul_batch, k_ec_sum_strided: point_kernels.hpp) -- from there on L and R are MSMs
// over <= 4 097 pairs on the one-launch small-MSM kernel, both in one launch.  Option "ipa_small_m": 0 default, 1 never, else the length.
#define IPA_SMALL_M_DEFAULT 4096u

extern "C" {

static int ipa_alloc(bpmi_ctx *ctx, uint64_t n, bpmi_ipa **out) {
  bpmi_ipa *st = new bpmi_ipa();
  st->ctx = ctx; st->n0 = st->n = st->M = n; st->d = 0; st->cur = 0; st->lr_done = false; st->prep_ready = false;
  st->big_m = ctx->opt_ipa_big > 0 ? (uint64_t)ctx->opt_ipa_big : IPA_BIG_M_DEFAULT;
  if (st->big_m < 32) st->big_m = 32;
  st->small_m = ctx->opt_ipa_small == 1 ? 0 : (ctx->opt_ipa_small > 1 ? (uint64_t)ctx->opt_ipa_small : IPA_SMALL_M_DEFAULT);
  const size_t pts = align_up(64 * n, 256), scs = align_up(32 * n, 256);
  // the targets of a fold: n / 16 points (the ladder's 16-way fold) or small_m points (the product fold of bases that were never folded)
  const size_t pts2 = align_up(64 * std::max<uint64_t>(n / 16 + 1, std::min<uint64_t>(n, st->small_m)), 256), coef = align_up(32 * n, 256);
  const size_t bytes = pts * 2 + pts2 * 2 + scs * 7 + coef * 4 + 256 * 3 + 2 * 32 * SC_DOT_MAX_BLOCKS + 2 * align_up(sizeof(NafK), 256);
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
  st->hscale = nullptr;
  u32 *hscale_buf = (u32 *)p; p += scs;
  for (int k = 0; k < 2; k++) { st->cg[k] = (u32 *)p; p += coef; st->ch[k] = (u32 *)p; p += coef; }
  st->u = (u32 *)p; p += 256;
  st->cl = (u32 *)p; p += 256;
  st->cr = (u32 *)p; p += 256;
  st->partial = (u32 *)p; p += 2 * 32 * SC_DOT_MAX_BLOCKS;
  for (int k = 0; k < 2; k++) { st->nafk[k] = (NafK *)p; p += align_up(sizeof(NafK), 256); }
  // coefficient tables start as [1]
  uint8_t one[32] = {1};
  e = h2d(ctx, st->cg[0], one, 32, ctx->stream);
  if (e == hipSuccess) e = h2d(ctx, st->ch[0], one, 32, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { (void)hipFree(st->block); delete st; return fail(ctx, BPMI_E_HIP, std::string("ipa_alloc: ") + hipGetErrorString(e)); }
  sc o; memset(&o, 0, sizeof(o)); o.v[0] = 1;
  st->hcg.assign(1, o); st->hch.assign(1, o);
  st->hscale_buf = hscale_buf;
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
  if (e == hipSuccess) e = h2d(ctx, st->u, u, 64, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { (void)hipFree(st->block); delete st; return fail(ctx, BPMI_E_HIP, std::string("ipa_create copy: ") + hipGetErrorString(e)); }
  st->src_g = d_g; st->src_h = d_h;
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
  const bool check = ctx->opt_validate >= 1;
  if (check) {
    rc = validate_host(ctx, u, 1, "bpmi_ipa_create", "u");
    if (rc == BPMI_OK) rc = validate_begin(ctx, s);
    if (rc) { (void)hipFree(st->block); delete st; return rc; }
  }
  hipError_t e = h2d(ctx, st->g, g, 64 * n, s);
  if (e == hipSuccess) e = h2d(ctx, st->h, h, 64 * n, s);
  if (e == hipSuccess) e = h2d(ctx, st->a, a, 32 * n, s);
  if (e == hipSuccess) e = h2d(ctx, st->b, b, 32 * n, s);
  if (e == hipSuccess) e = h2d(ctx, st->u, u, 64, s);
  if (e == hipSuccess && check) {
    validate_enqueue(ctx, st->g, n, 0, s);
    validate_enqueue(ctx, st->h, n, 1, s);
    e = hipMemcpyAsync(ctx->vflag, ctx->vflag_dev, 4, hipMemcpyDeviceToHost, s);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) { (void)hipFree(st->block); delete st; return fail(ctx, BPMI_E_HIP, std::string("ipa_create copy: ") + hipGetErrorString(e)); }
  if (check) {
    static const char *const names[] = {"g", "h"};
    rc = validate_end(ctx, "bpmi_ipa_create", names);
    if (rc) { (void)hipFree(st->block); delete st; return rc; }
  }
  *out = st;
  return BPMI_OK;
}
int bpmi_ipa_create_scaled(bpmi_ctx *ctx, const uint8_t *g, const uint8_t *h, const uint8_t *a, const uint8_t *b, uint64_t n,
                           const uint8_t u[64], const uint8_t *h_scale, bpmi_ipa **out) {
  int rc = bpmi_ipa_create(ctx, g, h, a, b, n, u, out);
  if (rc || !h_scale) return rc;
  bpmi_ipa *st = *out;
  if (n >= st->big_m) {
    // large bases get folded by the shared-scalar ladder, which needs real points: scale them once
    HIPCHK(ctx, h2d(ctx, st->hscale_buf, h_scale, 32 * n, ctx->stream));
    rc = bpmi_ec_mul_batch_dev(ctx, st->h, st->hscale_buf, n, st->h);
    if (rc == BPMI_OK) { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) rc = fail(ctx, BPMI_E_HIP, hipGetErrorString(e)); }
  } else {
    // never folded: the factors ride in the scalars of every L / R MSM
    hipError_t e = h2d(ctx, st->hscale_buf, h_scale, 32 * n, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(ctx, BPMI_E_HIP, std::string("ipa_create_scaled: ") + hipGetErrorString(e));
    else st->hscale = st->hscale_buf;
  }
  if (rc) { bpmi_ipa_destroy(st); *out = nullptr; }
  return rc;
}
uint64_t bpmi_ipa_len(const bpmi_ipa *st) { return st ? st->n : 0; }

int bpmi_ipa_round_LR(bpmi_ipa *st, uint8_t L[64], uint8_t R[64]) {
  if (!st || !L || !R) return BPMI_E_ARG;
  bpmi_ctx *ctx = st->ctx;
  if (st->n < 2) return fail(ctx, BPMI_E_STATE, "ipa already reduced to length 1");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t np = st->n / 2;
  u32 *a_lo = st->a, *a_hi = st->a + 8 * np, *b_lo = st->b, *b_hi = st->b + 8 * np;
  const bool prepared = st->prep_ready;          // k_ipa_small_step left c_L, c_R and the expanded scalars of this round on the device
  st->prep_ready = false;
  // cl = <a_lo, b_hi>, cr = <a_hi, b_lo>  (inner_product_prover.py:96-97), kept on the device
  if (!prepared) {
    DotJobs dj;
    dj.a[0] = a_lo; dj.b[0] = b_hi; dj.out[0] = st->cl;
    dj.a[1] = a_hi; dj.b[1] = b_lo; dj.out[1] = st->cr;
    sc_dot_jobs(ctx, dj, 2, np, st->partial);                 // both in one pair of launches
  }
  int rc;
  if (st->d == 0 && !st->hscale) {
    // bases are the current generators: L = <a_lo, g_hi> + <b_hi, h_lo> + cl*u  (:98) as ONE
    // three-segment MSM, R = <a_hi, g_lo> + <b_lo, h_hi> + cr*u  (:99)
    u32 *g_lo = st->g, *g_hi = st->g + 16 * np, *h_lo = st->h, *h_hi = st->h + 16 * np;
    Segs sL = segs_init();
    sL.pts[0] = g_hi; sL.sc[0] = a_lo; sL.n[0] = (u32)np;
    sL.pts[1] = h_lo; sL.sc[1] = b_hi; sL.n[1] = (u32)np;
    sL.pts[2] = st->u; sL.sc[2] = st->cl; sL.n[2] = 1;
    sL.total = (u32)(2 * np + 1);
    Segs sR = segs_init();
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
    if (!prepared) {
      StageTimer t(ctx, ST_SCFOLD);
      ExpandOut eo;
      eo.eg[0] = st->eg; eo.eh[0] = st->eh; eo.eg[1] = st->eg2; eo.eh[1] = st->eh2;
      hipLaunchKernelGGL(k_ipa_expand, dim3((u32)((st->M + 255) / 256), 2), dim3(256), 0, ctx->stream, st->a, st->b, st->cg[st->cur], st->ch[st->cur],
                         st->hscale, (u32)st->M, logm, eo);           // the scalars of L and of R in one launch
    }
    for (int right = 0; right < 2; right++) {
      u32 *eg = right ? st->eg2 : st->eg, *eh = right ? st->eh2 : st->eh;
      // only the non-zero half of every block of m logical positions takes part: for L the
      // upper g-halves and lower h-halves (g_hi with a_lo, h_lo with b_hi), for R the opposite
      sg[right] = segs_init();
      sg[right].pts[0] = st->g; sg[right].sc[0] = eg; sg[right].n[0] = (u32)(st->M / 2);
      sg[right].pts[1] = st->h; sg[right].sc[1] = eh; sg[right].n[1] = (u32)(st->M / 2);
      sg[right].hlog[0] = sg[right].hlog[1] = logm - 1;
      sg[right].phase[0] = right ? 0u : 1u;
      sg[right].phase[1] = right ? 1u : 0u;
      sg[right].pts[2] = st->u; sg[right].sc[2] = right ? st->cr : st->cl; sg[right].n[2] = 1;
      sg[right].total = (u32)(st->M + 1);
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
  st->prep_ready = false;
  // short vectors: the fold of a and b, the coefficient tables AND the next round's c_L, c_R and expanded scalars in one launch
  // (only where no generator fold can come any more: not above the ladder's threshold, not above the product fold's length)
  const bool small_step = ctx->opt_ipa_step && st->M <= 4096 && st->M == st->n << st->d && st->M < st->big_m && !(st->small_m && st->M > st->small_m);
  if (small_step) {
    const u32 K = 1u << st->d;
    IpaStep ps;
    ps.a = st->a; ps.b = st->b;
    ps.cg = st->cg[st->cur]; ps.ch = st->ch[st->cur]; ps.cg2 = st->cg[st->cur ^ 1]; ps.ch2 = st->ch[st->cur ^ 1];
    memcpy(ps.x_xinv.k1, x, 32); memcpy(ps.x_xinv.k2, xinv, 32);
    ps.np = (u32)np; ps.K = K; ps.M = (u32)st->M;
    ps.hscale = st->hscale;
    ps.cl = st->cl; ps.cr = st->cr;
    ps.eg[0] = st->eg; ps.eh[0] = st->eh; ps.eg[1] = st->eg2; ps.eh[1] = st->eh2;
    {
      StageTimer t(ctx, ST_SCFOLD);
      hipLaunchKernelGGL(k_ipa_small_step, dim3(1), dim3(1024), 0, ctx->stream, ps);
    }
    HIPCHK(ctx, hipGetLastError());
    st->cur ^= 1;
    st->d += 1;
    st->n = np;
    st->lr_done = false;
    st->prep_ready = np >= 2;
    return BPMI_OK;
  }
  // a' = x a_lo + x^-1 a_hi ; b' = x^-1 b_lo + x b_hi  (:109-110)
  {
    FoldJobs fj;
    fj.lo[0] = st->a; fj.hi[0] = st->a + 8 * np; fj.out[0] = st->a;
    fj.lo[1] = st->b; fj.hi[1] = st->b + 8 * np; fj.out[1] = st->b;
    memcpy(fj.xy[0].k1, x, 32); memcpy(fj.xy[0].k2, xinv, 32);
    memcpy(fj.xy[1].k1, xinv, 32); memcpy(fj.xy[1].k2, x, 32);
    StageTimer t(ctx, ST_SCFOLD);
    hipLaunchKernelGGL(k_sc_fold, dim3((u32)((np + 255) / 256), 2), dim3(256), 0, ctx->stream, fj, (u32)np);       // a and b in one launch
  }
  (void)rc;
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
  const bool track_host = ((2 * K <= MULTIFOLD_MAXK) && st->M >= st->big_m) ||
                          (2 * K <= GLVF_MAXK && st->small_m && st->M < st->big_m && st->M > st->small_m && !st->hscale && st->hcg.size() == K);
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
    MultifoldJob ja = {st->g, st->g2}, jb = {st->h, st->h2};
    const uint64_t npts = st->M;
    // width-4 NAF over affine tables of 3P, 5P, 7P: 2 x 216 B + 2 x 432 B of scratch per base point (1.4 GB at 2^20), allocated
    // HERE, when a fold is actually reached, and kept by the ctx for the next proof (round 3 allocated them with every state,
    // folded or not); if they do not fit the fold uses the plain NAF ladder
    void *wtab = nullptr;
    const uint64_t nthr = (npts + ODDMUL_PER_THREAD - 1) / ODDMUL_PER_THREAD;
    const bool glv = ctx->opt_fold_wnaf >= 2 && st->n % 64 == 0;          // k_ec_multifold_w4g reads its digits per WAVE: a wave must not straddle g and h
    const size_t tab_bytes = align_up(3ull * npts * 72, 256), scr_bytes = align_up(2ull * nthr * ODDMUL_PER_THREAD * 3 * 144, 256),
                 wn_bytes = align_up(glv ? sizeof(WnafG) : sizeof(WnafK), 256), tabx_bytes = glv ? align_up(4ull * npts * 36, 256) : 0;
    if (ctx->opt_fold_wnaf && npts >= 256) {
      const size_t need = 2 * tab_bytes + scr_bytes + 2 * wn_bytes + 2 * tabx_bytes;
      if (need > ctx->fold_tab_bytes) {
        ctx->fold_key_g = ctx->fold_key_h = nullptr;
        if (ctx->fold_tab) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->fold_tab); ctx->fold_tab = nullptr; ctx->fold_tab_bytes = 0; }
        if (hipMalloc(&ctx->fold_tab, need) == hipSuccess) ctx->fold_tab_bytes = need; else { (void)hipGetLastError(); ctx->fold_tab = nullptr; }
      }
      wtab = ctx->fold_tab;
    }
    if (wtab) {
      u32 *tab_a = (u32 *)wtab, *tab_b = (u32 *)((char *)wtab + tab_bytes), *scr = (u32 *)((char *)wtab + 2 * tab_bytes);
      char *dw = (char *)wtab + 2 * tab_bytes + scr_bytes;
      u32 *tabx_a = (u32 *)(dw + 2 * wn_bytes), *tabx_b = (u32 *)(dw + 2 * wn_bytes + tabx_bytes);
      if (glv) {
        // the coefficients in two 128-bit halves each (k = k1 + k2 lambda): row 2 t + half, a negative half with its digits negated
        // the coefficients in two 128-bit halves each, as the ladder's operation list (fold_ops_host.hpp)
        static thread_local WnafG hga, hgb;
        if (!glv_fold_ops(hga, st->hcg.data(), K2) || !glv_fold_ops(hgb, st->hch.data(), K2))
          return fail(ctx, BPMI_E_STATE, "fold: operation list overflow");
        HIPCHK(ctx, h2d(ctx, dw, &hga, sizeof(WnafG), ctx->stream));
        HIPCHK(ctx, h2d(ctx, dw + wn_bytes, &hgb, sizeof(WnafG), ctx->stream));
        {
          StageTimer t(ctx, ST_LINCOMB2);
          // Option ipa_fixed_generators: the generators of a deployment are constants, and so are the tables of their odd multiples
          // (3P, 5P, 7P and the beta x column: 1.1 ms of k_ec_odd_multiples at 2^20).  They are kept between proofs that name the SAME
          // caller arrays (bpmi_ipa_create_dev: d_g, d_h, n) -- the caller's promise that the arrays have not changed.
          // Only a fold whose input IS the caller's unfolded arrays (the first one: M == n0) may keep or reuse tables: a second 16-way fold (n0 >= 16
          // big_m) runs over the already folded, challenge-dependent bases, and tables recorded under (src_g, src_h, n0 / 16) would be served to a later
          // proof over a PREFIX of the same generator arrays.
          const bool from_source = ctx->opt_ipa_fixed && st->src_g && !st->hscale && st->M == st->n0;
          const bool kept = from_source && ctx->fold_key_g == st->src_g && ctx->fold_key_h == st->src_h && ctx->fold_key_n == npts;
          if (!kept) {
            hipLaunchKernelGGL(k_ec_odd_multiples<ODDMUL_PER_THREAD>, dim3((u32)((2 * nthr + 255) / 256)), dim3(256), 0, ctx->stream, st->g, st->h, (u32)npts, tab_a, tab_b, scr,
                               tabx_a, tabx_b);
            if (from_source) { ctx->fold_key_g = st->src_g; ctx->fold_key_h = st->src_h; ctx->fold_key_n = npts; }
            else ctx->fold_key_g = ctx->fold_key_h = nullptr;
          }
          hipLaunchKernelGGL(k_ec_multifold_w4g, dim3((u32)((2 * st->n + 255) / 256)), dim3(256), 0, ctx->stream, ja, jb, tab_a, tab_b, tabx_a, tabx_b,
                             (const WnafG *)dw, (const WnafG *)(dw + wn_bytes), (u32)st->n, K2);
        }
      } else {
        WnafK *dwa = (WnafK *)dw, *dwb = (WnafK *)(dw + wn_bytes);
        static thread_local WnafK hwa, hwb;
        memset(&hwa, 0, sizeof(hwa)); memset(&hwb, 0, sizeof(hwb));
        hwa.top = hwb.top = -1;
        for (u32 t = 0; t < K2; t++) {
          host_wnaf4((const uint8_t *)st->hcg[t].v, hwa.dg[t], hwa.top);
          host_wnaf4((const uint8_t *)st->hch[t].v, hwb.dg[t], hwb.top);
        }
        HIPCHK(ctx, h2d(ctx, dwa, &hwa, sizeof(WnafK), ctx->stream));
        HIPCHK(ctx, h2d(ctx, dwb, &hwb, sizeof(WnafK), ctx->stream));
        {
          StageTimer t(ctx, ST_LINCOMB2);
          ctx->fold_key_g = ctx->fold_key_h = nullptr;
          hipLaunchKernelGGL(k_ec_odd_multiples<ODDMUL_PER_THREAD>, dim3((u32)((2 * nthr + 255) / 256)), dim3(256), 0, ctx->stream, st->g, st->h, (u32)npts, tab_a, tab_b, scr,
                             (u32 *)nullptr, (u32 *)nullptr);
          hipLaunchKernelGGL(k_ec_multifold_w4, dim3((u32)((2 * st->n + 255) / 256)), dim3(256), 0, ctx->stream, ja, jb, tab_a, tab_b, dwa, dwb,
                             (u32)st->n, K2);
        }
      }
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // the host digit tables are thread-local statics
    }
    if (!wtab) {
      NafK ha, hb;
      memset(&ha, 0, sizeof(ha)); memset(&hb, 0, sizeof(hb));
      ha.top = hb.top = -1;
      for (u32 t = 0; t < K2; t++) {
        host_naf((const uint8_t *)st->hcg[t].v, ha.nz[t], ha.sg[t], ha.top);
        host_naf((const uint8_t *)st->hch[t].v, hb.nz[t], hb.sg[t], hb.top);
      }
      HIPCHK(ctx, h2d(ctx, st->nafk[0], &ha, sizeof(NafK), ctx->stream));
      HIPCHK(ctx, h2d(ctx, st->nafk[1], &hb, sizeof(NafK), ctx->stream));
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));     // ha / hb are stack objects
      {
        StageTimer t(ctx, ST_LINCOMB2);
        hipLaunchKernelGGL(k_ec_multifold, dim3((u32)((2 * st->n + 255) / 256)), dim3(256), 0, ctx->stream, ja, jb, st->nafk[0], st->nafk[1],
                           (u32)st->n, K2);
      }
    }
    HIPCHK(ctx, hipGetLastError());
    // the folded generators become the new bases
    std::swap(st->g, st->g2);
    std::swap(st->h, st->h2);
    st->M = st->n;
    st->d = 0;
    uint8_t one[32] = {1};
    HIPCHK(ctx, h2d(ctx, st->cg[st->cur], one, 32, ctx->stream));
    HIPCHK(ctx, h2d(ctx, st->ch[st->cur], one, 32, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    sc o; memset(&o, 0, sizeof(o)); o.v[0] = 1;
    st->hcg.assign(1, o); st->hch.assign(1, o);
  } else if (st->small_m && st->M < st->big_m && st->n == st->small_m && st->M > st->n && st->n > 1 &&
             (2 * st->M >= (uint64_t)MULB_MIN_N || ctx->opt_ipa_small > 1)) {
    // the product fold (point_kernels.hpp): out[i] = sum_t coef[t] (hscale) base[i + t m] with one thread per TERM
    const uint64_t M = st->M, m = st->n;
    const u32 K = (u32)(M / m);
    u32 logm = 0;
    while ((1ull << logm) < m) logm++;
    if (!st->hscale && st->hcg.size() == K && K % GLVF_TERMS == 0 && K <= GLVF_MAXK && m % 64 == 0 && ctx->opt_fold_shared) {
      // shared coefficients: GLV halves in non-adjacent form from the host, two terms per thread (k_ec_fold_glv)
      static thread_local GlvFoldK hk;
      memset(&hk, 0, sizeof(hk));
      hk.top = -1;
      for (int side = 0; side < 2; side++)
        for (u32 t = 0; t < K; t++) {
          const sc &cf = side ? st->hch[t] : st->hcg[t];
          u32 k1[4], k2[4];
          bool n1, n2;
          glv_split(k1, n1, k2, n2, cf);
          for (int hf = 0; hf < 2; hf++) {
            uint8_t k32[32] = {0};
            memcpy(k32, hf ? k2 : k1, 16);
            u32 nz[9], sg[9];
            host_naf(k32, nz, sg, hk.top);
            const bool neg = hf ? n2 : n1;
            for (int wd = 0; wd < 5; wd++) { hk.nz[side][2 * t + hf][wd] = nz[wd]; hk.sg[side][2 * t + hf][wd] = neg ? (nz[wd] & ~sg[wd]) : sg[wd]; }
          }
        }
      const u32 G = K / GLVF_TERMS;
      const size_t o_part = align_up(sizeof(GlvFoldK), 256);
      rc = ensure_stage_in(ctx, o_part + 4ull * XYZZ_WORDS * 2 * G * m + 512);
      if (rc) return rc;
      char *buf = (char *)ctx->stage_in;
      HIPCHK(ctx, h2d(ctx, buf, &hk, sizeof(hk), ctx->stream));
      {
        StageTimer t(ctx, ST_LINCOMB2);
        hipLaunchKernelGGL(k_ec_fold_glv, dim3((u32)((2ull * G * m + 255) / 256)), dim3(256), 0, ctx->stream, st->g, st->h, (u32)m, K, (const GlvFoldK *)buf,
                           (u32 *)(buf + o_part));
        hipLaunchKernelGGL(k_ec_sum_partials, dim3((u32)((2 * m + 255) / 256)), dim3(256), 0, ctx->stream, (const u32 *)(buf + o_part), (u32)m, G, st->g2, st->h2);
      }
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));          // the digit table is a thread-local static
    } else {
    const size_t o_sc = align_up(128 * M, 256), o_pr = o_sc + align_up(64 * M, 256);
    rc = ensure_stage_in(ctx, o_pr + 128 * M + 512);
    if (rc) return rc;
    char *buf = (char *)ctx->stage_in;
    u32 *d_pts = (u32 *)buf, *d_sc = (u32 *)(buf + o_sc), *d_prod = (u32 *)(buf + o_pr);
    HIPCHK(ctx, hipMemcpyAsync(d_pts, st->g, 64 * M, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_pts + 16 * M, st->h, 64 * M, hipMemcpyDeviceToDevice, ctx->stream));
    {
      StageTimer t(ctx, ST_SCFOLD);
      hipLaunchKernelGGL(k_ipa_fold_scalars, dim3((u32)((M + 255) / 256)), dim3(256), 0, ctx->stream, st->cg[st->cur], st->ch[st->cur], st->hscale, (u32)M, logm,
                         d_sc, d_sc + 8 * M);
    }
    rc = bpmi_ec_mul_batch_dev(ctx, d_pts, d_sc, 2 * M, d_prod);
    if (rc) return rc;
    {
      StageTimer t(ctx, ST_LINCOMB2);
      hipLaunchKernelGGL(k_ec_sum_strided, dim3((u32)((2 * m + 255) / 256)), dim3(256), 0, ctx->stream, d_prod, (u32)m, K, st->g2, st->h2);
    }
    HIPCHK(ctx, hipGetLastError());
    }
    std::swap(st->g, st->g2);
    std::swap(st->h, st->h2);
    st->M = m;
    st->d = 0;
    st->hscale = nullptr;                  // the factors are in the folded generators now
    static const uint8_t one[32] = {1};
    HIPCHK(ctx, h2d(ctx, st->cg[st->cur], one, 32, ctx->stream));
    HIPCHK(ctx, h2d(ctx, st->ch[st->cur], one, 32, ctx->stream));
    sc o; memset(&o, 0, sizeof(o)); o.v[0] = 1;
    st->hcg.assign(1, o); st->hch.assign(1, o);
  }
  return BPMI_OK;
}

// The whole halving loop of FastNIProver2.prove (/root/reference/src/innerproduct/inner_product_prover.py:94-110) in ONE call, the
// Fiat-Shamir edge included: per round L, R (bpmi_ipa_round_LR), the transcript items of the two points (base64 of the compressed
// point, '&'), the challenge x = mod_hash(transcript, q) (src/utils/utils.py:84-97), its decimal item, the fold (bpmi_ipa_fold with
// x and 1/x).  Byte for byte what utils/transcript.py builds -- the golden proofs pin it -- without a trip through the interpreter
// per round (20 rounds x ~40 us at n = 2^20).  The sharded prover keeps the round-by-round entry points (it exchanges L and R).
//   digest / digest_len   the transcript so far;  digest_out (capacity cap) receives the transcript after the last round
//   xs, Ls, Rs            32 / 64 / 64 bytes per round (little-endian scalars, 64-byte points), max_rounds entries each
int bpmi_ipa_prove_rounds(bpmi_ipa *st, const uint8_t *digest, uint64_t digest_len, uint8_t *digest_out, uint64_t cap, uint64_t *out_len, uint8_t *xs,
                          uint8_t *Ls, uint8_t *Rs, uint32_t max_rounds, uint32_t *rounds) {
  if (!st || (!digest && digest_len) || !digest_out || !out_len || !xs || !Ls || !Rs || !rounds) return BPMI_E_ARG;
  bpmi_ctx *ctx = st->ctx;
  std::vector<uint8_t> dg(digest, digest + digest_len);
  dg.reserve(digest_len + 256 * 24);
  uint32_t r = 0;
  while (st->n > 1) {
    if (r >= max_rounds) return fail(ctx, BPMI_E_ARG, "more rounds than max_rounds");
    uint8_t *L = Ls + 64 * (size_t)r, *R = Rs + 64 * (size_t)r;
    int rc = bpmi_ipa_round_LR(st, L, R);
    if (rc) return rc;
    rpt::append_point(dg, L);
    rpt::append_point(dg, R);
    rp::Sq x, xi;
    rpt::challenge(x, dg);
    rp::q_inv(xi, x);
    uint8_t xb[32], xib[32];
    rp::q_to_le(xb, x); rp::q_to_le(xib, xi);
    memcpy(xs + 32 * (size_t)r, xb, 32);
    rc = bpmi_ipa_fold(st, xb, xib);
    if (rc) return rc;
    r++;
  }
  if (!rpt::export_digest(dg, digest_out, cap, out_len)) return fail(ctx, BPMI_E_ARG, "digest_out too small");
  *rounds = r;
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

#define IPA_EXPORT_MAX 64
int bpmi_ipa_export(bpmi_ipa *st, uint8_t *g, uint8_t *h, uint8_t *a, uint8_t *b) {
  if (!st || !g || !h || !a || !b) return BPMI_E_ARG;
  bpmi_ctx *ctx = st->ctx;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint64_t m = st->n;
  HIPCHK(ctx, hipMemcpyAsync(a, st->a, 32 * m, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(b, st->b, 32 * m, hipMemcpyDeviceToHost, ctx->stream));
  if (st->d == 0 && !st->hscale) {
    // the bases are the current generators
    HIPCHK(ctx, hipMemcpyAsync(g, st->g, 64 * m, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h, st->h, 64 * m, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return BPMI_OK;
  }
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if (m > IPA_EXPORT_MAX) return fail(ctx, BPMI_E_STATE, "ipa export with deferred folds needs a current length <= 64");
  st->prep_ready = false;                 // the export's scalars go through the buffers a prepared round would read
  // deferred folds: every current generator is one MSM over the unfolded bases
  u32 logm = 0;
  while ((1ull << logm) < m) logm++;
  for (u32 pos = 0; pos < (u32)m; pos++) {
    {
      StageTimer t(ctx, ST_SCFOLD);
      hipLaunchKernelGGL(k_ipa_export_scalars, dim3((u32)((st->M + 255) / 256)), dim3(256), 0, ctx->stream, st->cg[st->cur],
                         st->ch[st->cur], st->hscale, (u32)st->M, logm, pos, st->eg, st->eh);
    }
    Segs sg = segs_init(), sh = segs_init();
    sg.pts[0] = st->g; sg.sc[0] = st->eg; sg.n[0] = (u32)st->M; sg.total = (u32)st->M;
    sh.pts[0] = st->h; sh.sc[0] = st->eh; sh.n[0] = (u32)st->M; sh.total = (u32)st->M;
    int rc = msm_run_pair(ctx, sg, g + 64 * pos, sh, h + 64 * pos);
    if (rc) return rc;
  }
  return BPMI_OK;
}

void bpmi_ipa_destroy(bpmi_ipa *st) {
  if (!st) return;
  (void)hipSetDevice(st->ctx->device);
  (void)hipStreamSynchronize(st->ctx->stream);
  (void)hipFree(st->block);
  delete st;
}

// ---- batch verification of range proofs: host-side preparation ---------------------------------------
int bpmi_rp_batch_prepare(uint32_t n_gens, uint32_t values_per_proof, uint64_t n_proofs, const uint8_t *blobs, uint64_t blobs_len, const uint64_t *blob_off,
                          const uint8_t *weights, const uint8_t *seed, int threads, uint8_t *v_scalars, uint8_t *pt_scalars, uint8_t *shared, uint8_t *comp_out,
                          int64_t *first_bad) {
  if (!blobs || !blob_off || (!weights && !seed) || !v_scalars || !pt_scalars || !shared || !first_bad) return BPMI_E_ARG;
  if (n_gens < 2 || (n_gens & (n_gens - 1)) || n_gens > 65536) return BPMI_E_ARG;
  const uint32_t m = values_per_proof;
  if (m < 1 || n_gens % m) return BPMI_E_ARG;
  uint32_t k = 0;
  while ((1u << k) < n_gens) k++;
  *first_bad = -1;
  // the offset table comes from the caller, the proofs from the network: never read outside blobs[0, blobs_len)
  if (blob_off[0] > blobs_len) return BPMI_E_ARG;
  for (uint64_t g = 0; g < n_proofs; g++) if (blob_off[g] > blob_off[g + 1] || blob_off[g + 1] > blobs_len) return BPMI_E_ARG;
  // wire formats 2 and 3 (rp_wire_v2_host.hpp): expanded to format 1 here (format 3's y coordinates checked), then everything below
  // runs as before
  std::vector<uint8_t> expanded;
  std::vector<uint64_t> expanded_off;
  bool any_v2 = false;
  for (uint64_t g = 0; g < n_proofs && !any_v2; g++)
    any_v2 = blob_off[g + 1] >= blob_off[g] + 5 && (blobs[blob_off[g] + 4] == '2' || blobs[blob_off[g] + 4] == '3');
  if (any_v2) {
    expanded_off.assign(n_proofs + 1, 0);
    std::vector<uint8_t> one;
    for (uint64_t g = 0; g < n_proofs; g++) {
      const uint8_t *b = blobs + blob_off[g];
      const size_t len = (size_t)(blob_off[g + 1] - blob_off[g]);
      // a format-1 proof among format-2 ones is taken as it is; so is a blob that claims format 2 and does not expand -- it is not a
      // format-1 proof either, so the checks below reject it AT ITS INDEX, behind any earlier bad proof (returning here at once made
      // the host name a later proof than the device: tools/fuzz_batch_prepare.py, round 5)
      if (len >= 5 && (b[4] == '2' || b[4] == '3') && rpw::expand_v2(b, len, one)) expanded.insert(expanded.end(), one.begin(), one.end());
      else expanded.insert(expanded.end(), b, b + len);
      expanded_off[g + 1] = expanded.size();
    }
    blobs = expanded.data(); blobs_len = expanded.size(); blob_off = expanded_off.data();
  }
  const size_t nacc = 5 + 2 * (size_t)n_gens;
  if (threads < 1) threads = 1;
  if ((uint64_t)threads > n_proofs) threads = n_proofs ? (int)n_proofs : 1;
  std::vector<uint64_t> pt_off(n_proofs + 1);
  for (uint64_t g = 0; g <= n_proofs; g++) pt_off[g] = g * (6 + 2 * (uint64_t)k);
  std::vector<std::vector<rp::Sq>> acc(threads, std::vector<rp::Sq>(nacc, rp::q_small(0)));
  std::vector<uint64_t> bad(threads, UINT64_MAX);
  auto work = [&](int t) {
    const uint64_t lo = n_proofs * t / threads, hi = n_proofs * (t + 1) / threads;
    // sub-chunks bound the scratch memory and keep one modular inversion per ~512 proofs
    for (uint64_t a = lo; a < hi; a += 512) {
      const uint64_t b = a + 512 < hi ? a + 512 : hi;
      uint64_t bd = UINT64_MAX;
      if (!rp::run_chunk(n_gens, k, m, blobs, blob_off, weights, a, b, pt_off.data(), v_scalars, pt_scalars, comp_out, acc[t].data(), &bd, seed)) { bad[t] = bd; return; }
    }
  };
  if (threads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  for (int t = 0; t < threads; t++) if (bad[t] != UINT64_MAX && (*first_bad < 0 || (int64_t)bad[t] < *first_bad)) *first_bad = (int64_t)bad[t];
  for (size_t i = 0; i < nacc; i++) {
    rp::Sq sum = rp::q_small(0);
    for (int t = 0; t < threads; t++) rp::q_add(sum, sum, acc[t][i]);
    rp::q_to_le(shared + 32 * i, sum);
  }
  return BPMI_OK;
}

// The same preparation on the GPU (rp_batch_kernels.hpp): the wire proofs are uploaded (in a few slices, so that the decoding of
// the points of slice c runs on the second lane while slice c + 1 is still on the link), one lane per proof and role parses,
// hashes and checks them, the weighted scalars are written straight into the caller's device scalar arrays, the proofs' points
// are decoded where they lie in the blobs into d_points, and only the (5 + 2n) shared coefficients and the verdict come back.
struct RpQueued { u32 *d_shared; unsigned long long *d_bad; u32 ncols; };
#define RP_UPLOAD_SLICES 4
// A batch is read in the wire format of its FIRST proof.  A well-formed proof of the OTHER format inside it is not a forged proof: the
// device paths report it as an argument error ("mixed wire formats"), not as a verdict -- a verifier must be able to tell a
// sender's mix-up from an attack (the host path, bpmi_rp_batch_prepare, takes the formats proof by proof).
static int rp_mixed_formats(bpmi_ctx *ctx, const uint8_t *blobs, uint64_t blobs_len, const uint64_t *blob_off, int64_t first_bad) {
  if (first_bad < 0) return BPMI_OK;
  const uint64_t a0 = blob_off[0], a = blob_off[first_bad], e = blob_off[first_bad + 1];
  if (a0 + 5 > blobs_len || e > blobs_len || e < a + 5) return BPMI_OK;
  const uint8_t *b = blobs + a;
  const uint8_t call = (blobs[a0 + 4] == '2' || blobs[a0 + 4] == '3') ? blobs[a0 + 4] : (uint8_t)'1';
  if (!(b[0] == 'B' && b[1] == 'P' && b[2] == 'R' && b[3] == 'P' && b[4] >= '1' && b[4] <= '3' && b[4] != call)) return BPMI_OK;
  // WELL-FORMED in the format it claims?  (a format-1 proof whose magic a flipped bit turned into "BPRP3" is a bad proof, not a mix-up)
  const size_t len = (size_t)(e - a);
  rp::Parsed parsed;
  const bool well_formed = b[4] == '1' ? rp::parse_blob(parsed, b, len) : (len > 0 && rpw::v2_length(b, len) == len);
  if (well_formed)
    return fail(ctx, BPMI_E_ARG, "mixed wire formats: proof " + std::to_string(first_bad) + " is format " + std::string(1, (char)b[4]) + " in a format-" +
                                     std::string(1, (char)call) + " batch (one format per call; bpmi_rp_wire_v2_to_v1 converts)");
  return BPMI_OK;
}
// queues everything on the ctx's two lanes and returns without waiting; the results stay on the device (d_shared: 5 + 2n
// scalars of 8 words, *d_bad behind them).  The caller waits for both lanes whatever this returns.
static int rp_prepare_enqueue(bpmi_ctx *ctx, uint32_t n_gens, uint32_t m, uint64_t n_proofs, const uint8_t *blobs, uint64_t blobs_len, const uint64_t *blob_off,
                              const uint8_t *weights, const uint8_t *seed, void *d_v_scalars, void *d_pt_scalars, void *d_points, RpQueued &Q) {
  if (n_gens < 2 || (n_gens & (n_gens - 1)) || n_gens > 65536) return fail(ctx, BPMI_E_ARG, "n_gens must be a power of two in [2, 65536]");
  if (m < 1 || n_gens % m) return fail(ctx, BPMI_E_ARG, "values_per_proof must divide n_gens");
  if (n_proofs == 0 || n_proofs > (1ull << 22)) return fail(ctx, BPMI_E_ARG, "n_proofs must be in [1, 2^22]");
  if (blobs_len > (1ull << 32)) return fail(ctx, BPMI_E_ARG, "at most 4 GiB of proofs per call");
  uint32_t k = 0;
  while ((1u << k) < n_gens) k++;
  if (blob_off[0] > blobs_len) return fail(ctx, BPMI_E_ARG, "offset table leaves the buffer");
  // wire format 2 (rp_wire_v2_host.hpp): told by the first proof's magic; every proof of the call must then be format 2 (the device
  // expander refuses the others).  The array the roles read holds the EXPANDED proofs: its rows are sized by the longest expansion
  const uint8_t fmt0 = (blob_off[1] >= blob_off[0] + 5 && blob_off[0] + 5 <= blobs_len) ? blobs[blob_off[0] + 4] : (uint8_t)'1';
  const bool v2 = fmt0 == '2' || fmt0 == '3';
  const uint64_t hint_bytes = fmt0 == '3' ? 32ull * (6 + 2 * k) : 0;             // format 3: the points' y coordinates behind the format-2 proof
  // The expander, the roles and the element kernel are chains of dependent instructions, one wave per SIMD; on formats 1 and 2 they run beside
  // the second lane's square roots (k_ec_decompress_wire: every issue slot it can get).  Raised issue priority lets the chains run at their
  // own speed: format 2 alone 1.71-1.73 -> 1.64-1.68 ms, ten in flight +2.4-2.8 % (two boxes, profiles/r06_C5_preparation_priority_ab.txt);
  // format 3 has no square roots beside it and gains nothing (option "rp_priority": 0 off, 1 = formats 1 and 2 (default), 2 = always)
  const u32 rp_prio = (ctx->opt_rp_prio == 2 || (ctx->opt_rp_prio == 1 && fmt0 != '3')) ? 1u : 0u;
  uint64_t maxlen = 0;
  for (uint64_t g = 0; g < n_proofs; g++) {
    if (blob_off[g] > blob_off[g + 1] || blob_off[g + 1] > blobs_len) return fail(ctx, BPMI_E_ARG, "offset table leaves the buffer");
    uint64_t len = blob_off[g + 1] - blob_off[g];
    if (v2) {
      // longest expansion a format-2 proof of this length can have (its seeds' lengths are NOT read here: 2^14 scattered reads of
      // the receive buffer cost more than the upload saves): S seed bytes in all, base64 of them at most three times (the
      // Protocol-1 seed appears in two transcripts), every point item 45 bytes, every decimal item 79
      const uint64_t body = 6 + 32ull * (5 + k) + 33ull * (6 + 2 * k);
      const uint64_t S = len > body + 132 + hint_bytes ? len - body - 132 - hint_bytes : 0;
      len = body + 2 + 12 + 3 * (4 * ((S + 2) / 3) + 1) + 4 * 45 + 3 * 79 + 2 * 79 + 1 + (uint64_t)k * (45 + 45 + 79);
    }
    maxlen = std::max(maxlen, len);
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const u32 P = (u32)n_proofs, ncols = 5 + 2 * n_gens, per = 6 + 2 * k;
  // the proofs as 8-byte words, word-major (k_rp_transpose): W rows of P words, 16 rows of zero padding for loads that run past a proof
  const u32 W = (u32)((std::min<uint64_t>(maxlen, RP_MAX_PROOF_BYTES) + 7) / 8) + 16;
  // device staging: blobs | offsets | weights | status
  const size_t o_off = align_up(blobs_len + 128, 256);      // 128 bytes of slack: the kernel's batched 8-byte loads may run past the last proof
  const size_t o_w = o_off + align_up(8 * ((size_t)P + 1), 256), o_st = o_w + (weights ? align_up(128 * (size_t)P, 256) : 0);
  int rc = ensure_stage_in(ctx, o_st + align_up(RP_ROLES * (size_t)P, 256));
  if (rc) return rc;
  rc = ensure_lane(ctx, 1);
  if (rc) return rc;
  rc = ensure_pin(ctx, 32 * (size_t)ncols + 64);
  if (rc) return rc;
  for (int c = 0; c < RP_UPLOAD_SLICES; c++)
    if (!ctx->ev_slice[c]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_slice[c], hipEventDisableTiming));
  char *din = (char *)ctx->stage_in;
  // contributions + contexts: at most ~256 MB of cells per launch
  const u32 nslots = CTX_SLOTS(k, m);
  const size_t cell_row = 36 * ((size_t)ncols + nslots), out_row = 32 * (size_t)ncols;       // scratch cells are 9 limbs, the result 8 words
  u32 rows = (u32)std::min<size_t>(P, std::max<size_t>(1, ((size_t)256 << 20) / cell_row));
  if (ctx->opt_rp_rows > 0) rows = std::min<u32>(rows, (u32)ctx->opt_rp_rows);
  const size_t o_ctx = align_up(36 * (size_t)ncols * rows, 256), o_shared = o_ctx + align_up(36 * (size_t)nslots * rows, 256),
               o_T = o_shared + align_up(2 * out_row + 256, 256), T_bytes = 8 * (size_t)W * P,      // summed columns | verdict | MSM scalars of the shared generators
               o_lens = o_T + align_up(T_bytes, 256);                                               // format 2: lengths of the expanded proofs
  const size_t need = o_lens + 4 * (size_t)P + 256;
  if (need > ctx->rp_buf_bytes) {
    if (ctx->rp_buf) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(ctx->rp_buf)); ctx->rp_buf = nullptr; ctx->rp_buf_bytes = 0; }
    HIPCHK(ctx, hipMalloc(&ctx->rp_buf, need));
    ctx->rp_buf_bytes = need;
  }
  u32 *d_contrib = (u32 *)ctx->rp_buf, *d_ctx = (u32 *)((char *)ctx->rp_buf + o_ctx), *d_shared = (u32 *)((char *)ctx->rp_buf + o_shared);
  u64 *d_T = (u64 *)((char *)ctx->rp_buf + o_T);
  unsigned long long *d_bad = (unsigned long long *)(d_shared + 8 * (size_t)ncols);
  Q.d_shared = d_shared; Q.d_bad = d_bad; Q.ncols = ncols;
  HIPCHK(ctx, h2d(ctx, din + o_off, blob_off, 8 * ((size_t)P + 1), ctx->stream));
  if (weights) HIPCHK(ctx, h2d(ctx, din + o_w, weights, 128 * (size_t)P, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(d_shared, 0, out_row, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(d_bad, 0xFF, 8, ctx->stream));
  // Upload in slices of whole proofs; the point decoding of a slice (second lane; it reads only the wire bytes) starts as soon as the
  // slice has arrived and runs beside the upload of the next one and, for the last slice, beside the preparation kernels.
  // Option rp_overlap = 0 (measurements only): one decoding launch BEHIND the preparation kernels on the same stream, so that
  // every kernel's duration is its own.
  auto decode = [&](hipStream_t st, u32 g0, u32 g1) {
    StageTimer t(ctx, ST_DECOMP, st);
    const u64 npts = (u64)(g1 - g0) * per;
    hipLaunchKernelGGL(k_ec_decompress_wire, dim3((u32)((npts + 255) / 256)), dim3(256), 0, st, (const uint8_t *)din, (const u64 *)(din + o_off) + g0,
                       k, g1 - g0, (u64)g0, (u32)RP_MAX_PROOF_BYTES, (u32 *)d_points + 16ull * per * g0, d_bad);
  };
  // (format 3's points are checked, not computed: 0.02 ms for 2^14 proofs -- nothing to hide behind an upload, and four uploads with their
  // events cost more than one: a batch alone 1.63 ms against 1.72; the one check still runs on the second lane, beside the expander)
  const u32 nsl = (P >= 4096 && ctx->opt_rp_overlap) ? (ctx->opt_rp_slices > 0 ? (u32)ctx->opt_rp_slices : (fmt0 != '3' ? RP_UPLOAD_SLICES : 1u)) : 1u;
  for (u32 c = 0; c < nsl; c++) {
    const u32 g0 = (u32)((uint64_t)P * c / nsl), g1 = (u32)((uint64_t)P * (c + 1) / nsl);
    const uint64_t b0 = c == 0 ? 0 : blob_off[g0], b1 = c + 1 == nsl ? blobs_len : blob_off[g1];
    if (b1 > b0) HIPCHK(ctx, h2d(ctx, din + b0, blobs + b0, b1 - b0, ctx->stream));
    if (ctx->opt_rp_overlap) {
      HIPCHK(ctx, hipEventRecord(ctx->ev_slice[c], ctx->stream));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->stream1, ctx->ev_slice[c], 0));
      decode(ctx->stream1, g0, g1);
    }
  }
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream1));
  u32 *d_lens = v2 ? (u32 *)((char *)ctx->rp_buf + o_lens) : nullptr;
  if (v2) {
    // the expander writes the format-1 proofs word-major itself; what it does not write must read as zero
    HIPCHK(ctx, hipMemsetAsync(d_T, 0, T_bytes, ctx->stream));
    StageTimer t(ctx, ST_RPPREP);
    hipLaunchKernelGGL(rpd::k_rp_expand_v2, dim3((P + 3) / 4), dim3(64), 0, ctx->stream, (const uint8_t *)din, (const u64 *)(din + o_off), P, k, W, d_T, d_lens,
                       (u32)fmt0, rp_prio);
  } else {
    StageTimer t(ctx, ST_RPPREP);
    hipLaunchKernelGGL(rpd::k_rp_transpose, dim3((P + 63) / 64, (W + 63) / 64), dim3(256), 0, ctx->stream, (const uint8_t *)din, (const u64 *)(din + o_off), P, W, d_T);
  }
  rpd::Params q;
  q.Tstride = P;
  q.weights = weights ? (const uint8_t *)(din + o_w) : nullptr;
  for (int i = 0; i < 8; i++) q.seed[i] = seed ? ((u32)seed[4 * i] << 24) | ((u32)seed[4 * i + 1] << 16) | ((u32)seed[4 * i + 2] << 8) | seed[4 * i + 3] : 0;
  q.n = n_gens; q.k = k; q.m = m; q.Pall = P; q.only_role = ctx->opt_rp_only_role;
  q.contrib = d_contrib;
  q.prio = rp_prio;
  q.ctx = d_ctx;
  q.bad = d_bad;
  const size_t lds_bytes = ((size_t)k + 1) * 9 * 64 * sizeof(u32);            // role 2: k + 1 prefix products of 9 limbs per lane
  u32 lb = 0;
  while ((1u << lb) < n_gens / m) lb++;
  rpd::ElemGeom eg;
  eg.el_log = std::min<u32>(3u, lb);
  eg.ranges = n_gens >> eg.el_log;
  for (u32 base = 0; base < P; base += rows) {
    const u32 cnt = std::min(rows, P - base);
    u32 lanes = (u32)ctx->opt_rp_lanes;
    if (!lanes) lanes = 64;
    q.off = (const u64 *)(din + o_off) + base;
    q.lens = v2 ? d_lens + base : nullptr;
    q.T = d_T + base;
    q.P = cnt; q.lanes = lanes; q.first = base;
    q.v_scalars = (u32 *)d_v_scalars + 8 * (size_t)base * m;
    q.pt_scalars = (u32 *)d_pt_scalars + 8 * (size_t)base * per;
    q.status = (uint8_t *)(din + o_st) + base;
    {
      StageTimer t(ctx, ST_RPPREP);
      hipLaunchKernelGGL(rpd::k_rp_roles, dim3(RP_ROLES * ((cnt + lanes - 1) / lanes)), dim3(64), lds_bytes, ctx->stream, q);
    }
    {
      StageTimer t(ctx, ST_RPELEM);
      hipLaunchKernelGGL(rpd::k_rp_elements, dim3(2 * eg.ranges * ((cnt + 63) / 64)), dim3(64), 0, ctx->stream, q, eg);
      hipLaunchKernelGGL(rpd::k_rp_colsum, dim3(ncols), dim3(256), 0, ctx->stream, (const u32 *)d_contrib, cnt, d_shared);
    }
  }
  if (!ctx->opt_rp_overlap) decode(ctx->stream, 0, P);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
  return BPMI_OK;
}
// both lanes idle again; the first error of (rc, the two waits)
static int rp_wait_lanes(bpmi_ctx *ctx, int rc) {
  const hipError_t e0 = wait_stream(ctx, ctx->stream), e1 = ctx->stream1 ? wait_stream(ctx, ctx->stream1) : hipSuccess;      // (lane 2 never takes part in a batch)
  if (rc) return rc;
  HIPCHK(ctx, e0);
  HIPCHK(ctx, e1);
  return BPMI_OK;
}
int bpmi_rp_batch_prepare_dev(bpmi_ctx *ctx, uint32_t n_gens, uint32_t values_per_proof, uint64_t n_proofs, const uint8_t *blobs, uint64_t blobs_len,
                              const uint64_t *blob_off, const uint8_t *weights, const uint8_t *seed, void *d_v_scalars, void *d_pt_scalars, void *d_points,
                              uint8_t *shared, int64_t *first_bad) {
  if (!ctx) return BPMI_E_ARG;
  if (!blobs || !blob_off || (!weights && !seed) || !d_v_scalars || !d_pt_scalars || !d_points || !shared || !first_bad) return fail(ctx, BPMI_E_ARG, "null argument");
  *first_bad = -1;
  RpQueued Q;
  int rc = rp_prepare_enqueue(ctx, n_gens, values_per_proof, n_proofs, blobs, blobs_len, blob_off, weights, seed, d_v_scalars, d_pt_scalars, d_points, Q);
  const size_t out_row = rc ? 0 : 32 * (size_t)Q.ncols;
  if (!rc && hipMemcpyAsync(ctx->pin, Q.d_shared, out_row + 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, BPMI_E_HIP, "copy of the shared coefficients failed");
  // an error in the middle must not return while the second lane is still writing into the caller's point array
  rc = rp_wait_lanes(ctx, rc);
  if (rc) return rc;
  memcpy(shared, ctx->pin, out_row);
  unsigned long long bad;
  memcpy(&bad, (char *)ctx->pin + out_row, 8);
  *first_bad = bad == ~0ull ? -1 : (int64_t)bad;
  rc = rp_mixed_formats(ctx, blobs, blobs_len, blob_off, *first_bad);
  if (rc) return rc;
  if (ctx->opt_rp_only_role >= 0) *first_bad = 0;        // a profiling run checked part of every proof: it must never read as "all valid"
  return BPMI_OK;
}

// The whole batch verification in ONE call: preparation as above, the shared coefficients folded on the device into the scalars of
// the 3 + 2n shared generators (k_rp_shared_scalars), and the batch's one MSM over [shared generators | commitments | proof
// points] -- no host round trip between the preparation and the MSM.  out = the 64-byte value of the combination (the identity
// for a valid batch; a sharded caller folds the ranks' values), *first_bad as above (then `out` means nothing).
//   v_points  HOST, n_proofs x values_per_proof x 64 B: the commitments, in proof order
//   d_gens    DEVICE, (3 + 2 n_gens) x 64 B: g, h, u, gs, hs (uploaded once per verifier)
//   d_points  DEVICE scratch, (n_proofs (values_per_proof + 6 + 2k)) x 64 B;  d_scalars  DEVICE scratch, the same count x 32 B
int bpmi_rp_batch_verify_dev(bpmi_ctx *ctx, uint32_t n_gens, uint32_t values_per_proof, uint64_t n_proofs, const uint8_t *blobs, uint64_t blobs_len,
                             const uint64_t *blob_off, const uint8_t *weights, const uint8_t *seed, const uint8_t *v_points, const void *d_gens, void *d_points,
                             void *d_scalars, uint8_t out[64], int64_t *first_bad) {
  if (!ctx) return BPMI_E_ARG;
  if (!blobs || !blob_off || (!weights && !seed) || !v_points || !d_gens || !d_points || !d_scalars || !out || !first_bad) return fail(ctx, BPMI_E_ARG, "null argument");
  *first_bad = -1;
  uint32_t k = 0;
  while ((1u << k) < n_gens) k++;
  const uint64_t nv = n_proofs * values_per_proof, npts = n_proofs * (6 + 2 * (uint64_t)k);
  if (3 + 2 * (uint64_t)n_gens + nv + npts > (1ull << 23)) return fail(ctx, BPMI_E_ARG, "at most 2^23 points in the batch's MSM");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // commitments: the first nv points / scalars of the per-proof arrays
  HIPCHK(ctx, h2d(ctx, d_points, v_points, 64 * nv, ctx->stream));
  const bool check = ctx->opt_validate >= 1;
  if (check) {
    int vrc = validate_begin(ctx, ctx->stream);
    if (vrc) return vrc;
    validate_enqueue(ctx, d_points, nv, 0, ctx->stream);
    if (ctx->opt_validate >= 2) validate_enqueue(ctx, d_gens, 3 + 2 * (uint64_t)n_gens, 1, ctx->stream);
    vrc = validate_fetch(ctx, ctx->stream);
    if (vrc) return vrc;
  }
  RpQueued Q;
  int rc = rp_prepare_enqueue(ctx, n_gens, values_per_proof, n_proofs, blobs, blobs_len, blob_off, weights, seed, d_scalars, (char *)d_scalars + 32 * nv,
                              (char *)d_points + 64 * nv, Q);
  u32 *d_fin = nullptr;
  if (!rc) {
    // scalars of g, h, u, gs_i, hs_i: c_g, c_h, c_u, c_gs[i] + gs_const, c_hs[i] + hs_const -- in place behind the raw sums
    d_fin = Q.d_shared + 8 * (size_t)Q.ncols + 32;
    {
      StageTimer t(ctx, ST_RPELEM);
      hipLaunchKernelGGL(rpd::k_rp_shared_scalars, dim3((3 + 2 * n_gens + 255) / 256), dim3(256), 0, ctx->stream, (const u32 *)Q.d_shared, n_gens, d_fin);
    }
    if (hipMemcpyAsync(ctx->pin, Q.d_bad, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, BPMI_E_HIP, "copy of the verdict failed");
  }
  if (rc) return rp_wait_lanes(ctx, rc);
  Segs s = segs_init();
  s.pts[0] = (const u32 *)d_gens; s.sc[0] = d_fin; s.n[0] = 3 + 2 * n_gens;
  s.pts[1] = (const u32 *)d_points; s.sc[1] = (const u32 *)d_scalars; s.n[1] = (u32)(nv + npts);
  s.total = s.n[0] + s.n[1];
  rc = msm_run(ctx, s, out);                      // waits for the MSM (ctx stream: behind everything queued above)
  rc = rp_wait_lanes(ctx, rc);
  if (rc) return rc;
  if (check) {
    static const char *const names[] = {"v_points", "d_gens"};
    rc = validate_end(ctx, "bpmi_rp_batch_verify_dev", names);
    if (rc) { memset(out, 0xFF, 64); return rc; }                  // (never the identity)
  }
  unsigned long long bad;
  memcpy(&bad, ctx->pin, 8);
  *first_bad = bad == ~0ull ? -1 : (int64_t)bad;
  rc = rp_mixed_formats(ctx, blobs, blobs_len, blob_off, *first_bad);
  if (rc) { memset(out, 0xFF, 64); return rc; }
  if (ctx->opt_rp_only_role >= 0) *first_bad = 0;
  return BPMI_OK;
}
// page-locked host memory for buffers that are handed to the library again and again (e.g. the receive buffer of wire proofs:
// uploads from it run at link speed and without a staging copy)
int bpmi_host_alloc(bpmi_ctx *ctx, size_t bytes, void **out) {
  if (!ctx || !out || !bytes) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return BPMI_OK;
}
int bpmi_host_free(bpmi_ctx *ctx, void *p) {
  if (!ctx) return BPMI_E_ARG;
  if (!p) return BPMI_OK;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipHostFree(p));
  return BPMI_OK;
}

// ---- self-test hook ----------------------------------------------------------------------------------------------
int bpmi_debug_fe_op(bpmi_ctx *ctx, int op, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d, uint64_t n, uint32_t *out) {
  if (!ctx || !a || !b || !c || !d || !out) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n == 0) return BPMI_OK;
  if (n > (1u << 22)) return fail(ctx, BPMI_E_ARG, "n too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = 36 * n, stride = align_up(bytes, 256);
  int rc = ensure_stage_in(ctx, 5 * stride + 512);
  if (rc) return rc;
  char *base = (char *)ctx->stage_in;
  const uint32_t *src[4] = {a, b, c, d};
  for (int k = 0; k < 4; k++) HIPCHK(ctx, h2d(ctx, base + k * stride, src[k], bytes, ctx->stream));
  hipLaunchKernelGGL(k_debug_fe_op, dim3((u32)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, (const u32 *)base, (const u32 *)(base + stride),
                     (const u32 *)(base + 2 * stride), (const u32 *)(base + 3 * stride), (u32)n, (u32 *)(base + 4 * stride));
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, base + 4 * stride, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

// the four-lane point addition of the bucket reduction (quad_add, msm_kernels.hpp) on n pairs of 144-byte XYZZ records
int bpmi_debug_quad_add(bpmi_ctx *ctx, const uint32_t *a, const uint32_t *b, uint64_t n, uint32_t *out) {
  if (!ctx || !a || !b || !out) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  if (n == 0) return BPMI_OK;
  if (n > (1u << 20)) return fail(ctx, BPMI_E_ARG, "n too large");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = 4 * XYZZ_WORDS * n, stride = align_up(bytes, 256);
  int rc = ensure_stage_in(ctx, 3 * stride + 512);
  if (rc) return rc;
  char *base = (char *)ctx->stage_in;
  HIPCHK(ctx, h2d(ctx, base, a, bytes, ctx->stream));
  HIPCHK(ctx, h2d(ctx, base + stride, b, bytes, ctx->stream));
  hipLaunchKernelGGL(k_debug_quad_add, dim3((u32)((4 * n + 255) / 256)), dim3(256), 0, ctx->stream, (const u32 *)base, (const u32 *)(base + stride), (u32)n,
                     (u32 *)(base + 2 * stride));
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipMemcpyAsync(out, base + 2 * stride, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return BPMI_OK;
}

// ---- profiling -----------------------------------------------------------------------------------
static void prof_drain(bpmi_ctx *ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  for (auto &e : ctx->evs) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { ctx->prof_ms[e.stage] += ms; ctx->prof_calls[e.stage]++; }
    ctx->ev_pool.push_back(e.a); ctx->ev_pool.push_back(e.b);
  }
  ctx->evs.clear();
}
int bpmi_profile(bpmi_ctx *ctx, int enable) {
  if (!ctx) return BPMI_E_ARG;
  prof_drain(ctx);
  ctx->prof = enable != 0;
  ctx->prof_only = enable == 2 ? ST_ACCUM : -1;
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

#include "rp_prove_host.hpp"
