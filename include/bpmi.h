/*
 * bpmi.h -- C-ABI of libbpmi.so: the MI355X (gfx950) multi-scalar-multiplication +
 * inner-product-argument engine that drops in behind src/pippenger and
 * src/innerproduct of wborgeaud/python-bulletproofs.
 *
 * The reference has no FFI layer of its own: its hot path is a Python call
 * surface that bottoms out in the third-party `fastecdsa` C extension.  Each
 * entry point below names the reference interface it replaces (paths relative to
 * /root/reference); INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross this boundary;
 *   - field elements / scalars: 32 bytes little-endian (= 4 x u64 LE limbs);
 *     affine point: x || y (64 bytes); the identity is 64 zero bytes
 *     ((0,0) is not on y^2 = x^3 + 7);
 *   - scalars of the MSM / scalar-multiplication entry points (bpmi_msm*, bpmi_ec_mul_batch*)
 *     may be any 256-bit value: the kernels reduce them mod q as they load them (the `% order`
 *     of pippenger.py:26; the Python wrapper also does it, with the length check of :23-24).
 *     The mod-q entry points (bpmi_sc_*, the a / b vectors and challenges of bpmi_ipa_*) expect
 *     values in [0, q);
 *   - every function returns 0 on success or a negative BPMI_E_* code, never
 *     throws, never aborts; bpmi_last_error(ctx) gives the message
 *     (ctx == NULL: the message of the last failed bpmi_ctx_create);
 *   - a ctx is for use by one thread at a time; distinct ctxs are independent;
 *   - a ctx, the objects made from it and the library's host worker threads do not survive fork(): a child process creates
 *     its own ctx (the HIP runtime does not survive a fork either);
 *   - points handed in through HOST pointers are checked to be the identity or on the curve (option "validate_points", default 1;
 *     BPMI_E_ARG names the first bad index and no result is written).  Points behind DEVICE pointers are the caller's
 *     responsibility unless the option is 2: each must be 64 zero bytes or (x, y) with x, y < p and y^2 = x^3 + 7; anything else
 *     gives an unspecified (never out-of-bounds) result;
 *   - `*_dev` variants take DEVICE pointers (hipMalloc'd, or torch tensors'
 *     data_ptr()) on the ctx's device and enqueue on the ctx's stream; results
 *     written to host pointers are complete when the call returns;
 *   - there is NO CPU fallback: without a usable gfx950 device bpmi_ctx_create
 *     fails with BPMI_E_NODEVICE.
 */
#ifndef BPMI_H
#define BPMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPMI_OK 0
#define BPMI_E_NODEVICE -1   /* no HIP device / wrong architecture */
#define BPMI_E_HIP -2        /* a HIP runtime call failed (message has the call) */
#define BPMI_E_ARG -3        /* bad argument (NULL pointer, n too large, ...) */
#define BPMI_E_NOMEM -4      /* device or host allocation failed */
#define BPMI_E_STATE -5      /* call not valid in the object's current state */

#define BPMI_MAX_N (1ull << 26) /* largest supported vector length */

typedef struct bpmi_ctx bpmi_ctx;
typedef struct bpmi_ipa bpmi_ipa;
typedef struct bpmi_rp_prover bpmi_rp_prover;

/* ---- library / context ---------------------------------------------------- */
int bpmi_version(void);
/* number of visible HIP devices (0 if none); does not initialise a device */
int bpmi_device_count(void);
/* `stream`: a hipStream_t to enqueue on (e.g. torch.cuda.current_stream().cuda_stream),
 * or NULL for a stream owned by the ctx.  Returns NULL on failure.
 * One ctx drives ONE device.  A multi-GPU caller creates one ctx per device (one process per GPU is
 * what the Python side and bench.py do; a single C process may equally hold several ctxs, one thread
 * each) and folds the 64-byte partial results itself (bpmi_ec_sum): the library has no multi-device
 * entry point and no collective of its own.
 * Host memory a ctx allocates lazily: a 32 MB page-locked staging ring at the first upload of 4 KB .. 8 MB
 * from pageable memory (uploads from bpmi_host_alloc memory bypass it), 16 KB of page-locked result
 * buffer per pending-MSM slot. */
bpmi_ctx *bpmi_ctx_create(int device, void *stream);
void bpmi_ctx_destroy(bpmi_ctx *ctx);
const char *bpmi_last_error(const bpmi_ctx *ctx);
int bpmi_sync(bpmi_ctx *ctx);

/* tuning knobs (0 = automatic):
 *   "window_bits"  MSM window bits c in [2,16]
 *   "chunk"        sorted entries added per thread in the accumulate kernel
 *   "tail"         where the O(256) sequential window-combine tail runs: 1 device kernel,
 *                  2 host thread (default; the result is consumed on the host anyway)
 *   "small_n"      largest n that runs on the one-launch small-MSM kernel (-1: never; default 4608)
 *   "split"        1: run one MSM as two window groups on the ctx's two lanes (default 0)
 *   "async_lanes"  1: slot s of bpmi_msm_dev_enqueue runs on the ctx's lane s (own stream and workspace; three lanes), so
 *                  the tail stages of one MSM overlap the sort / accumulate of the next (inputs must be complete
 *                  before the first enqueue of a burst; default 0)
 *   "fused_scan"   1 (default): the accumulate kernel folds the partial results of the 64 threads of a wave itself (two
 *                  records per wave go to the segmented scan); 0: two records per thread (round 3; kept for A/B runs and tests)
 *   "rp_lanes"     bpmi_rp_batch_prepare_dev: proofs per 64-lane wave (power of two; 0 = 64, the fastest measured)
 *   "rp_only_role" profiling only: 0..3 runs just that role of the preparation kernel (Protocol-2 hash chain | the other
 *                  transcript checks | inversion and factor tables | inverse-free scalars); such a call reports proof 0 as bad
 *                  whatever it saw, so it can never pass for a verification; -1 (default) all four
 *   "rp_overlap"   0 (measurements only): the point decoding runs BEHIND the preparation kernels instead of beside them, and the
 *                  upload is not sliced: every kernel's duration is then its own; 1 default
 *   "glv"          1: the bucket pipeline runs on GLV-split scalars (2n pairs of 128-bit scalars, half the windows).  An
 *                  experiment that lost (profiles/r03_glv_msm_on_off.txt); default 0 = off
 *   "priority"     the MSM's stages around the accumulation raise their waves' issue priority (s_setprio 3): 0 none, 1 (default from round 6) all of
 *                  them, 16 + mask the stages of the mask (1 sort, 2 segmented scan, 4 stage 1 of the bucket reduction, 8 its finish).  Beside round
 *                  3's one-round accumulation it lost (profiles/r03_wave_priority_ab.txt); beside the multi-round one ("rounds") it gains 2-8 %
 *                  with two MSMs in flight (profiles/r06_wave_priority_and_chunk_ab.txt)
 *   "rounds"       where another MSM's kernels run beside an accumulation (the asynchronous pipeline, the slices of a large MSM, a synchronous
 *                  pair from 2^19 pairs) its chunk length is ceil(W n / (64 x 3072 x rounds)), at least 20: `rounds` rounds of three waves per
 *                  SIMD.  0 (default) 3; 1: rounds 2-5's one-round accumulation (86 entries at 2^20).  "pair_rounds" = 1 keeps one round for pairs
 *   "pair_sched"   1: a synchronous pair of MSMs of 2^19 pairs or more as both sorts, then the accumulations one after the other.  Measured
 *                  neutral (profiles/r06_C3_pair_sched_ab.txt); default 0
 *   "accum_stream" / "lane_priority" / "accum_chain"   round-6 experiments on WHERE the pipeline's accumulations are queued (a low-priority stream
 *                  of their own; queue priorities of the lanes; unchained lanes): nothing measurable (profiles/r06_accum_stream_and_chunk_ab.txt); off
 *   "spin_wait"    polls of a completion event before the calling thread sleeps in the runtime (per ctx; default 0)
 *   "mul_batch_glv" bpmi_ec_mul_batch[_dev] from 32 768 points: 1 (default) GLV halves on fixed signed three-bit windows over affine
 *                  3P, 5P, 7P (k_ec_odd_multiples + k_ec_mul_batch_glv; workspace 1 080 B per point of a 196 608-point slice);
 *                  0 the bit-serial ladder at every size
 *   "rp_rows"      bpmi_rp_batch_prepare_dev: proofs per kernel launch (0 = as many as fit ~256 MB of scratch cells)
 *   "ipa_big_m"    base length from which the IPA prover folds its generators 16-way at
 *                  once instead of deferring the fold into the MSM scalars (default 2^18)
 *   "ipa_small_m"  logical length at which bases below "ipa_big_m" are folded once more, through per-term products
 *                  (0 = default 4096, 1 = never, else a power of two; the later rounds then run on the one-launch small-MSM kernel)
 *   "pair_phases"  1: bpmi_msm2 on the bucket pipeline queues both MSMs' sorts before either accumulation.  An experiment that came out
 *                  neutral (profiles/r04_C3_pair_phases_ab.txt); default 0
 *   "prover_table_bits" window bits of the fixed-base tables a bpmi_rp_prover builds, read by bpmi_rp_prover_create: 0 = default 16, else 4 .. 16
 *                  (wider: fewer additions per scalar multiplication, a larger table -- 16 bits: 64 KB x 32 768 entries per (generator, window),
 *                  4.4 GB and 72 ms to build for 64-bit proofs; 12 bits: 378 MB, 17 ms, 22 % slower proving; profiles/r06_batch_prover_table_bits.txt)
 *   "rp_priority"  the batch preparation's chain kernels (expander, roles, elements) raise their waves' issue priority: 0 never, 1 (default) on wire
 *                  formats 1 and 2, where the point decoding's square roots run beside them (format 2: a batch alone 1.72 -> 1.66 ms, ten in
 *                  flight +2.4-2.8 %), 2 always (format 3: nothing; profiles/r06_C5_preparation_priority_ab.txt)
 *   "rp_slices"    uploads of a batch of >= 4 096 proofs in bpmi_rp_batch_prepare_dev / bpmi_rp_batch_verify_dev: 1 .. 4, 0 (default) = 4 for wire
 *                  formats 1 and 2 (the points of a slice are decoded on the second lane while the next slice is on the link), 1 for format 3
 *                  (its points are checked, not computed; profiles/r06_C5_upload_slices_ab.txt)
 *   "prover_wire_format" 2 (default) or 3: the wire format bpmi_rp_prove_batch writes (3: with the points' y coordinates, see
 *                  bpmi_rp_wire_v2_to_v1; 32 (6 + 2k) bytes more per proof -- bpmi_rp_prove_batch_proof_bytes counts them)
 *   "ipa_fixed_generators" 1: the generator arrays handed to bpmi_ipa_create_dev are deployment constants.  The tables of odd multiples that
 *                  the prover's 16-way generator fold builds from them (1.1 ms at n = 2^20) are then kept between proofs for as long as the
 *                  calls name the same d_g, d_h and n: the caller's promise that the arrays were not modified.  Default 0
 *   "validate_points" on-curve check of input points: 0 never, 1 (default) every entry point that takes HOST pointers to points (bpmi_msm,
 *                  bpmi_msm2, bpmi_ec_mul_batch, bpmi_ec_lincomb2_batch, bpmi_ec_sum, bpmi_ipa_create[_scaled], the extra points of
 *                  bpmi_ipa_verify_dev, the commitments of bpmi_rp_batch_verify_dev, bpmi_rp_prover_create), 2 also the synchronous entry
 *                  points that take DEVICE pointers (bpmi_msm_dev, the generators of bpmi_ipa_verify_dev and bpmi_rp_batch_verify_dev).
 *                  What fastecdsa's Point constructor does for the reference (reached from /root/reference/src/utils/utils.py:119-131).
 *                  Cost: one kernel behind the upload, no extra wait (profiles/r05_validate_points_cost.txt)
 *   "slice_n"      an MSM of more than "slice_min" pairs (the synchronous entry points: bpmi_msm[_dev], bpmi_msm_segs_dev, the one MSM of
 *                  bpmi_ipa_verify_dev and of the batch verifier) runs as ceil(n / (slice_n 17/16)) equal slices, two in flight on the ctx's
 *                  lanes 0 / 1 with their accumulations chained, the slices' results added on the host: the engine peaks at ~2^20 pairs per
 *                  MSM (profiles/r06_msm_big_n.txt).  0 (default) 2^20; 2^16 .. 2^23; -1: one MSM up to the sort's 2^23-pair limit
 *   "slice_min"    ... the size from which it does: 0 (default) 1.625 x slice_n, the measured crossover of one MSM against two slices
 *   "mid_parts"    blocks per window of the one-block-per-window kernel (k_msm_mid: MSMs of 1 536 .. 8 448 pairs in the inner-product rounds,
 *                  2 560 .. 8 448 one at a time): 0 (default) three from 3 000 pairs, else one; 1 .. 4 forced.  Every part leaves its own window
 *                  sum, the host tail adds them (profiles/r05_mid_kernel_parts_ab.txt: C4's argument 4.05 -> 3.2 ms)
 *   "mixed_windows" 1 (default): window bits c = 10 .. 14 as 256 / c windows of which the last 256 - (256 / c) c are c + 1 bits wide with twice
 *                  the buckets -- the windows cover the 256 bit positions exactly: no carry window, no short top window (13 bits: 10 + 9
 *                  windows) --, and the window table that goes with it (12 bits above the one-block kernel's 8 448 pairs, 13 from 19 000, 16 from 185 000);
 *                  0: uniform windows and the earlier table
 *   "top_window_unsigned" 1 (default): the same for window_bits = 15 (16 + 1 windows); 0: every width uniform, with its carry window
 *   "sort_inblock" 1 (default): up to 2^17 pairs the sort's second level handles partitions of any size in one block (two launches fewer)
 *   "reduce_fit"   1 (default): stage 1 of the bucket reduction gives every sum as many lanes (any number up to 64, not only powers of two)
 *                  as keep its waves within the chip's 1 024 SIMDs at one wave each; 0: round 4's rule (total buckets / 2^15 elements
 *                  per lane, power-of-two groups).  "reduce_epl" > 0 forces the elements per lane
 *   "final_spread" the finish of the bucket reduction (12 dependent point additions on quads of lanes): 0 one 16-wave block per (window,
 *                  array) (k_digit_final_quad: the 16 waves share one CU, 49 us); 3 (default) the 16 second-level sums of every array as
 *                  waves of their own, in blocks of four, and a second launch of one wave per array for the last 8 additions (18 + 21 us);
 *                  2 the same with one-wave blocks; 1 one launch, the wave that draws an array's last ticket runs the finish (60 us: the
 *                  tickets' device-scope fences cost more than the launch they save; profiles/r05_bucket_reduction_finish_ab.txt)
 *   "segscan_fused" / "hist_scan_fused" 1: the segmented scan's last level / the scan of the sort's partition counts run in the block that
 *                  finishes the level / the histogram kernel last (one launch fewer each).  Experiments that lost to the cost of the
 *                  device-scope fence every block needs (profiles/r05_last_block_fusions_ab.txt); default 0
 *   "fold_wnaf"    the ladder of that 16-way fold: 2 (default) width-4 non-adjacent forms of the coefficients' GLV halves over affine
 *                  tables of 3P, 5P, 7P and of beta x (k_ec_multifold_w4g; 792 B of workspace per generator, kept by the ctx after the first fold),
 *                  1 of the whole coefficients (k_ec_multifold_w4), 0 the plain NAF ladder without tables (k_ec_multifold) */
int bpmi_set_option(bpmi_ctx *ctx, const char *name, int64_t value);

/* ---- device buffers (so callers need no other GPU runtime) ------------------- */
int bpmi_malloc(bpmi_ctx *ctx, size_t bytes, void **dptr);
int bpmi_free(bpmi_ctx *ctx, void *dptr);
int bpmi_upload(bpmi_ctx *ctx, void *dptr, const void *host, size_t bytes);
int bpmi_download(bpmi_ctx *ctx, void *host, const void *dptr, size_t bytes);
/* device -> device, enqueued on the ctx stream (ordered with the ctx's other work) */
int bpmi_memcpy_dev(bpmi_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);

/* ---- multi-scalar multiplication ------------------------------------------------
 * out = sum_i scalars[i] * pts[i].
 * Replaces Pippenger.multiexp on EC(secp256k1) = the PipSECP256k1 singleton
 * (src/pippenger/pippenger.py:22-61, src/pippenger/__init__.py:5; group law
 * src/pippenger/group.py:27-32), i.e. every call site listed in SURVEY.md
 * section 8b: src/utils/commitments.py:13, src/innerproduct/inner_product_verifier.py:134,140,
 * src/rangeproofs/rangeproof_prover.py:81, rangeproof_verifier.py:92, ...
 * n == 0 gives the identity (pippenger.py:28-29). */
int bpmi_msm(bpmi_ctx *ctx, const uint8_t *pts, const uint8_t *scalars, uint64_t n, uint8_t out[64]);
int bpmi_msm_dev(bpmi_ctx *ctx, const void *d_pts, const void *d_scalars, uint64_t n, uint8_t out[64]);
/* The same MSM (n <= 2^23, device pointers) split into an asynchronous pair: `enqueue` queues every kernel
 * and the device->host copy of the window sums on the ctx stream and returns; `finish` waits for that
 * MSM only (its own completion event), runs the host part of the tail and writes the result.  Three slots
 * (0, 1, 2) may be in flight: a caller that rotates them overlaps the host tail -- and its own work between
 * two MSMs, e.g. the exchange of per-GPU partial results -- of MSM k with the kernels of MSM k + 1 (and, three
 * deep with "async_lanes", the sort of MSM k + 2).  The
 * inputs must stay valid and unmodified until `finish`; while a slot is pending, a synchronous call that
 * needs it fails with BPMI_E_STATE. */
int bpmi_msm_dev_enqueue(bpmi_ctx *ctx, int slot, const void *d_pts, const void *d_scalars, uint64_t n);
int bpmi_msm_finish(bpmi_ctx *ctx, int slot, uint8_t out[64]);
/* What an MSM of n pairs runs as under the ctx's current options; no GPU work.  pipelined: 1 = as bpmi_msm_dev_enqueue runs it, 0 = as
 * the synchronous entry points do (which slice large inputs: option "slice_n").  geom[0] kernel family (0 the bucket pipeline, 1 the
 * one-launch kernel, 2 one block per window), [1] window bits c, [2] windows W, [3] how many of them are c + 1 bits wide, [4] buckets,
 * [5] sorted entries per thread of the accumulation, [6] slices K, [7] pairs per slice (the other fields describe ONE slice).
 * The bucket additions of the call are K x W x (pairs per slice); bench.py prices its multiply-adds from this, not from a constant. */
int bpmi_msm_geometry(bpmi_ctx *ctx, uint64_t n, int pipelined, uint32_t geom[8]);
/* One MSM over up to three (points, scalars) arrays in different device buffers (BPMI_MAX_N pairs in total at most):
 * `multiexp(gs + hs + ..., a + b + ...)` without the list concatenation of src/utils/commitments.py:13. */
int bpmi_msm_segs_dev(bpmi_ctx *ctx, uint32_t nseg, const void *const *d_pts, const void *const *d_scalars, const uint64_t *n, uint8_t out[64]);
/* Two independent MSMs (n0, n1 <= 2^23) from host buffers, overlapped on the ctx's two lanes: pairs
 * such as A / S (rangeproof_prover.py:52,60) and T1 / T2 (:71-72) cost one round trip instead of two. */
int bpmi_msm2(bpmi_ctx *ctx, const uint8_t *pts0, const uint8_t *scalars0, uint64_t n0, uint8_t out0[64], const uint8_t *pts1,
              const uint8_t *scalars1, uint64_t n1, uint8_t out1[64]);

/* ---- batched point operations ----------------------------------------------------
 * out[i] = scalars[i] * pts[i]       replaces `ModP * Point` / `int * Point`
 * (src/utils/utils.py:43-44), e.g. hsp[i] = y^-i * hs[i]
 * (src/rangeproofs/rangeproof_prover.py:77, rangeproof_verifier.py:72). */
int bpmi_ec_mul_batch(bpmi_ctx *ctx, const uint8_t *pts, const uint8_t *scalars, uint64_t n, uint8_t *out);
int bpmi_ec_mul_batch_dev(bpmi_ctx *ctx, const void *d_pts, const void *d_scalars, uint64_t n, void *d_out);
/* out[i] = k1 * p1[i] + k2 * p2[i]   replaces the generator fold
 * g' = x^-1 * g_lo + x * g_hi (src/innerproduct/inner_product_prover.py:107-108). */
int bpmi_ec_lincomb2_batch(bpmi_ctx *ctx, const uint8_t *p1, const uint8_t *p2, const uint8_t k1[32],
                           const uint8_t k2[32], uint64_t n, uint8_t *out);
int bpmi_ec_lincomb2_batch_dev(bpmi_ctx *ctx, const void *d_p1, const void *d_p2, const uint8_t k1[32],
                               const uint8_t k2[32], uint64_t n, void *d_out);
/* out = sum_i pts[i]                 replaces chains of `Point + Point`
 * (fastecdsa Point.__add__; e.g. `A + x*S + multiexp(...)`, rangeproof_prover.py:78-87),
 * and folds the per-GPU partial results of a sharded MSM. */
int bpmi_ec_sum(bpmi_ctx *ctx, const uint8_t *pts, uint64_t n, uint8_t out[64]);
/* the same for points already in device memory (e.g. the receive buffer of an all_gather) */
int bpmi_ec_sum_dev(bpmi_ctx *ctx, const void *d_pts, uint64_t n, uint8_t out[64]);
/* ... and without the wait: the sum goes to d_out (64 bytes of device memory), ordered on the ctx stream; read it after bpmi_sync */
int bpmi_ec_sum_dev_enqueue(bpmi_ctx *ctx, const void *d_pts, uint64_t n, void *d_out);

/* out[i] = the point encoded by comp[33*i .. 33*i+33) in SEC1 compressed form (0x02 | 0x03,
 * then x big-endian; 33 zero bytes = identity); ok[i] = 1 when the encoding is valid
 * (known tag, x < p, x^3 + 7 a square), else out[i] = identity and ok[i] = 0.
 * Replaces bytes_to_point / b64_to_point (src/utils/utils.py:114-131) in bulk, e.g. for
 * the 19 points of every proof of a batch that arrives as bytes. */
int bpmi_ec_decompress_batch(bpmi_ctx *ctx, const uint8_t *comp, uint64_t n, uint8_t *out, uint8_t *ok);
/* the same with the decoded points left on the device (d_out: n x 64 bytes of device memory), ready to be MSM input */
int bpmi_ec_decompress_batch_dev(bpmi_ctx *ctx, const uint8_t *comp, uint64_t n, void *d_out, uint8_t *ok);

/* ---- bulk scalar (mod q) operations ------------------------------------------------
 * out = sum_i a[i] * b[i] mod q       replaces inner_product (src/utils/utils.py:134-137) */
int bpmi_sc_dot(bpmi_ctx *ctx, const uint8_t *a, const uint8_t *b, uint64_t n, uint8_t out[32]);
int bpmi_sc_dot_dev(bpmi_ctx *ctx, const void *d_a, const void *d_b, uint64_t n, uint8_t out[32]);
/* out[i] = x * lo[i] + y * hi[i] mod q  replaces the a / b fold
 * (src/innerproduct/inner_product_prover.py:109-110) */
int bpmi_sc_fold(bpmi_ctx *ctx, const uint8_t *lo, const uint8_t *hi, const uint8_t x[32], const uint8_t y[32],
                 uint64_t n, uint8_t *out);
int bpmi_sc_fold_dev(bpmi_ctx *ctx, const void *d_lo, const void *d_hi, const uint8_t x[32], const uint8_t y[32],
                     uint64_t n, void *d_out);

/* The verifier's s-vector with the proof's final scalars folded in, n = 2^k entries each:
 *   s_i = prod_j xs[j]^(+1 if bit (k-1-j) of i is set, else -1)   (Verifier2.get_ss,
 *   src/innerproduct/inner_product_verifier.py:91-102),  sa[i] = a s_i,  sb[i] = b s_i^-1 scale[i]
 * (the `a * s_i` / `b * s_i^-1` lists of :131-133; scale: n scalars or NULL -- hsp_i = y^-i hs_i of
 * src/rangeproofs/rangeproof_verifier.py:72 rides in the scalars).  xs, xinvs: k challenges and their
 * inverses.  2-3 multiplications per element on the device instead of the reference's k per element. */
int bpmi_sc_svector(bpmi_ctx *ctx, const uint8_t *xs, const uint8_t *xinvs, uint32_t k, const uint8_t a[32], const uint8_t b[32],
                    const uint8_t *scale, uint8_t *sa, uint8_t *sb);

/* Verifier2.verify's two sides (src/innerproduct/inner_product_verifier.py:127-147) as ONE multi-scalar
 * multiplication over generators that are already in device memory (n = 2^k points each, d_hscale: n scalars or NULL):
 *   out = sum_i sa[i] g_i + sum_i sb[i] h_i + sum_t extra_scalars[t] * extra_pts[t]
 * with sa, sb as in bpmi_sc_svector, computed on the device and consumed there (no host loop over n, no
 * round trip of the 2n scalars).  The caller passes u, the L_j, R_j and P with their scalars
 * (a b, -x_j^2, -x_j^-2, -1) as the extra terms and accepts iff out is the identity (64 zero bytes). */
int bpmi_ipa_verify_dev(bpmi_ctx *ctx, const void *d_g, const void *d_h, const void *d_hscale, uint64_t n, const uint8_t *xs,
                        const uint8_t *xinvs, uint32_t k, const uint8_t a[32], const uint8_t b[32], const uint8_t *extra_pts,
                        const uint8_t *extra_scalars, uint64_t n_extra, uint8_t out[64]);

/* ---- inner-product argument prover, split at the Fiat-Shamir edge ----------------------
 * One object = one run of FastNIProver2.prove (src/innerproduct/inner_product_prover.py:70-110).
 * g, h: n points; a, b: n scalars; u: one point; all copied to the device once and kept
 * there.  a and b are halved in place every round; the generator fold is DEFERRED: L and R
 * are MSMs over the unfolded generators with the fold coefficients multiplied into the
 * scalars (the prover's outputs never contain the folded generators), and large bases are
 * materialised 16-way at once.
 *   bpmi_ipa_round_LR : cl, cr, L, R of the current round          (:96-99)
 *   --- host: transcript.add_list_points([L, R]); x = H(transcript) (:102-106) ---
 *   bpmi_ipa_fold     : g, h, a, b <- folded with x, x^-1           (:107-110)
 *   bpmi_ipa_finish   : the final a[0], b[0]                         (:85-94)     */
int bpmi_ipa_create(bpmi_ctx *ctx, const uint8_t *g, const uint8_t *h, const uint8_t *a, const uint8_t *b,
                    uint64_t n, const uint8_t u[64], bpmi_ipa **out);
int bpmi_ipa_create_dev(bpmi_ctx *ctx, const void *d_g, const void *d_h, const void *d_a, const void *d_b,
                        uint64_t n, const uint8_t u[64], bpmi_ipa **out);
/* As bpmi_ipa_create, for the generators h_scale[i] * h[i] (h_scale: n scalars, or NULL):
 * the range-proof provers run the argument over hsp[i] = y^-i * hs[i]
 * (src/rangeproofs/rangeproof_prover.py:77,88); the factors are multiplied into the MSM
 * scalars instead of n point multiplications (large n: one batched multiplication). */
int bpmi_ipa_create_scaled(bpmi_ctx *ctx, const uint8_t *g, const uint8_t *h, const uint8_t *a, const uint8_t *b,
                           uint64_t n, const uint8_t u[64], const uint8_t *h_scale, bpmi_ipa **out);
uint64_t bpmi_ipa_len(const bpmi_ipa *st);
int bpmi_ipa_round_LR(bpmi_ipa *st, uint8_t L[64], uint8_t R[64]);
int bpmi_ipa_fold(bpmi_ipa *st, const uint8_t x[32], const uint8_t xinv[32]);
/* the whole halving loop in one call, Fiat-Shamir included: rounds of (L, R) -> transcript items -> x = mod_hash(transcript, q) -> fold
 * until the state has length 1 (inner_product_prover.py:94-110, utils.py:84-97).  digest: the transcript so far; digest_out
 * (capacity cap): the transcript after the last round; xs / Ls / Rs: 32 / 64 / 64 bytes per round, max_rounds entries each. */
int bpmi_ipa_prove_rounds(bpmi_ipa *st, const uint8_t *digest, uint64_t digest_len, uint8_t *digest_out, uint64_t cap, uint64_t *out_len, uint8_t *xs,
                          uint8_t *Ls, uint8_t *Rs, uint32_t max_rounds, uint32_t *rounds);
int bpmi_ipa_finish(bpmi_ipa *st, uint8_t a[32], uint8_t b[32]);
/* The CURRENT (folded) g, h (len points each) and a, b (len scalars each), len = bpmi_ipa_len:
 * what the reference holds in gp, hp, ap, bp at the top of its loop (:84).  With folds
 * deferred, every generator costs one MSM, so this is meant for short states (len <= 64):
 * a prover sharded over GPUs hands its last element to the other ranks through it. */
int bpmi_ipa_export(bpmi_ipa *st, uint8_t *g, uint8_t *h, uint8_t *a, uint8_t *b);
void bpmi_ipa_destroy(bpmi_ipa *st);

/* ---- batch verification of range proofs: host-side preparation (no GPU work, no ctx) ---------------
 * For n_proofs range proofs -- values_per_proof = 1: single-value proofs; m > 1: aggregated proofs of m values
 * each, n_gens = m x bits -- over n_gens generator pairs in the wire format of
 * python-bulletproofs_amd/rangeproofs/codec.py (blob i = blobs[blob_off[i] .. blob_off[i+1]); the offsets must be
 * non-decreasing and end inside blobs[0 .. blobs_len), else BPMI_E_ARG):
 * parses every proof, runs the byte-level transcript checks of RangeVerifier / Verifier1 / Verifier2
 * (src/rangeproofs/rangeproof_verifier.py:42-53, src/innerproduct/inner_product_verifier.py:31-43,
 * 104-125) and computes, with the caller's random weights (4 scalars per proof, LE, in [1, q); or weights = NULL and
 * a fresh random 32-byte `seed`: the weights are then derived inside, SHA-256(seed || proof index || 0..3) cut to 248 bits), the
 * scalars of the ONE multi-scalar multiplication that is the identity iff every proof verifies:
 *   v_scalars   n_proofs x values_per_proof x 32 B   for the commitments V_i (V_i,0 .. V_i,m-1)
 *   pt_scalars  n_proofs x (6 + 2k) x 32 B, k = log2 n_gens, for each proof's points in wire order
 *               (T1 T2 A S u_new P_new L_1..L_k R_1..R_k)
 *   shared      (5 + 2 n_gens) x 32 B: coefficients of g, h, u, the constant added to every gs_i and
 *               to every hs_i, then of gs_0.. and hs_0..
 *   comp_out    (optional, may be NULL) n_proofs x (6 + 2k) x 33 B: the proofs' compressed points, gathered
 *               in the same order, ready for bpmi_ec_decompress_batch
 * *first_bad = index of the first proof that failed parsing or a transcript check, or -1.  `threads`
 * host threads share the proofs.  The native twin of BatchRangeVerifier.add (rangeproofs/batch.py). */
int bpmi_rp_batch_prepare(uint32_t n_gens, uint32_t values_per_proof, uint64_t n_proofs, const uint8_t *blobs, uint64_t blobs_len, const uint64_t *blob_off,
                          const uint8_t *weights, const uint8_t *seed, int threads, uint8_t *v_scalars, uint8_t *pt_scalars, uint8_t *shared, uint8_t *comp_out, int64_t *first_bad);

/* out[i - lo] = mod_hash(str(i) || tail, q) for i in [lo, hi), 32 bytes little-endian each -- the reference's seeded
 * challenge / blinding derivation (src/utils/utils.py:84-97: the first counter c >= 1 with SHA-256(str(c) || msg) in [1, q)),
 * in bulk: the range-proof provers draw 2 n m blinding scalars this way (rangeproof_prover.py:57-60).  Host code. */
int bpmi_mod_hash_range(const uint8_t *tail, uint64_t tail_len, uint64_t lo, uint64_t hi, int threads, uint8_t *out);

/* The O(n m) scalar algebra of the range-proof provers and verifiers in native HOST code (csrc/rp_algebra_host.hpp; no elliptic-curve
 * arithmetic, no GPU): vectors of scalars mod q in and out, 32 bytes little-endian each, `threads` host threads.  aL: one byte per
 * bit (0 / 1); n bits per value, m values (m = 1 and aggregated = 0: the single proof's z^2 2^i terms).
 *   bpmi_rp_poly_coeffs       t1, t2 of `_get_polynomial_coeffs` (src/rangeproofs/rangeproof_prover.py:93-101,
 *                             rangeproof_aggreg_prover.py:117-130)
 *   bpmi_rp_final_vectors     l, r, t_hat of `_final_compute` (:103-112 / :132-146), plus yscale_i = y^-i and
 *                             hsc_i = (z y^i + zt_i) y^-i: the hs-scalars of P (:78-87 / :95-101) over the UNSCALED generators
 *   bpmi_rp_verifier_vectors  yscale, hsc as above and ysum = sum_{i < n m} y^i for delta(y, z) (rangeproof_verifier.py:55-99,
 *                             rangeproof_aggreg_verifier.py:55-108) */
int bpmi_rp_poly_coeffs(uint32_t n, uint32_t m, int aggregated, const uint8_t *aL, const uint8_t *sL, const uint8_t *sR, const uint8_t y[32],
                        const uint8_t z[32], int threads, uint8_t t1[32], uint8_t t2[32]);
int bpmi_rp_final_vectors(uint32_t n, uint32_t m, int aggregated, const uint8_t *aL, const uint8_t *sL, const uint8_t *sR, const uint8_t y[32],
                          const uint8_t z[32], const uint8_t x[32], int threads, uint8_t *ls, uint8_t *rs, uint8_t t_hat[32], uint8_t *hsc, uint8_t *yscale);
int bpmi_rp_verifier_vectors(uint32_t n, uint32_t m, int aggregated, const uint8_t y[32], const uint8_t z[32], int threads, uint8_t *hsc, uint8_t *yscale,
                             uint8_t ysum[32]);

/* Wire format 2 of a range proof (csrc/rp_wire_v2_host.hpp, rangeproofs/codec.py): format 1 without the three transcripts -- they
 * are functions of the other fields (/root/reference/src/utils/transcript.py:13-33, rangeproof_verifier.py:42-53) -- plus the four
 * challenges y, z, x, x_ip in binary and the two transcript seeds: 1.09 KB instead of 2.56 KB for a 64-bit proof.  This entry point
 * expands n_proofs format-2 proofs (proof g = blobs[off[g], off[g + 1])) into format-1 proofs packed in out[0, cap), out_off[0 ..
 * n_proofs]; HOST code.  *first_bad = first proof that is not a well-formed format-2 proof, or -1.  bpmi_rp_batch_prepare_dev and
 * bpmi_rp_batch_verify_dev take EITHER format (all proofs of a call in the same one, told by the magic of the first): format 2 is
 * uploaded as it is and expanded on the device.  A format-2 proof is valid exactly when its expansion is.
 * Wire format 3 (round 6) is a format-2 proof with the magic "BPRP3", followed by the y coordinate of each of its 6 + 2k points (32 B
 * big-endian, 0 for the identity; 1.67 KB for a 64-bit proof).  The verifiers CHECK each y -- below p, the parity of the encoding's
 * tag, on the curve with x: then it is the y a decompression would compute -- instead of taking 6 + 2k square roots per proof, which
 * were a quarter of a batch verification's device time.  A format-3 proof is valid exactly when its format-2 part is and every y is
 * right; a wrong y is an invalid proof (*first_bad names it).  This entry point and the three batch entry points take format 3
 * wherever they take format 2; bpmi_rp_prove_batch writes it under option "prover_wire_format" = 3 (the prover holds the points in
 * affine form anyway). */
int bpmi_rp_wire_v2_to_v1(const uint8_t *blobs, uint64_t blobs_len, const uint64_t *off, uint64_t n_proofs, uint8_t *out, uint64_t cap,
                          uint64_t *out_off, int64_t *first_bad);

/* The same preparation on the GPU (csrc/rp_batch_kernels.hpp; one lane per proof parses, re-hashes the transcripts and computes
 * the weighted scalars).  `blobs` / `blob_off` / `weights` / `seed` / `shared` are HOST pointers with the meaning above (blobs may be
 * page-locked memory from bpmi_host_alloc: the upload then runs at link speed); the outputs that feed the MSM stay on the device:
 *   d_v_scalars   n_proofs x values_per_proof x 32 B      d_pt_scalars  n_proofs x (6 + 2k) x 32 B
 *   d_points      n_proofs x (6 + 2k) x 64 B: the proofs' points, decoded where they lie in the blobs (wire order)
 * *first_bad = smallest index of a proof that failed parsing, a transcript check, or has an invalid point encoding; -1 if none.
 * When *first_bad >= 0 the device arrays and `shared` hold no usable result (the batch is rejected).  Otherwise: the same numbers
 * as bpmi_rp_batch_prepare for the same weights / seed, byte for byte (tests/test_gpu_batch_dev.py).  A wire proof longer
 * than 32 KiB is invalid (in both functions).  n_proofs <= 2^22, blobs_len <= 4 GiB per call.  One call at a time per ctx. */
int bpmi_rp_batch_prepare_dev(bpmi_ctx *ctx, uint32_t n_gens, uint32_t values_per_proof, uint64_t n_proofs, const uint8_t *blobs, uint64_t blobs_len,
                              const uint64_t *blob_off, const uint8_t *weights, const uint8_t *seed, void *d_v_scalars, void *d_pt_scalars, void *d_points,
                              uint8_t *shared, int64_t *first_bad);
/* The whole batch verification in ONE call (replaces a loop of RangeVerifier.verify, /root/reference/src/rangeproofs/
 * rangeproof_verifier.py:55-99 with src/innerproduct/inner_product_verifier.py:44-58, :127-147, over proofs that share their
 * generators): the preparation above (wire bytes uploaded in slices, the points of a slice decoded while the next slice is on the
 * link), the shared coefficients folded into the scalars of g, h, u, gs, hs on the device, and the batch's one MSM over
 * [g h u gs hs | commitments | proof points] -- no host round trip in between.
 *   v_points  HOST   n_proofs x values_per_proof x 64 B: the commitments, in proof order
 *   d_gens    DEVICE (3 + 2 n_gens) x 64 B: g, h, u, gs[0..n), hs[0..n)  (uploaded once per verifier)
 *   d_points / d_scalars  DEVICE scratch for n_proofs (values_per_proof + 6 + 2k) points of 64 B / scalars of 32 B
 * out = the 64-byte value of the random linear combination: the identity (64 zero bytes) exactly when every equation of every
 * proof holds (up to the 1/q soundness error of the random weights); ranks that verified disjoint shards fold their values.
 * *first_bad as for bpmi_rp_batch_prepare_dev; when it is >= 0 `out` means nothing.  At most 2^23 points in the MSM. */
int bpmi_rp_batch_verify_dev(bpmi_ctx *ctx, uint32_t n_gens, uint32_t values_per_proof, uint64_t n_proofs, const uint8_t *blobs, uint64_t blobs_len,
                             const uint64_t *blob_off, const uint8_t *weights, const uint8_t *seed, const uint8_t *v_points, const void *d_gens, void *d_points,
                             void *d_scalars, uint8_t out[64], int64_t *first_bad);
/* ---- a BATCH of single-value range proofs, proved on the device in one call (csrc/rp_prove_kernels.hpp, rp_prove_host.hpp) -------------
 * Replaces a LOOP of NIRangeProver(v, n, g, h, gs, hs, gamma, u, group, seed).prove() (/root/reference/src/rangeproofs/
 * rangeproof_prover.py:35-91) with NIProver.prove / FastNIProver2.prove inside (/root/reference/src/innerproduct/
 * inner_product_prover.py:27-44, :84-110) over proofs that share their generators: the same transcripts
 * (/root/reference/src/utils/transcript.py:13-33), the same seeded blinding scalars (/root/reference/src/utils/utils.py:84-97), the
 * same proofs byte for byte -- but every protocol step is ONE launch over all proofs, every scalar multiplication a lookup in tables
 * of the fixed generators, and the Fiat-Shamir hashes run on the device.
 *   bpmi_rp_prover_create   nbits: a power of two in [2, 128]; g, h, u: 64-byte points; gs, hs: nbits points each.  Builds the tables
 *                           (windows of "prover_table_bits" bits, default 16: 4.4 GB of device memory and ~72 ms for 64-bit proofs;
 *                           12 bits: 378 MB, 17 ms, 22 % slower proving) and keeps them for the prover's lifetime.
 *                           The points are checked to be on the curve (option "validate_points" >= 1, the default): BPMI_E_ARG names the first bad one.
 *   bpmi_rp_prove_batch     values, gammas: n_proofs x 32 B little-endian, in [0, q) -- checked: BPMI_E_ARG names the first index that is not -- (of a value only the low nbits bits enter the
 *                           proof, as in rangeproof_prover.py:40); seeds: proof i's transcript seed = seeds[seed_off[i] .. seed_off[i+1])
 *                           (at most 65 535 bytes).  out[out_off[i] .. out_off[i+1]) = proof i in wire format 2 (3 under option "prover_wire_format")
 *                           (python-bulletproofs_amd/rangeproofs/codec.py; bpmi_rp_wire_v2_to_v1 expands it, bpmi_rp_batch_verify_dev
 *                           takes it as it is); out_off has n_proofs + 1 entries; cap >= n_proofs x bpmi_rp_prove_batch_proof_bytes
 *                           (of the longest seed).  At most 2^20 proofs per call; one call at a time per prover and ctx.
 *   bpmi_rp_prover_last_ms  device milliseconds of the last batch: A and S | y, z, T1, T2 | x, the vectors, P_new | the rounds of the
 *                           inner-product argument | the wire bytes | their copy to the host | the whole batch */
/* The same for AGGREGATED proofs (round 6): a proof covers m values of nbits bits each (AggregNIRangeProver,
 * /root/reference/src/rangeproofs/rangeproof_aggreg_prover.py:36-146); nbits and m powers of two with 2 <= nbits x m <= 128, gs and hs nbits x m
 * points each; bpmi_rp_prove_batch then takes n_proofs x m values and blinding factors (proof p: entries p m .. p m + m - 1).  m = 1 is
 * bpmi_rp_prover_create.  Longer vectors (the 128 x 64-bit proof of config C4) are the single-proof prover's: one proof fills the chip there. */
int bpmi_rp_prover_create_aggregated(bpmi_ctx *ctx, uint32_t nbits, uint32_t m, const uint8_t g[64], const uint8_t h[64], const uint8_t u[64], const uint8_t *gs,
                                     const uint8_t *hs, bpmi_rp_prover **out);
int bpmi_rp_prover_create(bpmi_ctx *ctx, uint32_t nbits, const uint8_t g[64], const uint8_t h[64], const uint8_t u[64], const uint8_t *gs, const uint8_t *hs,
                          bpmi_rp_prover **out);
void bpmi_rp_prover_destroy(bpmi_rp_prover *pv);
uint64_t bpmi_rp_prove_batch_proof_bytes(const bpmi_rp_prover *pv, uint64_t seed_len);
int bpmi_rp_prove_batch(bpmi_rp_prover *pv, uint64_t n_proofs, const uint8_t *values, const uint8_t *gammas, const uint8_t *seeds, const uint64_t *seed_off,
                        uint8_t *out, uint64_t cap, uint64_t *out_off);
int bpmi_rp_prover_last_ms(const bpmi_rp_prover *pv, double ms[7]);

/* Page-locked host memory (hipHostMalloc) for buffers handed to the library repeatedly, e.g. the receive buffer of wire proofs. */
int bpmi_host_alloc(bpmi_ctx *ctx, size_t bytes, void **out);
int bpmi_host_free(bpmi_ctx *ctx, void *p);

/* ---- self-test hook (not part of the drop-in surface) ---------------------------------------------------
 * Runs one member of the device's field-multiplication family on n operand tuples given as RAW 9 x 29-bit
 * limb vectors (9 uint32 each, any lazy magnitude the routine allows) and returns the raw result limbs:
 * op 0 a*b, 1 a^2, 2 a*b + c, 3 a^2 + c, 4 a*b + c*d, 5 carry(a), 6 canonical(a).  tests/test_gpu_field.py
 * compares them with the host build of the same header, limb for limb -- the arithmetic under every
 * `Point + Point` of the reference (src/pippenger/group.py:31-32) is pinned at its worst-case bounds.
 * op 10..15: the mod-q limb arithmetic of the batch-preparation kernel (csrc/scalar.hpp "sq"): 10 a*b, 11 a+b, 12 a-b,
 * 13 -a (raw limbs out); 14 canonical 8 words of a; 15 inverse of a's canonical value (8 words, binary Euclid). */
int bpmi_debug_fe_op(bpmi_ctx *ctx, int op, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d, uint64_t n,
                     uint32_t *out);
/* self-test hook: out[i] = a[i] + b[i] on 144-byte XYZZ records (4 x 9 limbs of 29 bits: X, Y, ZZ, ZZZ; all zero = identity) with the
 * four-lane point addition of the bucket reduction (csrc/msm_kernels.hpp quad_add) */
int bpmi_debug_quad_add(bpmi_ctx *ctx, const uint32_t *a, const uint32_t *b, uint64_t n, uint32_t *out);

/* ---- per-stage device timing (HIP events on the ctx's stream) --------------------------
 * After bpmi_profile(ctx, 1) every MSM records HIP events around each kernel
 * stage (enable = 2: around the dominant stage, msm_accumulate, only -- every recorded event costs
 * a ~10 us bubble between two kernels, so a timed run should use 2); bpmi_profile_read returns accumulated milliseconds and launch counts
 * per stage since the last reset.  Stage names: bpmi_profile_stage_name(i). */
#define BPMI_NSTAGES 15
int bpmi_profile(bpmi_ctx *ctx, int enable);
int bpmi_profile_reset(bpmi_ctx *ctx);
int bpmi_profile_read(bpmi_ctx *ctx, double ms[BPMI_NSTAGES], uint64_t calls[BPMI_NSTAGES]);
const char *bpmi_profile_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif /* BPMI_H */
