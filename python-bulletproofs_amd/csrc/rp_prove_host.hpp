// rp_prove_host.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).  HOST code.
// The batched range-proof prover's object and launch sequence (kernels: rp_prove_kernels.hpp): bpmi_rp_prover_create builds the
// fixed-base tables of a generator set once; bpmi_rp_prove_batch proves any number of single-value proofs over them, every protocol
// step as one launch over the whole batch, and returns the proofs as wire format 2.
#pragma once

#define PV_SPLIT_MIN 2048u               // a batch of 2 x this many proofs or more runs as two halves on two lanes
#define PROVER_TW_DEFAULT 16u          // 16 additions per term, 4.4 GB and 72 ms to build for 64-bit proofs (profiles/r06_batch_prover_table_bits.txt); 12: 22 additions, 378 MB, 17 ms
struct bpmi_rp_prover {
  bpmi_ctx *ctx = nullptr;
  u32 n = 0, k = 0, nbases = 0;                // n: elements of a proof's vectors = nb m
  u32 nb = 0, m = 1;                           // bits per value, values per proof (m > 1: aggregated proofs)
  u32 *table = nullptr;                        // [(3 + 2n)][wt][bt] affine points
  u32 tw = 16, wt = 16, bt = 32768;            // table windows: tw bits, wt = ceil(256 / tw) per scalar, bt = 2^(tw-1) entries each
  unsigned short *bases = nullptr;             // device: the base lists of every job kind (offsets below, in entries)
  u32 off_S = 0, off_T = 0, off_P = 0, off_round = 0;
  rpp::sc x_ip;                                // mod_hash(b"&", q): the Protocol-1 challenge of an empty seed (the same for every proof)
  std::string ip_prefix;                       // "&&" str(x_ip) "&"
  unsigned char *d_ip_prefix = nullptr;
  u32 u_new[16];                               // x_ip u, affine words
  void *buf = nullptr; size_t buf_bytes = 0;   // the batch's device arrays (grown on demand)
  void *pin = nullptr; size_t pin_bytes = 0;   // page-locked staging of the inputs / the proofs
  hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // phase boundaries of a batch, created once with the prover
  hipEvent_t ev_chain[2] = {nullptr, nullptr};                                            // two halves of a batch: the last multi-scalar multiplication of each
  double last_ms[8] = {0};                     // device milliseconds of the last batch: blind+A/S | y,z+T | x+final+P_new | rounds | emit+copy | total
};

namespace rpp_host {

static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// decimal text of a 256-bit little-endian value
static std::string decimal_of(const uint8_t le[32]) {
  rp::Sq v;
  rp::q_from_le(v, le);
  std::vector<uint8_t> dg;
  rpt::append_decimal(dg, v);
  return std::string((const char *)dg.data(), dg.size() - 1);
}
// base64 of a seed, written where it is needed (a batch encodes one seed per proof: no string per proof); returns the length
static size_t b64_into(char *o, const uint8_t *p, size_t n) {
  static const char T[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  char *q = o;
  for (size_t i = 0; i < n; i += 3) {
    const uint32_t a = p[i], b = i + 1 < n ? p[i + 1] : 0, c = i + 2 < n ? p[i + 2] : 0;
    const uint32_t v = (a << 16) | (b << 8) | c;
    *q++ = T[v >> 18]; *q++ = T[(v >> 12) & 63];
    *q++ = i + 1 < n ? T[(v >> 6) & 63] : '='; *q++ = i + 2 < n ? T[v & 63] : '=';
  }
  return (size_t)(q - o);
}

}  // namespace rpp_host

extern "C" {

void bpmi_rp_prover_destroy(bpmi_rp_prover *pv) {
  if (!pv) return;
  if (pv->ctx) { (void)hipSetDevice(pv->ctx->device); (void)hipStreamSynchronize(pv->ctx->stream); }
  if (pv->table) (void)hipFree(pv->table);
  if (pv->bases) (void)hipFree(pv->bases);
  if (pv->d_ip_prefix) (void)hipFree(pv->d_ip_prefix);
  if (pv->buf) (void)hipFree(pv->buf);
  if (pv->pin) (void)hipHostFree(pv->pin);
  for (auto e : pv->ev) if (e) (void)hipEventDestroy(e);
  for (auto e : pv->ev_chain) if (e) (void)hipEventDestroy(e);
  delete pv;
}

static int rp_prover_create_impl(bpmi_ctx *ctx, uint32_t vbits, uint32_t m, const uint8_t g[64], const uint8_t h[64], const uint8_t u[64], const uint8_t *gs, const uint8_t *hs,
                                 bpmi_rp_prover **out, bpmi_rp_prover *&partial);
// (the C ABI never throws: a failed host allocation inside -- the 64 (3 + 2n)-byte point list, the base lists, a std::string -- is BPMI_E_NOMEM)
int bpmi_rp_prover_create_aggregated(bpmi_ctx *ctx, uint32_t nbits, uint32_t m, const uint8_t g[64], const uint8_t h[64], const uint8_t u[64], const uint8_t *gs,
                                     const uint8_t *hs, bpmi_rp_prover **out) {
  bpmi_rp_prover *partial = nullptr;
  try {
    return rp_prover_create_impl(ctx, nbits, m, g, h, u, gs, hs, out, partial);
  } catch (const std::bad_alloc &) {
    if (partial) bpmi_rp_prover_destroy(partial);
    if (out) *out = nullptr;
    return ctx ? fail(ctx, BPMI_E_NOMEM, "bpmi_rp_prover_create: out of host memory") : BPMI_E_NOMEM;
  }
}
int bpmi_rp_prover_create(bpmi_ctx *ctx, uint32_t nbits, const uint8_t g[64], const uint8_t h[64], const uint8_t u[64], const uint8_t *gs, const uint8_t *hs,
                          bpmi_rp_prover **out) {
  return bpmi_rp_prover_create_aggregated(ctx, nbits, 1, g, h, u, gs, hs, out);
}
static int rp_prover_create_impl(bpmi_ctx *ctx, uint32_t vbits, uint32_t m, const uint8_t g[64], const uint8_t h[64], const uint8_t u[64], const uint8_t *gs, const uint8_t *hs,
                                 bpmi_rp_prover **out, bpmi_rp_prover *&partial) {
  if (!ctx || !g || !h || !u || !gs || !hs || !out) return ctx ? fail(ctx, BPMI_E_ARG, "null argument") : BPMI_E_ARG;
  *out = nullptr;
  if (vbits < 1 || vbits > 128 || (vbits & (vbits - 1)) || m < 1 || (m & (m - 1)) || (uint64_t)vbits * m < 2 || (uint64_t)vbits * m > 128)
    return fail(ctx, BPMI_E_ARG, "the bit width and the number of values must be powers of two with 2 <= bits x values <= 128");
  const uint32_t nbits = vbits * m;                      // elements of a proof's vectors
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->opt_validate >= 1) {
    int vrc = validate_host(ctx, g, 1, "bpmi_rp_prover_create", "g");
    if (!vrc) vrc = validate_host(ctx, h, 1, "bpmi_rp_prover_create", "h");
    if (!vrc) vrc = validate_host(ctx, u, 1, "bpmi_rp_prover_create", "u");
    if (!vrc) vrc = validate_host(ctx, gs, nbits, "bpmi_rp_prover_create", "gs");
    if (!vrc) vrc = validate_host(ctx, hs, nbits, "bpmi_rp_prover_create", "hs");
    if (vrc) return vrc;
  }
  bpmi_rp_prover *pv = new bpmi_rp_prover();
  partial = pv;
  pv->ctx = ctx; pv->n = nbits; pv->nb = vbits; pv->m = m; pv->k = 0;
  while ((1u << pv->k) < nbits) pv->k++;
  const u32 n = nbits, nb = 3 + 2 * n;
  pv->nbases = nb;
  auto bail = [&](int rc) { bpmi_rp_prover_destroy(pv); partial = nullptr; return rc; };
  for (int i = 0; i < 7; i++) { hipError_t ee = hipEventCreate(&pv->ev[i]); if (ee != hipSuccess) return bail(fail(ctx, BPMI_E_HIP, std::string("bpmi_rp_prover_create: ") + hipGetErrorString(ee))); }
  for (int i = 0; i < 2; i++) { hipError_t ee = hipEventCreateWithFlags(&pv->ev_chain[i], hipEventDisableTiming); if (ee != hipSuccess) return bail(fail(ctx, BPMI_E_HIP, std::string("bpmi_rp_prover_create: ") + hipGetErrorString(ee))); }
  // the Protocol-1 challenge of the empty seed: transcript b"&" (inner_product_prover.py:33; transcript.py:13-14)
  {
    const uint8_t amp = '&';
    rp::Sha one;
    rp::sha_init(one);
    rp::sha_update(one, (const uint8_t *)"1", 1);
    rp::sha_update(one, &amp, 1);
    rp::Sq x;
    rp::mod_hash_q(x, one, &amp, 1);
    uint8_t le[32];
    rp::q_to_le(le, x);
    memcpy(pv->x_ip.v, le, 32);
    pv->ip_prefix = "&&" + rpp_host::decimal_of(le) + "&";
  }
  // points of the bases, the table's inputs and the table: entry (b, k, d) = d 2^(8k) base_b by the engine's batched multiplication
  // table windows (ctx option "prover_table_bits", read HERE): wider windows are fewer additions per term and a larger table
  // (measured at 2^14 64-bit proofs, profiles/r05_batch_prover_table_bits.txt: 8 bits 36.0 ms / 34 MB / 16 ms to build, 10: 31.0 / 112 / 29,
  // 11: 29.3 / 206 / 44, 12: 28.1 / 378 / 72, 13: 26.6 / 687 / 126 with round 5's builder; round 6's builder and 14 .. 16 bits:
  // profiles/r06_batch_prover_table_bits.txt)
  pv->tw = ctx->opt_prover_tw ? (u32)ctx->opt_prover_tw : PROVER_TW_DEFAULT;
  pv->wt = (256u + pv->tw - 1u) / pv->tw; pv->bt = 1u << (pv->tw - 1u);
  const size_t entries = (size_t)nb * pv->wt * pv->bt;
  const u32 nbk = nb * pv->wt;                       // (base, window) pairs
  std::vector<uint8_t> basepts(64 * (size_t)nb);
  memcpy(&basepts[0], g, 64); memcpy(&basepts[64], h, 64); memcpy(&basepts[128], u, 64);
  memcpy(&basepts[192], gs, 64 * (size_t)n); memcpy(&basepts[192 + 64 * (size_t)n], hs, 64 * (size_t)n);
  u32 *d_base = nullptr, *d_pts = nullptr, *d_sc = nullptr, *d_wb = nullptr;
  hipError_t e = hipMalloc(&d_base, 64 * (size_t)nb + 96);
  if (e == hipSuccess) e = hipMalloc(&d_pts, 64 * (size_t)nbk);
  if (e == hipSuccess) e = hipMalloc(&d_sc, 32 * (size_t)nbk);
  if (e == hipSuccess) e = hipMalloc(&d_wb, 64 * (size_t)nbk);
  if (e == hipSuccess) e = hipMalloc(&pv->table, 64 * entries);
  auto free_tmp = [&]() { if (d_base) (void)hipFree(d_base); if (d_pts) (void)hipFree(d_pts); if (d_sc) (void)hipFree(d_sc); if (d_wb) (void)hipFree(d_wb); };
  if (e != hipSuccess) { free_tmp(); return bail(fail(ctx, BPMI_E_NOMEM, std::string("bpmi_rp_prover_create: ") + hipGetErrorString(e))); }
  e = hipMemcpyAsync(d_base, basepts.data(), basepts.size(), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // (basepts is pageable and local)
  if (e != hipSuccess) { free_tmp(); return bail(fail(ctx, BPMI_E_HIP, std::string("bpmi_rp_prover_create: ") + hipGetErrorString(e))); }
  // the window bases 2^(tw k) base_b by the engine's batched multiplication, then the multiples level by level (rp_prove_kernels.hpp)
  rpp::Tab T0 = {nullptr, pv->tw, pv->wt, pv->bt};
  hipLaunchKernelGGL(rpp::k_pv_window_scalars, dim3((nbk + 255) / 256), dim3(256), 0, ctx->stream, d_base, nb, T0, d_pts, d_sc);
  int rc = bpmi_ec_mul_batch_dev(ctx, d_pts, d_sc, nbk, d_wb);
  if (rc == BPMI_OK) {
    hipLaunchKernelGGL(rpp::k_pv_table_seed, dim3((nbk + 255) / 256), dim3(256), 0, ctx->stream, d_wb, nbk, T0, pv->table);
    for (u32 j = 0; j + 1 < pv->tw; j++) {
      const size_t threads = (size_t)nbk << j;
      hipLaunchKernelGGL(rpp::k_pv_table_level, dim3((u32)((threads + 255) / 256)), dim3(256), 0, ctx->stream, nbk, T0, j, pv->table);
    }
    e = hipGetLastError();
    if (e != hipSuccess) rc = fail(ctx, BPMI_E_HIP, std::string("bpmi_rp_prover_create: ") + hipGetErrorString(e));
  }
  // u_new = x_ip u
  if (rc == BPMI_OK) {
    e = hipMemcpyAsync((char *)d_base + 64 * (size_t)nb, pv->x_ip.v, 32, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) rc = fail(ctx, BPMI_E_HIP, hipGetErrorString(e));
  }
  if (rc == BPMI_OK) rc = bpmi_ec_mul_batch_dev(ctx, (char *)d_base + 128, (char *)d_base + 64 * (size_t)nb, 1, (char *)d_base + 64 * (size_t)nb + 32);
  if (rc == BPMI_OK) {
    e = hipMemcpyAsync(pv->u_new, (char *)d_base + 64 * (size_t)nb + 32, 64, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(ctx, BPMI_E_HIP, hipGetErrorString(e));
  }
  free_tmp();
  if (rc) return bail(rc);
  // base lists: S / P_new: gs_0.., hs_0.., then h (S) or u (P_new); T: g, h; round r: L then R (rp_prove_kernels.hpp k_pv_round_wide)
  std::vector<unsigned short> bl;
  pv->off_S = 0;
  for (u32 i = 0; i < n; i++) bl.push_back((unsigned short)(3 + i));
  for (u32 i = 0; i < n; i++) bl.push_back((unsigned short)(3 + n + i));
  bl.push_back(1);
  pv->off_T = (u32)bl.size();
  bl.push_back(0); bl.push_back(1);
  pv->off_P = (u32)bl.size();
  for (u32 i = 0; i < n; i++) bl.push_back((unsigned short)(3 + i));
  for (u32 i = 0; i < n; i++) bl.push_back((unsigned short)(3 + n + i));
  bl.push_back(2);
  pv->off_round = (u32)bl.size();
  for (u32 r = 0; r < pv->k; r++) {
    const u32 len = n >> r, half = len >> 1;
    for (int side = 0; side < 2; side++) {               // 0: L, 1: R
      for (u32 j = 0; j < n; j++) if (((j & (len - 1)) >= half) == (side == 0)) bl.push_back((unsigned short)(3 + j));
      for (u32 j = 0; j < n; j++) if (((j & (len - 1)) < half) == (side == 0)) bl.push_back((unsigned short)(3 + n + j));
      bl.push_back(2);
    }
  }
  e = hipMalloc(&pv->bases, 2 * bl.size());
  if (e == hipSuccess) e = hipMemcpy(pv->bases, bl.data(), 2 * bl.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&pv->d_ip_prefix, pv->ip_prefix.size() + 16);
  if (e == hipSuccess) e = hipMemcpy(pv->d_ip_prefix, pv->ip_prefix.data(), pv->ip_prefix.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) return bail(fail(ctx, BPMI_E_HIP, std::string("bpmi_rp_prover_create: ") + hipGetErrorString(e)));
  *out = pv;
  partial = nullptr;
  return BPMI_OK;
}

uint64_t bpmi_rp_prove_batch_proof_bytes(const bpmi_rp_prover *pv, uint64_t seed_len) {
  if (!pv) return 0;
  const uint64_t k = pv->k;
  return 6 + 32 * (5 + k) + 33 * (6 + 2 * k) + 128 + 2 + seed_len + 2 + (pv->ctx->opt_prover_wire == 3 ? 32 * (6 + 2 * k) : 0);
}

static int rp_prove_batch_impl(bpmi_rp_prover *pv, uint64_t n_proofs, const uint8_t *values, const uint8_t *gammas, const uint8_t *seeds, const uint64_t *seed_off,
                               uint8_t *out, uint64_t cap, uint64_t *out_off);
int bpmi_rp_prove_batch(bpmi_rp_prover *pv, uint64_t n_proofs, const uint8_t *values, const uint8_t *gammas, const uint8_t *seeds, const uint64_t *seed_off,
                        uint8_t *out, uint64_t cap, uint64_t *out_off) {
  try {
    return rp_prove_batch_impl(pv, n_proofs, values, gammas, seeds, seed_off, out, cap, out_off);
  } catch (const std::bad_alloc &) {
    return pv && pv->ctx ? fail(pv->ctx, BPMI_E_NOMEM, "bpmi_rp_prove_batch: out of host memory") : BPMI_E_NOMEM;
  }
}
// a 32-byte little-endian value below the group order?  (the kernels' mod-q arithmetic takes reduced operands: taux = ... + z^2 gamma)
static bool rp_scalar_reduced(const uint8_t le[32]) {
  static const uint8_t QLE[32] = {0x41, 0x41, 0x36, 0xD0, 0x8C, 0x5E, 0xD2, 0xBF, 0x3B, 0xA0, 0x48, 0xAF, 0xE6, 0xDC, 0xAE, 0xBA,
                                  0xFE, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
  for (int i = 31; i >= 0; i--) if (le[i] != QLE[i]) return le[i] < QLE[i];
  return false;
}
static int rp_prove_batch_impl(bpmi_rp_prover *pv, uint64_t n_proofs, const uint8_t *values, const uint8_t *gammas, const uint8_t *seeds, const uint64_t *seed_off,
                               uint8_t *out, uint64_t cap, uint64_t *out_off) {
  if (!pv) return BPMI_E_ARG;
  bpmi_ctx *ctx = pv->ctx;
  if (!values || !gammas || !seed_off || !out || !out_off || (!seeds && seed_off[n_proofs] != seed_off[0])) return fail(ctx, BPMI_E_ARG, "null argument");
  if (n_proofs == 0) { out_off[0] = 0; return BPMI_OK; }
  if (n_proofs > (1u << 20)) return fail(ctx, BPMI_E_ARG, "at most 2^20 proofs per call");
  // the reference's prover takes ModP values: reduced by construction (/root/reference/src/utils/utils.py:24-27); raw bytes are checked here
  for (uint64_t p = 0; p < n_proofs * pv->m; p++) {
    if (!rp_scalar_reduced(values + 32 * p)) return fail(ctx, BPMI_E_ARG, "bpmi_rp_prove_batch: values[" + std::to_string(p) + "] is not below the group order");
    if (!rp_scalar_reduced(gammas + 32 * p)) return fail(ctx, BPMI_E_ARG, "bpmi_rp_prove_batch: gammas[" + std::to_string(p) + "] is not below the group order");
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const u32 P = (u32)n_proofs, n = pv->n, k = pv->k, npt = 6 + 2 * k;
  const int wire_fmt = ctx->opt_prover_wire;                  // 2, or 3: with the points' y coordinates (bpmi_rp_prove_batch_proof_bytes counts them)
  // the proofs' seeds: base64(seed) || '&' starts every range-proof transcript (transcript.py:13-14)
  uint64_t max_seed = 0, total_out = 0;
  for (u32 p = 0; p < P; p++) {
    if (seed_off[p + 1] < seed_off[p]) return fail(ctx, BPMI_E_ARG, "seed offsets must not decrease");
    const uint64_t sl = seed_off[p + 1] - seed_off[p];
    if (sl > 0xFFFF) return fail(ctx, BPMI_E_ARG, "a seed is longer than 65535 bytes");
    max_seed = std::max(max_seed, sl);
    out_off[p] = total_out;
    total_out += bpmi_rp_prove_batch_proof_bytes(pv, sl);
  }
  out_off[P] = total_out;
  if (total_out > cap) return fail(ctx, BPMI_E_ARG, "the output buffer is too small (bpmi_rp_prove_batch_proof_bytes per proof)");
  const u32 dig0_stride = (u32)((((max_seed + 2) / 3) * 4 + 1 + 15) & ~15ull);
  const u32 tr_stride = (u32)((std::max<uint64_t>(dig0_stride + 4 * 45 + 3 * 80, pv->ip_prefix.size() + (uint64_t)k * (90 + 80)) + 64 + 15) & ~15ull);
  // ---- device layout
  using rpp_host::up256;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += up256(bytes); return at; };
  const size_t o_dig0 = take((size_t)P * dig0_stride), o_dlen = take(4ull * P), o_val = take(32ull * P * pv->m), o_gam = take(32ull * P * pv->m);
  const size_t o_seeds = take(seed_off[P] - seed_off[0] + 16), o_soff = take(8ull * (P + 1)), o_ooff = take(8ull * (P + 1));
  const size_t in_bytes = o;                                   // everything above is uploaded in one copy
  const size_t o_tr = take((size_t)P * tr_stride), o_trlen = take(4ull * P);
  const size_t o_slr = take(32ull * P * (2 * n + 1)), o_alpha = take(32ull * P), o_chal = take(128ull * P), o_tau = take(64ull * P), o_tsc = take(128ull * P);
  const size_t o_res = take(160ull * P), o_xs = take(32ull * P * k), o_xr = take(64ull * P);
  const size_t o_a = take(32ull * P * n), o_b = take(32ull * P * n), o_cg = take(32ull * P * n), o_hf = take(32ull * P * n);
  const size_t o_jsc = take(32ull * P * (2 * n + 2)), o_jout = take(144ull * 2 * P), o_pts = take(64ull * P * npt);
  const size_t o_out = take(total_out + 16);
  if (o > pv->buf_bytes) {
    if (pv->buf) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(pv->buf)); pv->buf = nullptr; pv->buf_bytes = 0; }
    HIPCHK(ctx, hipMalloc(&pv->buf, o + o / 8));
    pv->buf_bytes = o + o / 8;
  }
  const size_t pin_need = std::max(in_bytes, (size_t)total_out);      // (the output half is only used when `out` is pageable)
  if (pin_need > pv->pin_bytes) {
    if (pv->pin) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipHostFree(pv->pin)); pv->pin = nullptr; pv->pin_bytes = 0; }
    HIPCHK(ctx, hipHostMalloc(&pv->pin, pin_need + pin_need / 8, hipHostMallocDefault));
    pv->pin_bytes = pin_need + pin_need / 8;
  }
  char *d = (char *)pv->buf, *hp = (char *)pv->pin;
  // ---- inputs into the staging buffer
  memset(hp, 0, in_bytes);
  for (u32 p = 0; p < P; p++) {
    char *dg = hp + o_dig0 + (size_t)p * dig0_stride;                 // (dig0_stride >= 4 ceil(len / 3) + 1)
    size_t tl = rpp_host::b64_into(dg, seeds + seed_off[p], seed_off[p + 1] - seed_off[p]);
    dg[tl++] = '&';
    ((u32 *)(hp + o_dlen))[p] = (u32)tl;
    ((uint64_t *)(hp + o_soff))[p] = seed_off[p] - seed_off[0];
    ((uint64_t *)(hp + o_ooff))[p] = out_off[p];
  }
  ((uint64_t *)(hp + o_soff))[P] = seed_off[P] - seed_off[0];
  ((uint64_t *)(hp + o_ooff))[P] = out_off[P];
  memcpy(hp + o_val, values, 32ull * P * pv->m);
  memcpy(hp + o_gam, gammas, 32ull * P * pv->m);
  if (seed_off[P] > seed_off[0]) memcpy(hp + o_seeds, seeds + seed_off[0], seed_off[P] - seed_off[0]);
  hipStream_t st = ctx->stream;
  HIPCHK(ctx, hipMemcpyAsync(d, hp, in_bytes, hipMemcpyHostToDevice, st));
  rpp::Batch B;
  memset(&B, 0, sizeof(B));
  B.P = P; B.n = n; B.k = k; B.nb = pv->nb; B.m = pv->m; B.table = rpp::Tab{pv->table, pv->tw, pv->wt, pv->bt};
  B.dig0 = (const unsigned char *)(d + o_dig0); B.dig0_stride = dig0_stride; B.dig0_len = (const u32 *)(d + o_dlen);
  B.values = (const u32 *)(d + o_val); B.gammas = (const u32 *)(d + o_gam);
  B.ip_prefix = pv->d_ip_prefix; B.ip_prefix_len = (u32)pv->ip_prefix.size(); B.x_ip = pv->x_ip;
  memcpy(B.u_new, pv->u_new, 64);
  B.tr = (unsigned char *)(d + o_tr); B.tr_stride = tr_stride; B.tr_len = (u32 *)(d + o_trlen);
  B.slr = (u32 *)(d + o_slr); B.alpha = (u32 *)(d + o_alpha); B.chal = (u32 *)(d + o_chal); B.tau = (u32 *)(d + o_tau); B.tsc = (u32 *)(d + o_tsc);
  B.res = (u32 *)(d + o_res); B.xs = (u32 *)(d + o_xs); B.xr = (u32 *)(d + o_xr);
  B.a = (u32 *)(d + o_a); B.b = (u32 *)(d + o_b); B.cg = (u32 *)(d + o_cg); B.hf = (u32 *)(d + o_hf);
  B.jsc = (u32 *)(d + o_jsc); B.jout = (u32 *)(d + o_jout); B.pts = (u32 *)(d + o_pts);
  hipEvent_t *ev = pv->ev;
  auto blocks = [](uint64_t threads, u32 per) { return dim3((u32)((threads + per - 1) / per)); };
  // Round 6 experiment (option "prover_split", off): a large batch as TWO halves on the ctx's two lanes.  Between two multi-scalar
  // multiplications a half is a chain of one-lane-per-proof kernels (transcript hash, inversion: a quarter of the SIMDs busy) and short vector
  // kernels, 0.45 ms of a 2.3 ms round, which the other half's additions could hide.  Measured twice -- the halves free-running, and their
  // multiplications alternating through events (the form below) -- 19.2-19.4 ms per 2^14 proofs either way against 19.1-19.2 unsplit
  // (profiles/r06_batch_prover_table_bits.txt): the chains' instructions are issue slots the additions lose.
  struct Half { rpp::Batch H; hipStream_t st; u32 p0; bool rec; };
  auto make_half = [&](u32 p0, u32 cnt, hipStream_t hs, bool rec) {
    Half h;
    h.H = B; h.st = hs; h.p0 = p0; h.rec = rec;
    rpp::Batch &H = h.H;
    H.P = cnt;
    H.dig0 += (size_t)p0 * dig0_stride; H.dig0_len += p0;
    H.values += 8ull * p0 * pv->m; H.gammas += 8ull * p0 * pv->m;
    H.tr += (size_t)p0 * tr_stride; H.tr_len += p0;
    H.slr += 8ull * p0 * (2 * n + 1); H.alpha += 8ull * p0; H.chal += 32ull * p0; H.tau += 16ull * p0; H.tsc += 32ull * p0;
    H.res += 40ull * p0; H.xs += 8ull * p0 * k; H.xr += 16ull * p0;
    H.a += 8ull * p0 * n; H.b += 8ull * p0 * n; H.cg += 8ull * p0 * n; H.hf += 8ull * p0 * n;
    H.jsc += 8ull * p0 * (2 * n + 2); H.jout += 36ull * 2 * p0; H.pts += 16ull * p0 * npt;
    return h;
  };
  // step s of a half: 0 A and S; 1 y, z, T1, T2; 2 x, the vectors, P_new; 3 .. 2 + k the rounds; 3 + k the wire bytes.  Every step is
  // [kernels before] [ONE multi-scalar multiplication] [kernels behind]; with two halves (`other` set) the multiplication waits for the other
  // half's previous one and is followed by an event of its own: the additions of the two halves alternate on the chip, and a half's
  // one-lane-per-proof chains (hash, inversion) run beside the OTHER half's additions instead of beside their own twin's.
  const u32 per_block = 256u / n;                        // proofs per block of the n-lanes-per-proof kernels
  const u32 nsteps = 4 + k;
  auto step = [&](Half &h, u32 sidx, hipEvent_t mine, hipEvent_t other) {
    rpp::Batch &H = h.H;
    hipStream_t hs = h.st;
    const u32 Pc = H.P;
    auto msm = [&](u32 njobs, u32 ntypes, u32 T, u32 base_off, const u32 *scalars, u32 stride, int gl) {
      rpp::MsmJobs J;
      J.njobs = njobs; J.ntypes = ntypes; J.T = T; J.bases = pv->bases + base_off; J.scalars = scalars; J.stride = stride; J.out = H.jout;
      const uint64_t threads = (uint64_t)njobs << gl;
      if (other) (void)hipStreamWaitEvent(hs, other, 0);
      if (gl == 4) hipLaunchKernelGGL(rpp::k_pv_msm<4>, blocks(threads, 256), dim3(256), 0, hs, J, H.table);
      else hipLaunchKernelGGL(rpp::k_pv_msm<1>, blocks(threads, 256), dim3(256), 0, hs, J, H.table);
      if (mine) (void)hipEventRecord(mine, hs);
    };
    auto affine = [&](u32 count, u32 per, u32 slot0, u32 stp) {
      hipLaunchKernelGGL(rpp::k_pv_affine, blocks(count, 256), dim3(256), 0, hs, (const u32 *)H.jout, count, per, H.pts, npt, slot0, stp);
    };
    if (sidx == 0) {                                     // A, S (rangeproof_prover.py:40-59)
      if (h.rec) (void)hipEventRecord(ev[0], hs);
      hipLaunchKernelGGL(rpp::k_pv_blind, blocks((uint64_t)Pc * (2 * n + 2), 256), dim3(256), 0, hs, H);
      hipLaunchKernelGGL(rpp::k_pv_commit_A, blocks((uint64_t)Pc * 16, 256), dim3(256), 0, hs, H, H.jout);
      affine(Pc, 1, PV_PT_A, 0);
      msm(Pc, 1, 2 * n + 1, pv->off_S, H.slr, 2 * n + 1, 4);
      affine(Pc, 1, PV_PT_S, 0);
      if (h.rec) (void)hipEventRecord(ev[1], hs);
    } else if (sidx == 1) {                              // y, z, tau1, tau2; t1, t2; T1, T2 (:60-67)
      hipLaunchKernelGGL(rpp::k_pv_chal_yz, blocks(Pc, 64), dim3(64), 0, hs, H);
      hipLaunchKernelGGL(rpp::k_pv_poly, dim3((Pc + per_block - 1) / per_block), dim3(256), 0, hs, H);
      msm(2 * Pc, 1, 2, pv->off_T, H.tsc, 2, 1);
      affine(2 * Pc, 2, PV_PT_T1, 1);
      if (h.rec) (void)hipEventRecord(ev[2], hs);
    } else if (sidx == 2) {                              // x; l, r, t_hat, taux, mu; P_new (:68-90; inner_product_prover.py:33-37)
      hipLaunchKernelGGL(rpp::k_pv_final_chal, blocks(Pc, 64), dim3(64), 0, hs, H);
      hipLaunchKernelGGL(rpp::k_pv_final_wide, dim3((Pc + per_block - 1) / per_block), dim3(256), 0, hs, H);
      msm(Pc, 1, 2 * n + 1, pv->off_P, H.jsc, 2 * n + 1, 4);
      affine(Pc, 1, PV_PT_PNEW, 0);
      if (h.rec) (void)hipEventRecord(ev[3], hs);
      hipLaunchKernelGGL(rpp::k_pv_round_wide, dim3((Pc + per_block - 1) / per_block), dim3(256), 0, hs, H, 0u, 1u);
    } else if (sidx < 3 + k) {                           // a round of Protocol 2 (inner_product_prover.py:94-110)
      const u32 r = sidx - 3;
      msm(2 * Pc, 2, n + 1, pv->off_round + r * 2 * (n + 1), H.jsc, n + 1, 4);
      affine(2 * Pc, 2, 6 + r, k);
      hipLaunchKernelGGL(rpp::k_pv_round_chal, blocks(Pc, 64), dim3(64), 0, hs, H, r);
      hipLaunchKernelGGL(rpp::k_pv_round_wide, dim3((Pc + per_block - 1) / per_block), dim3(256), 0, hs, H, r, 0u);
    } else {
      if (h.rec) (void)hipEventRecord(ev[4], hs);
      hipLaunchKernelGGL(rpp::k_pv_emit, blocks(Pc, 64), dim3(64), 0, hs, H, (const unsigned char *)(d + o_seeds), (const uint64_t *)(d + o_soff) + h.p0,
                         (unsigned char *)(d + o_out), (const uint64_t *)(d + o_ooff) + h.p0, (u32)wire_fmt);
    }
  };
  const bool split = ctx->opt_prover_split && P >= 2u * (ctx->opt_prover_split > 1 ? (u32)ctx->opt_prover_split : PV_SPLIT_MIN);
  if (split) {
    int rc = ensure_lane(ctx, 1);
    if (rc) return rc;
    const u32 P0 = (P + 1) / 2;
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, st));                  // (the inputs are up)
    HIPCHK(ctx, hipStreamWaitEvent(ctx->stream1, ctx->ev_fork, 0));
    Half ha = make_half(0, P0, st, true), hb = make_half(P0, P - P0, ctx->stream1, false);
    for (u32 sidx = 0; sidx < nsteps; sidx++) {
      step(ha, sidx, pv->ev_chain[0], sidx ? pv->ev_chain[1] : nullptr);          // (queued alternately: an event is recorded before it is waited for)
      step(hb, sidx, pv->ev_chain[1], pv->ev_chain[0]);
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->stream1));
    HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
  } else {
    Half ha = make_half(0, P, st, true);
    for (u32 sidx = 0; sidx < nsteps; sidx++) step(ha, sidx, nullptr, nullptr);
  }
  (void)hipEventRecord(ev[5], st);
  hipError_t e = hipGetLastError();
  // the proofs go straight into the caller's buffer when it is page-locked (bpmi_host_alloc: what BatchRangeProver hands in), else
  // through the prover's staging buffer and one host copy (18 MB for 2^14 64-bit proofs: 2-3 ms of a 27 ms batch)
  bool direct = false;
  {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, out) == hipSuccess && attr.type == hipMemoryTypeHost) direct = true;
    else (void)hipGetLastError();                              // an ordinary host pointer is "invalid value" to the query
  }
  if (e == hipSuccess) e = hipMemcpyAsync(direct ? (void *)out : (void *)hp, d + o_out, total_out, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipEventRecord(ev[6], st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return fail(ctx, BPMI_E_HIP, std::string("bpmi_rp_prove_batch: ") + hipGetErrorString(e));
  if (!direct) memcpy(out, hp, total_out);
  for (int i = 0; i < 6; i++) { float ms = 0; (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]); pv->last_ms[i] = ms; }
  { float ms = 0; (void)hipEventElapsedTime(&ms, ev[0], ev[6]); pv->last_ms[6] = ms; }
  return BPMI_OK;
}

int bpmi_rp_prover_last_ms(const bpmi_rp_prover *pv, double ms[7]) {
  if (!pv || !ms) return BPMI_E_ARG;
  for (int i = 0; i < 7; i++) ms[i] = pv->last_ms[i];
  return BPMI_OK;
}

}  // extern "C"
