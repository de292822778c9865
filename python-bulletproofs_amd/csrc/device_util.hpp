// device_util.hpp -- part of libbpmi (included by bpmi.hip; one translation unit).
// Device-side load/store helpers, the multi-segment input descriptor.
#pragma once

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
  // Optional half-block selection per segment: with hlog[s] = h < 32 the logical element j of
  // segment s is the physical element ((j >> h) << (h + 1)) | (phase[s] << h) | (j & (2^h - 1)),
  // i.e. only the lower (phase 0) or upper (phase 1) half of every block of 2^(h+1) elements.
  // The deferred-fold MSMs of the IPA use it to skip the half of the scalars that is zero by
  // construction.  hlog[s] >= 32: dense (the default, set by segs_init).
  u32 hlog[3];
  u32 phase[3];
  // GLV (scalar.hpp glv_split; prepared per MSM by k_glv_prepare, csrc/msm_kernels.hpp): when glv_sub is set the MSM runs over
  // 2 * total VIRTUAL pairs -- virtual pair 2i is (P_i, |k1_i|), 2i + 1 is (lambda P_i, |k2_i|) -- with 128-bit magnitudes
  // glv_sub[4 v ..] and signs glv_neg[v]; lambda P_i = (beta x_i, y_i), the x coordinates precomputed in glv_bx[8 i ..].
  const u32 *glv_sub;
  const unsigned char *glv_neg;
  const u32 *glv_bx;
};
static inline Segs segs_init() {
  Segs s;
  memset(&s, 0, sizeof(s));
  s.hlog[0] = s.hlog[1] = s.hlog[2] = 0xFFu;
  return s;
}
__device__ __forceinline__ u32 seg_phys(const Segs &s, int k, u32 j) {
  const u32 h = s.hlog[k];
  if (h >= 32u) return j;
  return ((j >> h) << (h + 1u)) | (s.phase[k] << h) | (j & ((1u << h) - 1u));
}

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
  if (i < s.n[0]) return s.pts[0] + 16ull * seg_phys(s, 0, i);
  i -= s.n[0];
  if (i < s.n[1]) return s.pts[1] + 16ull * seg_phys(s, 1, i);
  i -= s.n[1];
  return s.pts[2] + 16ull * seg_phys(s, 2, i);
}
__device__ __forceinline__ const u32 *seg_scalar(const Segs &s, u32 i) {
  if (i < s.n[0]) return s.sc[0] + 8ull * seg_phys(s, 0, i);
  i -= s.n[0];
  if (i < s.n[1]) return s.sc[1] + 8ull * seg_phys(s, 1, i);
  i -= s.n[1];
  return s.sc[2] + 8ull * seg_phys(s, 2, i);
}
// the 16 words (x, y) of the point an MSM entry refers to: a logical index, or with GLV a virtual one
template <bool GLV> __device__ __forceinline__ void load_entry_point(u32 w[16], const Segs &s, u32 idx) {
  if (GLV) {
    const u32 i = idx >> 1;
    const u32 *p = seg_point(s, i);
    load_words8(w, (idx & 1u) ? s.glv_bx + 8ull * i : p);
    load_words8(w + 8, p + 8);
  } else {
    load_words16(w, seg_point(s, idx));
  }
}
__device__ __forceinline__ void load_affine(affine &P, const u32 *p) {
  u32 w[16];
  load_words16(w, p);
  affine_from_words(P, w);
}
__device__ __forceinline__ void load_affine_y(fe &y, const u32 *p) {    // the y half of a 64-byte wire point
  const uint4 *q = reinterpret_cast<const uint4 *>(p);
  const uint4 a = q[2], b = q[3];
  const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  fe_from_words(y, w);
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
