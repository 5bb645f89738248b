/*
 * ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin extern "C" door onto the REFERENCE's own CPU/SSE implementation, used
 * to pin oracle/viso_oracle.c and to generate tests/golden/.  It is compiled
 * together with the reference sources *where they lie* under
 * /root/reference/src (see oracle/Makefile, target _ref); nothing from the
 * reference is copied into this repository and the resulting
 * oracle/_ref/libviso_ref.so is git-ignored.
 *
 * The SSE functions are private members of Matcher (src/matcher.h:150-238);
 * the harness reaches them by re-declaring `private` as `public` for the one
 * include below (standard headers are included first so they are unaffected).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <emmintrin.h>
#include <algorithm>
#include <iostream>
#include <limits>
#include <cmath>
#include <vector>

/* Matcher::findMatch and computeDescriptor are `inline` members defined in
 * matcher.cpp (src/matcher.cpp:216,470), so no other translation unit can link
 * to them: the harness therefore compiles the reference's matcher.cpp as part
 * of THIS translation unit, straight from the read-only mount (-I$(REF)/src). */
#define private public
#include "matcher.cpp"
#include "viso_stereo.h"   /* estimateMotion is private (src/viso_stereo.h:75) */
#include "viso_mono.h"     /* ... and src/viso_mono.h:74 */
#undef private
#include "filter.h"

namespace {

struct HParams { /* same layout as vo_params / vh_params */
  int32_t nms_n, nms_tau, match_binsize, match_radius, match_disp_tolerance,
      outlier_disp_tolerance, outlier_flow_tolerance, multi_stage, half_resolution, refinement;
  double f, cu, cv, base;
};

Matcher::parameters to_ref(const HParams *h) {
  Matcher::parameters p;
  p.nms_n = h->nms_n;
  p.nms_tau = h->nms_tau;
  p.match_binsize = h->match_binsize;
  p.match_radius = h->match_radius;
  p.match_disp_tolerance = h->match_disp_tolerance;
  p.outlier_disp_tolerance = h->outlier_disp_tolerance;
  p.outlier_flow_tolerance = h->outlier_flow_tolerance;
  p.multi_stage = h->multi_stage;
  p.half_resolution = h->half_resolution;
  p.refinement = h->refinement;
  p.f = h->f; p.cu = h->cu; p.cv = h->cv; p.base = h->base;
  return p;
}

/* Matcher holds a 672 KB p_matched_2 array: keep it on the heap. */
struct Holder {
  Matcher *m;
  explicit Holder(const HParams *h) : m(new Matcher(to_ref(h))) {}
  ~Holder() { delete m; }
};

uint8_t *aligned_copy(const uint8_t *src, size_t n, size_t slack) {
  uint8_t *p = (uint8_t *)_mm_malloc(n + slack, 16);
  memset(p, 0, n + slack);
  memcpy(p, src, n);
  return p;
}

}  // namespace

extern "C" {

/* filter::sobel5x5 / blob5x5 / checkerboard5x5 (src/filter.cpp:418,445,433)
 * on a stride-bpl image.  Outputs are raw planes including the SSE border
 * artefacts; compare on the valid interior only. */
void ref_filters(const uint8_t *I, int32_t bpl, int32_t H, uint8_t *du, uint8_t *dv,
                 int16_t *f1, int16_t *f2) {
  const size_t n = (size_t)bpl * H;
  uint8_t *Ia = aligned_copy(I, n, 64);
  /* the row passes write 2 bytes past the plane (filter.cpp:86,125) */
  uint8_t *a = (uint8_t *)_mm_malloc(n + 64, 16), *b = (uint8_t *)_mm_malloc(n + 64, 16);
  int16_t *c = (int16_t *)_mm_malloc(2 * n + 64, 16), *d = (int16_t *)_mm_malloc(2 * n + 64, 16);
  memset(a, 0, n + 64); memset(b, 0, n + 64); memset(c, 0, 2 * n + 64); memset(d, 0, 2 * n + 64);
  filter::sobel5x5(Ia, a, b, bpl, H);
  filter::blob5x5(Ia, c, bpl, H);
  filter::checkerboard5x5(Ia, d, bpl, H);
  if (du) memcpy(du, a, n);
  if (dv) memcpy(dv, b, n);
  if (f1) memcpy(f1, c, 2 * n);
  if (f2) memcpy(f2, d, 2 * n);
  _mm_free(Ia); _mm_free(a); _mm_free(b); _mm_free(c); _mm_free(d);
}

/* Matcher::computeFeatures (src/matcher.cpp:585-672). */
int32_t ref_compute_features(const HParams *hp, const uint8_t *I, const int32_t *dims,
                             int32_t *max1, int32_t cap1, int32_t *num1, int32_t *max2,
                             int32_t cap2, int32_t *num2, uint8_t *du, uint8_t *dv) {
  Holder h(hp);
  const size_t n = (size_t)dims[2] * dims[1];
  uint8_t *Ia = aligned_copy(I, n, 64);
  int32_t *m1 = 0, *m2 = 0, n1 = 0, n2 = 0;
  uint8_t *I_du = 0, *I_dv = 0, *I_du_full = 0, *I_dv_full = 0;
  h.m->computeFeatures(Ia, dims, m1, n1, m2, n2, I_du, I_dv, I_du_full, I_dv_full);
  if (num1) *num1 = n1;
  if (num2) *num2 = n2;
  if (max1 && m1) memcpy(max1, m1, sizeof(int32_t) * 12 * (size_t)std::min(n1, cap1));
  if (max2 && m2) memcpy(max2, m2, sizeof(int32_t) * 12 * (size_t)std::min(n2, cap2));
  int32_t dm[3] = {dims[0], dims[1], dims[2]};
  if (hp->half_resolution) h.m->getHalfResolutionDimensions(dims, dm);
  if (du) memcpy(du, I_du, (size_t)dm[2] * dm[1]);
  if (dv) memcpy(dv, I_dv, (size_t)dm[2] * dm[1]);
  if (m1) _mm_free(m1);
  if (m2) _mm_free(m2);
  _mm_free(I_du); _mm_free(I_dv);
  if (I_du_full) _mm_free(I_du_full);
  if (I_dv_full) _mm_free(I_dv_full);
  _mm_free(Ia);
  return 0;
}

/* Matcher::createHalfResolutionImage (src/matcher.cpp:572-583). */
void ref_half_resolution(const HParams *hp, const uint8_t *I, const int32_t *dims,
                         int32_t *dims_half, uint8_t *out) {
  Holder h(hp);
  h.m->getHalfResolutionDimensions(dims, dims_half);
  if (!out) return;
  uint8_t *r = h.m->createHalfResolutionImage(const_cast<uint8_t *>(I), dims);
  /* padding columns are uninitialised in the reference: copy the valid ones */
  for (int32_t v = 0; v < dims_half[1]; v++)
    memcpy(out + (size_t)v * dims_half[2], r + (size_t)v * dims_half[2], dims_half[0]);
  _mm_free(r);
}

/* Matcher::createIndexVector (src/matcher.cpp:194-214), flattened to CSR. */
void ref_create_index(const HParams *hp, const int32_t *m, int32_t n, int32_t u_bin_num,
                      int32_t v_bin_num, int32_t *bin_start, int32_t *list) {
  Holder h(hp);
  const int32_t nb = 4 * u_bin_num * v_bin_num;
  std::vector<int32_t> *k = new std::vector<int32_t>[nb];
  h.m->createIndexVector(const_cast<int32_t *>(m), n, k, u_bin_num, v_bin_num);
  int32_t pos = 0;
  for (int32_t b = 0; b < nb; b++) {
    bin_start[b] = pos;
    for (size_t t = 0; t < k[b].size(); t++) list[pos++] = k[b][t];
  }
  bin_start[nb] = pos;
  delete[] k;
}

/* Matcher::findMatch (src/matcher.cpp:216-272) for every i1 in [0,n1).
 * dims = {W,H,*}; feature arrays must be 16-byte aligned copies. */
void ref_match_all(const HParams *hp, const int32_t *dims, const int32_t *m1, int32_t n1,
                   const int32_t *m2, int32_t n2, double u_, double v_, int32_t *best) {
  Holder h(hp);
  const int32_t ubn = (int32_t)ceil((float)dims[0] / (float)hp->match_binsize);
  const int32_t vbn = (int32_t)ceil((float)dims[1] / (float)hp->match_binsize);
  int32_t *a1 = (int32_t *)aligned_copy((const uint8_t *)m1, 48 * (size_t)n1, 64);
  int32_t *a2 = (int32_t *)aligned_copy((const uint8_t *)m2, 48 * (size_t)n2, 64);
  std::vector<int32_t> *k2 = new std::vector<int32_t>[4 * ubn * vbn];
  h.m->createIndexVector(a2, n2, k2, ubn, vbn);
  for (int32_t i = 0; i < n1; i++) {
    int32_t mi = -12345;
    h.m->findMatch(a1, i, a2, 12, k2, ubn, vbn, 0, mi, 0, true, false, u_, v_);
    best[i] = mi;
  }
  delete[] k2;
  _mm_free(a1); _mm_free(a2);
}

/* Matcher::matching (src/matcher.cpp:274-344): flow is the only method the
 * reference implements.  The caller must provide room for n1c records (the
 * reference performs no bounds check). */
int32_t ref_matching_flow(const HParams *hp, const int32_t *dims, const int32_t *m1p, int32_t n1p,
                          const int32_t *m1c, int32_t n1c, void *out, int32_t cap, int32_t *n_out) {
  if (cap < n1c) return -1;
  Holder h(hp);
  h.m->dims_c[0] = dims[0]; h.m->dims_c[1] = dims[1]; h.m->dims_c[2] = dims[2];
  int32_t *ap = (int32_t *)aligned_copy((const uint8_t *)m1p, 48 * (size_t)n1p, 64);
  int32_t *ac = (int32_t *)aligned_copy((const uint8_t *)m1c, 48 * (size_t)n1c, 64);
  int32_t cnt = 0;
  h.m->matching(ap, 0, ac, 0, n1p, 0, n1c, 0, (Matcher::p_match *)out, cnt, 0, false, 0);
  *n_out = cnt;
  _mm_free(ap); _mm_free(ac);
  return 0;
}

/* Matcher::bucketFeatures (src/matcher.cpp:140-187).  The reference keeps
 * buckets[126][256] on the stack and p_matched_2[POINT_L]: the caller must
 * stay within 126 buckets x 256 matches and POINT_L matches. */
int32_t ref_bucket_features(const HParams *hp, void *pm, int32_t n, int32_t max_features,
                            float bw, float bh) {
  if (n > POINT_L) return -1;
  Holder h(hp);
  memcpy(h.m->p_matched_2, pm, sizeof(Matcher::p_match) * (size_t)n);
  h.m->p_matched_2_cnt = n;
  h.m->bucketFeatures(max_features, bw, bh);
  memcpy(pm, h.m->p_matched_2, sizeof(Matcher::p_match) * (size_t)h.m->p_matched_2_cnt);
  return h.m->p_matched_2_cnt;
}

/* removeOutliers (src/remove_outliers.cpp:4-94), the step Matcher::matchFeatures
 * runs right after matching (src/matcher.cpp:108).  Works on a POINT_L-sized
 * copy because the reference's signature is p_match[POINT_L]; its working set
 * (Delaunator + copy + support counters, about 2 MB) lives on the stack. */
int32_t ref_remove_outliers(void *pm, int32_t n) {
  if (n > POINT_L) return -1;
  Matcher::p_match *buf = new Matcher::p_match[POINT_L];
  memcpy(buf, pm, sizeof(Matcher::p_match) * (size_t)n);
  int32_t cnt = n;
  removeOutliers(buf, cnt);
  memcpy(pm, buf, sizeof(Matcher::p_match) * (size_t)cnt);
  delete[] buf;
  return cnt;
}

/* delaunator::Delaunator (src/delaunator.cpp:183-407) on its own: returns the
 * number of triangle corners written (3 per triangle). */
int32_t ref_delaunay(const float *xy, int32_t n, int32_t *tri, int32_t cap) {
  if (n > POINT_L || n < 3) return -1;
  delaunator::Delaunator *d = new delaunator::Delaunator();
  for (int32_t i = 0; i < n; i++) d->read_point(xy[2 * i], xy[2 * i + 1]);
  d->delaunat();
  const int32_t cnt = d->triangles_cnt;
  if (cnt > cap) { delete d; return -2; }
  memcpy(tri, d->triangles, sizeof(int32_t) * (size_t)cnt);
  delete d;
  return cnt;
}

int32_t ref_sizeof_p_match(void) { return (int32_t)sizeof(Matcher::p_match); }
int32_t ref_point_l(void) { return POINT_L; }


/* VisualOdometryStereo::estimateMotion (src/viso_stereo.cpp:54-157) on
 * caller-supplied matches, with a freshly constructed object (srand(0) in the
 * constructor, src/viso.cpp:35, so the samples are rand()'s first 3*ransac_iters values).
 * ego: {int32 ransac_iters, int32 reweighting, double inlier_threshold, f, cu, cv, base}.
 * Returns 1/0 (success / empty result); tr[6], inliers (capacity n), *n_inliers. */
struct HEgo { int32_t ransac_iters, reweighting; double inlier_threshold, f, cu, cv, base; };
int32_t ref_estimate_motion_stereo(const void *ego, const void *pm, int32_t n, double *tr, int32_t *inliers, int32_t *n_inliers) {
  const HEgo *e = (const HEgo *)ego;
  VisualOdometryStereo::parameters param;
  param.ransac_iters = e->ransac_iters; param.reweighting = e->reweighting != 0; param.inlier_threshold = e->inlier_threshold;
  param.calib.f = e->f; param.calib.cu = e->cu; param.calib.cv = e->cv; param.base = e->base;
  VisualOdometryStereo *vo = new VisualOdometryStereo(param);
  std::vector<Matcher::p_match> v((const Matcher::p_match *)pm, (const Matcher::p_match *)pm + n);
  std::vector<double> r = vo->estimateMotion(v);
  std::vector<int32_t> inl = vo->getInlierIndices();
  *n_inliers = (int32_t)inl.size();
  for (size_t i = 0; i < inl.size(); i++) inliers[i] = inl[i];
  for (int i = 0; i < 6; i++) tr[i] = r.size() == 6 ? r[i] : 0.0;
  delete vo;
  return r.size() == 6 ? 1 : 0;
}

/* SURVEY 8(f-4), mono half: VisualOdometryMono::estimateMotion (src/viso_mono.cpp:41-160) on
 * caller-supplied flow matches, with a freshly constructed object (srand(0) in the constructor, so
 * the samples are getRandomSample(N,8) on rand()'s first 8*ransac_iters values).
 * mono: {int32 ransac_iters, pad; double inlier_threshold, motion_threshold, height, pitch, f, cu, cv}. */
struct HMono { int32_t ransac_iters, pad; double inlier_threshold, motion_threshold, height, pitch, f, cu, cv; };
int32_t ref_estimate_motion_mono(const void *mono, const void *pm, int32_t n, double *tr, int32_t *inliers, int32_t *n_inliers) {
  const HMono *e = (const HMono *)mono;
  VisualOdometryMono::parameters param;
  param.ransac_iters = e->ransac_iters; param.inlier_threshold = e->inlier_threshold; param.motion_threshold = e->motion_threshold;
  param.height = e->height; param.pitch = e->pitch;
  param.calib.f = e->f; param.calib.cu = e->cu; param.calib.cv = e->cv;
  VisualOdometryMono *vo = new VisualOdometryMono(param);
  std::vector<Matcher::p_match> v((const Matcher::p_match *)pm, (const Matcher::p_match *)pm + n);
  std::vector<double> r = vo->estimateMotion(v);
  std::vector<int32_t> inl = vo->getInlierIndices();
  *n_inliers = (int32_t)inl.size();
  for (size_t i = 0; i < inl.size(); i++) inliers[i] = inl[i];
  for (int i = 0; i < 6; i++) tr[i] = r.size() == 6 ? r[i] : 0.0;
  delete vo;
  return r.size() == 6 ? 1 : 0;
}

/* Matrix::svd (src/matrix.cpp:579-802) on a row-major m x n array: U2 m x m, W min(m,n), V n x n. */
void ref_svd(const double *a, int32_t m, int32_t n, double *U2, double *W, double *V) {
  Matrix A(m, n, a), U, S, Vm;
  A.svd(U, S, Vm);
  for (int32_t i = 0; i < m; i++) for (int32_t j = 0; j < m; j++) U2[i * m + j] = U.val[i][j];
  for (int32_t i = 0; i < S.m; i++) W[i] = S.val[i][0];
  for (int32_t i = 0; i < n; i++) for (int32_t j = 0; j < n; j++) V[i * n + j] = Vm.val[i][j];
}

}  // extern "C"
