/*
 * viso_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's detect+match hot path (the
 * libviso2-style SSE Matcher of Chang-Tun-Yu/HLS-final-Visual-Odometry).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this; the product path (libviso_hip.so) never does.
 *
 * Pinning: every function marked [pinned] is checked bit-for-bit against the
 * reference's own sources compiled in-container (oracle/_ref, see
 * oracle/Makefile + oracle/ref_harness.cpp) and against the committed golden
 * vectors in tests/golden/ that were generated from that build.
 * Functions marked [unpinned] restate stock-libviso2 behaviour that is absent
 * from the reference tree (stereo / quad compositions): "parity unpinned".
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef VISO_ORACLE_H
#define VISO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* POD mirror of Matcher::parameters (src/matcher.h:45-72), field for field. */
typedef struct vo_params {
  int32_t nms_n;
  int32_t nms_tau;
  int32_t match_binsize;
  int32_t match_radius;
  int32_t match_disp_tolerance;
  int32_t outlier_disp_tolerance;
  int32_t outlier_flow_tolerance;
  int32_t multi_stage;
  int32_t half_resolution;
  int32_t refinement;
  double f, cu, cv, base;
} vo_params;

/* Mirror of Matcher::p_match (src/matcher.h:89-104): 48 bytes. */
typedef struct vo_p_match {
  float u1p, v1p; int32_t i1p;
  float u2p, v2p; int32_t i2p;
  float u1c, v1c; int32_t i1c;
  float u2c, v2c; int32_t i2c;
} vo_p_match;

/* Matcher::parameters() defaults (src/matcher.h:60-71). */
void vo_default_params(vo_params *p);

/* [pinned] Exact-integer 5x5 filters on the valid interior:
 *   du,dv : src/filter.cpp:418-426 (+:269-321,:132-171,:79-127)
 *   f1    : src/filter.cpp:445-467 (blob)
 *   f2    : src/filter.cpp:433-438 (+:323-370) (checkerboard, code sign)
 * Planes have stride bpl; pixels outside the valid interior
 * (x in [2,bpl-3], y in [2,H-3]; f1 additionally x,y>=3) are written 0.
 * Any of the four outputs may be NULL. */
void vo_filters(const uint8_t *I, int32_t bpl, int32_t H,
                uint8_t *du, uint8_t *dv, int16_t *f1, int16_t *f2);

/* [pinned] Neubeck/Van Gool NMS on f1,f2 (src/matcher.cpp:366-468).
 * out4 receives {u,v,val,c} per maximum, in reference output order.
 * Returns the number found (may exceed cap; only cap are written). */
int32_t vo_nms(const int16_t *f1, const int16_t *f2, const int32_t dims[3],
               int32_t nms_n, int32_t nms_tau, int32_t *out4, int32_t cap);

/* [pinned] 32-byte descriptor (src/matcher.cpp:470-514). */
void vo_descriptor(const uint8_t *du, const uint8_t *dv, int32_t bpl,
                   int32_t u, int32_t v, uint8_t desc[32]);

/* [pinned] 2x2 box half-resolution image (src/matcher.cpp:566-583).
 * dims_half[3] is filled; out must hold dims_half[2]*dims_half[1] bytes
 * (query with out==NULL first). */
void vo_half_resolution(const uint8_t *I, const int32_t dims[3],
                        int32_t dims_half[3], uint8_t *out);

/* [pinned] Matcher::computeFeatures (src/matcher.cpp:585-672).
 * max2 (dense) / max1 (sparse, multi_stage only) receive int32[12] records
 * {u*s, v*s, 0, c, d1..d8}. num1/num2 return the true counts even when they
 * exceed the capacities (records beyond cap are dropped).  max1/num1 and the
 * du/dv plane outputs (matching-resolution planes, stride dims_matching[2])
 * may be NULL. Returns 0, or -1 on bad dims. */
int32_t vo_compute_features(const vo_params *p, const uint8_t *I,
                            const int32_t dims[3],
                            int32_t *max1, int32_t cap1, int32_t *num1,
                            int32_t *max2, int32_t cap2, int32_t *num2,
                            uint8_t *du, uint8_t *dv);

/* [pinned] Matcher::createIndexVector (src/matcher.cpp:194-214) as CSR:
 * bin_start has 4*u_bin_num*v_bin_num+1 entries, list has n entries. */
void vo_create_index(const int32_t *m, int32_t n, int32_t binsize,
                     int32_t u_bin_num, int32_t v_bin_num,
                     int32_t *bin_start, int32_t *list);

/* [pinned] Matcher::findMatch (src/matcher.cpp:216-272), including the
 * optional u_,v_ distance term (pass u_<0 to disable, the only form the
 * reference's callers use).  flow==0 narrows the v-range to
 * +-match_disp_tolerance ([unpinned], stock libviso2; the reference ignores
 * the flag). Returns min_ind (0 when no candidate). */
int32_t vo_find_match(const vo_params *p, const int32_t *m1, int32_t i1,
                      const int32_t *m2, const int32_t *bin_start2,
                      const int32_t *list2, int32_t u_bin_num,
                      int32_t v_bin_num, int32_t flow, double u_, double v_);

/* Matcher::matching (src/matcher.cpp:274-344).
 *   method 0 (flow)  [pinned]   uses m1p,m1c; includes the fork's M dedup mask.
 *   method 1 (stereo)[unpinned] uses m1c,m2c  (SURVEY App. A.7).
 *   method 2 (quad)  [unpinned] uses all four.
 * dims = {W,H,*} of the current frame (dims_c in the reference).
 * Writes at most cap records, returns the true count in *n_out.
 * Returns 0, or -1 on invalid method. */
int32_t vo_matching(const vo_params *p, const int32_t dims[3], int32_t method,
                    const int32_t *m1p, int32_t n1p,
                    const int32_t *m2p, int32_t n2p,
                    const int32_t *m1c, int32_t n1c,
                    const int32_t *m2c, int32_t n2c,
                    vo_p_match *out, int32_t cap, int32_t *n_out);
/* quad matching with stock libviso2's motion prior on the hop previous right -> current right [upstream-recollection] */
int32_t vo_matching_quad_prior(const vo_params *p, const int32_t dims[3], const double *tr16,
                               const int32_t *m1p, int32_t n1p, const int32_t *m2p, int32_t n2p,
                               const int32_t *m1c, int32_t n1c, const int32_t *m2c, int32_t n2c,
                               vo_p_match *out, int32_t cap, int32_t *n_out);

/* For all i1 in set 1: best match index in set 2 (vo_find_match for every
 * query).  Used to check the GPU's whole-set match tables. */
void vo_match_all(const vo_params *p, const int32_t dims[3],
                  const int32_t *m1, int32_t n1,
                  const int32_t *m2, int32_t n2,
                  int32_t flow, int32_t *best);

/* [pinned] Matcher::bucketFeatures + LFSR shuffle (src/matcher.cpp:113-187),
 * restated without the fixed buckets[126][256] capacity (the reference
 * overflows it beyond 1024x284; results are identical whenever the reference
 * stays in bounds). Operates in place, returns the new count. */
int32_t vo_bucket_features(vo_p_match *pm, int32_t n, int32_t max_features,
                           float bucket_width, float bucket_height);

/* [pinned] delaunator::Delaunator::delaunat (src/delaunator.cpp:183-407) on n
 * points xy = {x0,y0,x1,y1,...}: writes min(result, cap) triangle corners
 * (3 per triangle, reference order) and returns the corner count.  *max_depth
 * (nullable) receives the deepest flip stack used; the reference's stack has
 * 13 slots (delaunator.hpp:13) and is undefined beyond.  See viso_outliers.c. */
int32_t vo_delaunay(const float *xy, int32_t n, int32_t *tri_out, int32_t cap, int32_t *max_depth);

/* [pinned] removeOutliers (src/remove_outliers.cpp:4-94): Delaunay-neighbour
 * flow-consistency vote on (u1c,v1c), hard-coded tolerance 5, keep >= 4 votes.
 * In place, order preserved, returns the new count. */
int32_t vo_remove_outliers(vo_p_match *pm, int32_t n, int32_t *max_depth);

/* FNV-1a-64 over raw bytes. */
uint64_t vo_fnv1a64(const void *data, uint64_t nbytes);

/* ---- stereo egomotion (viso_egomotion.c) ------------------- */
/* VisualOdometryStereo::parameters (src/viso_stereo.h:31-43) + the calibration it uses
 * (src/viso.h:41-50, param.base). */
typedef struct vo_ego_params {
  int32_t ransac_iters;      /* 200 */
  int32_t reweighting;       /* 1 */
  double inlier_threshold;   /* 2.0 */
  double f, cu, cv, base;
} vo_ego_params;
void vo_default_ego_params(vo_ego_params *e);
/* [pinned] VisualOdometry::getRandomSample(N,3) x iters from successive rand() values r[3*iters] (src/viso.cpp:86-106). */
void vo_draw_samples(int32_t N, int32_t iters, const int32_t *r, int32_t *samples);
/* [pinned] VisualOdometryStereo::estimateMotion (src/viso_stereo.cpp:54-157) with given 3-point samples. */
int32_t vo_estimate_motion_stereo(const vo_ego_params *e, const vo_p_match *pm, int32_t n, const int32_t *samples,
                                  double tr[6], int32_t *inliers, int32_t *n_inliers);

/* ---- monocular egomotion (viso_mono.c) ---------------------- */
/* VisualOdometryMono::parameters (src/viso_mono.h:32-46) + the calibration it uses (src/viso.h:41-50). */
typedef struct vo_mono_params {
  int32_t ransac_iters;      /* 2000 */
  int32_t pad_;
  double inlier_threshold;   /* 0.00001 */
  double motion_threshold;   /* 100.0 */
  double height, pitch;      /* 1.0, 0.0 */
  double f, cu, cv;
} vo_mono_params;
void vo_default_mono_params(vo_mono_params *e);
/* [pinned] Matrix::svd (src/matrix.cpp:579-802) as its callers see it: a m x n row-major -> U2 m x m (may be NULL),
 * W[min(m,n)], V n x n. */
void vo_svd(const double *a, int32_t m, int32_t n, double *U2, double *W, double *V);
/* [pinned] VisualOdometry::getRandomSample(N,num) x iters from successive rand() values r[num*iters]. */
void vo_draw_samples_n(int32_t N, int32_t num, int32_t iters, const int32_t *r, int32_t *samples);
/* [pinned] VisualOdometryMono::estimateMotion (src/viso_mono.cpp:41-160) with given 8-point samples [iters][8]. */
int32_t vo_estimate_motion_mono(const vo_mono_params *e, const vo_p_match *pm, int32_t n, const int32_t *samples,
                                double tr[6], int32_t *inliers, int32_t *n_inliers);

#ifdef __cplusplus
}
#endif
#endif
