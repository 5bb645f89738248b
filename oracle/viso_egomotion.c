/*
 * viso_egomotion.c -- TEST INFRASTRUCTURE ONLY (see viso_oracle.h).
 *
 * Plain-C restatement of the reference's stereo egomotion estimate, SURVEY 8(f-4):
 *   VisualOdometryStereo::estimateMotion            src/viso_stereo.cpp:54-157
 *   VisualOdometryStereo::getInlier                 src/viso_stereo.cpp:159-177
 *   VisualOdometryStereo::updateParameters          src/viso_stereo.cpp:179-227
 *   VisualOdometryStereo::computeObservations       src/viso_stereo.cpp:229-238
 *   VisualOdometryStereo::computeResidualsAndJacobian  src/viso_stereo.cpp:240-330
 *   Matrix::solve (Gauss-Jordan, full pivoting)     src/matrix.cpp:417-504
 *   VisualOdometry::getRandomSample                 src/viso.cpp:86-106
 * Same operations in the same order in double precision, so that with the same
 * 3-point samples the result is bit-identical to the reference compiled here
 * (oracle/_ref; tests/test_egomotion.py pins it) -- [pinned].
 *
 * The reference draws its samples with rand() after srand(0) in the constructor
 * (src/viso.cpp:35); here the samples are an input (vo_draw_samples reproduces the
 * reference's drawing from any rand()-like source the caller passes values from).
 */
#include "viso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  double ransac_iters_unused;
} ego_unused;

/* Matrix::solve for a 6x6 system with one right-hand side (src/matrix.cpp:417-504);
 * A and b are overwritten; returns 0 when singular (|pivot| < 1e-20). */
static int solve6(double A[6][6], double b[6]) {
  int indxc[6], indxr[6], ipiv[6];
  int i, icol = 0, irow = 0, j, k, l, ll;
  double big, dum, pivinv, temp;
  for (j = 0; j < 6; j++) ipiv[j] = 0;
  for (i = 0; i < 6; i++) {
    big = 0.0;
    for (j = 0; j < 6; j++)
      if (ipiv[j] != 1)
        for (k = 0; k < 6; k++)
          if (ipiv[k] == 0)
            if (fabs(A[j][k]) >= big) { big = fabs(A[j][k]); irow = j; icol = k; }
    ++(ipiv[icol]);
    if (irow != icol) {
      for (l = 0; l < 6; l++) { temp = A[irow][l]; A[irow][l] = A[icol][l]; A[icol][l] = temp; }
      temp = b[irow]; b[irow] = b[icol]; b[icol] = temp;
    }
    indxr[i] = irow; indxc[i] = icol;
    if (fabs(A[icol][icol]) < 1e-20) return 0;
    pivinv = 1.0 / A[icol][icol];
    A[icol][icol] = 1.0;
    for (l = 0; l < 6; l++) A[icol][l] *= pivinv;
    b[icol] *= pivinv;
    for (ll = 0; ll < 6; ll++)
      if (ll != icol) {
        dum = A[ll][icol];
        A[ll][icol] = 0.0;
        for (l = 0; l < 6; l++) A[ll][l] -= A[icol][l] * dum;
        b[ll] -= b[icol] * dum;
      }
  }
  (void)indxr; (void)indxc;  /* the column unscrambling only touches A, which is discarded */
  return 1;
}

typedef struct {
  const vo_p_match *pm;
  int32_t n;
  const vo_ego_params *e;
  double *X, *Y, *Z;          /* 3d points of the previous frame */
  double *J, *predict, *observe, *residual;
} ego_ctx;

/* computeObservations + computeResidualsAndJacobian for `na` active matches */
static void residuals_jacobian(ego_ctx *c, const double tr[6], const int32_t *active, int32_t na) {
  const vo_ego_params *e = c->e;
  int32_t i, j;
  for (i = 0; i < na; i++) {
    c->observe[4 * i + 0] = c->pm[active[i]].u1c;
    c->observe[4 * i + 1] = c->pm[active[i]].v1c;
    c->observe[4 * i + 2] = c->pm[active[i]].u2c;
    c->observe[4 * i + 3] = c->pm[active[i]].v2c;
  }
  const double rx = tr[0], ry = tr[1], rz = tr[2], tx = tr[3], ty = tr[4], tz = tr[5];
  const double sx = sin(rx), cx = cos(rx), sy = sin(ry), cy = cos(ry), sz = sin(rz), cz = cos(rz);
  const double r00 = +cy * cz, r01 = -cy * sz, r02 = +sy;
  const double r10 = +sx * sy * cz + cx * sz, r11 = -sx * sy * sz + cx * cz, r12 = -sx * cy;
  const double r20 = -cx * sy * cz + sx * sz, r21 = +cx * sy * sz + sx * cz, r22 = +cx * cy;
  const double rdrx10 = +cx * sy * cz - sx * sz, rdrx11 = -cx * sy * sz - sx * cz, rdrx12 = -cx * cy;
  const double rdrx20 = +sx * sy * cz + cx * sz, rdrx21 = -sx * sy * sz + cx * cz, rdrx22 = -sx * cy;
  const double rdry00 = -sy * cz, rdry01 = +sy * sz, rdry02 = +cy;
  const double rdry10 = +sx * cy * cz, rdry11 = -sx * cy * sz, rdry12 = +sx * sy;
  const double rdry20 = -cx * cy * cz, rdry21 = +cx * cy * sz, rdry22 = -cx * sy;
  const double rdrz00 = -cy * sz, rdrz01 = -cy * cz;
  const double rdrz10 = -sx * sy * sz + cx * cz, rdrz11 = -sx * sy * cz - cx * sz;
  const double rdrz20 = +cx * sy * sz + sx * cz, rdrz21 = +cx * sy * cz - sx * sz;
  for (i = 0; i < na; i++) {
    const double X1p = c->X[active[i]], Y1p = c->Y[active[i]], Z1p = c->Z[active[i]];
    const double X1c = r00 * X1p + r01 * Y1p + r02 * Z1p + tx;
    const double Y1c = r10 * X1p + r11 * Y1p + r12 * Z1p + ty;
    const double Z1c = r20 * X1p + r21 * Y1p + r22 * Z1p + tz;
    double weight = 1.0;
    if (e->reweighting) weight = 1.0 / (fabs(c->observe[4 * i + 0] - e->cu) / fabs(e->cu) + 0.05);
    const double X2c = X1c - e->base;
    for (j = 0; j < 6; j++) {
      double X1cd = 0, Y1cd = 0, Z1cd = 0;
      switch (j) {
        case 0: X1cd = 0; Y1cd = rdrx10 * X1p + rdrx11 * Y1p + rdrx12 * Z1p; Z1cd = rdrx20 * X1p + rdrx21 * Y1p + rdrx22 * Z1p; break;
        case 1: X1cd = rdry00 * X1p + rdry01 * Y1p + rdry02 * Z1p; Y1cd = rdry10 * X1p + rdry11 * Y1p + rdry12 * Z1p;
                Z1cd = rdry20 * X1p + rdry21 * Y1p + rdry22 * Z1p; break;
        case 2: X1cd = rdrz00 * X1p + rdrz01 * Y1p; Y1cd = rdrz10 * X1p + rdrz11 * Y1p; Z1cd = rdrz20 * X1p + rdrz21 * Y1p; break;
        case 3: X1cd = 1; Y1cd = 0; Z1cd = 0; break;
        case 4: X1cd = 0; Y1cd = 1; Z1cd = 0; break;
        case 5: X1cd = 0; Y1cd = 0; Z1cd = 1; break;
      }
      c->J[(4 * i + 0) * 6 + j] = weight * e->f * (X1cd * Z1c - X1c * Z1cd) / (Z1c * Z1c);
      c->J[(4 * i + 1) * 6 + j] = weight * e->f * (Y1cd * Z1c - Y1c * Z1cd) / (Z1c * Z1c);
      c->J[(4 * i + 2) * 6 + j] = weight * e->f * (X1cd * Z1c - X2c * Z1cd) / (Z1c * Z1c);
      c->J[(4 * i + 3) * 6 + j] = weight * e->f * (Y1cd * Z1c - Y1c * Z1cd) / (Z1c * Z1c);
    }
    c->predict[4 * i + 0] = e->f * X1c / Z1c + e->cu;
    c->predict[4 * i + 1] = e->f * Y1c / Z1c + e->cv;
    c->predict[4 * i + 2] = e->f * X2c / Z1c + e->cu;
    c->predict[4 * i + 3] = e->f * Y1c / Z1c + e->cv;
    c->residual[4 * i + 0] = weight * (c->observe[4 * i + 0] - c->predict[4 * i + 0]);
    c->residual[4 * i + 1] = weight * (c->observe[4 * i + 1] - c->predict[4 * i + 1]);
    c->residual[4 * i + 2] = weight * (c->observe[4 * i + 2] - c->predict[4 * i + 2]);
    c->residual[4 * i + 3] = weight * (c->observe[4 * i + 3] - c->predict[4 * i + 3]);
  }
}

enum { EGO_UPDATED = 0, EGO_FAILED = 1, EGO_CONVERGED = 2 };

/* updateParameters (src/viso_stereo.cpp:179-227) */
static int update_parameters(ego_ctx *c, const int32_t *active, int32_t na, double tr[6], double step_size, double eps) {
  if (na < 3) return EGO_FAILED;
  residuals_jacobian(c, tr, active, na);
  double A[6][6], B[6];
  int32_t m, n, i;
  for (m = 0; m < 6; m++) {
    for (n = 0; n < 6; n++) {
      double a = 0;
      for (i = 0; i < 4 * na; i++) a += c->J[i * 6 + m] * c->J[i * 6 + n];
      A[m][n] = a;
    }
    double b = 0;
    for (i = 0; i < 4 * na; i++) b += c->J[i * 6 + m] * c->residual[i];
    B[m] = b;
  }
  if (!solve6(A, B)) return EGO_FAILED;
  int converged = 1;
  for (m = 0; m < 6; m++) {
    tr[m] += step_size * B[m];
    if (fabs(B[m]) > eps) converged = 0;
  }
  return converged ? EGO_CONVERGED : EGO_UPDATED;
}

/* getInlier (src/viso_stereo.cpp:159-177): indices of the matches whose squared
 * reprojection error is below inlier_threshold^2 */
static int32_t get_inliers(ego_ctx *c, const double tr[6], const int32_t *all, int32_t *out) {
  residuals_jacobian(c, tr, all, c->n);
  int32_t k = 0, i;
  const double thr = c->e->inlier_threshold * c->e->inlier_threshold;
  for (i = 0; i < c->n; i++) {
    const double d0 = c->observe[4 * i + 0] - c->predict[4 * i + 0], d1 = c->observe[4 * i + 1] - c->predict[4 * i + 1];
    const double d2 = c->observe[4 * i + 2] - c->predict[4 * i + 2], d3 = c->observe[4 * i + 3] - c->predict[4 * i + 3];
    if (d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3 < thr) out[k++] = i;   /* pow(x,2) == x*x */
  }
  return k;
}

void vo_default_ego_params(vo_ego_params *e) {
  memset(e, 0, sizeof(*e));
  e->ransac_iters = 200; e->inlier_threshold = 2.0; e->reweighting = 1;   /* src/viso_stereo.h:39-41 */
  e->f = 1; e->cu = 0; e->cv = 0; e->base = 1;                              /* src/viso.h:41-50, src/viso_stereo.h:38 */
}

/* VisualOdometry::getRandomSample(N, 3) (src/viso.cpp:86-106) `iters` times; r[] holds
 * the successive values of rand() (3 per iteration are consumed). */
void vo_draw_samples(int32_t N, int32_t iters, const int32_t *r, int32_t *samples) {
  int32_t *total = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int32_t k, i, t;
  for (k = 0; k < iters; k++) {
    int32_t left = N;
    for (i = 0; i < N; i++) total[i] = i;
    for (i = 0; i < 3; i++) {
      const int32_t j = r[3 * k + i] % left;
      samples[3 * k + i] = total[j];
      for (t = j; t + 1 < left; t++) total[t] = total[t + 1];   /* vector::erase */
      left--;
    }
  }
  free(total);
}

/* estimateMotion (src/viso_stereo.cpp:54-157).  samples: [ransac_iters][3] match indices.
 * tr[6] = (rx,ry,rz,tx,ty,tz); inliers (capacity n) / *n_inliers = the inlier set of the
 * best hypothesis (VisualOdometry::inliers).  Returns 1 on success, 0 where the reference
 * returns an empty vector (N < 6, fewer than 6 inliers, refinement not converged). */
int32_t vo_estimate_motion_stereo(const vo_ego_params *e, const vo_p_match *pm, int32_t n, const int32_t *samples,
                                  double tr[6], int32_t *inliers, int32_t *n_inliers) {
  int32_t i, k;
  *n_inliers = 0;
  for (i = 0; i < 6; i++) tr[i] = 0;
  if (n < 6) return 0;
  ego_ctx c;
  c.pm = pm; c.n = n; c.e = e;
  c.X = (double *)malloc(sizeof(double) * (size_t)n * 3); c.Y = c.X + n; c.Z = c.Y + n;
  c.J = (double *)malloc(sizeof(double) * (size_t)n * (24 + 12));
  c.predict = c.J + (size_t)24 * n; c.observe = c.predict + (size_t)4 * n; c.residual = c.observe + (size_t)4 * n;
  int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)n * 2), *cur = all + n;
  for (i = 0; i < n; i++) all[i] = i;
  for (i = 0; i < n; i++) {
    /* max(u1p - u2p, 0.0001f): a float expression (src/viso_stereo.cpp:81) */
    const float df = pm[i].u1p - pm[i].u2p;
    const double d = (double)(df > 0.0001f ? df : 0.0001f);
    c.X[i] = (pm[i].u1p - e->cu) * e->base / d;
    c.Y[i] = (pm[i].v1p - e->cv) * e->base / d;
    c.Z[i] = e->f * e->base / d;
  }
  double best_tr[6] = {0, 0, 0, 0, 0, 0};
  int have_tr = 0;
  int32_t nbest = 0;
  for (k = 0; k < e->ransac_iters; k++) {
    double t6[6] = {0, 0, 0, 0, 0, 0};
    int result = EGO_UPDATED, iter = 0;
    while (result == EGO_UPDATED) {
      result = update_parameters(&c, samples + 3 * k, 3, t6, 1, 1e-6);
      if (iter++ > 20 || result == EGO_CONVERGED) break;
    }
    if (result != EGO_FAILED) {
      const int32_t nc = get_inliers(&c, t6, all, cur);
      if (nc > nbest) {
        nbest = nc; memcpy(inliers, cur, sizeof(int32_t) * (size_t)nc);
        memcpy(best_tr, t6, sizeof(t6)); have_tr = 1;
      }
    }
  }
  int32_t success = 1;
  if (nbest >= 6) {
    int iter = 0, result = EGO_UPDATED;
    while (result == EGO_UPDATED) {
      result = update_parameters(&c, inliers, nbest, best_tr, 1, 1e-8);
      if (iter++ > 100 || result == EGO_CONVERGED) break;
    }
    if (result != EGO_CONVERGED) success = 0;
  } else {
    success = 0;
  }
  (void)have_tr;
  *n_inliers = nbest;
  for (i = 0; i < 6; i++) tr[i] = success ? best_tr[i] : 0.0;   /* the reference returns an empty vector on failure */
  free(c.X); free(c.J); free(all);
  return success;
}
