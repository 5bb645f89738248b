/*
 * viso_mono.c -- TEST INFRASTRUCTURE ONLY (see viso_oracle.h).
 *
 * Plain-C restatement of the reference's monocular egomotion estimate, SURVEY 8(f-4):
 *   VisualOdometryMono::estimateMotion          src/viso_mono.cpp:41-160
 *   VisualOdometryMono::smallerThanMedian       src/viso_mono.cpp:162-185  (only its median is used)
 *   VisualOdometryMono::normalizeFeaturePoints  src/viso_mono.cpp:187-233
 *   VisualOdometryMono::fundamentalMatrix       src/viso_mono.cpp:235-266
 *   VisualOdometryMono::getInlier               src/viso_mono.cpp:268-315
 *   VisualOdometryMono::EtoRt                   src/viso_mono.cpp:317-362
 *   VisualOdometryMono::triangulateChieral      src/viso_mono.cpp:364-399
 *   Matrix::svd (+ pythag)                      src/matrix.cpp:579-802, :846-854
 *   Matrix::operator* / lu / det                src/matrix.cpp:263-277, :514-572, :400-415
 *   VisualOdometry::getRandomSample(N,8)        src/viso.cpp:86-106
 * Same operations in the same order and the same precisions (the normalised feature
 * points live in the float fields of p_match, and products of two of them are float
 * products), so that with the same 8-point samples the result is bit-identical to the
 * reference compiled here (oracle/_ref; tests/test_mono.py pins it) -- [pinned].
 *
 * Where the reference would call exit(0) (division of a Matrix by |s| < 1e-20, getMat on an
 * empty Matrix when no chirality solution has a point in front of both cameras) this
 * restatement returns "no estimate" (0); tests/ keep clear of those inputs on the reference.
 */
#include "viso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double sign_of(double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); }

static double pythag(double a, double b) { /* src/matrix.cpp:846-854 */
  const double absa = fabs(a), absb = fabs(b);
  if (absa > absb) { const double q = absb / absa; return absa * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q)); }
  if (absb == 0.0) return 0.0;
  { const double q = absa / absb; return absb * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q)); }
}

/* Matrix::svd (src/matrix.cpp:579-802): a is m x n row-major, replaced by the m x n factor U;
 * w[n] the singular values, v n x n; columns sorted by decreasing singular value, signs flipped so
 * that most elements of (U column, V column) are non-negative.  Returns 0 when an iteration limit
 * was hit (the reference only prints a message and goes on; so does this). */
int vo_svd_raw(double *a, int m, int n, double *w, double *v) {
  double *rv1 = (double *)malloc(sizeof(double) * (size_t)n);
  int flag, i, its, j, jj, k, l = 0, nm = 0, converged = 1;
  double anorm, c, f, g, h, s, scale, x, y, z;
#define A(r, q) a[(size_t)(r) * n + (q)]
#define V(r, q) v[(size_t)(r) * n + (q)]
  memset(v, 0, sizeof(double) * (size_t)n * n);
  g = scale = anorm = 0.0;
  for (i = 0; i < n; i++) { /* Householder reduction to bidiagonal form */
    l = i + 1;
    rv1[i] = scale * g;
    g = s = scale = 0.0;
    if (i < m) {
      for (k = i; k < m; k++) scale += fabs(A(k, i));
      if (scale) {
        for (k = i; k < m; k++) { A(k, i) /= scale; s += A(k, i) * A(k, i); }
        f = A(i, i);
        g = -sign_of(sqrt(s), f);
        h = f * g - s;
        A(i, i) = f - g;
        for (j = l; j < n; j++) {
          for (s = 0.0, k = i; k < m; k++) s += A(k, i) * A(k, j);
          f = s / h;
          for (k = i; k < m; k++) A(k, j) += f * A(k, i);
        }
        for (k = i; k < m; k++) A(k, i) *= scale;
      }
    }
    w[i] = scale * g;
    g = s = scale = 0.0;
    if (i < m && i != n - 1) {
      for (k = l; k < n; k++) scale += fabs(A(i, k));
      if (scale) {
        for (k = l; k < n; k++) { A(i, k) /= scale; s += A(i, k) * A(i, k); }
        f = A(i, l);
        g = -sign_of(sqrt(s), f);
        h = f * g - s;
        A(i, l) = f - g;
        for (k = l; k < n; k++) rv1[k] = A(i, k) / h;
        for (j = l; j < m; j++) {
          for (s = 0.0, k = l; k < n; k++) s += A(j, k) * A(i, k);
          for (k = l; k < n; k++) A(j, k) += s * rv1[k];
        }
        for (k = l; k < n; k++) A(i, k) *= scale;
      }
    }
    { const double t = fabs(w[i]) + fabs(rv1[i]); anorm = anorm > t ? anorm : t; }
  }
  for (i = n - 1; i >= 0; i--) { /* accumulation of right-hand transformations */
    if (i < n - 1) {
      if (g) {
        for (j = l; j < n; j++) V(j, i) = (A(i, j) / A(i, l)) / g;
        for (j = l; j < n; j++) {
          for (s = 0.0, k = l; k < n; k++) s += A(i, k) * V(k, j);
          for (k = l; k < n; k++) V(k, j) += s * V(k, i);
        }
      }
      for (j = l; j < n; j++) V(i, j) = V(j, i) = 0.0;
    }
    V(i, i) = 1.0;
    g = rv1[i];
    l = i;
  }
  for (i = (m < n ? m : n) - 1; i >= 0; i--) { /* accumulation of left-hand transformations */
    l = i + 1;
    g = w[i];
    for (j = l; j < n; j++) A(i, j) = 0.0;
    if (g) {
      g = 1.0 / g;
      for (j = l; j < n; j++) {
        for (s = 0.0, k = l; k < m; k++) s += A(k, i) * A(k, j);
        f = (s / A(i, i)) * g;
        for (k = i; k < m; k++) A(k, j) += f * A(k, i);
      }
      for (j = i; j < m; j++) A(j, i) *= g;
    } else
      for (j = i; j < m; j++) A(j, i) = 0.0;
    ++A(i, i);
  }
  for (k = n - 1; k >= 0; k--) { /* diagonalisation of the bidiagonal form */
    for (its = 0; its < 30; its++) {
      flag = 1;
      for (l = k; l >= 0; l--) {
        nm = l - 1;
        if ((double)(fabs(rv1[l]) + anorm) == anorm) { flag = 0; break; }
        if ((double)(fabs(w[nm]) + anorm) == anorm) break;
      }
      if (flag) {
        c = 0.0;
        s = 1.0;
        for (i = l; i <= k; i++) {
          f = s * rv1[i];
          rv1[i] = c * rv1[i];
          if ((double)(fabs(f) + anorm) == anorm) break;
          g = w[i];
          h = pythag(f, g);
          w[i] = h;
          h = 1.0 / h;
          c = g * h;
          s = -f * h;
          for (j = 0; j < m; j++) {
            y = A(j, nm); z = A(j, i);
            A(j, nm) = y * c + z * s;
            A(j, i) = z * c - y * s;
          }
        }
      }
      z = w[k];
      if (l == k) {
        if (z < 0.0) {
          w[k] = -z;
          for (j = 0; j < n; j++) V(j, k) = -V(j, k);
        }
        break;
      }
      if (its == 29) converged = 0;
      x = w[l];
      nm = k - 1;
      y = w[nm];
      g = rv1[nm];
      h = rv1[k];
      f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y);
      g = pythag(f, 1.0);
      f = ((x - z) * (x + z) + h * ((y / (f + sign_of(g, f))) - h)) / x;
      c = s = 1.0;
      for (j = l; j <= nm; j++) {
        i = j + 1;
        g = rv1[i];
        y = w[i];
        h = s * g;
        g = c * g;
        z = pythag(f, h);
        rv1[j] = z;
        c = f / z;
        s = h / z;
        f = x * c + g * s;
        g = g * c - x * s;
        h = y * s;
        y *= c;
        for (jj = 0; jj < n; jj++) {
          x = V(jj, j); z = V(jj, i);
          V(jj, j) = x * c + z * s;
          V(jj, i) = z * c - x * s;
        }
        z = pythag(f, h);
        w[j] = z;
        if (z) { z = 1.0 / z; c = f * z; s = h * z; }
        f = c * g + s * y;
        x = c * y - s * g;
        for (jj = 0; jj < m; jj++) {
          y = A(jj, j); z = A(jj, i);
          A(jj, j) = y * c + z * s;
          A(jj, i) = z * c - y * s;
        }
      }
      rv1[l] = 0.0;
      rv1[k] = f;
      w[k] = x;
    }
  }
  { /* shell sort by decreasing singular value, columns of U and V along (src/matrix.cpp:770-790) */
    int inc = 1;
    double sw, *su = (double *)malloc(sizeof(double) * (size_t)m), *sv = (double *)malloc(sizeof(double) * (size_t)n);
    do { inc *= 3; inc++; } while (inc <= n);
    do {
      inc /= 3;
      for (i = inc; i < n; i++) {
        sw = w[i];
        for (k = 0; k < m; k++) su[k] = A(k, i);
        for (k = 0; k < n; k++) sv[k] = V(k, i);
        j = i;
        while (w[j - inc] < sw) {
          w[j] = w[j - inc];
          for (k = 0; k < m; k++) A(k, j) = A(k, j - inc);
          for (k = 0; k < n; k++) V(k, j) = V(k, j - inc);
          j -= inc;
          if (j < inc) break;
        }
        w[j] = sw;
        for (k = 0; k < m; k++) A(k, j) = su[k];
        for (k = 0; k < n; k++) V(k, j) = sv[k];
      }
    } while (inc > 1);
    for (k = 0; k < n; k++) { /* flip signs */
      int s2 = 0;
      for (i = 0; i < m; i++) if (A(i, k) < 0.0) s2++;
      for (j = 0; j < n; j++) if (V(j, k) < 0.0) s2++;
      if (s2 > (m + n) / 2) {
        for (i = 0; i < m; i++) A(i, k) = -A(i, k);
        for (j = 0; j < n; j++) V(j, k) = -V(j, k);
      }
    }
    free(su); free(sv);
  }
  free(rv1);
#undef A
#undef V
  return converged;
}

/* Matrix::svd as its callers see it: U2 m x m (columns beyond min(m,n) zero), W min(m,n), V n x n. */
void vo_svd(const double *a_in, int32_t m, int32_t n, double *U2, double *W, double *V) {
  double *a = (double *)malloc(sizeof(double) * (size_t)m * n), *w = (double *)malloc(sizeof(double) * (size_t)n);
  const int mn = m < n ? m : n;
  int i, j;
  memcpy(a, a_in, sizeof(double) * (size_t)m * n);
  vo_svd_raw(a, m, n, w, V);
  if (U2) {
    memset(U2, 0, sizeof(double) * (size_t)m * m);
    for (i = 0; i < m; i++) for (j = 0; j < mn; j++) U2[(size_t)i * m + j] = a[(size_t)i * n + j];
  }
  for (i = 0; i < mn; i++) W[i] = w[i];
  free(a); free(w);
}

/* C = A (ma x na) * B (na x nb), Matrix::operator* (src/matrix.cpp:263-277): sums start at 0, k ascending */
static void matmul(const double *A, int ma, int na, const double *B, int nb, double *C) {
  int i, j, k;
  for (i = 0; i < ma; i++)
    for (j = 0; j < nb; j++) {
      double c = 0.0;
      for (k = 0; k < na; k++) c += A[i * na + k] * B[k * nb + j];
      C[i * nb + j] = c;
    }
}
static void transpose3(const double *A, double *T) {
  int i, j;
  for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) T[j * 3 + i] = A[i * 3 + j];
}
/* U * diag(W) * ~V for 3x3 factors (src/viso_mono.cpp:91-94, :262-265) */
static void recompose3(const double *U, const double *W, const double *V, double *out) {
  double D[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, UD[9], Vt[9];
  D[0] = W[0]; D[4] = W[1]; D[8] = W[2];
  matmul(U, 3, 3, D, 3, UD);
  transpose3(V, Vt);
  matmul(UD, 3, 3, Vt, 3, out);
}

/* Matrix::det of a 3x3 (src/matrix.cpp:400-415 over lu, :514-572) */
static double det3(const double *M) {
  double a[3][3], vv[3], big, dum, sum, temp, d = 1.0;
  int i, imax = 0, j, k, ok = 1;
  for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) a[i][j] = M[i * 3 + j];
  for (i = 0; i < 3 && ok; i++) {
    big = 0.0;
    for (j = 0; j < 3; j++) if ((temp = fabs(a[i][j])) > big) big = temp;
    if (big == 0.0) { ok = 0; break; }  /* lu returns false; det multiplies the untouched diagonal */
    vv[i] = 1.0 / big;
  }
  if (ok) {
    for (j = 0; j < 3; j++) {
      for (i = 0; i < j; i++) {
        sum = a[i][j];
        for (k = 0; k < i; k++) sum -= a[i][k] * a[k][j];
        a[i][j] = sum;
      }
      big = 0.0;
      for (i = j; i < 3; i++) {
        sum = a[i][j];
        for (k = 0; k < j; k++) sum -= a[i][k] * a[k][j];
        a[i][j] = sum;
        if ((dum = vv[i] * fabs(sum)) >= big) { big = dum; imax = i; }
      }
      if (j != imax) {
        for (k = 0; k < 3; k++) { dum = a[imax][k]; a[imax][k] = a[j][k]; a[j][k] = dum; }
        d = -d;
        vv[imax] = vv[j];
      }
      if (j != 2) {
        dum = 1.0 / a[j][j];
        for (i = j + 1; i < 3; i++) a[i][j] *= dum;
      }
    }
  }
  for (i = 0; i < 3; i++) d *= a[i][i];
  return d;
}

void vo_default_mono_params(vo_mono_params *e) { /* src/viso_mono.h:39-45, src/viso.h:46-48 */
  memset(e, 0, sizeof(*e));
  e->ransac_iters = 2000; e->inlier_threshold = 0.00001; e->motion_threshold = 100.0;
  e->height = 1.0; e->pitch = 0.0; e->f = 1; e->cu = 0; e->cv = 0;
}

/* VisualOdometry::getRandomSample(N,num) x iters (src/viso.cpp:86-106): draw k of a sample takes
 * element r % (N-k) of the ordered list of indices not yet taken. */
void vo_draw_samples_n(int32_t N, int32_t num, int32_t iters, const int32_t *r, int32_t *samples) {
  int32_t it, k, q;
  for (it = 0; it < iters; it++) {
    int32_t taken[64], nt = 0;  /* ascending */
    for (k = 0; k < num && k < 64; k++) {
      int32_t j = (int32_t)((uint32_t)r[(size_t)it * num + k] % (uint32_t)(N - k));
      for (q = 0; q < nt; q++) if (j >= taken[q]) j++;
      samples[(size_t)it * num + k] = j;
      for (q = nt; q > 0 && taken[q - 1] > j; q--) taken[q] = taken[q - 1];
      taken[q] = j; nt++;
    }
  }
}

/* fundamentalMatrix (src/viso_mono.cpp:235-266) on the normalised points */
static void fundamental(const vo_p_match *pn, const int32_t *active, int32_t N, double *F) {
  double *A = (double *)malloc(sizeof(double) * (size_t)N * 9), w[9], V[81], U3[9], W3[3], V3[9], F0[9];
  int32_t i;
  for (i = 0; i < N; i++) {
    const vo_p_match m = pn[active[i]];
    double *a = A + (size_t)i * 9;
    a[0] = m.u1c * m.u1p; a[1] = m.u1c * m.v1p; a[2] = m.u1c;   /* float products, widened */
    a[3] = m.v1c * m.u1p; a[4] = m.v1c * m.v1p; a[5] = m.v1c;
    a[6] = m.u1p; a[7] = m.v1p; a[8] = 1;
  }
  vo_svd_raw(A, N, 9, w, V);
  for (i = 0; i < 9; i++) F0[i] = V[i * 9 + 8];  /* reshape(V(:,8),3,3), row-major */
  vo_svd(F0, 3, 3, U3, W3, V3);
  W3[2] = 0;
  recompose3(U3, W3, V3, F);
  free(A);
}

/* getInlier (src/viso_mono.cpp:268-315) */
static int32_t sampson_inliers(const vo_p_match *pn, int32_t n, const double *F, double thr, int32_t *out) {
  const double f00 = F[0], f01 = F[1], f02 = F[2], f10 = F[3], f11 = F[4], f12 = F[5], f20 = F[6], f21 = F[7], f22 = F[8];
  int32_t i, cnt = 0;
  for (i = 0; i < n; i++) {
    const double u1 = pn[i].u1p, v1 = pn[i].v1p, u2 = pn[i].u1c, v2 = pn[i].v1c;
    const double Fx1u = f00 * u1 + f01 * v1 + f02, Fx1v = f10 * u1 + f11 * v1 + f12, Fx1w = f20 * u1 + f21 * v1 + f22;
    const double Ftx2u = f00 * u2 + f10 * v2 + f20, Ftx2v = f01 * u2 + f11 * v2 + f21;
    const double x2tFx1 = u2 * Fx1u + v2 * Fx1v + Fx1w;
    const double d = x2tFx1 * x2tFx1 / (Fx1u * Fx1u + Fx1v * Fx1v + Ftx2u * Ftx2u + Ftx2v * Ftx2v);
    if (fabs(d) < thr) { if (out) out[cnt] = i; cnt++; }
  }
  return cnt;
}

/* triangulateChieral (src/viso_mono.cpp:364-399): X 4 x n (row-major), returns the number of
 * points in front of both cameras */
static int32_t triangulate(const vo_p_match *pm, int32_t n, const double *K, const double *R, const double *t, double *X) {
  double P1[12] = {0}, Rt[12] = {0}, P2[12], J[16], U4[16], W4[4], V4[16];
  int32_t i, j, num = 0;
  for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) { P1[i * 4 + j] = K[i * 3 + j]; Rt[i * 4 + j] = R[i * 3 + j]; }
  for (i = 0; i < 3; i++) Rt[i * 4 + 3] = t[i];
  matmul(K, 3, 3, Rt, 4, P2);
  for (i = 0; i < n; i++) {
    for (j = 0; j < 4; j++) {
      J[0 * 4 + j] = P1[2 * 4 + j] * pm[i].u1p - P1[0 * 4 + j];
      J[1 * 4 + j] = P1[2 * 4 + j] * pm[i].v1p - P1[1 * 4 + j];
      J[2 * 4 + j] = P2[2 * 4 + j] * pm[i].u1c - P2[0 * 4 + j];
      J[3 * 4 + j] = P2[2 * 4 + j] * pm[i].v1c - P2[1 * 4 + j];
    }
    vo_svd(J, 4, 4, U4, W4, V4);
    for (j = 0; j < 4; j++) X[(size_t)j * n + i] = V4[j * 4 + 3];
  }
  for (i = 0; i < n; i++) {
    double a = 0.0, b = 0.0;
    for (j = 0; j < 4; j++) a += P1[2 * 4 + j] * X[(size_t)j * n + i];
    for (j = 0; j < 4; j++) b += P2[2 * 4 + j] * X[(size_t)j * n + i];
    if (a * X[(size_t)3 * n + i] > 0 && b * X[(size_t)3 * n + i] > 0) num++;
  }
  return num;
}

static int cmp_double(const void *a, const void *b) {
  const double x = *(const double *)a, y = *(const double *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

/* VisualOdometryMono::estimateMotion (src/viso_mono.cpp:41-160).  samples[ransac_iters][8] as
 * getRandomSample(N,8) returns them.  Returns 1 and tr[6] = (rx,ry,rz,tx,ty,tz), or 0 where the
 * reference returns an empty vector; inliers/n_inliers = VisualOdometry::inliers at return. */
int32_t vo_estimate_motion_mono(const vo_mono_params *e, const vo_p_match *pm, int32_t N, const int32_t *samples, double tr[6],
                                int32_t *inliers, int32_t *n_inliers) {
  int32_t i, j, k, ok = 0, nbest = 0;
  vo_p_match *pn = NULL;
  int32_t *cur = NULL, *best = NULL;
  double *X = NULL, *Xc = NULL, *d = NULL, *dist = NULL;
  for (i = 0; i < 6; i++) tr[i] = 0;
  *n_inliers = 0;
  if (N < 10) return 0;
  {
    const double K[9] = {e->f, 0, e->cu, 0, e->f, e->cv, 0, 0, 1};
    double Tp[9], Tc[9], F[9], E[9], T1[9], T2[9], Kt[9];
    pn = (vo_p_match *)malloc(sizeof(vo_p_match) * (size_t)N);
    memcpy(pn, pm, sizeof(vo_p_match) * (size_t)N);
    { /* normalizeFeaturePoints (src/viso_mono.cpp:187-233) */
      double cpu = 0, cpv = 0, ccu = 0, ccv = 0, sp = 0, sc = 0;
      for (i = 0; i < N; i++) { cpu += pn[i].u1p; cpv += pn[i].v1p; ccu += pn[i].u1c; ccv += pn[i].v1c; }
      cpu /= (double)N; cpv /= (double)N; ccu /= (double)N; ccv /= (double)N;
      for (i = 0; i < N; i++) {
        pn[i].u1p = (float)(pn[i].u1p - cpu); pn[i].v1p = (float)(pn[i].v1p - cpv);
        pn[i].u1c = (float)(pn[i].u1c - ccu); pn[i].v1c = (float)(pn[i].v1c - ccv);
      }
      for (i = 0; i < N; i++) {  /* float expressions: sqrt(float) is the float overload in the reference's C++ */
        sp += sqrtf(pn[i].u1p * pn[i].u1p + pn[i].v1p * pn[i].v1p);
        sc += sqrtf(pn[i].u1c * pn[i].u1c + pn[i].v1c * pn[i].v1c);
      }
      if (fabs(sp) < 1e-10 || fabs(sc) < 1e-10) goto done;
      sp = sqrt(2.0) * (double)N / sp;
      sc = sqrt(2.0) * (double)N / sc;
      for (i = 0; i < N; i++) {
        pn[i].u1p = (float)(pn[i].u1p * sp); pn[i].v1p = (float)(pn[i].v1p * sp);
        pn[i].u1c = (float)(pn[i].u1c * sc); pn[i].v1c = (float)(pn[i].v1c * sc);
      }
      { const double tp[9] = {sp, 0, -sp * cpu, 0, sp, -sp * cpv, 0, 0, 1}, tc[9] = {sc, 0, -sc * ccu, 0, sc, -sc * ccv, 0, 0, 1};
        memcpy(Tp, tp, sizeof(tp)); memcpy(Tc, tc, sizeof(tc)); }
    }
    cur = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    best = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    for (k = 0; k < e->ransac_iters; k++) { /* RANSAC (src/viso_mono.cpp:63-77) */
      int32_t nc;
      fundamental(pn, samples + (size_t)k * 8, 8, F);
      nc = sampson_inliers(pn, N, F, e->inlier_threshold, cur);
      if (nc > nbest) { nbest = nc; memcpy(best, cur, sizeof(int32_t) * (size_t)nc); }
    }
    *n_inliers = nbest;
    if (inliers) memcpy(inliers, best, sizeof(int32_t) * (size_t)nbest);
    if (nbest < 10) goto done;
    fundamental(pn, best, nbest, F);  /* refit on all inliers */
    transpose3(Tc, T1); matmul(T1, 3, 3, F, 3, T2); matmul(T2, 3, 3, Tp, 3, F);   /* F = ~Tc*F*Tp */
    transpose3(K, Kt); matmul(Kt, 3, 3, F, 3, T2); matmul(T2, 3, 3, K, 3, E);     /* E = ~K*F*K */
    { double U[9], W[3], V[9]; vo_svd(E, 3, 3, U, W, V); W[2] = 0; recompose3(U, W, V, E); }
    { /* EtoRt (src/viso_mono.cpp:317-362) */
      const double Wm[9] = {0, -1, 0, +1, 0, 0, 0, 0, 1}, Zm[9] = {0, +1, 0, -1, 0, 0, 0, 0, 0};
      double U[9], S[3], V[9], Ut[9], Vt[9], Wt[9], T[9], Ra[9], Rb[9], t0[3], R[9], t[3];
      int32_t max_in = 0, have = 0;
      vo_svd(E, 3, 3, U, S, V);
      transpose3(U, Ut); transpose3(V, Vt); transpose3(Wm, Wt);
      matmul(U, 3, 3, Zm, 3, T1); matmul(T1, 3, 3, Ut, 3, T);
      matmul(U, 3, 3, Wm, 3, T1); matmul(T1, 3, 3, Vt, 3, Ra);
      matmul(U, 3, 3, Wt, 3, T1); matmul(T1, 3, 3, Vt, 3, Rb);
      t0[0] = T[2 * 3 + 1]; t0[1] = T[0 * 3 + 2]; t0[2] = T[1 * 3 + 0];
      if (det3(Ra) < 0) for (i = 0; i < 9; i++) Ra[i] = -Ra[i];
      if (det3(Rb) < 0) for (i = 0; i < 9; i++) Rb[i] = -Rb[i];
      X = (double *)malloc(sizeof(double) * 4 * (size_t)N);
      Xc = (double *)malloc(sizeof(double) * 4 * (size_t)N);
      for (i = 0; i < 4; i++) {
        const double *Ri = i < 2 ? Ra : Rb;
        double ti[3];
        int32_t num;
        for (j = 0; j < 3; j++) ti[j] = (i & 1) ? -t0[j] : t0[j];
        num = triangulate(pm, N, K, Ri, ti, Xc);
        if (num > max_in) { max_in = num; memcpy(X, Xc, sizeof(double) * 4 * (size_t)N); memcpy(R, Ri, sizeof(R)); memcpy(t, ti, sizeof(t)); have = 1; }
      }
      if (!have) goto done;  /* (the reference dereferences an empty X here) */
      { /* X = X / X(3,:), points in front, median distance, ground plane (src/viso_mono.cpp:100-141) */
        int32_t np = 0, half, best_idx = 0;
        double median, sigma, weight, best_sum = 0, n0, n1;
        for (i = 0; i < N; i++) {
          const double wv = X[(size_t)3 * N + i];
          for (j = 0; j < 4; j++) X[(size_t)j * N + i] = wv != 0 ? X[(size_t)j * N + i] / wv : 0.0;
        }
        d = (double *)malloc(sizeof(double) * (size_t)N);
        dist = (double *)malloc(sizeof(double) * (size_t)N);
        n0 = cos(-e->pitch); n1 = sin(-e->pitch);
        for (i = 0; i < N; i++)
          if (X[(size_t)2 * N + i] > 0) {
            const double x = X[i], y = X[(size_t)N + i], z = X[(size_t)2 * N + i];
            double dd = 0.0;
            dist[np] = fabs(x) + fabs(y) + fabs(z);
            dd += n0 * y; dd += n1 * z;   /* ~n * x_plane */
            d[np] = dd;
            np++;
          }
        if (np < 10) goto done;
        half = np / 2;
        qsort(dist, (size_t)np, sizeof(double), cmp_double);
        median = dist[half];
        if (median > e->motion_threshold) goto done;
        sigma = median / 50.0;
        weight = 1.0 / (2.0 * sigma * sigma);
        for (i = 0; i < np; i++)
          if (d[i] > median / e->motion_threshold) {
            double sum = 0;
            for (j = 0; j < np; j++) { const double q = d[j] - d[i]; sum += exp(-q * q * weight); }
            if (sum > best_sum) { best_sum = sum; best_idx = i; }
          }
        if (fabs(d[best_idx]) < 1e-20) goto done;  /* (Matrix::operator/ exits the reference here) */
        for (j = 0; j < 3; j++) t[j] = (t[j] * e->height) / d[best_idx];
        {
          const double ry = asin(R[0 * 3 + 2]);
          tr[0] = asin(-R[1 * 3 + 2] / cos(ry));
          tr[1] = ry;
          tr[2] = asin(-R[0 * 3 + 1] / cos(ry));
          tr[3] = t[0]; tr[4] = t[1]; tr[5] = t[2];
          ok = 1;
        }
      }
    }
  }
done:
  free(pn); free(cur); free(best); free(X); free(Xc); free(d); free(dist);
  if (!ok) for (i = 0; i < 6; i++) tr[i] = 0;
  return ok;
}
