// svd_static.h -- Matrix::svd (reference src/matrix.cpp:579-802) for a matrix of compile-time shape M x N with EVERY
// array index a compile-time constant after unrolling, so that U, V, w, rv1 can live in registers.
// Data-dependent indices of the original (the split point l of the QR phase, the column pair of a
// cancellation rotation, the moves of the shell sort) are turned into predicates over static loops;
// the sequence of floating-point operations applied to the data is the original's.
#pragma once
#ifndef SVD_HD
#define SVD_HD inline
#endif

// Everything up to (not including) the sort: U, w, V as the diagonalisation leaves them.
// WANT_U = false leaves out what only U needs -- the accumulation of the left-hand transformations and the
// column rotations of U in the diagonalisation; neither feeds back into w, rv1 or V (the rotation angles are
// functions of w and rv1 alone), so w and V come out bit for bit as with WANT_U = true and U is left as the
// Householder reduction's scratch.
// Returns whether the decomposition of -A may fail to be the exact mirror image of that of A (what the callers of
// svd_static_last_v_unsigned rely on): a Householder pivot f that is exactly zero (sgn(a, f) takes +0 and -0 alike),
// or an exactly vanishing w[i] / rv1[i] in the accumulation of the transformations, where the column is replaced by a
// unit vector whose sign is not tied to the sign of A, a cancellation rotation of the diagonalisation (taken when
// a w[l-1] is negligible: from then on the sign of that column of U is arbitrary and later sweeps mix it back in
// at the 1e-16 level), or a vanishing z in a sweep.  All of these need a numerically singular A; tests/cpp/
// svd_static_check.cpp checks that unflagged 3x3 inputs -- random, integer, skew-like, exactly rank-deficient --
// have exact mirror images.  (With WANT_U = false the w[i] of the left-hand accumulation is not looked at.)
template <int M, int N, bool WANT_U = true>
SVD_HD bool svd_static_core(double (&U)[M][N], double (&w)[N], double (&V)[N][N]) {
  constexpr int MN = M < N ? M : N;
  double rv1[N];
  bool zero_pivot = false;
  double anorm = 0.0, g = 0.0, scale = 0.0;
  auto sgn = [](double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); };
  auto pyth = [](double a, double b) {
    const double absa = fabs(a), absb = fabs(b);
    if (absa > absb) { const double q = absb / absa; return absa * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q)); }
    if (absb == 0.0) return 0.0;
    const double q = absa / absb;
    return absb * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q));
  };
#pragma unroll
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int j = 0; j < N; j++) V[i][j] = 0.0;
  // Householder reduction to bidiagonal form
#pragma unroll
  for (int i = 0; i < N; i++) {
    const int l = i + 1;
    rv1[i] = scale * g;
    g = 0.0; scale = 0.0;
    double s = 0.0;
    if (i < M) {
#pragma unroll
      for (int k = i; k < M; k++) scale += fabs(U[k][i]);
      if (scale) {
#pragma unroll
        for (int k = i; k < M; k++) { U[k][i] /= scale; s += U[k][i] * U[k][i]; }
        double f = U[i][i];
        zero_pivot = zero_pivot || f == 0.0;
        g = -sgn(sqrt(s), f);
        const double h = f * g - s;
        U[i][i] = f - g;
#pragma unroll
        for (int j = l; j < N; j++) {
          double t = 0.0;
#pragma unroll
          for (int k = i; k < M; k++) t += U[k][i] * U[k][j];
          f = t / h;
#pragma unroll
          for (int k = i; k < M; k++) U[k][j] += f * U[k][i];
        }
#pragma unroll
        for (int k = i; k < M; k++) U[k][i] *= scale;
      }
    }
    w[i] = scale * g;
    g = 0.0; scale = 0.0; s = 0.0;
    if (i < M && i != N - 1) {
#pragma unroll
      for (int k = l; k < N; k++) scale += fabs(U[i][k]);
      if (scale) {
#pragma unroll
        for (int k = l; k < N; k++) { U[i][k] /= scale; s += U[i][k] * U[i][k]; }
        const double f = U[i][l < N ? l : 0];
        zero_pivot = zero_pivot || f == 0.0;
        g = -sgn(sqrt(s), f);
        const double h = f * g - s;
        U[i][l < N ? l : 0] = f - g;
#pragma unroll
        for (int k = l; k < N; k++) rv1[k] = U[i][k] / h;
#pragma unroll
        for (int j = l; j < M; j++) {
          double t = 0.0;
#pragma unroll
          for (int k = l; k < N; k++) t += U[j][k] * U[i][k];
#pragma unroll
          for (int k = l; k < N; k++) U[j][k] += t * rv1[k];
        }
#pragma unroll
        for (int k = l; k < N; k++) U[i][k] *= scale;
      }
    }
    const double t = fabs(w[i]) + fabs(rv1[i]);
    anorm = anorm > t ? anorm : t;
  }
  // accumulation of right-hand transformations (g carries over from the last Householder step)
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    const int l = i + 1;
    if (i < N - 1) {
      zero_pivot = zero_pivot || g == 0.0;
      if (g) {
#pragma unroll
        for (int j = l; j < N; j++) V[j][i] = (U[i < M ? i : 0][j] / U[i < M ? i : 0][l < N ? l : 0]) / g;
#pragma unroll
        for (int j = l; j < N; j++) {
          double t = 0.0;
#pragma unroll
          for (int k = l; k < N; k++) t += U[i < M ? i : 0][k] * V[k][j];
#pragma unroll
          for (int k = l; k < N; k++) V[k][j] += t * V[k][i];
        }
      }
#pragma unroll
      for (int j = l; j < N; j++) { V[i][j] = 0.0; V[j][i] = 0.0; }
    }
    V[i][i] = 1.0;
    g = rv1[i];
  }
  // accumulation of left-hand transformations
  if (WANT_U)
#pragma unroll
  for (int i = MN - 1; i >= 0; i--) {
    const int l = i + 1;
    g = w[i];
    zero_pivot = zero_pivot || g == 0.0;
#pragma unroll
    for (int j = l; j < N; j++) U[i][j] = 0.0;
    if (g) {
      g = 1.0 / g;
#pragma unroll
      for (int j = l; j < N; j++) {
        double t = 0.0;
#pragma unroll
        for (int k = l; k < M; k++) t += U[k][i] * U[k][j];
        const double f = (t / U[i][i]) * g;
#pragma unroll
        for (int k = i; k < M; k++) U[k][j] += f * U[k][i];
      }
#pragma unroll
      for (int j = i; j < M; j++) U[j][i] *= g;
    } else {
#pragma unroll
      for (int j = i; j < M; j++) U[j][i] = 0.0;
    }
    U[i][i] += 1.0;
  }
  // diagonalisation of the bidiagonal form
#pragma unroll
  for (int k = N - 1; k >= 0; k--) {
    bool done = false;
    for (int its = 0; its < 30 && !done; its++) {
      // test for splitting: l = the first index (from k downwards) where rv1[l] is negligible (flag = 0) or w[l-1] is
      int l = 0;
      bool flag = true, found = false;
#pragma unroll
      for (int ll = k; ll >= 0; ll--) {
        if (!found) {
          if ((double)(fabs(rv1[ll]) + anorm) == anorm) { flag = false; l = ll; found = true; }
          else if (ll > 0 && (double)(fabs(w[ll > 0 ? ll - 1 : 0]) + anorm) == anorm) { l = ll; found = true; }
        }
      }
      // (rv1[0] is always 0, so the scan always ends with found; kept total for safety)
      if (!found) { l = 0; flag = false; }
      if (flag) {  // cancellation of rv1[l]: rotations of columns (l-1, i), i = l..k
        double c = 0.0, s = 1.0;
        bool brk = false;
#pragma unroll
        for (int i = 1; i <= k; i++) {
          if (i >= l && !brk) {
            const double f = s * rv1[i];
            rv1[i] = c * rv1[i];
            if ((double)(fabs(f) + anorm) == anorm) brk = true;
            else {
              zero_pivot = true;  // w[l-1] was negligible, not zero: the sign of column l-1 is no longer tied to that of A
              g = w[i];
              double h = pyth(f, g);
              w[i] = h;
              h = 1.0 / h;
              c = g * h;
              s = -f * h;
              if (WANT_U)
#pragma unroll
              for (int a = 0; a < i; a++)
                if (a == l - 1) {
#pragma unroll
                  for (int r = 0; r < M; r++) { const double y = U[r][a], z = U[r][i]; U[r][a] = y * c + z * s; U[r][i] = z * c - y * s; }
                }
            }
          }
        }
      }
      double z = w[k];
      if (l == k) {  // convergence
        if (z < 0.0) {
          w[k] = -z;
#pragma unroll
          for (int r = 0; r < N; r++) V[r][k] = -V[r][k];
        }
        done = true;
      } else {
        // shift from the bottom 2-by-2 minor; x = w[l] (dynamic l: select)
        double x = w[0];
#pragma unroll
        for (int q = 1; q <= k; q++) if (q == l) x = w[q];
        constexpr int nmk = 0;  // (placeholder to keep the structure of the original visible)
        (void)nmk;
        double y = w[k > 0 ? k - 1 : 0];
        g = rv1[k > 0 ? k - 1 : 0];
        double h = rv1[k];
        double f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y);
        g = pyth(f, 1.0);
        f = ((x - z) * (x + z) + h * ((y / (f + sgn(g, f))) - h)) / x;
        double c = 1.0, s = 1.0;
#pragma unroll
        for (int j = 0; j < k; j++) {
          if (j >= l) {
            constexpr int dummy = 0; (void)dummy;
            const int i = j + 1;
            g = rv1[i];
            y = w[i];
            h = s * g;
            g = c * g;
            z = pyth(f, h);
            rv1[j] = z;
            c = f / z;
            s = h / z;
            f = x * c + g * s;
            g = g * c - x * s;
            h = y * s;
            y *= c;
#pragma unroll
            for (int r = 0; r < N; r++) { const double xx = V[r][j], zz = V[r][i]; V[r][j] = xx * c + zz * s; V[r][i] = zz * c - xx * s; }
            z = pyth(f, h);
            w[j] = z;
            if (z) { z = 1.0 / z; c = f * z; s = h * z; } else zero_pivot = true;  // (the rotation of U keeps V's c, s)
            f = c * g + s * y;
            x = c * y - s * g;
            if (WANT_U)
#pragma unroll
            for (int r = 0; r < M; r++) { const double yy = U[r][j], zz = U[r][i]; U[r][j] = yy * c + zz * s; U[r][i] = zz * c - yy * s; }
          }
        }
        // rv1[l] = 0 (dynamic l)
#pragma unroll
        for (int q = 0; q <= k; q++) if (q == l) rv1[q] = 0.0;
        rv1[k] = f;
        w[k] = x;
      }
    }
  }
  return zero_pivot;
}

// The shell sort of the singular values (src/matrix.cpp:770-790) on w alone: w sorted (decreasing), perm[d] = the
// column that ends up at position d.  (The moves of the original on a copy of w with an index array; the dynamic
// indices of two 9-element arrays are resolved by selects.)
template <int N>
SVD_HD void svd_static_sort(double (&w)[N], int (&perm)[N]) {
  double ws[N];
#pragma unroll
  for (int q = 0; q < N; q++) { ws[q] = w[q]; perm[q] = q; }
  int inc = 1;
  do { inc *= 3; inc++; } while (inc <= N);
  do {
    inc /= 3;
    for (int i = inc; i < N; i++) {
      double sw = 0.0; int sp = 0;
#pragma unroll
      for (int q = 0; q < N; q++) if (q == i) { sw = ws[q]; sp = perm[q]; }
      int j = i;
      while (true) {
        double wj = 0.0; int pj = 0;
#pragma unroll
        for (int q = 0; q < N; q++) if (q == j - inc) { wj = ws[q]; pj = perm[q]; }
        if (!(wj < sw)) break;
#pragma unroll
        for (int q = 0; q < N; q++) if (q == j) { ws[q] = wj; perm[q] = pj; }
        j -= inc;
        if (j < inc) break;
      }
#pragma unroll
      for (int q = 0; q < N; q++) if (q == j) { ws[q] = sw; perm[q] = sp; }
    }
  } while (inc > 1);
#pragma unroll
  for (int q = 0; q < N; q++) w[q] = ws[q];
}

// The whole Matrix::svd: U (M x N factor), w, V sorted and sign-normalised as the reference returns them.
template <int M, int N>
SVD_HD bool svd_static(double (&U)[M][N], double (&w)[N], double (&V)[N][N]) {
  const bool zero_pivot = svd_static_core<M, N>(U, w, V);
  int perm[N];
  svd_static_sort<N>(w, perm);
  {
    double Uo[M][N], Vo[N][N];
#pragma unroll
    for (int r = 0; r < M; r++)
#pragma unroll
      for (int q = 0; q < N; q++) Uo[r][q] = U[r][q];
#pragma unroll
    for (int r = 0; r < N; r++)
#pragma unroll
      for (int q = 0; q < N; q++) Vo[r][q] = V[r][q];
#pragma unroll
    for (int d = 0; d < N; d++) {
#pragma unroll
      for (int sidx = 0; sidx < N; sidx++)
        if (perm[d] == sidx) {
#pragma unroll
          for (int r = 0; r < M; r++) U[r][d] = Uo[r][sidx];
#pragma unroll
          for (int r = 0; r < N; r++) V[r][d] = Vo[r][sidx];
        }
    }
  }
  // flip signs so that most elements of (U column, V column) are non-negative
#pragma unroll
  for (int k = 0; k < N; k++) {
    int s2 = 0;
#pragma unroll
    for (int r = 0; r < M; r++) s2 += U[r][k] < 0.0 ? 1 : 0;
#pragma unroll
    for (int r = 0; r < N; r++) s2 += V[r][k] < 0.0 ? 1 : 0;
    if (s2 > (M + N) / 2) {
#pragma unroll
      for (int r = 0; r < M; r++) U[r][k] = -U[r][k];
#pragma unroll
      for (int r = 0; r < N; r++) V[r][k] = -V[r][k];
    }
  }
  return zero_pivot;
}

// Only the LAST column of the sorted, sign-normalised V (the direction of the smallest singular value): what the
// 8-point system and the triangulation take from Matrix::svd.  No column is moved; the source column is selected.
template <int M, int N>
SVD_HD void svd_static_last_v(double (&U)[M][N], double (&w)[N], double (&V)[N][N], double (&out)[N]) {
  svd_static_core<M, N>(U, w, V);
  int perm[N];
  svd_static_sort<N>(w, perm);
#pragma unroll
  for (int sidx = 0; sidx < N; sidx++)
    if (perm[N - 1] == sidx) {
      int s2 = 0;
#pragma unroll
      for (int r = 0; r < M; r++) s2 += U[r][sidx] < 0.0 ? 1 : 0;
#pragma unroll
      for (int r = 0; r < N; r++) s2 += V[r][sidx] < 0.0 ? 1 : 0;
      const bool flip = s2 > (M + N) / 2;
#pragma unroll
      for (int r = 0; r < N; r++) out[r] = flip ? -V[r][sidx] : V[r][sidx];
    }
}

// The direction of the smallest singular value up to SIGN: the column of V that svd_static_last_v would return, without
// the sign normalisation (which needs U) and therefore without U.  For callers whose result does not depend on the sign
// of the vector -- the inlier count of a fundamental-matrix hypothesis: F -> -F negates every intermediate of the rank-2
// projection and of the Sampson test exactly (IEEE arithmetic is symmetric under negation) and the test squares it.
template <int M, int N>
SVD_HD void svd_static_last_v_unsigned(double (&U)[M][N], double (&w)[N], double (&V)[N][N], double (&out)[N]) {
  svd_static_core<M, N, false>(U, w, V);
  int perm[N];
  svd_static_sort<N>(w, perm);
#pragma unroll
  for (int sidx = 0; sidx < N; sidx++)
    if (perm[N - 1] == sidx) {
#pragma unroll
      for (int r = 0; r < N; r++) out[r] = V[r][sidx];
    }
}
