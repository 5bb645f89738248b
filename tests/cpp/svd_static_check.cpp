// CPU check of hls-final-visual-odometry_amd/csrc/svd_static.h (the register-resident, static-index form of
// Matrix::svd the mono kernels use for the 3x3, 4x4 and -- optionally -- 8x9 factorizations): the header is plain
// C++, so it is compiled for the host here and compared BIT FOR BIT with the oracle's vo_svd (itself pinned to
// the reference's Matrix::svd) on random, rank-deficient, small-integer, zero-column and 8-point-like inputs.
// Built and run by tests/test_mono.py::test_static_svd_header_equals_the_oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include "../../hls-final-visual-odometry_amd/csrc/svd_static.h"
extern "C" void vo_svd(const double *a, int32_t m, int32_t n, double *U2, double *W, double *V);
template <int M, int N> int run(int trials, unsigned seed) {
  std::mt19937 rng(seed); std::normal_distribution<double> nd;
  int bad = 0;
  for (int t = 0; t < trials; t++) {
    double a[M * N];
    for (auto &x : a) x = nd(rng);
    if (t % 5 == 1) for (int j = 0; j < N; j++) a[(M - 1) * N + j] = a[j] * 2 - a[N + j];       // rank deficient
    if (t % 5 == 2) for (auto &x : a) x = std::round(x * 2);                                       // exact zeros, ties
    if (t % 5 == 3) for (int i = 0; i < M; i++) a[i * N] = 0;                                      // zero column
    if (t % 5 == 4) for (int i = 0; i < M; i++) { double u = nd(rng)*300, v = nd(rng)*100, up = u + 5, vp = v + 1;  // like an 8-point system
        if (N == 9) { double r[9] = {u*up, u*vp, u, v*up, v*vp, v, up, vp, 1}; memcpy(a + i * N, r, sizeof(r)); } }
    double U[M][N], w[N], V[N][N];
    for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) U[i][j] = a[i * N + j];
    svd_static<M, N>(U, w, V);
    double U2[M * M], W[N], Vr[N * N];
    vo_svd(a, M, N, U2, W, Vr);
    const int mn = M < N ? M : N;
    bool ok = true;
    for (int i = 0; i < M && ok; i++) for (int j = 0; j < mn; j++) if (memcmp(&U[i][j], &U2[i * M + j], 8)) { ok = false; break; }
    for (int j = 0; j < mn && ok; j++) if (memcmp(&w[j], &W[j], 8)) ok = false;
    for (int i = 0; i < N && ok; i++) for (int j = 0; j < N; j++) if (memcmp(&V[i][j], &Vr[i * N + j], 8)) { ok = false; break; }
    {  // the last-column variant
      double U3[M][N], w3[N], V3[N][N], lv[N];
      for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) U3[i][j] = a[i * N + j];
      svd_static_last_v<M, N>(U3, w3, V3, lv);
      for (int i = 0; i < N && ok; i++) if (memcmp(&lv[i], &Vr[i * N + N - 1], 8)) ok = false;
      // ... and the one without U and without the sign normalisation: the same column up to sign, the same w
      double U4[M][N], w4[N], V4[N][N], lu[N];
      for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) U4[i][j] = a[i * N + j];
      svd_static_last_v_unsigned<M, N>(U4, w4, V4, lu);
      bool same = true, opposite = true;
      for (int i = 0; i < N; i++) { const double neg = -lv[i]; same = same && !memcmp(&lu[i], &lv[i], 8); opposite = opposite && !memcmp(&lu[i], &neg, 8); }
      if (!same && !opposite) ok = false;
      for (int j = 0; j < N && ok; j++) if (memcmp(&w4[j], &w3[j], 8)) ok = false;
    }
    if (!ok) { bad++; if (bad < 4) printf("  mismatch M=%d N=%d trial %d (kind %d)\n", M, N, t, t % 5); }
  }
  printf("svd_static<%d,%d>: %d trials, %d mismatching\n", M, N, trials, bad);
  return bad;
}
// The mirror-image property fundamental8<false> rests on (csrc/kernels_mono.hip): the rank-2 projection
// U diag(w0, w1, 0) V^T of -F is exactly minus that of F -- unless svd_static reports a zero Householder pivot.
static bool rank2(const double *m, double *out) {
  double U[3][3], w[3], V[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) U[i][j] = m[i * 3 + j];
  const bool zp = svd_static<3, 3>(U, w, V);
  const double d[3] = {w[0], w[1], 0.0};
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double ud[3], acc = 0.0;
    for (int k = 0; k < 3; k++) { ud[k] = 0.0; for (int q = 0; q < 3; q++) ud[k] += U[i][q] * (q == k ? d[k] : 0.0); }
    for (int k = 0; k < 3; k++) acc += ud[k] * V[j][k];
    out[i * 3 + j] = acc;
  }
  return zp;
}
static int mirror(int trials, unsigned seed) {
  std::mt19937 rng(seed); std::normal_distribution<double> nd;
  int bad = 0, flagged = 0, broken_flagged = 0, per_kind[6] = {0, 0, 0, 0, 0, 0};
  for (int t = 0; t < trials; t++) {
    double m[9], n[9], a[9], b[9];
    for (auto &x : m) x = nd(rng);
    if (t % 6 == 1) for (auto &x : m) x = std::round(x);              // exact zeros among the entries, often singular
    if (t % 6 == 2) { m[0] = 0; m[4] = 0; m[8] = 0; }                 // skew-like: zero diagonal
    if (t % 6 == 3) { m[t % 9] = 0; m[(t / 9) % 9] = -0.0; }
    if (t % 6 == 4) { double p[3], q[3], r[3], s[3]; for (int i = 0; i < 3; i++) { p[i] = nd(rng); q[i] = nd(rng); r[i] = nd(rng); s[i] = nd(rng); }
                      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i * 3 + j] = p[i] * q[j] + r[i] * s[j]; }   // rank 2 up to rounding (a noise-free F)
    if (t % 6 == 5) { double p[3], q[3]; for (int i = 0; i < 3; i++) { p[i] = std::round(nd(rng) * 3); q[i] = nd(rng); }
                      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i * 3 + j] = p[i] * q[j]; }                 // rank 1
    for (int i = 0; i < 9; i++) n[i] = -m[i];
    const bool za = rank2(m, a), zb = rank2(n, b);
    bool mirror_ok = true;
    for (int i = 0; i < 9; i++) { const double neg = -a[i]; if (memcmp(&neg, &b[i], 8) && !(a[i] == 0.0 && b[i] == 0.0)) mirror_ok = false; }
    if (za != zb) { bad++; if (bad < 4) printf("  zero-pivot flag differs between F and -F, trial %d\n", t); }
    flagged += za ? 1 : 0; per_kind[t % 6] += za ? 1 : 0;
    if (!mirror_ok) { if (za) broken_flagged++; else { bad++; if (bad < 4) printf("  mirror image broken without a flag, trial %d\n", t); } }
  }
  printf("rank-2 mirror image: %d trials, %d flagged (%d of them really differ), %d unflagged failures\n", trials, flagged, broken_flagged, bad);
  printf("  flagged per family (random, integer, zero diagonal, two zeros, rank 2, rank 1): %d %d %d %d %d %d of %d each\n", per_kind[0], per_kind[1], per_kind[2],
         per_kind[3], per_kind[4], per_kind[5], trials / 6);
  if (per_kind[0]) { printf("  a generic matrix was flagged\n"); bad++; }
  return bad;
}
int main() {
  int bad = mirror(120000, 9);
  bad += run<8, 9>(1500, 1); bad += run<3, 3>(3000, 2); bad += run<4, 4>(3000, 3);
  return bad != 0;
}
