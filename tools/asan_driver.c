/* asan_driver.c -- SURVEY section 5, row 2: AddressSanitizer + UBSan job over the host-side C/C++ of this
 * repository (GPU sanitizers are not available on the pool).  Built and run by `make -C oracle asan`:
 *   oracle/viso_oracle.c, viso_outliers.c, viso_egomotion.c, viso_mono.c   (the CPU restatement: test infrastructure)
 *   hls-final-visual-odometry_amd/csrc/outliers.cpp                         (the product's host-side removeOutliers)
 * with -fsanitize=address,undefined -fno-sanitize-recover, on synthetic frames of several sizes and
 * parameter sets incl. the degenerate ones (tiny images, empty sets, capacity overflow, collinear points).
 * Prints one line per stage and "asan driver: ok" at the end; any finding aborts with the sanitizer's report. */
#include "../oracle/viso_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int32_t vh_remove_outliers_pm(void *pm, int32_t n, int32_t *n_out); /* csrc/outliers.cpp (include/viso_hip.h) */

/* SURVEY App. B generator */
static uint8_t *synth(int W, int H, int bpl, int dx, int dy, int blur, int gain, uint32_t seed) {
  const int Wc = W + 64, Hc = H + 64;
  uint8_t *base = (uint8_t *)malloc((size_t)Wc * Hc), *t = (uint8_t *)malloc((size_t)Wc * Hc);
  uint8_t *I = (uint8_t *)calloc((size_t)bpl * H + 64, 1);
  uint32_t s = seed;
  for (int i = 0; i < Wc * Hc; i++) { s = s * 1664525u + 1013904223u; base[i] = (uint8_t)(s >> 24); }
  for (int b = 0; b < blur; b++) {
    memcpy(t, base, (size_t)Wc * Hc);
    for (int y = 1; y < Hc - 1; y++)
      for (int x = 1; x < Wc - 1; x++) {
        int a = 0;
        for (int j = -1; j <= 1; j++) for (int i = -1; i <= 1; i++) a += t[(y + j) * Wc + x + i];
        base[y * Wc + x] = (uint8_t)(a / 9);
      }
  }
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      int v = ((int)base[(y + 32 + dy) * Wc + x + 32 + dx] - 128) * gain + 128;
      I[y * bpl + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
  free(base); free(t);
  return I;
}

static int32_t features(const vo_params *p, const uint8_t *I, const int32_t dims[3], int32_t **out, int32_t cap) {
  int32_t n1 = 0, n2 = 0;
  int32_t *m2 = (int32_t *)malloc(sizeof(int32_t) * 12 * (size_t)(cap > 0 ? cap : 1));
  int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * 12 * (size_t)(cap > 0 ? cap : 1));
  uint8_t *du = (uint8_t *)malloc((size_t)dims[2] * dims[1] + 64), *dv = (uint8_t *)malloc((size_t)dims[2] * dims[1] + 64);
  if (vo_compute_features(p, I, dims, m1, cap, &n1, m2, cap, &n2, du, dv) != 0) { fprintf(stderr, "compute_features failed\n"); exit(2); }
  free(m1); free(du); free(dv);
  *out = m2;
  return n2 < cap ? n2 : cap;
}

int main(void) {
  static const int sizes[][2] = {{15, 15}, {33, 20}, {64, 48}, {321, 97}, {640, 200}, {1241, 376}};
  long total_feat = 0, total_match = 0;
  for (unsigned k = 0; k < sizeof(sizes) / sizeof(sizes[0]); k++) {
    const int W = sizes[k][0], H = sizes[k][1], bpl = W + 15 - (W - 1) % 16;
    const int32_t dims[3] = {W, H, bpl};
    for (int variant = 0; variant < 4; variant++) {
      vo_params p;
      vo_default_params(&p);
      if (variant == 1) { p.nms_n = 1; p.nms_tau = 10; p.match_binsize = 17; p.match_radius = 33; p.multi_stage = 1; }
      if (variant == 2) { p.nms_n = 5; p.half_resolution = 1; p.match_disp_tolerance = 0; }
      if (variant == 3) { p.nms_n = 3; p.match_binsize = 300; p.match_radius = 1000; p.multi_stage = 1; }
      uint8_t *img[4];
      int32_t *f[4], n[4];
      const int cap = variant == 1 ? 64 : 1 << 16;  /* variant 1: capacity below the feature count */
      for (int q = 0; q < 4; q++) {
        img[q] = synth(W, H, bpl, (q >> 1) * 5 + (q & 1) * 9, q >> 1, 3 + variant, 1 + (variant & 1), 7 + k);
        n[q] = features(&p, img[q], dims, &f[q], cap);
        total_feat += n[q];
      }
      for (int method = 0; method <= 2; method++) {
        const int32_t mc = (n[0] > n[2] ? n[0] : n[2]) + 1;
        vo_p_match *pm = (vo_p_match *)malloc(sizeof(vo_p_match) * (size_t)mc);
        int32_t nm = 0;
        if (vo_matching(&p, dims, method, f[0], n[0], f[1], n[1], f[2], n[2], f[3], n[3], pm, mc, &nm) != 0) { fprintf(stderr, "matching failed\n"); exit(2); }
        if (nm > mc) nm = mc;
        total_match += nm;
        if (method != 1) {
          vo_p_match *a = (vo_p_match *)malloc(sizeof(vo_p_match) * (size_t)(nm + 1)), *b = (vo_p_match *)malloc(sizeof(vo_p_match) * (size_t)(nm + 1));
          memcpy(a, pm, sizeof(vo_p_match) * (size_t)nm); memcpy(b, pm, sizeof(vo_p_match) * (size_t)nm);
          int32_t depth = 0, kb = 0;
          const int32_t ka = vo_remove_outliers(a, nm, &depth);
          if (vh_remove_outliers_pm(b, nm, &kb) != 0 || ka != kb || memcmp(a, b, sizeof(vo_p_match) * (size_t)ka)) {
            fprintf(stderr, "removeOutliers: product and oracle differ (%d vs %d)\n", ka, kb); exit(3);
          }
          vo_bucket_features(a, ka, 2, 50.0f, 50.0f);
          free(a); free(b);
        }
        if (nm >= 10) { /* egomotion on whatever came out (mostly "no estimate": the point is the memory behaviour) */
          double tr[6]; int32_t ninl = 0;
          int32_t *inl = (int32_t *)malloc(sizeof(int32_t) * (size_t)nm);
          const int iters = 40;
          int32_t *r = (int32_t *)malloc(sizeof(int32_t) * 8 * iters), *smp = (int32_t *)malloc(sizeof(int32_t) * 8 * iters);
          for (int i = 0; i < 8 * iters; i++) r[i] = rand();
          if (method == 2) {
            vo_ego_params e; vo_default_ego_params(&e); e.ransac_iters = iters; e.f = 645.24; e.cu = W / 2.0; e.cv = H / 2.0; e.base = 0.57;
            vo_draw_samples(nm, iters, r, smp);
            vo_estimate_motion_stereo(&e, pm, nm, smp, tr, inl, &ninl);
          } else if (method == 0) {
            vo_mono_params e; vo_default_mono_params(&e); e.ransac_iters = iters; e.f = 645.24; e.cu = W / 2.0; e.cv = H / 2.0; e.height = 1.65;
            vo_draw_samples_n(nm, 8, iters, r, smp);
            vo_estimate_motion_mono(&e, pm, nm, smp, tr, inl, &ninl);
          }
          free(inl); free(r); free(smp);
        }
        free(pm);
      }
      for (int q = 0; q < 4; q++) { free(img[q]); free(f[q]); }
    }
    printf("size %dx%d: done (%ld features, %ld matches so far)\n", W, H, total_feat, total_match);
  }
  { /* degenerate point sets for the triangulation: collinear, duplicate, tiny */
    for (int n = 0; n <= 40; n += (n < 6 ? 1 : 17)) {
      vo_p_match *a = (vo_p_match *)calloc((size_t)n + 1, sizeof(vo_p_match)), *b = (vo_p_match *)calloc((size_t)n + 1, sizeof(vo_p_match));
      for (int i = 0; i < n; i++) { a[i].u1c = (float)(i * 3); a[i].v1c = (float)((n & 1) ? 7 : i % 5); a[i].u1p = a[i].u1c + 2; a[i].v1p = a[i].v1c; a[i].i1c = a[i].i1p = i; }
      memcpy(b, a, sizeof(vo_p_match) * (size_t)n);
      int32_t depth = 0, kb = 0;
      const int32_t ka = vo_remove_outliers(a, n, &depth);
      if (vh_remove_outliers_pm(b, n, &kb) != 0 || ka != kb) { fprintf(stderr, "degenerate removeOutliers differs at n=%d\n", n); exit(3); }
      free(a); free(b);
    }
    double a9[72], U[64], Wv[9], V[81];
    for (int i = 0; i < 72; i++) a9[i] = (i % 7) - 3;  /* rank-deficient 8x9 */
    vo_svd(a9, 8, 9, U, Wv, V);
    printf("degenerate cases: done\n");
  }
  printf("asan driver: ok (%ld features, %ld matches)\n", total_feat, total_match);
  return 0;
}
