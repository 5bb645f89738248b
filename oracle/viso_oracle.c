/*
 * viso_oracle.c -- TEST INFRASTRUCTURE ONLY (see viso_oracle.h).
 *
 * Plain-C, exact-integer restatement of the reference's CPU/SSE detect+match
 * path.  Nothing here is SSE; every function states the arithmetic the SSE
 * code performs on the region that is observable through features/matches
 * (SURVEY.md Appendix A).  Citations are relative to /root/reference/.
 */
#include "viso_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define VO_MARGIN 7 /* src/matcher.cpp:38 */

void vo_default_params(vo_params *p) {
  /* src/matcher.h:60-71 */
  memset(p, 0, sizeof(*p));
  p->nms_n = 2;
  p->nms_tau = 50;
  p->match_binsize = 50;
  p->match_radius = 200;
  p->match_disp_tolerance = 2;
  p->outlier_disp_tolerance = 5;
  p->outlier_flow_tolerance = 5;
  p->multi_stage = 0;
  p->half_resolution = 0;
  p->refinement = 0;
}

uint64_t vo_fnv1a64(const void *data, uint64_t nbytes) {
  const uint8_t *b = (const uint8_t *)data;
  uint64_t h = 1469598103934665603ULL;
  for (uint64_t i = 0; i < nbytes; i++) {
    h ^= b[i];
    h *= 1099511628211ULL;
  }
  return h;
}

/* ------------------------------------------------------------------ filters */

static inline int32_t px(const uint8_t *I, int32_t bpl, int32_t x, int32_t y) {
  return (int32_t)I[(int64_t)y * bpl + x];
}

void vo_filters(const uint8_t *I, int32_t bpl, int32_t H, uint8_t *du,
                uint8_t *dv, int16_t *f1, int16_t *f2) {
  static const int32_t S[5] = {1, 4, 6, 4, 1};   /* smoothing taps  */
  static const int32_t D[5] = {1, 2, 0, -2, -1}; /* derivative taps: k=-2..2 */
  static const int32_t C[5] = {1, 1, 0, -1, -1}; /* checkerboard taps */
  const int64_t n = (int64_t)bpl * H;
  if (du) memset(du, 0, (size_t)n);
  if (dv) memset(dv, 0, (size_t)n);
  if (f1) memset(f1, 0, (size_t)n * sizeof(int16_t));
  if (f2) memset(f2, 0, (size_t)n * sizeof(int16_t));
  for (int32_t y = 2; y <= H - 3; y++) {
    for (int32_t x = 2; x <= bpl - 3; x++) {
      int32_t a_du = 0, a_dv = 0, a_f2 = 0, s5 = 0, s3 = 0;
      for (int32_t ky = -2; ky <= 2; ky++) {
        for (int32_t kx = -2; kx <= 2; kx++) {
          const int32_t v = px(I, bpl, x + kx, y + ky);
          /* du: column (1,4,6,4,1), row (1,2,0,-2,-1): filter.cpp:288-318 then :132-171 */
          a_du += S[ky + 2] * D[kx + 2] * v;
          /* dv: column (1,2,0,-2,-1), row (1,4,6,4,1): filter.cpp:288-318 then :79-127 */
          a_dv += D[ky + 2] * S[kx + 2] * v;
          /* f2: (1,1,0,-1,-1)^T x (1,1,0,-1,-1): filter.cpp:339-347,:365-367 */
          a_f2 += C[ky + 2] * C[kx + 2] * v;
          s5 += v;
          if (ky >= -1 && ky <= 1 && kx >= -1 && kx <= 1) s3 += v;
        }
      }
      const int64_t o = (int64_t)y * bpl + x;
      if (du) {
        /* arithmetic >>7 (floor), +128, unsigned saturation: filter.cpp:159-160,168 */
        int32_t r = (a_du >> 7) + 128;
        du[o] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
      }
      if (dv) {
        int32_t r = (a_dv >> 7) + 128; /* filter.cpp:114-115,124 */
        dv[o] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
      }
      if (f2) f2[o] = (int16_t)a_f2;
      /* blob: -S5x5 + 2*S3x3 + 7*centre (filter.cpp:461-463); first written
       * pixel is (3,3) (filter.cpp:448) */
      if (f1 && x >= 3 && y >= 3) f1[o] = (int16_t)(-s5 + 2 * s3 + 7 * px(I, bpl, x, y));
    }
  }
}

/* ---------------------------------------------------------------------- NMS */

typedef struct {
  int32_t u, v, val, c;
} vo_max;

static inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* Dominance test of one block extremum (matcher.cpp:420-426 and siblings).
 * sign=+1: candidate is a minimum (fails if a strictly smaller value lies
 * in the (2n+1)^2 window outside the block); sign=-1: maximum. */
static int vo_nms_survives(const int16_t *f, int32_t bpl, int32_t W, int32_t H,
                           int32_t n, int32_t bi, int32_t bj, int32_t ci,
                           int32_t cj, int32_t cval, int32_t sign) {
  const int32_t j_hi = imin(cj + n, H - 1 - VO_MARGIN);
  const int32_t i_hi = imin(ci + n, W - 1 - VO_MARGIN);
  for (int32_t j2 = cj - n; j2 <= j_hi; j2++) {
    for (int32_t i2 = ci - n; i2 <= i_hi; i2++) {
      const int32_t cur = f[(int64_t)j2 * bpl + i2];
      const int outside = (i2 < bi || i2 > bi + n || j2 < bj || j2 > bj + n);
      if (outside && (sign > 0 ? cur < cval : cur > cval)) return 0;
    }
  }
  return 1;
}

int32_t vo_nms(const int16_t *f1, const int16_t *f2, const int32_t dims[3],
               int32_t nms_n, int32_t nms_tau, int32_t *out4, int32_t cap) {
  const int32_t W = dims[0], H = dims[1], bpl = dims[2];
  const int32_t n = nms_n, tau = nms_tau;
  int32_t cnt = 0;
  for (int32_t j = n + VO_MARGIN; j < H - n - VO_MARGIN; j += n + 1) {
    for (int32_t i = n + VO_MARGIN; i < W - n - VO_MARGIN; i += n + 1) {
      vo_max e[4]; /* f1min, f1max, f2min, f2max */
      const int64_t a0 = (int64_t)j * bpl + i;
      e[0].u = e[1].u = e[2].u = e[3].u = i;
      e[0].v = e[1].v = e[2].v = e[3].v = j;
      e[0].val = e[1].val = f1[a0];
      e[2].val = e[3].val = f2[a0];
      /* block scan, first extremum in scan order wins (matcher.cpp:393-417) */
      for (int32_t j2 = j; j2 <= j + n; j2++) {
        for (int32_t i2 = i; i2 <= i + n; i2++) {
          const int64_t a = (int64_t)j2 * bpl + i2;
          int32_t cur = f1[a];
          if (cur < e[0].val) { e[0].u = i2; e[0].v = j2; e[0].val = cur; }
          else if (cur > e[1].val) { e[1].u = i2; e[1].v = j2; e[1].val = cur; }
          cur = f2[a];
          if (cur < e[2].val) { e[2].u = i2; e[2].v = j2; e[2].val = cur; }
          else if (cur > e[3].val) { e[3].u = i2; e[3].v = j2; e[3].val = cur; }
        }
      }
      for (int32_t c = 0; c < 4; c++) {
        const int16_t *f = (c < 2) ? f1 : f2;
        const int32_t sign = (c & 1) ? -1 : +1;
        if (!vo_nms_survives(f, bpl, W, H, n, i, j, e[c].u, e[c].v, e[c].val, sign)) continue;
        /* threshold after the dominance test (matcher.cpp:427,439,451,463) */
        if (sign > 0 ? (e[c].val <= -tau) : (e[c].val >= tau)) {
          if (cnt < cap && out4) {
            out4[4 * cnt + 0] = e[c].u;
            out4[4 * cnt + 1] = e[c].v;
            out4[4 * cnt + 2] = e[c].val;
            out4[4 * cnt + 3] = c;
          }
          cnt++;
        }
      }
    }
  }
  return cnt;
}

/* --------------------------------------------------------------- descriptor */

void vo_descriptor(const uint8_t *du, const uint8_t *dv, int32_t bpl, int32_t u,
                   int32_t v, uint8_t desc[32]) {
  /* (column offset, row offset) of the 16 sample points in byte order
   * (matcher.cpp:482-513); each contributes du then dv. */
  static const int8_t off[16][2] = {
      {-3, -1}, {-3, +1}, {-1, -1}, {-1, +1}, {+3, -1}, {+3, +1}, {+1, -1}, {+1, +1},
      {-1, -5}, {-1, +5}, {+1, -5}, {+1, +5}, {-5, -3}, {-5, +3}, {+5, -3}, {+5, +3}};
  for (int k = 0; k < 16; k++) {
    const int64_t a = (int64_t)(v + off[k][1]) * bpl + (u + off[k][0]);
    desc[2 * k + 0] = du[a];
    desc[2 * k + 1] = dv[a];
  }
}

/* ---------------------------------------------------------- half resolution */

void vo_half_resolution(const uint8_t *I, const int32_t dims[3], int32_t dims_half[3],
                        uint8_t *out) {
  /* matcher.cpp:566-570 */
  dims_half[0] = dims[0] / 2;
  dims_half[1] = dims[1] / 2;
  dims_half[2] = dims_half[0] + 15 - (dims_half[0] - 1) % 16;
  if (!out) return;
  /* the reference leaves padding columns uninitialised; zero them here */
  memset(out, 0, (size_t)dims_half[2] * dims_half[1]);
  for (int32_t v = 0; v < dims_half[1]; v++)
    for (int32_t u = 0; u < dims_half[0]; u++) {
      const int64_t r0 = (int64_t)(2 * v) * dims[2] + 2 * u, r1 = r0 + dims[2];
      out[(int64_t)v * dims_half[2] + u] =
          (uint8_t)(((int32_t)I[r0] + I[r0 + 1] + I[r1] + I[r1 + 1]) / 4); /* :578-581 */
    }
}

/* ----------------------------------------------------------- computeFeatures */

static void vo_pack(const int32_t *max4, int32_t n, int32_t cap, int32_t s,
                    const uint8_t *du, const uint8_t *dv, int32_t bpl, int32_t *out12) {
  if (!out12) return;
  for (int32_t k = 0; k < n && k < cap; k++) {
    int32_t *r = out12 + 12 * (int64_t)k;
    const int32_t u = max4[4 * k + 0], v = max4[4 * k + 1];
    r[0] = u * s; /* matcher.cpp:667 */
    r[1] = v * s;
    r[2] = 0; /* val is zeroed on packing */
    r[3] = max4[4 * k + 3];
    vo_descriptor(du, dv, bpl, u, v, (uint8_t *)(r + 4));
  }
}

int32_t vo_compute_features(const vo_params *p, const uint8_t *I, const int32_t dims[3],
                            int32_t *max1, int32_t cap1, int32_t *num1, int32_t *max2,
                            int32_t cap2, int32_t *num2, uint8_t *du_out, uint8_t *dv_out) {
  if (!I || dims[0] <= 0 || dims[1] <= 0 || dims[2] < dims[0]) return -1;
  int32_t dm[3] = {dims[0], dims[1], dims[2]};
  const uint8_t *Im = I;
  uint8_t *half = NULL;
  if (p->half_resolution) { /* matcher.cpp:603-616 */
    vo_half_resolution(I, dims, dm, NULL);
    half = (uint8_t *)malloc((size_t)dm[2] * dm[1] + 16);
    vo_half_resolution(I, dims, dm, half);
    Im = half;
  }
  const int64_t np = (int64_t)dm[2] * dm[1];
  uint8_t *du = (uint8_t *)malloc((size_t)np), *dv = (uint8_t *)malloc((size_t)np);
  int16_t *f1 = (int16_t *)malloc((size_t)np * 2), *f2 = (int16_t *)malloc((size_t)np * 2);
  vo_filters(Im, dm[2], dm[1], du, dv, f1, f2);
  const int32_t s = p->half_resolution ? 2 : 1;

  /* an upper bound on maxima: 4 per NMS block */
  if (num1) *num1 = 0;
  if (p->multi_stage) { /* matcher.cpp:621-628 */
    int32_t ns = p->nms_n * 4;
    if (ns > 10) ns = imax(p->nms_n, 10);
    int32_t cnt = vo_nms(f1, f2, dm, ns, p->nms_tau, NULL, 0);
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)(cnt + 1));
    vo_nms(f1, f2, dm, ns, p->nms_tau, tmp, cnt);
    vo_pack(tmp, cnt, cap1, s, du, dv, dm[2], max1);
    if (num1) *num1 = cnt;
    free(tmp);
  }
  {
    int32_t cnt = vo_nms(f1, f2, dm, p->nms_n, p->nms_tau, NULL, 0);
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)(cnt + 1));
    vo_nms(f1, f2, dm, p->nms_n, p->nms_tau, tmp, cnt);
    vo_pack(tmp, cnt, cap2, s, du, dv, dm[2], max2);
    if (num2) *num2 = cnt;
    free(tmp);
  }
  if (du_out) memcpy(du_out, du, (size_t)np);
  if (dv_out) memcpy(dv_out, dv, (size_t)np);
  free(du); free(dv); free(f1); free(f2); free(half);
  return 0;
}

/* ------------------------------------------------------------------ binning */

static inline int32_t vo_bin_of(int32_t x, int32_t binsize, int32_t nbin) {
  /* min((int)floor((float)x/(float)binsize), nbin-1)  (matcher.cpp:208-209).
   * Kept in float exactly as the reference does. */
  return imin((int32_t)floorf((float)x / (float)binsize), nbin - 1);
}

void vo_create_index(const int32_t *m, int32_t n, int32_t binsize, int32_t u_bin_num,
                     int32_t v_bin_num, int32_t *bin_start, int32_t *list) {
  const int32_t nb = 4 * u_bin_num * v_bin_num;
  memset(bin_start, 0, sizeof(int32_t) * (size_t)(nb + 1));
  for (int32_t i = 0; i < n; i++) {
    const int32_t *r = m + 12 * (int64_t)i;
    const int32_t b = (r[3] * v_bin_num + vo_bin_of(r[1], binsize, v_bin_num)) * u_bin_num +
                      vo_bin_of(r[0], binsize, u_bin_num);
    bin_start[b + 1]++;
  }
  for (int32_t b = 0; b < nb; b++) bin_start[b + 1] += bin_start[b];
  int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nb + 1));
  memcpy(fill, bin_start, sizeof(int32_t) * (size_t)(nb + 1));
  for (int32_t i = 0; i < n; i++) { /* ascending i == push_back order */
    const int32_t *r = m + 12 * (int64_t)i;
    const int32_t b = (r[3] * v_bin_num + vo_bin_of(r[1], binsize, v_bin_num)) * u_bin_num +
                      vo_bin_of(r[0], binsize, u_bin_num);
    list[fill[b]++] = i;
  }
  free(fill);
}

/* ---------------------------------------------------------------- findMatch */

static inline int32_t vo_sad32(const uint8_t *a, const uint8_t *b) {
  int32_t s = 0;
  for (int k = 0; k < 32; k++) s += abs((int32_t)a[k] - (int32_t)b[k]);
  return s; /* two _mm_sad_epu8 + add (matcher.cpp:252-255) */
}

int32_t vo_find_match(const vo_params *p, const int32_t *m1, int32_t i1, const int32_t *m2,
                      const int32_t *bin_start2, const int32_t *list2, int32_t u_bin_num,
                      int32_t v_bin_num, int32_t flow, double u_, double v_) {
  int32_t min_ind = 0; /* matcher.cpp:221 */
  double min_cost = 10000000;
  const int32_t *q = m1 + 12 * (int64_t)i1;
  const int32_t u1 = q[0], v1 = q[1], c = q[3];
  const uint8_t *d1 = (const uint8_t *)(q + 4);
  const float bs = (float)p->match_binsize;

  float u_min = (float)(u1 - p->match_radius), u_max = (float)(u1 + p->match_radius);
  float v_min = (float)(v1 - p->match_radius), v_max = (float)(v1 + p->match_radius);
  if (!flow) { /* stock libviso2 1-d stereo search; absent from the reference */
    v_min = (float)(v1 - p->match_disp_tolerance);
    v_max = (float)(v1 + p->match_disp_tolerance);
  }
  /* matcher.cpp:237-240 */
  const int32_t ub0 = imin(imax((int32_t)floorf(u_min / bs), 0), u_bin_num - 1);
  const int32_t ub1 = imin(imax((int32_t)floorf(u_max / bs), 0), u_bin_num - 1);
  const int32_t vb0 = imin(imax((int32_t)floorf(v_min / bs), 0), v_bin_num - 1);
  const int32_t vb1 = imin(imax((int32_t)floorf(v_max / bs), 0), v_bin_num - 1);

  for (int32_t ub = ub0; ub <= ub1; ub++) {
    for (int32_t vb = vb0; vb <= vb1; vb++) {
      const int32_t k = (c * v_bin_num + vb) * u_bin_num + ub;
      for (int32_t t = bin_start2[k]; t < bin_start2[k + 1]; t++) {
        const int32_t i2 = list2[t];
        const int32_t *r = m2 + 12 * (int64_t)i2;
        const float u2 = (float)r[0], v2 = (float)r[1];
        if (u2 >= u_min && u2 <= u_max && v2 >= v_min && v2 <= v_max) {
          double cost = (double)vo_sad32(d1, (const uint8_t *)(r + 4));
          if (u_ >= 0 && v_ >= 0) { /* matcher.cpp:257-262 */
            const double ddu = (double)r[0] - u_, ddv = (double)r[1] - v_;
            cost += 4 * sqrt(ddu * ddu + ddv * ddv);
          }
          if (cost < min_cost) { min_ind = i2; min_cost = cost; }
        }
      }
    }
  }
  return min_ind;
}

/* ----------------------------------------------------------------- matching */

typedef struct {
  int32_t *bin_start, *list;
} vo_index;

static vo_index vo_index_make(const vo_params *p, const int32_t *m, int32_t n, int32_t ubn,
                              int32_t vbn) {
  vo_index k;
  k.bin_start = (int32_t *)malloc(sizeof(int32_t) * (size_t)(4 * ubn * vbn + 1));
  k.list = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  vo_create_index(m, n, p->match_binsize, ubn, vbn, k.bin_start, k.list);
  return k;
}
static void vo_index_free(vo_index *k) { free(k->bin_start); free(k->list); }

static void vo_emit(vo_p_match *out, int32_t cap, int32_t *cnt, float u1p, float v1p,
                    int32_t i1p, float u2p, float v2p, int32_t i2p, float u1c, float v1c,
                    int32_t i1c, float u2c, float v2c, int32_t i2c) {
  if (*cnt < cap && out) {
    vo_p_match *m = out + *cnt;
    m->u1p = u1p; m->v1p = v1p; m->i1p = i1p; m->u2p = u2p; m->v2p = v2p; m->i2p = i2p;
    m->u1c = u1c; m->v1c = v1c; m->i1c = i1c; m->u2c = u2c; m->v2c = v2c; m->i2c = i2c;
  }
  (*cnt)++;
}

int32_t vo_matching(const vo_params *p, const int32_t dims[3], int32_t method,
                    const int32_t *m1p, int32_t n1p, const int32_t *m2p, int32_t n2p,
                    const int32_t *m1c, int32_t n1c, const int32_t *m2c, int32_t n2c,
                    vo_p_match *out, int32_t cap, int32_t *n_out) {
  /* matcher.cpp:282-284 */
  const int32_t ubn = (int32_t)ceilf((float)dims[0] / (float)p->match_binsize);
  const int32_t vbn = (int32_t)ceilf((float)dims[1] / (float)p->match_binsize);
  int32_t cnt = 0;
  *n_out = 0;
  if (method == 0) {
    /* flow: matcher.cpp:299-336.  With an empty previous/current set the
     * reference would dereference element 0 of an empty array; we return no
     * matches instead. */
    if (n1p <= 0 || n1c <= 0) return 0;
    vo_index k1p = vo_index_make(p, m1p, n1p, ubn, vbn);
    vo_index k1c = vo_index_make(p, m1c, n1c, ubn, vbn);
    uint8_t *M = (uint8_t *)calloc((size_t)dims[0] * dims[1], 1); /* :293 */
    for (int32_t i1c = 0; i1c < n1c; i1c++) {
      const int32_t u1c = m1c[12 * (int64_t)i1c], v1c = m1c[12 * (int64_t)i1c + 1];
      const int32_t i1p = vo_find_match(p, m1c, i1c, m1p, k1p.bin_start, k1p.list, ubn, vbn, 1, -1, -1);
      const int32_t i1c2 = vo_find_match(p, m1p, i1p, m1c, k1c.bin_start, k1c.list, ubn, vbn, 1, -1, -1);
      if (i1c2 == i1c) {
        const int32_t u1p = m1p[12 * (int64_t)i1p], v1p = m1p[12 * (int64_t)i1p + 1];
        uint8_t *mk = M + (int64_t)v1c * dims[0] + u1c; /* indexed with W: :331 */
        if (*mk == 0) {
          vo_emit(out, cap, &cnt, (float)u1p, (float)v1p, i1p, -1, -1, -1, (float)u1c,
                  (float)v1c, i1c, -1, -1, -1);
          *mk = 1;
        }
      }
    }
    free(M);
    vo_index_free(&k1p);
    vo_index_free(&k1c);
  } else if (method == 1) {
    /* stereo [unpinned]: SURVEY App. A.7 */
    if (n1c <= 0 || n2c <= 0) return 0;
    vo_index k1c = vo_index_make(p, m1c, n1c, ubn, vbn);
    vo_index k2c = vo_index_make(p, m2c, n2c, ubn, vbn);
    for (int32_t i1c = 0; i1c < n1c; i1c++) {
      const int32_t i2c = vo_find_match(p, m1c, i1c, m2c, k2c.bin_start, k2c.list, ubn, vbn, 0, -1, -1);
      const int32_t i1c2 = vo_find_match(p, m2c, i2c, m1c, k1c.bin_start, k1c.list, ubn, vbn, 0, -1, -1);
      if (i1c2 == i1c) {
        const int32_t u1c = m1c[12 * (int64_t)i1c], v1c = m1c[12 * (int64_t)i1c + 1];
        const int32_t u2c = m2c[12 * (int64_t)i2c], v2c = m2c[12 * (int64_t)i2c + 1];
        if (u1c >= u2c)
          vo_emit(out, cap, &cnt, -1, -1, -1, -1, -1, -1, (float)u1c, (float)v1c, i1c,
                  (float)u2c, (float)v2c, i2c);
      }
    }
    vo_index_free(&k1c);
    vo_index_free(&k2c);
  } else if (method == 2) {
    /* quad [unpinned]: SURVEY App. A.7 */
    if (n1p <= 0 || n2p <= 0 || n1c <= 0 || n2c <= 0) return 0;
    vo_index k1p = vo_index_make(p, m1p, n1p, ubn, vbn);
    vo_index k2p = vo_index_make(p, m2p, n2p, ubn, vbn);
    vo_index k1c = vo_index_make(p, m1c, n1c, ubn, vbn);
    vo_index k2c = vo_index_make(p, m2c, n2c, ubn, vbn);
    for (int32_t i1p = 0; i1p < n1p; i1p++) {
      const int32_t i2p = vo_find_match(p, m1p, i1p, m2p, k2p.bin_start, k2p.list, ubn, vbn, 0, -1, -1);
      const int32_t i2c = vo_find_match(p, m2p, i2p, m2c, k2c.bin_start, k2c.list, ubn, vbn, 1, -1, -1);
      const int32_t i1c = vo_find_match(p, m2c, i2c, m1c, k1c.bin_start, k1c.list, ubn, vbn, 0, -1, -1);
      const int32_t i1p2 = vo_find_match(p, m1c, i1c, m1p, k1p.bin_start, k1p.list, ubn, vbn, 1, -1, -1);
      if (i1p2 == i1p) {
        const int32_t u1p = m1p[12 * (int64_t)i1p], v1p = m1p[12 * (int64_t)i1p + 1];
        const int32_t u2p = m2p[12 * (int64_t)i2p], v2p = m2p[12 * (int64_t)i2p + 1];
        const int32_t u1c = m1c[12 * (int64_t)i1c], v1c = m1c[12 * (int64_t)i1c + 1];
        const int32_t u2c = m2c[12 * (int64_t)i2c], v2c = m2c[12 * (int64_t)i2c + 1];
        if (u1p >= u2p && u1c >= u2c)
          vo_emit(out, cap, &cnt, (float)u1p, (float)v1p, i1p, (float)u2p, (float)v2p, i2p,
                  (float)u1c, (float)v1c, i1c, (float)u2c, (float)v2c, i2c);
      }
    }
    vo_index_free(&k1p);
    vo_index_free(&k2p);
    vo_index_free(&k1c);
    vo_index_free(&k2c);
  } else {
    return -1;
  }
  *n_out = cnt;
  return 0;
}

/* Quad matching with the motion prior of stock libviso2's Matcher::matching [upstream-recollection: the code is
 * absent from the reference tree, whose matchFeatures accepts Tr_delta and ignores it, src/matcher.cpp:93-111; the
 * only part the reference pins is findMatch's u_,v_ cost term, src/matcher.cpp:257-262].  Hop 2 of the circle,
 * previous right -> current right, is searched around the position predicted for the current right image: the 3-d
 * point of the (1p, 2p) pair, moved by Tr_delta (row-major 4x4), projected with the intrinsics of `p`. */
int32_t vo_matching_quad_prior(const vo_params *p, const int32_t dims[3], const double *tr,
                               const int32_t *m1p, int32_t n1p, const int32_t *m2p, int32_t n2p,
                               const int32_t *m1c, int32_t n1c, const int32_t *m2c, int32_t n2c,
                               vo_p_match *out, int32_t cap, int32_t *n_out) {
  const int32_t ubn = (int32_t)ceilf((float)dims[0] / (float)p->match_binsize);
  const int32_t vbn = (int32_t)ceilf((float)dims[1] / (float)p->match_binsize);
  int32_t cnt = 0;
  *n_out = 0;
  if (n1p <= 0 || n2p <= 0 || n1c <= 0 || n2c <= 0) return 0;
  vo_index k1p = vo_index_make(p, m1p, n1p, ubn, vbn);
  vo_index k2p = vo_index_make(p, m2p, n2p, ubn, vbn);
  vo_index k1c = vo_index_make(p, m1c, n1c, ubn, vbn);
  vo_index k2c = vo_index_make(p, m2c, n2c, ubn, vbn);
  for (int32_t i1p = 0; i1p < n1p; i1p++) {
    const int32_t i2p = vo_find_match(p, m1p, i1p, m2p, k2p.bin_start, k2p.list, ubn, vbn, 0, -1, -1);
    const int32_t u1p = m1p[12 * (int64_t)i1p], v1p = m1p[12 * (int64_t)i1p + 1];
    const int32_t u2p = m2p[12 * (int64_t)i2p], v2p = m2p[12 * (int64_t)i2p + 1];
    double d = (double)u1p - (double)u2p;
    if (d < 1.0) d = 1.0;
    const double x1p = ((double)u1p - p->cu) * p->base / d;
    const double y1p = ((double)v1p - p->cv) * p->base / d;
    const double z1p = p->f * p->base / d;
    const double x2c = tr[0] * x1p + tr[1] * y1p + tr[2] * z1p + tr[3] - p->base;
    const double y2c = tr[4] * x1p + tr[5] * y1p + tr[6] * z1p + tr[7];
    const double z2c = tr[8] * x1p + tr[9] * y1p + tr[10] * z1p + tr[11];
    const double u2c_ = p->f * x2c / z2c + p->cu;
    const double v2c_ = p->f * y2c / z2c + p->cv;
    const int32_t i2c = vo_find_match(p, m2p, i2p, m2c, k2c.bin_start, k2c.list, ubn, vbn, 1, u2c_, v2c_);
    const int32_t i1c = vo_find_match(p, m2c, i2c, m1c, k1c.bin_start, k1c.list, ubn, vbn, 0, -1, -1);
    const int32_t i1p2 = vo_find_match(p, m1c, i1c, m1p, k1p.bin_start, k1p.list, ubn, vbn, 1, -1, -1);
    if (i1p2 == i1p) {
      const int32_t u1c = m1c[12 * (int64_t)i1c], v1c = m1c[12 * (int64_t)i1c + 1];
      const int32_t u2c = m2c[12 * (int64_t)i2c], v2c = m2c[12 * (int64_t)i2c + 1];
      if (u1p >= u2p && u1c >= u2c)
        vo_emit(out, cap, &cnt, (float)u1p, (float)v1p, i1p, (float)u2p, (float)v2p, i2p,
                (float)u1c, (float)v1c, i1c, (float)u2c, (float)v2c, i2c);
    }
  }
  vo_index_free(&k1p);
  vo_index_free(&k2p);
  vo_index_free(&k1c);
  vo_index_free(&k2c);
  *n_out = cnt;
  return 0;
}

void vo_match_all(const vo_params *p, const int32_t dims[3], const int32_t *m1, int32_t n1,
                  const int32_t *m2, int32_t n2, int32_t flow, int32_t *best) {
  const int32_t ubn = (int32_t)ceilf((float)dims[0] / (float)p->match_binsize);
  const int32_t vbn = (int32_t)ceilf((float)dims[1] / (float)p->match_binsize);
  vo_index k2 = vo_index_make(p, m2, n2, ubn, vbn);
  for (int32_t i = 0; i < n1; i++)
    best[i] = vo_find_match(p, m1, i, m2, k2.bin_start, k2.list, ubn, vbn, flow, -1, -1);
  vo_index_free(&k2);
}

/* ------------------------------------------------------------ bucketFeatures */

/* Matcher::rand_number (matcher.cpp:113-124): a 32-bit LFSR with taps
 * {32,22,2,1} evaluated in double arithmetic.  floor(number/2^k) is the
 * shift number>>k; the tap-32 term goes through int(double), which on the
 * reference's platform (x86-64 cvttsd2si) yields INT_MIN -- low bit 0 --
 * whenever number >= 2^31. */
static uint32_t vo_lfsr(uint32_t x) {
  uint32_t b = (x < 0x80000000u) ? (x & 1u) : 0u; /* tap 32: number / 2^0  */
  b ^= (x >> 10) & 1u;                            /* tap 22: number / 2^10 */
  b ^= (x >> 30) & 1u;                            /* tap 2 : number / 2^30 */
  b ^= (x >> 31) & 1u;                            /* tap 1 : number / 2^31 */
  return (x >> 1) + (b << 31);
}

int32_t vo_bucket_features(vo_p_match *pm, int32_t n, int32_t max_features,
                           float bucket_width, float bucket_height) {
  /* matcher.cpp:140-187 */
  float u_max = 0, v_max = 0;
  for (int32_t i = 0; i < n; i++) {
    if (pm[i].u1c > u_max) u_max = pm[i].u1c;
    if (pm[i].v1c > v_max) v_max = pm[i].v1c;
  }
  const int32_t cols = (int32_t)floorf(u_max / bucket_width) + 1;
  const int32_t rows = (int32_t)floorf(v_max / bucket_height) + 1;
  const int32_t nb = cols * rows;
  int32_t *start = (int32_t *)calloc((size_t)nb + 1, sizeof(int32_t));
  int32_t *which = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
  for (int32_t i = 0; i < n; i++) {
    const int32_t u = (int32_t)floorf(pm[i].u1c / bucket_width);
    const int32_t v = (int32_t)floorf(pm[i].v1c / bucket_height);
    which[i] = v * cols + u;
    start[which[i] + 1]++;
  }
  for (int32_t b = 0; b < nb; b++) start[b + 1] += start[b];
  vo_p_match *tmp = (vo_p_match *)malloc(sizeof(vo_p_match) * (size_t)(n > 0 ? n : 1));
  int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nb + 1));
  memcpy(fill, start, sizeof(int32_t) * (size_t)(nb + 1));
  for (int32_t i = 0; i < n; i++) tmp[fill[which[i]]++] = pm[i];
  int32_t out = 0;
  uint32_t rnd = 5;
  for (int32_t b = 0; b < nb; b++) {
    vo_p_match *bk = tmp + start[b];
    const int32_t len = start[b + 1] - start[b];
    for (int32_t i = 1; i < len; i++) { /* random_shuffle, matcher.cpp:126-138 */
      const int32_t j = (int32_t)(rnd % (uint32_t)(i + 1));
      rnd = vo_lfsr(rnd);
      vo_p_match t = bk[i]; bk[i] = bk[j]; bk[j] = t;
    }
    for (int32_t j = 0, k = 0; j < len; j++) {
      pm[out++] = bk[j];
      if (++k >= max_features) break;
    }
  }
  free(start); free(which); free(tmp); free(fill);
  return out;
}
