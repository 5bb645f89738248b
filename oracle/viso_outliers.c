/*
 * viso_outliers.c -- TEST INFRASTRUCTURE ONLY (see viso_oracle.h).
 *
 * CPU restatement of the reference's outlier filter, SURVEY.md row 8(f-1):
 *
 *   removeOutliers                 src/remove_outliers.cpp:4-94
 *   delaunator::Delaunator         src/delaunator.cpp:183-407 (sweep), :450-549
 *                                  (legalize), :551-603 (hash_key / add_triangle /
 *                                  link), predicates :23-181
 *
 * The reference's triangulator is an HLS-flavoured single-precision variant of
 * the sweep-hull "delaunator": points are visited in order of distance from the
 * bounding-box centre (stable insertion sort, :409-424), the seed points are
 * visited like any other point, a point that sees no hull edge is dropped, and
 * several sub-expressions are evaluated in double because of `0.5`, `3.0`, `4.0`
 * literals.  On integer pixel coordinates co-circular quadruples are the rule,
 * so the set of triangles depends on all of this; every rounding step below is
 * therefore kept where the reference has it (compile with -ffp-contract=off).
 *
 * Pinned: against the reference's own removeOutliers / Delaunator compiled
 * into oracle/_ref (tests/test_oracle.py) and tests/golden/outliers_*.npz.
 *
 * Deliberate differences (none changes a result the reference defines):
 *  - storage is sized by n instead of POINT_L/TRIANGLE_L/HASH_L (delaunator.hpp:10-13);
 *  - the flip stack grows on demand; the reference's has EDGE_L = 13 slots and
 *    writes past them when a legalisation goes deeper (undefined behaviour).
 *    vo_delaunay reports the depth reached so tests can tell when a comparison
 *    with the reference is meaningful;
 *  - degenerate input the reference would index out of bounds on (no seed
 *    triangle: all points identical or collinear) yields zero triangles.
 */
#include "viso_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NIL 2147483647 /* delaunator.cpp:13 INVALID_INDEX */

typedef struct {
  int32_t n;
  const float *x, *y;
  int32_t *tri, ntri;   /* triangle corners, 3 per triangle */
  int32_t *half, nhalf; /* half-edge twins */
  int32_t *hprev, *hnext, *htri;
  int32_t *hash, hsize;
  int32_t hull_start;
  float cx, cy; /* seed circumcentre: origin of the angular hash */
  int32_t *stack, stack_cap, max_depth;
} sweep_t;

/* Point::determinant of (p1-p0, p2-p0) with the reference's rejection of
 * near-degenerate triples (delaunator.cpp:99-121). */
static int ccw(float px, float py, float qx, float qy, float rx, float ry) {
  const float ax = qx - px, ay = qy - py, bx = rx - px, by = ry - py;
  const float det = ax * by - ay * bx;
  const float len = (ax * ax + ay * ay) + (bx * bx + by * by);
  if (det == 0) return 0;
  const float rel = fabsf(len / det);
  if ((double)rel > 1e14) return 0;
  return det > 0;
}

/* delaunator.cpp:23-39: squared circumradius, +inf for degenerate triples
 * (DOUBLE_MAX does not fit data_type = float). */
static float circumradius2(float ax, float ay, float bx, float by, float cx, float cy) {
  const float dx = bx - ax, dy = by - ay, ex = cx - ax, ey = cy - ay;
  const float bl = dx * dx + dy * dy, cl = ex * ex + ey * ey;
  const float det = dx * ey - dy * ex;
  /* float numerator, then `* 0.5 / det` in double, rounded to float by Point() */
  const float rx = (float)((double)(ey * bl - dy * cl) * 0.5 / (double)det);
  const float ry = (float)((double)(dx * cl - ex * bl) * 0.5 / (double)det);
  if (bl != 0 && cl != 0 && det != 0 && bl == bl && cl == cl && det == det) return rx * rx + ry * ry;
  return INFINITY;
}

/* delaunator.cpp:149-175 */
static int in_circle(const sweep_t *s, int32_t a, int32_t b, int32_t c, int32_t p) {
  const float dx = s->x[a] - s->x[p], dy = s->y[a] - s->y[p];
  const float ex = s->x[b] - s->x[p], ey = s->y[b] - s->y[p];
  const float fx = s->x[c] - s->x[p], fy = s->y[c] - s->y[p];
  const float ap = dx * dx + dy * dy, bp = ex * ex + ey * ey, cp = fx * fx + fy * fy;
  const float t1 = dx * (ey * cp - bp * fy);
  const float t2 = dy * (ex * cp - bp * fx);
  const float t3 = ap * (ex * fy - ey * fx);
  return ((t1 - t2) + t3) < 0.0f;
}

/* delaunator.cpp:178-182, :551-557: angular bucket of (x,y) about the seed
 * circumcentre.  The point coinciding with the centre has no angle (the
 * reference would index with INT_MIN); it gets bucket 0. */
static int32_t hash_key(const sweep_t *s, float x, float y) {
  const float dx = x - s->cx, dy = y - s->cy;
  const float p = dx / (fabsf(dx) + fabsf(dy));
  const float ang = (float)((dy > 0.0f ? 3.0 - (double)p : 1.0 + (double)p) / 4.0);
  const float k = floorf(ang * (float)s->hsize);
  if (!(k == k)) return 0;
  const int32_t i = (int32_t)k;
  return i >= s->hsize ? i % s->hsize : i;
}

/* delaunator.cpp:585-603 */
static void link(sweep_t *s, int32_t a, int32_t b) {
  if (a == s->nhalf) s->half[s->nhalf++] = b;
  else if (a < s->nhalf) s->half[a] = b;
  if (b != NIL) {
    if (b == s->nhalf) s->half[s->nhalf++] = a;
    else if (b < s->nhalf) s->half[b] = a;
  }
}

/* delaunator.cpp:566-583 */
static int32_t add_triangle(sweep_t *s, int32_t i0, int32_t i1, int32_t i2, int32_t a, int32_t b, int32_t c) {
  const int32_t t = s->ntri;
  s->tri[s->ntri++] = i0;
  s->tri[s->ntri++] = i1;
  s->tri[s->ntri++] = i2;
  link(s, t, a);
  link(s, t + 1, b);
  link(s, t + 2, c);
  return t;
}

static void push_edge(sweep_t *s, int32_t depth, int32_t e) {
  if (depth >= s->stack_cap) {
    s->stack_cap *= 2;
    s->stack = (int32_t *)realloc(s->stack, sizeof(int32_t) * (size_t)s->stack_cap);
  }
  s->stack[depth] = e;
  if (depth + 1 > s->max_depth) s->max_depth = depth + 1;
}

/* delaunator.cpp:450-549: restore the Delaunay condition across half-edge a,
 * following flips with an explicit stack; returns the half-edge `ar` of the
 * last triangle looked at. */
static int32_t legalize(sweep_t *s, int32_t a) {
  int32_t depth = 0, ar = 0;
  for (;;) {
    const int32_t b = s->half[a];
    const int32_t a0 = a - a % 3;
    ar = a0 + (a + 2) % 3;
    if (b == NIL) {
      if (depth == 0) break;
      a = s->stack[--depth];
      continue;
    }
    const int32_t b0 = b - b % 3;
    const int32_t al = a0 + (a + 1) % 3, bl = b0 + (b + 2) % 3;
    const int32_t p0 = s->tri[ar], pr = s->tri[a], pl = s->tri[al], p1 = s->tri[bl];
    if (in_circle(s, p0, pr, pl, p1)) {
      s->tri[a] = p1;
      s->tri[b] = p0;
      const int32_t hbl = s->half[bl];
      if (hbl == NIL) { /* the flipped edge was a hull edge: repoint its hull record */
        int32_t e = s->hull_start;
        do {
          if (s->htri[e] == bl) { s->htri[e] = a; break; }
          e = s->hprev[e];
        } while (e != s->hull_start);
      }
      link(s, a, hbl);
      link(s, b, s->half[ar]);
      link(s, ar, bl);
      push_edge(s, depth++, b0 + (b + 1) % 3);
    } else {
      if (depth == 0) break;
      a = s->stack[--depth];
    }
  }
  return ar;
}

int32_t vo_delaunay(const float *xy, int32_t n, int32_t *tri_out, int32_t cap, int32_t *max_depth) {
  if (max_depth) *max_depth = 0;
  if (n < 3) return 0;
  sweep_t s;
  memset(&s, 0, sizeof(s));
  float *x = (float *)malloc(sizeof(float) * (size_t)n), *y = (float *)malloc(sizeof(float) * (size_t)n);
  for (int32_t i = 0; i < n; i++) { x[i] = xy[2 * i]; y[i] = xy[2 * i + 1]; }
  s.n = n; s.x = x; s.y = y;
  const size_t ncorner = 6 * (size_t)n + 16; /* <= 2n-5 triangles in any planar triangulation */
  s.tri = (int32_t *)malloc(sizeof(int32_t) * ncorner);
  s.half = (int32_t *)malloc(sizeof(int32_t) * ncorner);
  s.hprev = (int32_t *)calloc((size_t)n, sizeof(int32_t));
  s.hnext = (int32_t *)calloc((size_t)n, sizeof(int32_t));
  s.htri = (int32_t *)calloc((size_t)n, sizeof(int32_t));
  s.hsize = (int32_t)ceil(sqrt((double)n)); /* :192 */
  s.hash = (int32_t *)malloc(sizeof(int32_t) * (size_t)s.hsize);
  s.stack_cap = 64;
  s.stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)s.stack_cap);
  int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  float *dist = (float *)malloc(sizeof(float) * (size_t)n);
  int32_t result = 0;

  /* bounding box, :195-215 */
  float max_x = -INFINITY, max_y = -INFINITY, min_x = INFINITY, min_y = INFINITY;
  for (int32_t i = 0; i < n; i++) {
    ids[i] = i;
    if (x[i] < min_x) min_x = x[i];
    if (y[i] < min_y) min_y = y[i];
    if (max_x < x[i]) max_x = x[i];
    if (max_y < y[i]) max_y = y[i];
  }
  for (int32_t i = 0; i < s.hsize; i++) s.hash[i] = NIL;
  const float width = max_x - min_x, height = max_y - min_y;
  const float span = width * width + height * height;
  const float ccx = (min_x + max_x) / 2, ccy = (min_y + max_y) / 2;

  /* first seed: closest to the box centre, :222-232 */
  int32_t i0 = NIL, i1 = NIL, i2 = NIL;
  float best = INFINITY;
  for (int32_t i = 0; i < n; i++) {
    const float dx = x[i] - ccx, dy = y[i] - ccy;
    const float d = dx * dx + dy * dy;
    dist[i] = d;
    if (d < best) { i0 = i; best = d; }
  }
  /* visiting order: ascending distance from the box centre, ties in input
   * order (insertion sort :409-424 moves an element past strictly larger keys only) */
  for (int32_t i = 1; i < n; i++) {
    const float key = dist[i];
    const int32_t id = ids[i];
    int32_t j = i - 1;
    while (j >= 0 && key < dist[j]) { ids[j + 1] = ids[j]; dist[j + 1] = dist[j]; j--; }
    dist[j + 1] = key;
    ids[j + 1] = id;
  }
  if (i0 == NIL) goto done;
  /* second seed: nearest distinct point, :240-249 */
  best = INFINITY;
  for (int32_t i = 0; i < n; i++) {
    if (i == i0) continue;
    const float dx = x[i] - x[i0], dy = y[i] - y[i0];
    const float d = dx * dx + dy * dy;
    if (d < best && d > 0.0f) { i1 = i; best = d; }
  }
  if (i1 == NIL) goto done;
  /* third seed: smallest circumcircle with the first two, :253-262 */
  best = INFINITY;
  for (int32_t i = 0; i < n; i++) {
    const float r = circumradius2(x[i0], y[i0], x[i1], y[i1], x[i], y[i]);
    if (r < best && !(i == i0 || i == i1)) { i2 = i; best = r; }
  }
  if (i2 == NIL) goto done;
  if (ccw(x[i0], y[i0], x[i1], y[i1], x[i2], y[i2])) { const int32_t t = i1; i1 = i2; i2 = t; } /* :264-268 */
  {
    /* circumcentre of the seed triangle, :124-146: `a + num * 0.5 / d` in double */
    const float ax = x[i0], ay = y[i0];
    const float dx = x[i1] - ax, dy = y[i1] - ay, ex = x[i2] - ax, ey = y[i2] - ay;
    const float bl = dx * dx + dy * dy, cl = ex * ex + ey * ey;
    const float d = dx * ey - dy * ex;
    s.cx = (float)((double)ax + (double)(ey * bl - dy * cl) * 0.5 / (double)d);
    s.cy = (float)((double)ay + (double)(dx * cl - ex * bl) * 0.5 / (double)d);
  }
  s.hull_start = i0;
  s.hnext[i0] = s.hprev[i2] = i1;
  s.hnext[i1] = s.hprev[i0] = i2;
  s.hnext[i2] = s.hprev[i1] = i0;
  s.htri[i0] = 0; s.htri[i1] = 1; s.htri[i2] = 2;
  s.hash[hash_key(&s, x[i0], y[i0])] = i0;
  s.hash[hash_key(&s, x[i1], y[i1])] = i1;
  s.hash[hash_key(&s, x[i2], y[i2])] = i2;
  add_triangle(&s, i0, i1, i2, NIL, NIL, NIL);

  /* the sweep, :303-404.  Every point is offered to the hull, the seeds too. */
  for (int32_t k = 0; k < n; k++) {
    const int32_t i = ids[k];
    const float px = x[i], py = y[i];
    /* a live hull vertex near the point's angle */
    int32_t start = 0;
    const int32_t key = hash_key(&s, px, py);
    for (int32_t j = 0; j < s.hsize; j++) {
      const int32_t slot = key + j;
      start = s.hash[slot >= s.hsize ? slot % s.hsize : slot];
      if (start != NIL && start != s.hnext[start]) break;
    }
    if (start == NIL) continue; /* unreachable while the hull has live vertices */
    start = s.hprev[start];
    int32_t e = start, q;
    /* first hull edge (e -> next e) that the point sees; coincident points and
     * points that see nothing are dropped */
    for (;;) {
      q = s.hnext[e];
      {
        const float ex = x[e] - px, ey = y[e] - py, qx = x[q] - px, qy = y[q] - py;
        if ((double)((ex * ex + ey * ey) / span) < 1e-20 || (double)((qx * qx + qy * qy) / span) < 1e-20) { e = NIL; break; }
      }
      if (ccw(px, py, x[e], y[e], x[q], y[q])) break;
      e = q;
      if (e == start) { e = NIL; break; }
    }
    if (e == NIL) continue;

    int32_t t = add_triangle(&s, e, i, s.hnext[e], NIL, NIL, s.htri[e]);
    s.htri[i] = legalize(&s, t + 2);
    s.htri[e] = t;

    /* forward along the hull while edges stay visible */
    int32_t next = s.hnext[e];
    for (;;) {
      q = s.hnext[next];
      if (!ccw(px, py, x[next], y[next], x[q], y[q])) break;
      t = add_triangle(&s, next, i, q, s.htri[i], NIL, s.htri[next]);
      s.htri[i] = legalize(&s, t + 2);
      s.hnext[next] = next; /* no longer a hull vertex */
      next = q;
    }
    /* backward, only when the walk above began at its starting vertex */
    if (e == start) {
      for (;;) {
        q = s.hprev[e];
        if (!ccw(px, py, x[q], y[q], x[e], y[e])) break;
        t = add_triangle(&s, q, i, e, NIL, s.htri[e], s.htri[q]);
        legalize(&s, t + 2);
        s.htri[q] = t;
        s.hnext[e] = e;
        e = q;
      }
    }
    s.hprev[i] = e;
    s.hull_start = e;
    s.hprev[next] = i;
    s.hnext[e] = i;
    s.hnext[i] = next;
    s.hash[hash_key(&s, px, py)] = i;
    s.hash[hash_key(&s, x[e], y[e])] = e;
    if ((size_t)s.ntri + 12 > ncorner) break; /* cannot happen for a planar triangulation */
  }
  result = s.ntri;
  if (tri_out) memcpy(tri_out, s.tri, sizeof(int32_t) * (size_t)(result < cap ? result : cap));
  if (max_depth) *max_depth = s.max_depth;
done:
  free(x); free(y); free(s.tri); free(s.half); free(s.hprev); free(s.hnext); free(s.htri);
  free(s.hash); free(s.stack); free(ids); free(dist);
  return result;
}

int32_t vo_remove_outliers(vo_p_match *pm, int32_t n, int32_t *max_depth) {
  if (max_depth) *max_depth = 0;
  if (n <= 3) return n; /* remove_outliers.cpp:6-7 */
  float *xy = (float *)malloc(sizeof(float) * 2 * (size_t)n);
  for (int32_t i = 0; i < n; i++) { xy[2 * i] = pm[i].u1c; xy[2 * i + 1] = pm[i].v1c; } /* :27 */
  const int32_t cap = 6 * n + 16;
  int32_t *tri = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
  const int32_t ncorner = vo_delaunay(xy, n, tri, cap, max_depth);
  int32_t *support = (int32_t *)calloc((size_t)n, sizeof(int32_t));
  const float tol = 5; /* hard-coded, :34 -- not param.outlier_flow_tolerance */
  for (int32_t t = 0; t + 2 < ncorner; t += 3) {
    const int32_t a = tri[t], b = tri[t + 1], c = tri[t + 2];
    const float au = pm[a].u1c - pm[a].u1p, av = pm[a].v1c - pm[a].v1p;
    const float bu = pm[b].u1c - pm[b].u1p, bv = pm[b].v1c - pm[b].v1p;
    const float cu = pm[c].u1c - pm[c].u1p, cv = pm[c].v1c - pm[c].v1p;
    /* an edge supports its end points when their flows agree in L1 (:66-68) */
    const int ab = fabsf(au - bu) + fabsf(av - bv) < tol;
    const int bc = fabsf(bu - cu) + fabsf(bv - cv) < tol;
    const int ac = fabsf(au - cu) + fabsf(av - cv) < tol;
    support[a] += ab + ac;
    support[b] += ab + bc;
    support[c] += bc + ac;
  }
  int32_t kept = 0;
  for (int32_t i = 0; i < n; i++)
    if (support[i] >= 4) pm[kept++] = pm[i]; /* :88-90 */
  free(xy); free(tri); free(support);
  return kept;
}
