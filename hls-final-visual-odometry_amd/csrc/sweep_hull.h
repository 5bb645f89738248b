// sweep_hull.h -- the sequential core of the reference's outlier vote, shared by the host build
// (outliers.cpp, g++-style C++) and the device build (kernels_vote.hip, one lane per match list):
// the single-precision sweep-hull triangulator of src/delaunator.cpp:183-407 with its legalisation
// (:450-549), the predicates (:23-181) and the angular hash (:551-557).
//
// The triangulation is one chain of hull updates and edge flips per match list, and on integer pixel
// coordinates (co-circular quadruples everywhere) its outcome is a property of that chain -- the
// visiting order and every float rounding step -- not of the point set.  So host and device run THIS
// code, statement for statement; what differs is who prepares the input (visiting order, seed points)
// and who reads the triangles afterwards.  Build every translation unit that includes it with
// -ffp-contract=off: `a * b + c` must round twice, as in the reference's x86 build.
//
// Storage is laid out for a machine where every dependent load costs hundreds of cycles and every
// instruction of a lone lane costs a full wavefront's issue slot:
//   * a hull node carries its point, so walking the hull is one load per step, not two;
//   * the mesh is an array of HALF-EDGE records {origin corner, its point, twin}, 16 bytes each, a
//     triangle = three consecutive records (4 t + 0..2; half-edge ids never leave this file, only the
//     corners do).  The circumcircle test of a legalisation step needs ONE 16-byte load -- the record
//     opposite the shared edge in the twin's triangle; the three records of the triangle being
//     legalised stay in registers across flips, and so do the records a flip leaves behind for the
//     edge it queues.  No slot of a record is ever selected by a computed index: which of the three
//     half-edges is meant is an address, not a select chain;
//   * the angular hash and the flip stack sit behind a pointer type of the includer's choice
//     (the device build puts them in LDS).
#pragma once
#include <cstdint>

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__) || defined(VH_SH_DEVICE)
#define VH_SH_FN __host__ __device__ inline
#else
#define VH_SH_FN inline
#endif

namespace vh_sh {
// (the two builds of this header live in one shared library: the inline namespace keeps their symbols apart)
#ifdef VH_SH_LINK16
inline namespace link16 {
#else
inline namespace link32 {
#endif

constexpr int32_t kNone = -1;  // the reference's INVALID_INDEX (delaunator.cpp:13); ids are non-negative

struct Pt { float x, y; };

// Hull links are point indices.  The device build (VH_SH_LINK16: lists of at most 65 535 points) keeps them in 16 bits,
// which makes a node 16 bytes -- one aligned load, four nodes per 64-byte sector instead of two.
#ifdef VH_SH_LINK16
typedef uint16_t link_t;
struct alignas(16) Node {  // one per point, valid from the point's insertion on
#else
typedef int32_t link_t;
struct alignas(4) Node {
#endif
  float x, y;
  link_t next, prev;   // hull links (next == own index: off the hull)
  int32_t edge_of;     // half-edge of the triangle behind the hull edge node -> next
};
VH_SH_FN Node make_node(float x, float y, int32_t next, int32_t prev, int32_t edge_of) {
  Node n;
  n.x = x; n.y = y; n.next = static_cast<link_t>(next); n.prev = static_cast<link_t>(prev); n.edge_of = edge_of;
  return n;
}

struct alignas(16) Half {  // one half-edge
  int32_t p;     // corner it starts at
  float x, y;    // that corner's point
  int32_t twin;  // the same edge seen from the neighbouring triangle (kNone: hull)
};
// Half-edge ids are 4 t + k (k = 0..2) so that next / previous are two instructions; in MEMORY a triangle is three
// consecutive records, 48 bytes (round 4 stored four slots per triangle: a third more footprint and traffic for nothing).
VH_SH_FN int32_t next_half(int32_t h) { return (h & 3) == 2 ? h - 2 : h + 1; }
VH_SH_FN int32_t prev_half(int32_t h) { return (h & 3) == 0 ? h + 2 : h - 1; }
VH_SH_FN int32_t half_slot(int32_t h) { return (h >> 2) * 3 + (h & 3); }

// orientation test of delaunator.cpp:99-121 (the `clockwise` twin is unused there)
VH_SH_FN bool turns_ccw(Pt p, Pt q, Pt r) {
  const float ux = q.x - p.x, uy = q.y - p.y, vx = r.x - p.x, vy = r.y - p.y;
  const float det = ux * vy - uy * vx;
  if (det == 0) return false;
  const float size = (ux * ux + uy * uy) + (vx * vx + vy * vy);
  // |size / det| > 1e14 is impossible when |det| >= 1 and size <= 9e13 (the quotient cannot exceed size):
  // on pixel coordinates det is a non-zero integer, so the division is only done for exotic inputs
  if (!(__builtin_fabsf(det) >= 1.0f && size <= 9e13f) && static_cast<double>(__builtin_fabsf(size / det)) > 1e14) return false;
  return det > 0;
}

// offset of the circumcentre of (a,b,c) from a; false for a degenerate triple
// (delaunator.cpp:23-39 and :124-146 share this arithmetic)
VH_SH_FN bool circum_offset(Pt a, Pt b, Pt c, double &ox, double &oy) {
  const float dx = b.x - a.x, dy = b.y - a.y, ex = c.x - a.x, ey = c.y - a.y;
  const float bl = dx * dx + dy * dy, cl = ex * ex + ey * ey;
  const float det = dx * ey - dy * ex;
  ox = static_cast<double>(ey * bl - dy * cl) * 0.5 / static_cast<double>(det);
  oy = static_cast<double>(dx * cl - ex * bl) * 0.5 / static_cast<double>(det);
  return (bl > 0 || bl < 0) && (cl > 0 || cl < 0) && (det > 0 || det < 0);
}

// squared circumradius of (a,b,c) as the seed search compares it (delaunator.cpp:23-39): +inf for a degenerate triple
VH_SH_FN float circum_r2(Pt a, Pt b, Pt c) {
  double ox, oy;
  if (!circum_offset(a, b, c, ox, oy)) return __builtin_inff();
  const float rx = static_cast<float>(ox), ry = static_cast<float>(oy);
  return rx * rx + ry * ry;
}

// delaunator.cpp:149-175
VH_SH_FN bool inside_circumcircle(Pt A, Pt B, Pt C, Pt Q) {
  const float dx = A.x - Q.x, dy = A.y - Q.y, ex = B.x - Q.x, ey = B.y - Q.y, fx = C.x - Q.x, fy = C.y - Q.y;
  const float ap = dx * dx + dy * dy, bp = ex * ex + ey * ey, cp = fx * fx + fy * fy;
  const float s1 = dx * (ey * cp - bp * fy), s2 = dy * (ex * cp - bp * fx), s3 = ap * (ex * fy - ey * fx);
  return (s1 - s2) + s3 < 0.0f;
}

// ceil(sqrt((double)n)) (delaunator.cpp:264) in integers: for n < 2^31 the two agree (a non-square's
// root is further than 2^-32 relative from an integer, a square's root is exact)
VH_SH_FN int32_t hash_size(int32_t n) {
  int32_t k = 0;
  while (static_cast<int64_t>(k) * k < n) k++;
  return k;
}

// FastI32: pointer to int32 for the angular hash and the flip stack (host: int32_t *; device: an LDS pointer)
template <class FastI32>
struct Sweep {
  // storage (caller-owned): node[n], half[6 n] (a triangulation of n points has fewer than 2 n triangles,
  // three records each: half_slot), bucket[hash_size(n)], pend[pend_cap]
  Node *node;
  Half *half;
  FastI32 bucket;
  FastI32 pend;
  int32_t pend_cap;
  const Pt *pts;
  // the visiting order: ascending distance from the bounding box's centre, stable (:192-232, :409-424); null: the points
  // are numbered in visiting order already (the device build: the hull's nodes are then the most recently visited
  // points, neighbours in memory, instead of being scattered over the whole node array)
  const int32_t *order;
  int32_t n;
  // state
  int32_t ntri;        // triangles emitted
  int32_t hull_entry;
  int32_t buckets;
  Pt origin;
  float span, span_eps;
  bool span_eps_ok;
  int32_t overflow;    // the flip stack went beyond pend_cap (the reference's has 13 slots and is undefined beyond)
  int32_t max_depth;
#ifdef VH_SH_STATS
  int64_t st_fix = 0, st_fix_steps = 0, st_flips = 0, st_legal_iters = 0, st_walk = 0, st_pop_miss = 0;
#endif

  VH_SH_FN Half &hf(int32_t h) const { return half[half_slot(h)]; }

  // delaunator.cpp:178-182 + :551-557
  VH_SH_FN int32_t bucket_of(Pt q) const {
    const float dx = q.x - origin.x, dy = q.y - origin.y;
    const float t = dx / (__builtin_fabsf(dx) + __builtin_fabsf(dy));
    const float turn = static_cast<float>((dy > 0.0f ? 3.0 - static_cast<double>(t) : 1.0 + static_cast<double>(t)) / 4.0);
    const float scaled = __builtin_floorf(turn * static_cast<float>(buckets));
    if (scaled != scaled) return 0;  // q == origin: the reference indexes out of bounds here
    const int32_t k = static_cast<int32_t>(scaled);
    return k >= buckets ? k % buckets : k;
  }

  // delaunator.cpp:566-583 (+ link, :585-603): the three new records are returned as well
  VH_SH_FN int32_t emit(int32_t i0, Pt P0, int32_t i1, Pt P1, int32_t i2, Pt P2, int32_t a, int32_t b, int32_t c, Half &H0, Half &H1, Half &H2) {
    const int32_t h = 4 * ntri++;
    H0 = Half{i0, P0.x, P0.y, a}; H1 = Half{i1, P1.x, P1.y, b}; H2 = Half{i2, P2.x, P2.y, c};
    Half *T = half + 3 * (h >> 2);
    T[0] = H0; T[1] = H1; T[2] = H2;
    if (a != kNone) hf(a).twin = h;
    if (b != kNone) hf(b).twin = h + 1;
    if (c != kNone) hf(c).twin = h + 2;
    return h;
  }

  // the edge that a flip moved was on the hull: the hull node that pointed at it (found by walking
  // the hull as it stood when the current point's insertion began, :490-500) points at `a` now
  VH_SH_FN void hull_fix(int32_t bl, int32_t a) {
    int32_t e = hull_entry;
#ifdef VH_SH_STATS
    st_fix++;
#endif
    do {
#ifdef VH_SH_STATS
      st_fix_steps++;
#endif
      if (node[e].edge_of == bl) { node[e].edge_of = a; break; }
      e = node[e].prev;
    } while (e != hull_entry);
  }

  // delaunator.cpp:450-549.  Ha, Hal, Har: the records of a, next(a), prev(a) as they stand in memory.
  VH_SH_FN int32_t legalize(int32_t a, Half Ha, Half Hal, Half Har) {
    int32_t depth = 0, top = kNone;  // the flip stack: its top in a register, the rest in pend[]
    // what the latest flip left of the twin's triangle, for the edge it queued (cb): the records of cb, next(cb), prev(cb)
    int32_t cb = kNone;
    Half Cs = Ha, Cn = Ha, Cp = Ha;
    while (true) {
#ifdef VH_SH_STATS
      st_legal_iters++;
#endif
      const int32_t b = Ha.twin;
      bool flipped = false;
      if (b != kNone) {
        const int32_t bl = prev_half(b);
        Half Hbl = hf(bl);
        if (inside_circumcircle(Pt{Har.x, Har.y}, Pt{Ha.x, Ha.y}, Pt{Hal.x, Hal.y}, Pt{Hbl.x, Hbl.y})) {
#ifdef VH_SH_STATS
          st_flips++;
#endif
          // tri[a] = p1, tri[b] = p0; link(a, twin[bl]); link(b, twin[ar]); link(ar, bl)
          const int32_t outer = Hbl.twin, har = Har.twin, ar = prev_half(a), br = next_half(b);
          if (outer == kNone) hull_fix(bl, a);
          Ha = Half{Hbl.p, Hbl.x, Hbl.y, outer};
          hf(a) = Ha;
          if (outer != kNone) hf(outer).twin = a;
          const Half Hb{Har.p, Har.x, Har.y, har};
          hf(b) = Hb;
          if (har != kNone) hf(har).twin = b;
          Har.twin = bl; hf(ar).twin = bl;
          Hbl.twin = ar; hf(bl).twin = ar;
          // push b0 + (b + 1) % 3
          if (depth > 0) {
            if (depth - 1 < pend_cap) pend[depth - 1] = top;
            else {
              // The stack is full (the reference's has 13 slots and is undefined beyond).  The flip above is complete and
              // the mesh consistent, but the edge `top` can no longer be remembered: STOP here -- nothing is popped and
              // nothing more is written for this list (insert_all checks `overflow` after every legalisation), so a
              // lost entry can never come back as an index.  The list is reported (VH_VOTE_STACK), not approximated.
              overflow = 1;
              return prev_half(a);
            }
          }
          top = br;
          depth++;
          if (depth > max_depth) max_depth = depth;
          cb = br; Cs = hf(br); Cn = Hbl; Cp = Hb;  // (nothing is written between here and the moment br is taken off the stack, unless another flip replaces these)
          flipped = true;  // and look at edge a again
        }
      }
      if (!flipped) {
        if (depth == 0) break;
        a = top;
        depth--;
        if (depth > 0) top = pend[depth - 1];  // (depth - 1 < pend_cap: a push beyond it ends the sweep)
        if (a == cb) { Ha = Cs; Hal = Cn; Har = Cp; }
        else {
#ifdef VH_SH_STATS
          st_pop_miss++;
#endif
          Ha = hf(a); Hal = hf(next_half(a)); Har = hf(prev_half(a));
        }
        cb = kNone;
      }
    }
    return prev_half(a);
  }

  // Point::equal, delaunator.hpp:64-71: (d2 / span) < 1e-20 in double.  d2 > span * 1e-18 (a normal float) puts
  // the quotient far above 1e-20 without dividing; anything else takes the literal form.
  VH_SH_FN bool coincides(Pt a, Pt b) const {
    const float dx = b.x - a.x, dy = b.y - a.y, d2 = dx * dx + dy * dy;
    if (span_eps_ok && d2 > span_eps) return false;
    return static_cast<double>(d2 / span) < 1e-20;
  }

  // :240-301: orientation of the seed triangle, origin of the angular hash, the first hull.
  // span_: squared diagonal of the bounding box (:192-205).
  VH_SH_FN bool seed(int32_t s0, int32_t s1, int32_t s2, float span_) {
    ntri = 0; overflow = 0; max_depth = 0;
    span = span_;
    span_eps = span * 1e-18f;
    span_eps_ok = span_eps >= 1.17549435e-38f && span_eps < __builtin_inff();
    if (s0 == kNone || s1 == kNone || s2 == kNone) return false;
    if (turns_ccw(pts[s0], pts[s1], pts[s2])) { const int32_t t = s1; s1 = s2; s2 = t; }
    const Pt P0 = pts[s0], P1 = pts[s1], P2 = pts[s2];
    {
      double ox, oy;
      circum_offset(P0, P1, P2, ox, oy);
      origin = Pt{static_cast<float>(static_cast<double>(P0.x) + ox), static_cast<float>(static_cast<double>(P0.y) + oy)};
    }
    buckets = hash_size(n);
    for (int32_t k = 0; k < buckets; k++) bucket[k] = kNone;
    hull_entry = s0;
    node[s0] = make_node(P0.x, P0.y, s1, s2, 0);
    node[s1] = make_node(P1.x, P1.y, s2, s0, 1);
    node[s2] = make_node(P2.x, P2.y, s0, s1, 2);
    bucket[bucket_of(P0)] = s0;
    bucket[bucket_of(P1)] = s1;
    bucket[bucket_of(P2)] = s2;
    Half H0, H1, H2;
    emit(s0, P0, s1, P1, s2, P2, kNone, kNone, kNone, H0, H1, H2);
    return true;
  }

  // :303-404; the three seeds are offered to the hull like every other point
  VH_SH_FN void insert_all() {
    int32_t i_next = n > 0 ? (order ? order[0] : 0) : 0;
    Pt q_next = n > 0 ? pts[i_next] : Pt{0, 0};
    for (int32_t k = 0; k < n; k++) {
      const int32_t i = i_next;
      const Pt q = q_next;
      if (k + 1 < n) { i_next = order ? order[k + 1] : k + 1; q_next = pts[i_next]; }  // (requested a whole insertion ahead of its use)
      int32_t at = kNone;
      Node N{};
      const int32_t first = bucket_of(q);
      for (int32_t j = 0; j < buckets; j++) {
        const int32_t slot = first + j;
        at = bucket[slot >= buckets ? slot % buckets : slot];
        if (at != kNone) {
          N = node[at];
          if (at != N.next) break;
        }
      }
      if (at == kNone) continue;
      const int32_t begin = N.prev;
      int32_t e = begin;
      Node E = node[e], F;
      int32_t f;
      while (true) {  // first hull edge e -> next[e] facing q
#ifdef VH_SH_STATS
        st_walk++;
#endif
        f = E.next;
        F = node[f];
        if (coincides(q, Pt{E.x, E.y}) || coincides(q, Pt{F.x, F.y})) { e = kNone; break; }
        if (turns_ccw(q, Pt{E.x, E.y}, Pt{F.x, F.y})) break;
        e = f;
        E = F;
        if (e == begin) { e = kNone; break; }
      }
      if (e == kNone) continue;  // duplicate, or nothing visible: the point is left out

      // emit(e, i, f): half-edges t: e -> i, t + 1: i -> f, t + 2: f -> e (the one shared with the old mesh)
      Half H0, H1, H2;
      int32_t t = emit(e, Pt{E.x, E.y}, i, q, f, Pt{F.x, F.y}, kNone, kNone, E.edge_of, H0, H1, H2);
      int32_t edge_i = legalize(t + 2, H2, H0, H1);
      if (overflow) return;  // flip stack exhausted: the sweep ends here, the list is refused
      node[e].edge_of = t;

      // (a legalisation may have re-pointed edge_of of any hull node: it is read again where it is used)
      int32_t fwd = f;
      Node FW = F;
      while (true) {
        const int32_t f2 = FW.next;
        const Node F2 = node[f2];
        const int32_t eo_fwd = node[fwd].edge_of;
        if (!turns_ccw(q, Pt{FW.x, FW.y}, Pt{F2.x, F2.y})) break;
        t = emit(fwd, Pt{FW.x, FW.y}, i, q, f2, Pt{F2.x, F2.y}, edge_i, kNone, eo_fwd, H0, H1, H2);
        edge_i = legalize(t + 2, H2, H0, H1);
        if (overflow) return;
        node[fwd].next = static_cast<link_t>(fwd);  // off the hull
        fwd = f2;
        FW = F2;
      }
      if (e == begin) {
        while (true) {
          const int32_t b = E.prev;
          const Node Bn = node[b];
          const int32_t eo_e = node[e].edge_of;
          if (!turns_ccw(q, Pt{Bn.x, Bn.y}, Pt{E.x, E.y})) break;
          t = emit(b, Pt{Bn.x, Bn.y}, i, q, e, Pt{E.x, E.y}, kNone, eo_e, Bn.edge_of, H0, H1, H2);
          legalize(t + 2, H2, H0, H1);
          if (overflow) return;
          node[b].edge_of = t;
          node[e].next = static_cast<link_t>(e);
          e = b;
          E = Bn;
        }
      }
      node[i] = make_node(q.x, q.y, fwd, e, edge_i);
      hull_entry = e;
      node[fwd].prev = static_cast<link_t>(i);
      node[e].next = static_cast<link_t>(i);
      bucket[first] = i;
      bucket[bucket_of(Pt{E.x, E.y})] = e;
    }
  }
};

}  // inline namespace
}  // namespace vh_sh
