// Host-side outlier vote that follows matching in the reference's
// Matcher::matchFeatures (src/matcher.cpp:108): removeOutliers
// (src/remove_outliers.cpp:4-94) on top of the reference's single-precision
// sweep-hull triangulator (src/delaunator.cpp:183-407, legalize :450-549).
//
// SURVEY.md row 8(f-1).  This stays on the host by design: the triangulation is
// one sequential chain of hull updates and edge flips per stream, and on integer
// pixel coordinates (co-circular quadruples everywhere) its outcome depends on
// the visiting order and on every float rounding step, so it cannot be
// re-associated into a data-parallel kernel and stay bit-identical.  Streams are
// independent, so callers with many streams run it from one host thread each.
//
// Numerics contract (tests/test_outliers.py checks it against vectors produced by
// the reference's own build, tests/golden/outliers.npz): data_type is float;
// the `0.5`, `3.0`, `4.0`, `1e14`, `1e-20` literals of the reference promote the
// enclosing sub-expression to double.  Build with -ffp-contract=off.
//
// Where the reference is undefined this file is defined instead: storage grows
// with n (reference: POINT_L/TRIANGLE_L/HASH_L arrays, delaunator.hpp:10-13),
// the flip stack grows (reference: 13 slots), and an input without a seed
// triangle produces no triangles, i.e. every match loses the vote.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include "../../include/viso_hip.h"

namespace {

constexpr int32_t kNone = std::numeric_limits<int32_t>::max();  // delaunator.cpp:13

struct Pt { float x, y; };

// orientation test of delaunator.cpp:99-121 (the `clockwise` twin is unused there)
inline bool turns_ccw(Pt p, Pt q, Pt r) {
  const float ux = q.x - p.x, uy = q.y - p.y, vx = r.x - p.x, vy = r.y - p.y;
  const float det = ux * vy - uy * vx;
  if (det == 0) return false;
  const float size = (ux * ux + uy * uy) + (vx * vx + vy * vy);
  // |size / det| > 1e14 is impossible when |det| >= 1 and size <= 9e13 (the quotient cannot exceed size):
  // on pixel coordinates det is a non-zero integer, so the division is only done for exotic inputs
  if (!(std::fabs(det) >= 1.0f && size <= 9e13f) && static_cast<double>(std::fabs(size / det)) > 1e14) return false;
  return det > 0;
}

// offset of the circumcentre of (a,b,c) from a; false for a degenerate triple
// (delaunator.cpp:23-39 and :124-146 share this arithmetic)
inline bool circum_offset(Pt a, Pt b, Pt c, double &ox, double &oy) {
  const float dx = b.x - a.x, dy = b.y - a.y, ex = c.x - a.x, ey = c.y - a.y;
  const float bl = dx * dx + dy * dy, cl = ex * ex + ey * ey;
  const float det = dx * ey - dy * ex;
  ox = static_cast<double>(ey * bl - dy * cl) * 0.5 / static_cast<double>(det);
  oy = static_cast<double>(dx * cl - ex * bl) * 0.5 / static_cast<double>(det);
  return (bl > 0 || bl < 0) && (cl > 0 || cl < 0) && (det > 0 || det < 0);
}

// Working storage of one triangulation.  One instance per host thread (thread_local below) and reused from call
// to call: a fresh std::vector set per call cost ~15 % of the call in allocations and first-touch page faults.
struct Scratch {
  std::vector<Pt> pts;
  std::vector<int32_t> tri, twin, prev, next, edge_of, bucket, pending, order, order2, votes;
  std::vector<uint32_t> key;
};

class SweepHull {
 public:
  SweepHull(Scratch &w, int32_t n) : w_(w), p_(w.pts), n_(n), tri_(w.tri), twin_(w.twin), prev_(w.prev), next_(w.next),
                                     edge_of_(w.edge_of), bucket_(w.bucket), pending_(w.pending) {}

  // triangle corners, three per triangle, in the reference's order: tri()[0 .. size())
  void run() {
    ntri_ = 0; ntwin_ = 0;
    if (n_ >= 3) sweep();
  }
  const int32_t *tri() const { return tri_.data(); }
  size_t size() const { return static_cast<size_t>(ntri_); }

 private:
  Scratch &w_;
  const std::vector<Pt> &p_;
  const int32_t n_;
  // tri_/twin_ are used as arrays of 6 n slots (a triangulation of n points has fewer than 2 n triangles) with their
  // own lengths ntri_/ntwin_ -- the reference's triangles_cnt / halfedges_cnt
  std::vector<int32_t> &tri_, &twin_, &prev_, &next_, &edge_of_, &bucket_, &pending_;
  int32_t ntri_ = 0, ntwin_ = 0;
  int32_t hull_entry_ = 0, buckets_ = 0;
  Pt origin_{0, 0};

  // delaunator.cpp:178-182 + :551-557
  int32_t bucket_of(Pt q) const {
    const float dx = q.x - origin_.x, dy = q.y - origin_.y;
    const float t = dx / (std::fabs(dx) + std::fabs(dy));
    const float turn = static_cast<float>((dy > 0.0f ? 3.0 - static_cast<double>(t) : 1.0 + static_cast<double>(t)) / 4.0);
    const float scaled = std::floor(turn * static_cast<float>(buckets_));
    if (std::isnan(scaled)) return 0;  // q == origin_: the reference indexes out of bounds here
    const int32_t k = static_cast<int32_t>(scaled);
    return k >= buckets_ ? k % buckets_ : k;
  }

  // delaunator.cpp:585-603
  void pair_up(int32_t a, int32_t b) {
    const auto set = [this](int32_t at, int32_t to) {
      if (at == ntwin_) twin_[ntwin_++] = to;
      else if (at < ntwin_) twin_[at] = to;
    };
    set(a, b);
    if (b != kNone) set(b, a);
  }

  // delaunator.cpp:566-583
  int32_t emit(int32_t i0, int32_t i1, int32_t i2, int32_t a, int32_t b, int32_t c) {
    const int32_t t = ntri_;
    tri_[t] = i0; tri_[t + 1] = i1; tri_[t + 2] = i2;
    ntri_ += 3;
    pair_up(t, a); pair_up(t + 1, b); pair_up(t + 2, c);
    return t;
  }

  // delaunator.cpp:149-175
  bool inside_circumcircle(int32_t a, int32_t b, int32_t c, int32_t q) const {
    const Pt A = p_[a], B = p_[b], C = p_[c], Q = p_[q];
    const float dx = A.x - Q.x, dy = A.y - Q.y, ex = B.x - Q.x, ey = B.y - Q.y, fx = C.x - Q.x, fy = C.y - Q.y;
    const float ap = dx * dx + dy * dy, bp = ex * ex + ey * ey, cp = fx * fx + fy * fy;
    const float s1 = dx * (ey * cp - bp * fy), s2 = dy * (ex * cp - bp * fx), s3 = ap * (ex * fy - ey * fx);
    return (s1 - s2) + s3 < 0.0f;
  }

  // delaunator.cpp:450-549
  int32_t legalize(int32_t a) {
    pending_.clear();
    int32_t ar = 0;
    while (true) {
      const int32_t b = twin_[a];
      const int32_t a0 = a - a % 3;
      ar = a0 + (a + 2) % 3;
      bool flipped = false;
      if (b != kNone) {
        const int32_t b0 = b - b % 3, al = a0 + (a + 1) % 3, bl = b0 + (b + 2) % 3;
        const int32_t p0 = tri_[ar], pr = tri_[a], pl = tri_[al], p1 = tri_[bl];
        if (inside_circumcircle(p0, pr, pl, p1)) {
          tri_[a] = p1;
          tri_[b] = p0;
          const int32_t outer = twin_[bl];
          if (outer == kNone) {  // the edge that moved was on the hull
            int32_t e = hull_entry_;
            do {
              if (edge_of_[e] == bl) { edge_of_[e] = a; break; }
              e = prev_[e];
            } while (e != hull_entry_);
          }
          pair_up(a, outer);
          pair_up(b, twin_[ar]);
          pair_up(ar, bl);
          pending_.push_back(b0 + (b + 1) % 3);
          flipped = true;  // and look at edge a again
        }
      }
      if (!flipped) {
        if (pending_.empty()) break;
        a = pending_.back();
        pending_.pop_back();
      }
    }
    return ar;
  }

  void sweep() {
    // :192-232 bounding box, visiting order by distance from its centre
    float lo_x = std::numeric_limits<float>::infinity(), lo_y = lo_x, hi_x = -lo_x, hi_y = -lo_x;
    for (const Pt &q : p_) {
      lo_x = std::min(q.x, lo_x); lo_y = std::min(q.y, lo_y);
      hi_x = std::max(q.x, hi_x); hi_y = std::max(q.y, hi_y);
    }
    const float w = hi_x - lo_x, h = hi_y - lo_y, span = w * w + h * h;
    const Pt mid{(lo_x + hi_x) / 2, (lo_y + hi_y) / 2};
    std::vector<uint32_t> &key = w_.key;
    std::vector<int32_t> &order = w_.order;
    key.resize(n_); order.resize(n_);
    int32_t s0 = kNone, s1 = kNone, s2 = kNone;
    float least = std::numeric_limits<float>::infinity();
    bool plain = true;  // every distance a non-negative, non-NaN float: its bit pattern orders like its value
    for (int32_t i = 0; i < n_; i++) {
      const float dx = p_[i].x - mid.x, dy = p_[i].y - mid.y, far = dx * dx + dy * dy;
      if (far < least) { s0 = i; least = far; }
      uint32_t bits;
      static_assert(sizeof(bits) == sizeof(far), "float is 32 bits");
      std::memcpy(&bits, &far, sizeof(bits));
      key[i] = bits;
      plain = plain && far >= 0.0f;  // (false for NaN)
      order[i] = i;
    }
    // the reference's insertion sort (:409-424) is stable; so are both of these
    if (plain && n_ >= 256) {
      // least-significant-digit radix sort on the distance's bit pattern, three 11-bit digits
      std::vector<int32_t> &tmp = w_.order2;
      tmp.resize(n_);
      int32_t *src = order.data(), *dst = tmp.data();
      for (int32_t pass = 0; pass < 3; pass++) {
        const int32_t shift = 11 * pass;
        uint32_t count[2048] = {0};
        for (int32_t i = 0; i < n_; i++) count[(key[i] >> shift) & 2047u]++;
        uint32_t run = 0;
        for (uint32_t &c : count) { const uint32_t k = c; c = run; run += k; }
        for (int32_t i = 0; i < n_; i++) { const int32_t v = src[i]; dst[count[(key[v] >> shift) & 2047u]++] = v; }
        std::swap(src, dst);
      }
      if (src != order.data()) std::memcpy(order.data(), src, sizeof(int32_t) * static_cast<size_t>(n_));
    } else {
      const auto far_of = [&key](int32_t i) { float f; std::memcpy(&f, &key[i], sizeof(f)); return f; };
      std::stable_sort(order.begin(), order.end(), [&far_of](int32_t a, int32_t b) { return far_of(a) < far_of(b); });
    }
    if (s0 == kNone) return;

    // :240-262 seed triangle
    least = std::numeric_limits<float>::infinity();
    for (int32_t i = 0; i < n_; i++) {
      if (i == s0) continue;
      const float dx = p_[i].x - p_[s0].x, dy = p_[i].y - p_[s0].y, d = dx * dx + dy * dy;
      if (d < least && d > 0.0f) { s1 = i; least = d; }
    }
    if (s1 == kNone) return;
    least = std::numeric_limits<float>::infinity();
    for (int32_t i = 0; i < n_; i++) {
      double ox, oy;
      float r2 = std::numeric_limits<float>::infinity();
      if (circum_offset(p_[s0], p_[s1], p_[i], ox, oy)) {
        const float rx = static_cast<float>(ox), ry = static_cast<float>(oy);
        r2 = rx * rx + ry * ry;
      }
      if (r2 < least && i != s0 && i != s1) { s2 = i; least = r2; }
    }
    if (s2 == kNone) return;
    if (turns_ccw(p_[s0], p_[s1], p_[s2])) std::swap(s1, s2);
    {
      double ox, oy;
      circum_offset(p_[s0], p_[s1], p_[s2], ox, oy);
      origin_ = Pt{static_cast<float>(static_cast<double>(p_[s0].x) + ox), static_cast<float>(static_cast<double>(p_[s0].y) + oy)};
    }

    buckets_ = static_cast<int32_t>(std::ceil(std::sqrt(static_cast<double>(n_))));
    bucket_.assign(buckets_, kNone);
    prev_.assign(n_, 0); next_.assign(n_, 0); edge_of_.assign(n_, 0);
    if (tri_.size() < 6 * static_cast<size_t>(n_)) { tri_.resize(6 * static_cast<size_t>(n_)); twin_.resize(6 * static_cast<size_t>(n_)); }
    hull_entry_ = s0;
    next_[s0] = prev_[s2] = s1;
    next_[s1] = prev_[s0] = s2;
    next_[s2] = prev_[s1] = s0;
    edge_of_[s0] = 0; edge_of_[s1] = 1; edge_of_[s2] = 2;
    bucket_[bucket_of(p_[s0])] = s0;
    bucket_[bucket_of(p_[s1])] = s1;
    bucket_[bucket_of(p_[s2])] = s2;
    emit(s0, s1, s2, kNone, kNone, kNone);

    // Point::equal, delaunator.hpp:64-71: (d2 / span) < 1e-20 in double.  d2 > span * 1e-18 (a normal float) puts
    // the quotient far above 1e-20 without dividing; anything else takes the literal form.
    const float span_eps = span * 1e-18f;
    const bool span_eps_ok = span_eps >= std::numeric_limits<float>::min() && std::isfinite(span_eps);
    const auto coincides = [span, span_eps, span_eps_ok](Pt a, Pt b) {
      const float dx = b.x - a.x, dy = b.y - a.y, d2 = dx * dx + dy * dy;
      if (span_eps_ok && d2 > span_eps) return false;
      return static_cast<double>(d2 / span) < 1e-20;
    };

    // :303-404; the three seeds are offered to the hull like every other point
    for (int32_t k = 0; k < n_; k++) {
      const int32_t i = order[k];
      const Pt q = p_[i];
      int32_t at = kNone;
      const int32_t first = bucket_of(q);
      for (int32_t j = 0; j < buckets_; j++) {
        const int32_t slot = first + j;
        at = bucket_[slot >= buckets_ ? slot % buckets_ : slot];
        if (at != kNone && at != next_[at]) break;
      }
      if (at == kNone) continue;
      const int32_t begin = prev_[at];
      int32_t e = begin;
      while (true) {  // first hull edge e -> next_[e] facing q
        const int32_t f = next_[e];
        if (coincides(q, p_[e]) || coincides(q, p_[f])) { e = kNone; break; }
        if (turns_ccw(q, p_[e], p_[f])) break;
        e = f;
        if (e == begin) { e = kNone; break; }
      }
      if (e == kNone) continue;  // duplicate, or nothing visible: the point is left out

      int32_t t = emit(e, i, next_[e], kNone, kNone, edge_of_[e]);
      edge_of_[i] = legalize(t + 2);
      edge_of_[e] = t;

      int32_t fwd = next_[e];
      while (true) {
        const int32_t f = next_[fwd];
        if (!turns_ccw(q, p_[fwd], p_[f])) break;
        t = emit(fwd, i, f, edge_of_[i], kNone, edge_of_[fwd]);
        edge_of_[i] = legalize(t + 2);
        next_[fwd] = fwd;  // off the hull
        fwd = f;
      }
      if (e == begin) {
        while (true) {
          const int32_t b = prev_[e];
          if (!turns_ccw(q, p_[b], p_[e])) break;
          t = emit(b, i, e, kNone, edge_of_[e], edge_of_[b]);
          legalize(t + 2);
          edge_of_[b] = t;
          next_[e] = e;
          e = b;
        }
      }
      prev_[i] = e;
      hull_entry_ = e;
      prev_[fwd] = i;
      next_[e] = i;
      next_[i] = fwd;
      bucket_[bucket_of(q)] = i;
      bucket_[bucket_of(p_[e])] = e;
    }
  }
};

}  // namespace

extern "C" int32_t vh_remove_outliers_pm(vh_p_match *pm, int32_t n, int32_t *n_out) {
  if (!n_out || n < 0 || (n > 0 && !pm)) return VH_ERR_INVALID_ARG;
  *n_out = n;
  if (n <= 3) return VH_OK;  // remove_outliers.cpp:6-7
  thread_local Scratch scratch;
  scratch.pts.resize(static_cast<size_t>(n));
  for (int32_t i = 0; i < n; i++) scratch.pts[i] = Pt{pm[i].u1c, pm[i].v1c};
  SweepHull hull(scratch, n);
  hull.run();
  const int32_t *tri = hull.tri();
  std::vector<int32_t> &votes = scratch.votes;
  votes.assign(static_cast<size_t>(n), 0);
  const float tol = 5;  // hard-coded in the reference (:34), not parameters::outlier_flow_tolerance
  const auto flow_agrees = [pm, tol](int32_t a, int32_t b) {
    const float au = pm[a].u1c - pm[a].u1p, av = pm[a].v1c - pm[a].v1p;
    const float bu = pm[b].u1c - pm[b].u1p, bv = pm[b].v1c - pm[b].v1p;
    return std::fabs(au - bu) + std::fabs(av - bv) < tol ? 1 : 0;
  };
  for (size_t t = 0; t + 2 < hull.size(); t += 3) {
    const int32_t a = tri[t], b = tri[t + 1], c = tri[t + 2];
    const int32_t ab = flow_agrees(a, b), bc = flow_agrees(b, c), ac = flow_agrees(a, c);
    votes[a] += ab + ac;
    votes[b] += ab + bc;
    votes[c] += bc + ac;
  }
  int32_t kept = 0;
  for (int32_t i = 0; i < n; i++)
    if (votes[i] >= 4) pm[kept++] = pm[i];
  *n_out = kept;
  return VH_OK;
}
