// Host-side outlier vote that follows matching in the reference's
// Matcher::matchFeatures (src/matcher.cpp:108): removeOutliers
// (src/remove_outliers.cpp:4-94) on top of the reference's single-precision
// sweep-hull triangulator (src/delaunator.cpp:183-407, legalize :450-549).
//
// SURVEY.md row 8(f-1), host form: what the drop-in Matcher::matchFeatures and the
// stateless vh_remove_outliers* run (one camera: 1.5 ms here, tens of ms as a
// single GPU lane).  The triangulation is one sequential chain of hull updates and
// edge flips per stream (csrc/sweep_hull.h, shared with the device form in
// kernels_vote.hip, which runs many streams and steps side by side instead).
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
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include "../../include/viso_hip.h"
#include "sweep_hull.h"

namespace {

using vh_sh::kNone;
using vh_sh::Pt;

// Working storage of one triangulation.  One instance per host thread (thread_local below) and reused from call
// to call: a fresh std::vector set per call cost ~15 % of the call in allocations and first-touch page faults.
struct Scratch {
  std::vector<Pt> pts;
  std::vector<vh_sh::Node> node;
  std::vector<vh_sh::Half> half;
  std::vector<int32_t> bucket, pending, order, order2, votes;
  std::vector<uint32_t> key;
};

// The part of Delaunator::delaunat before the sweep (delaunator.cpp:192-262): bounding box, visiting order by
// distance from its centre, the three seed points.  The device build does the same in parallel
// (kernels_vote.hip: vote_order); the sweep itself is vh_sh::Sweep, shared.
bool prepare(Scratch &w, int32_t n, int32_t seeds[3], float &span) {
  const std::vector<Pt> &p_ = w.pts;
  float lo_x = std::numeric_limits<float>::infinity(), lo_y = lo_x, hi_x = -lo_x, hi_y = -lo_x;
  for (const Pt &q : p_) {
    lo_x = std::min(q.x, lo_x); lo_y = std::min(q.y, lo_y);
    hi_x = std::max(q.x, hi_x); hi_y = std::max(q.y, hi_y);
  }
  const float bw = hi_x - lo_x, bh = hi_y - lo_y;
  span = bw * bw + bh * bh;
  const Pt mid{(lo_x + hi_x) / 2, (lo_y + hi_y) / 2};
  std::vector<uint32_t> &key = w.key;
  std::vector<int32_t> &order = w.order;
  key.resize(n); order.resize(n);
  int32_t s0 = kNone, s1 = kNone, s2 = kNone;
  float least = std::numeric_limits<float>::infinity();
  bool plain = true;  // every distance a non-negative, non-NaN float: its bit pattern orders like its value
  for (int32_t i = 0; i < n; i++) {
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
  if (plain && n >= 256) {
    // least-significant-digit radix sort on the distance's bit pattern, three 11-bit digits
    std::vector<int32_t> &tmp = w.order2;
    tmp.resize(n);
    int32_t *src = order.data(), *dst = tmp.data();
    for (int32_t pass = 0; pass < 3; pass++) {
      const int32_t shift = 11 * pass;
      uint32_t count[2048] = {0};
      for (int32_t i = 0; i < n; i++) count[(key[i] >> shift) & 2047u]++;
      uint32_t run = 0;
      for (uint32_t &c : count) { const uint32_t k = c; c = run; run += k; }
      for (int32_t i = 0; i < n; i++) { const int32_t v = src[i]; dst[count[(key[v] >> shift) & 2047u]++] = v; }
      std::swap(src, dst);
    }
    if (src != order.data()) std::memcpy(order.data(), src, sizeof(int32_t) * static_cast<size_t>(n));
  } else {
    const auto far_of = [&key](int32_t i) { float f; std::memcpy(&f, &key[i], sizeof(f)); return f; };
    std::stable_sort(order.begin(), order.end(), [&far_of](int32_t a, int32_t b) { return far_of(a) < far_of(b); });
  }
  seeds[0] = seeds[1] = seeds[2] = kNone;
  if (s0 == kNone) return false;

  // :240-262 seed triangle
  least = std::numeric_limits<float>::infinity();
  for (int32_t i = 0; i < n; i++) {
    if (i == s0) continue;
    const float dx = p_[i].x - p_[s0].x, dy = p_[i].y - p_[s0].y, d = dx * dx + dy * dy;
    if (d < least && d > 0.0f) { s1 = i; least = d; }
  }
  if (s1 == kNone) return false;
  least = std::numeric_limits<float>::infinity();
  for (int32_t i = 0; i < n; i++) {
    const float r2 = vh_sh::circum_r2(p_[s0], p_[s1], p_[i]);
    if (r2 < least && i != s0 && i != s1) { s2 = i; least = r2; }
  }
  if (s2 == kNone) return false;
  seeds[0] = s0; seeds[1] = s1; seeds[2] = s2;
  return true;
}

// triangulate w.pts[0, n): the corners of triangle t < return value are w.half[3 t + 0..2].p
int32_t triangulate(Scratch &w, int32_t n) {
  if (n < 3) return 0;
  int32_t seeds[3];
  float span = 0;
  if (!prepare(w, n, seeds, span)) return 0;
  if (w.node.size() < static_cast<size_t>(n)) { w.node.resize(n); w.half.resize(6 * static_cast<size_t>(n)); }
  vh_sh::Sweep<int32_t *> sw{};
  sw.node = w.node.data(); sw.half = w.half.data();
  w.bucket.resize(static_cast<size_t>(vh_sh::hash_size(n)));
  sw.bucket = w.bucket.data();
  if (w.pending.size() < 6 * static_cast<size_t>(n)) w.pending.resize(6 * static_cast<size_t>(n));  // (one entry per flip in flight: never more than there are half-edges)
  sw.pend = w.pending.data(); sw.pend_cap = static_cast<int32_t>(w.pending.size());
  sw.pts = w.pts.data(); sw.order = w.order.data(); sw.n = n;
  if (!sw.seed(seeds[0], seeds[1], seeds[2], span)) return 0;
  sw.insert_all();
#ifdef VH_SH_STATS
  fprintf(stderr, "sweep n=%d tri=%d fix=%lld fix_steps=%lld flips=%lld legal_iters=%lld walk=%lld pop_miss=%lld depth=%d\n", n, sw.ntri,
          (long long)sw.st_fix, (long long)sw.st_fix_steps, (long long)sw.st_flips, (long long)sw.st_legal_iters, (long long)sw.st_walk,
          (long long)sw.st_pop_miss, sw.max_depth);
#endif
  return sw.ntri;
}

}  // namespace

extern "C" int32_t vh_remove_outliers_pm(vh_p_match *pm, int32_t n, int32_t *n_out) {
  if (!n_out || n < 0 || (n > 0 && !pm)) return VH_ERR_INVALID_ARG;
  *n_out = n;
  if (n <= 3) return VH_OK;  // remove_outliers.cpp:6-7
  thread_local Scratch scratch;
  scratch.pts.resize(static_cast<size_t>(n));
  for (int32_t i = 0; i < n; i++) scratch.pts[i] = Pt{pm[i].u1c, pm[i].v1c};
  const int32_t ntri = triangulate(scratch, n);
  const vh_sh::Half *half = scratch.half.data();
  std::vector<int32_t> &votes = scratch.votes;
  votes.assign(static_cast<size_t>(n), 0);
  const float tol = 5;  // hard-coded in the reference (:34), not parameters::outlier_flow_tolerance
  const auto flow_agrees = [pm, tol](int32_t a, int32_t b) {
    const float au = pm[a].u1c - pm[a].u1p, av = pm[a].v1c - pm[a].v1p;
    const float bu = pm[b].u1c - pm[b].u1p, bv = pm[b].v1c - pm[b].v1p;
    return std::fabs(au - bu) + std::fabs(av - bv) < tol ? 1 : 0;
  };
  for (int32_t t = 0; t < ntri; t++) {
    const int32_t a = half[3 * t].p, b = half[3 * t + 1].p, c = half[3 * t + 2].p;
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
