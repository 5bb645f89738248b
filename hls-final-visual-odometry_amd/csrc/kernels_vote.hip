// kernels_vote.hip -- the steps between matching and the pose estimate on gfx950, for a batch of
// match lists at once:
//   removeOutliers                     (reference src/remove_outliers.cpp:4-94)
//   delaunator::Delaunator::delaunat   (src/delaunator.cpp:183-407, legalize :450-549)
//   Matcher::bucketFeatures + LFSR     (src/matcher.cpp:113-187)
//
// The triangulation under the vote is ONE sequential chain per list, and its result on integer pixel
// coordinates is a property of that chain (sweep_hull.h).  So it is not parallelised: it runs as it is,
// one list per lane, and the parallelism is the number of lists -- the streams of a group times the
// steps kept in flight (engine.hip: the vote batches).  Around the chain everything is data-parallel:
//
//   vote_prep    one thread per record: the step's lists leave the matcher's buffer (the next
//                emission overwrites it), points and flow vectors are split off
//   vote_order   one wave per list: bounding box, visiting order (a stable LSD radix sort of the
//                squared distances' bit patterns: the reference's insertion sort is stable), the
//                three seed points (first minimum in index order, as the reference's strict `<` scans)
//   vote_sweep   one LANE per list: vh_sh::Sweep -- hull walks and flips, a few dependent loads each;
//                the wave slots it holds are idle most of the time, which is the point: the chain's
//                latency is hidden by other lists and by the matcher's own kernels
//   vote_tally   one thread per triangle: three integer atomics (order-free)
//   vote_select  one wave per list: survivors (>= 4 votes) compacted in order and in place, then
//                bucketFeatures: stable counting sort by bucket, the reference's shuffle per bucket
//                (its LFSR sequence does not depend on the data, so bucket b starts at a known offset
//                of a precomputed table and the buckets are shuffled side by side), first
//                max_features of each bucket in bucket order.
//
// Built with -ffp-contract=off (Makefile): every float product-sum rounds as on the reference's x86 build.
#include "vh_dev.h"
#include "../../include/viso_hip.h"
#define VH_SH_DEVICE 1
#include "vh_vote.h"  // (includes sweep_hull.h in its 16-bit-link form)
#include <cstdlib>

namespace {

using vh_sh::kNone;
using vh_sh::Pt;

__device__ __forceinline__ uint64_t lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, d), hi = __shfl_xor((uint32_t)(v >> 32), d);
    const uint64_t o = ((uint64_t)hi << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int32_t wave_sum_i32(int32_t v) {
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
// inclusive prefix sum over the wave's lanes
__device__ __forceinline__ int32_t wave_scan_i32(int32_t v) {
  const int32_t lane = threadIdx.x & 63;
#pragma unroll
  for (int32_t d = 1; d < 64; d <<= 1) {
    const int32_t o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// Stable least-significant-digit radix sort of n (key, value) pairs by ONE wave, 8 bits per pass.
// a holds the input; a and b are the ping-pong buffers; the result is in the returned buffer.
// hist: LDS [4][256], base: LDS [256].  A pass in which every key has the same digit is skipped.
// Stability inside a 64-pair step: lanes with equal digits are ranked by lane number -- the mask of a
// lane's peers is the AND over the digit's bits of the ballots (or their complements).
__device__ uint2 *wave_radix_sort(uint2 *a, uint2 *b, int32_t n, uint32_t *hist, uint32_t *base) {
  const int32_t lane = threadIdx.x & 63;
  for (int32_t k = lane; k < 1024; k += 64) hist[k] = 0;
  __syncthreads();
  for (int32_t i = lane; i < n; i += 64) {
    const uint32_t key = a[i].x;
    atomicAdd(&hist[key & 255u], 1u);
    atomicAdd(&hist[256 + ((key >> 8) & 255u)], 1u);
    atomicAdd(&hist[512 + ((key >> 16) & 255u)], 1u);
    atomicAdd(&hist[768 + (key >> 24)], 1u);
  }
  __syncthreads();
  uint2 *src = a, *dst = b;
  for (int32_t pass = 0; pass < 4; pass++) {
    const uint32_t *h = hist + 256 * pass;
    const uint32_t c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
    if (__ballot(c0 == (uint32_t)n || c1 == (uint32_t)n || c2 == (uint32_t)n || c3 == (uint32_t)n)) continue;  // one digit holds them all
    const int32_t incl = wave_scan_i32((int32_t)(c0 + c1 + c2 + c3));
    uint32_t run = (uint32_t)incl - (c0 + c1 + c2 + c3);
    base[4 * lane] = run; run += c0;
    base[4 * lane + 1] = run; run += c1;
    base[4 * lane + 2] = run; run += c2;
    base[4 * lane + 3] = run;
    __syncthreads();
    const int32_t shift = 8 * pass;
    for (int32_t i0 = 0; i0 < n; i0 += 64) {
      const int32_t i = i0 + lane;
      const bool live = i < n;
      const uint2 kv = live ? src[i] : make_uint2(0, 0);
      const uint32_t digit = (kv.x >> shift) & 255u;
      uint64_t peers = __ballot(live);
#pragma unroll
      for (int32_t bit = 0; bit < 8; bit++) {
        const uint64_t m = __ballot((digit >> bit) & 1u);
        peers &= ((digit >> bit) & 1u) ? m : ~m;
      }
      const int32_t rank = __popcll(peers & lanes_below());
      const uint32_t at = live ? base[digit] : 0;
      __syncthreads();  // every lane has read its digit's cursor before the first lane of each group advances it
      if (live && rank == 0) base[digit] = at + (uint32_t)__popcll(peers);
      if (live) dst[at + rank] = kv;
      __syncthreads();
    }
    uint2 *t = src; src = dst; dst = t;
  }
  return src;
}

// ---- vote_prep ---------------------------------------------------------------------------------------
// grid (ceil(cap / 256), S): the lists of one step, problem p0 + blockIdx.y of the batch
__global__ __launch_bounds__(256) void vote_prep_kernel(VhVote vt, int32_t p0, const vh_p_match *src, int64_t src_stride, const int32_t *src_count,
                                                         int32_t src_cap, const int32_t *src_overflow, int32_t vote) {
  const int32_t s = blockIdx.y, p = p0 + s;
  const int32_t cnt = src_count[s];
  const int32_t n = min(min(cnt, src_cap), vt.cap);
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    VhVoteMeta &m = vt.meta[p];
    m.n = n;
    // a list that does not fit the batch's slots, or comes from a truncated match list / feature set
    m.status = (cnt > vt.cap || cnt > src_cap || (src_overflow && src_overflow[s])) ? VH_VOTE_TRUNCATED : (vote && n > 3 ? VH_VOTE_OK : VH_VOTE_SKIP);
    m.ntri = 0; m.kept = n; m.depth = 0; m.out = 0;
  }
  if (i >= n) return;
  const int4 *q = (const int4 *)(src + (int64_t)s * src_stride + i);
  const int4 w0 = q[0], w1 = q[1], w2 = q[2];
  int4 *d = (int4 *)(vt.pm + (int64_t)p * vt.cap + i);
  d[0] = w0; d[1] = w1; d[2] = w2;
  // {u1p v1p i1p u2p | v2p i2p u1c v1c | i1c u2c v2c i2c}
  vote_pts(vt, p)[i] = make_float2(__int_as_float(w1.z), __int_as_float(w1.w));  // (u1c, v1c): the point the vote triangulates
  vt.votes[(int64_t)p * vt.cap + i] = 0;
}

// ---- vote_order --------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void vote_order_kernel(VhVote vt) {
  __shared__ uint32_t hist[1024], base[256];
  const int32_t p = blockIdx.x, lane = threadIdx.x;
  VhVoteMeta &m = vt.meta[p];
  if (m.status != VH_VOTE_OK) return;
  const int32_t n = m.n;
  const float2 *pts = vote_pts(vt, p);
  const float inf = __builtin_inff();
  // :192-205 bounding box (std::min / std::max keep their second argument on an unordered comparison)
  float lo_x = inf, lo_y = inf, hi_x = -inf, hi_y = -inf;
  for (int32_t i = lane; i < n; i += 64) {
    const float2 q = pts[i];
    lo_x = q.x < lo_x ? q.x : lo_x; lo_y = q.y < lo_y ? q.y : lo_y;
    hi_x = hi_x < q.x ? q.x : hi_x; hi_y = hi_y < q.y ? q.y : hi_y;
  }
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) {
    const float a = __shfl_xor(lo_x, d), b = __shfl_xor(lo_y, d), c = __shfl_xor(hi_x, d), e = __shfl_xor(hi_y, d);
    lo_x = a < lo_x ? a : lo_x; lo_y = b < lo_y ? b : lo_y;
    hi_x = hi_x < c ? c : hi_x; hi_y = hi_y < e ? e : hi_y;
  }
  const float bw = hi_x - lo_x, bh = hi_y - lo_y;
  const float span = bw * bw + bh * bh;
  const float mid_x = (lo_x + hi_x) / 2, mid_y = (lo_y + hi_y) / 2;
  // distances from the centre: sort keys, and the first point (:207-232)
  uint2 *buf_a = vote_sort_buf(vt, p), *buf_b = buf_a + vt.cap;
  uint64_t best = ~0ull;
  bool plain = true;
  for (int32_t i = lane; i < n; i += 64) {
    const float2 q = pts[i];
    const float dx = q.x - mid_x, dy = q.y - mid_y, far = dx * dx + dy * dy;
    const uint32_t bits = __float_as_uint(far);
    buf_a[i] = make_uint2(bits, (uint32_t)i);
    plain = plain && bits <= 0x7f800000u;  // far >= 0: not NaN, not negative
    if (bits < 0x7f800000u) {              // far < +inf, the scan's initial minimum
      const uint64_t k = ((uint64_t)bits << 32) | (uint32_t)i;
      best = k < best ? k : best;
    }
  }
  best = wave_min_u64(best);
  if (__ballot(!plain) || best == ~0ull) {  // NaN or infinite coordinates: not a list this matcher produces
    if (lane == 0) { m.status = VH_VOTE_UNSUPPORTED; }
    return;
  }
  const int32_t s0 = (int32_t)(uint32_t)best;
  uint2 *sorted = wave_radix_sort(buf_a, buf_b, n, hist, base);
  // the points leave in visiting order: the sweep numbers them by their rank (vt.spts), vt.order maps a rank back to the
  // match it came from (the tally votes by match)
  int32_t *order = vt.order + (int64_t)p * vt.cap;
  float2 *spts = vt.spts + (int64_t)p * vt.cap;
  for (int32_t i = lane; i < n; i += 64) {
    const int32_t src = (int32_t)sorted[i].y;
    order[i] = src;
    spts[i] = pts[src];
  }
  // :240-262 the seed triangle: nearest point to the first, then the smallest circumcircle
  const float2 q0 = pts[s0];
  best = ~0ull;
  for (int32_t i = lane; i < n; i += 64) {
    const float2 q = pts[i];
    const float dx = q.x - q0.x, dy = q.y - q0.y, d = dx * dx + dy * dy;
    const uint32_t bits = __float_as_uint(d);
    if (i != s0 && d > 0.0f && bits < 0x7f800000u) {
      const uint64_t k = ((uint64_t)bits << 32) | (uint32_t)i;
      best = k < best ? k : best;
    }
  }
  best = wave_min_u64(best);
  int32_t s1 = kNone, s2 = kNone;
  if (best != ~0ull) {
    s1 = (int32_t)(uint32_t)best;
    const float2 q1 = pts[s1];
    best = ~0ull;
    for (int32_t i = lane; i < n; i += 64) {
      const float2 q = pts[i];
      const float r2 = vh_sh::circum_r2(Pt{q0.x, q0.y}, Pt{q1.x, q1.y}, Pt{q.x, q.y});
      const uint32_t bits = __float_as_uint(r2);
      if (i != s0 && i != s1 && bits < 0x7f800000u) {  // r2 < +inf (r2 is a sum of squares: never negative)
        const uint64_t k = ((uint64_t)bits << 32) | (uint32_t)i;
        best = k < best ? k : best;
      }
    }
    best = wave_min_u64(best);
    if (best != ~0ull) s2 = (int32_t)(uint32_t)best;
  }
  // the seeds' ranks
  int32_t r0 = kNone, r1 = kNone, r2 = kNone;
  for (int32_t i = lane; i < n; i += 64) {
    const int32_t src = (int32_t)sorted[i].y;
    if (src == s0) r0 = i;
    if (src == s1) r1 = i;
    if (src == s2) r2 = i;
  }
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) { r0 = max(r0, __shfl_xor(r0, d)); r1 = max(r1, __shfl_xor(r1, d)); r2 = max(r2, __shfl_xor(r2, d)); }
  if (lane == 0) { m.seeds[0] = r0; m.seeds[1] = s1 == kNone ? kNone : r1; m.seeds[2] = s2 == kNone ? kNone : r2; m.span = span; }
}

// ---- vote_sweep --------------------------------------------------------------------------------------
// `lanes` lists per wave, one per lane (the other lanes of the wave leave at once); the angular hash and the
// flip stack of a lane's list live in LDS: lanes * (hsize + VH_VOTE_PEND) words of dynamic shared memory
typedef __attribute__((address_space(3))) int32_t *LdsI32;
__global__ __launch_bounds__(64) void vote_sweep_kernel(VhVote vt, int32_t lanes, int32_t prio, int32_t pend_cap) {
  extern __shared__ int32_t sweep_lds[];
  const int32_t lane = threadIdx.x;
  // a wave of this kernel lives for a hundred milliseconds beside thousands of short-lived ones of the matcher: with
  // the default priority it gets one issue slot in eight on a busy SIMD and its latency -- the depth of the whole
  // post-stage pipeline -- grows by as much
  if (prio >= 3) __builtin_amdgcn_s_setprio(3);
  else if (prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (prio == 1) __builtin_amdgcn_s_setprio(1);
  if (lane >= lanes) return;
  const int32_t p = blockIdx.x * lanes + lane;
  if (p >= vt.P) return;
  VhVoteMeta &m = vt.meta[p];
  if (m.status != VH_VOTE_OK) return;
  vh_sh::Sweep<LdsI32> sw;
  sw.node = vt.node + (int64_t)p * vt.cap;
  sw.half = vote_half(vt, p);
  sw.bucket = (LdsI32)(sweep_lds + lane * (vt.hsize + VH_VOTE_PEND));
  sw.pend = sw.bucket + vt.hsize;
  sw.pend_cap = pend_cap;  // <= VH_VOTE_PEND, the slots the launch reserved
  sw.pts = (const Pt *)(vt.spts + (int64_t)p * vt.cap);
  sw.order = nullptr;  // (numbered in visiting order by vote_order)
  sw.n = m.n;
  if (sw.seed(m.seeds[0], m.seeds[1], m.seeds[2], m.span)) sw.insert_all();
  m.ntri = sw.ntri;
  m.depth = sw.max_depth;
  if (sw.overflow) m.status = VH_VOTE_STACK;
}

// ---- vote_tally --------------------------------------------------------------------------------------
// remove_outliers.cpp:36-80: every triangle votes for each of its corners once per agreeing neighbour
__device__ __forceinline__ void tally_triangle(const VhVote &vt, int32_t p, int32_t t, int32_t &a, int32_t &b, int32_t &c, int32_t &va, int32_t &vb, int32_t &vc) {
  const vh_sh::Half *T = vote_half(vt, p) + (int64_t)t * 3;
  const int32_t *order = vt.order + (int64_t)p * vt.cap;
  a = order[T[0].p]; b = order[T[1].p]; c = order[T[2].p];  // corner ranks -> matches
  // flow vectors (u1c - u1p, v1c - v1p), remove_outliers.cpp:40-47, from the records themselves
  const vh_p_match *pm = vt.pm + (int64_t)p * vt.cap;
  const auto flow_of = [pm](int32_t i) {
    const float2 prev = *(const float2 *)&pm[i].u1p, cur = *(const float2 *)&pm[i].u1c;
    return make_float2(cur.x - prev.x, cur.y - prev.y);
  };
  const float2 fa = flow_of(a), fb = flow_of(b), fc = flow_of(c);
  const float tol = 5;  // hard-coded in the reference (:34), not parameters::outlier_flow_tolerance
  const int32_t ab = fabsf(fa.x - fb.x) + fabsf(fa.y - fb.y) < tol ? 1 : 0;
  const int32_t bc = fabsf(fb.x - fc.x) + fabsf(fb.y - fc.y) < tol ? 1 : 0;
  const int32_t ac = fabsf(fa.x - fc.x) + fabsf(fa.y - fc.y) < tol ? 1 : 0;
  va = ab + ac; vb = ab + bc; vc = bc + ac;
}

// lists of up to 16 384 records: one workgroup per list, the counters in LDS (54 k votes per KITTI list: as global
// atomics they were the whole kernel), written out once
__global__ __launch_bounds__(1024) void vote_tally_kernel(VhVote vt) {
  extern __shared__ int32_t sv[];
  const int32_t p = blockIdx.x;
  const VhVoteMeta &m = vt.meta[p];
  if (m.status != VH_VOTE_OK) return;
  const int32_t n = m.n, ntri = m.ntri;
  for (int32_t i = threadIdx.x; i < n; i += 1024) sv[i] = 0;
  __syncthreads();
  for (int32_t t = threadIdx.x; t < ntri; t += 1024) {
    int32_t a, b, c, va, vb, vc;
    tally_triangle(vt, p, t, a, b, c, va, vb, vc);
    if (va) atomicAdd(&sv[a], va);
    if (vb) atomicAdd(&sv[b], vb);
    if (vc) atomicAdd(&sv[c], vc);
  }
  __syncthreads();
  int32_t *votes = vt.votes + (int64_t)p * vt.cap;
  for (int32_t i = threadIdx.x; i < n; i += 1024) votes[i] = sv[i];
}

// longer lists: a thread per triangle, global counters (zeroed by vote_prep)
__global__ __launch_bounds__(256) void vote_tally_global_kernel(VhVote vt) {
  const int32_t p = blockIdx.y;
  const VhVoteMeta &m = vt.meta[p];
  if (m.status != VH_VOTE_OK) return;
  const int32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= m.ntri) return;
  int32_t a, b, c, va, vb, vc;
  tally_triangle(vt, p, t, a, b, c, va, vb, vc);
  int32_t *votes = vt.votes + (int64_t)p * vt.cap;
  if (va) atomicAdd(&votes[a], va);
  if (vb) atomicAdd(&votes[b], vb);
  if (vc) atomicAdd(&votes[c], vc);
}

// ---- vote_select -------------------------------------------------------------------------------------
// One wave per list.  max_features < 1: the vote only (the survivors stay in vt.pm, their number in meta.kept).
__global__ __launch_bounds__(64) void vote_select_kernel(VhVote vt, int32_t max_features, float bw, float bh, const uint32_t *lfsr, int32_t lfsr_n,
                                                          vh_p_match *out, int32_t out_cap, int32_t *out_count) {
  __shared__ uint32_t hist[1024], base[256];
  const int32_t p = blockIdx.x, lane = threadIdx.x;
  VhVoteMeta &m = vt.meta[p];
  vh_p_match *pm = vt.pm + (int64_t)p * vt.cap;
  const int32_t n = m.n;
  int32_t kept = n;
  if (m.status == VH_VOTE_OK) {  // remove_outliers.cpp:82-91, in place: a record moves to a position not behind its own
    const int32_t *votes = vt.votes + (int64_t)p * vt.cap;
    kept = 0;
    for (int32_t i0 = 0; i0 < n; i0 += 64) {
      const int32_t i = i0 + lane;
      const bool keep = i < n && votes[i] >= 4;
      int4 w0 = make_int4(0, 0, 0, 0), w1 = w0, w2 = w0;
      if (keep) { const int4 *q = (const int4 *)(pm + i); w0 = q[0]; w1 = q[1]; w2 = q[2]; }
      const uint64_t mask = __ballot(keep);
      __syncthreads();  // (the step's loads have landed before its stores go out)
      if (keep) { int4 *d = (int4 *)(pm + kept + __popcll(mask & lanes_below())); d[0] = w0; d[1] = w1; d[2] = w2; }
      kept += __popcll(mask);
      __syncthreads();
    }
  }
  if (lane == 0) m.kept = kept;
  if (max_features < 1 || m.status == VH_VOTE_TRUNCATED || m.status == VH_VOTE_UNSUPPORTED || m.status == VH_VOTE_STACK) {
    if (lane == 0 && out_count) out_count[p] = 0;
    return;
  }
  __syncthreads();
  // ---- Matcher::bucketFeatures (matcher.cpp:140-187) on pm[0, kept) ----
  float u_max = 0, v_max = 0;
  for (int32_t i = lane; i < kept; i += 64) {
    const float u = pm[i].u1c, v = pm[i].v1c;
    u_max = u > u_max ? u : u_max; v_max = v > v_max ? v : v_max;
  }
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) {
    const float a = __shfl_xor(u_max, d), b = __shfl_xor(v_max, d);
    u_max = a > u_max ? a : u_max; v_max = b > v_max ? b : v_max;
  }
  const int32_t cols = (int32_t)floorf(u_max / bw) + 1, rows = (int32_t)floorf(v_max / bh) + 1;
  const int64_t nb64 = (int64_t)cols * rows;
  // the sort's ping-pong buffers in the half-edge storage (the tally is over); per bucket {start, shuffle offset, out offset}
  uint2 *buf_a = vote_sort_buf(vt, p), *buf_b = buf_a + vt.cap;
  int32_t *bstart = vt.bgrid + (int64_t)p * 3 * (vt.nb_max + 1);  // [nb + 1]
  if (cols < 1 || rows < 1 || nb64 > vt.nb_max) {
    if (lane == 0) { m.status = VH_VOTE_UNSUPPORTED; if (out_count) out_count[p] = 0; }
    return;
  }
  const int32_t nb = (int32_t)nb64;
  int32_t *boff = bstart + nb + 1, *bout = boff + nb + 1;
  for (int32_t b = lane; b <= nb; b += 64) bstart[b] = 0;
  __syncthreads();
  bool inside = true;
  for (int32_t i = lane; i < kept; i += 64) {
    const int32_t bu = (int32_t)floorf(pm[i].u1c / bw), bv = (int32_t)floorf(pm[i].v1c / bh);
    const bool ok = bu >= 0 && bu < cols && bv >= 0 && bv < rows;  // (negative or NaN coordinates: the reference indexes out of bounds)
    inside = inside && ok;
    const int32_t b = ok ? bv * cols + bu : 0;
    buf_a[i] = make_uint2((uint32_t)b, (uint32_t)i);
    atomicAdd(&bstart[b + 1], 1);  // (count of bucket b, shifted by one for the scan below)
  }
  if (__ballot(!inside)) {
    if (lane == 0) { m.status = VH_VOTE_UNSUPPORTED; if (out_count) out_count[p] = 0; }
    return;
  }
  __syncthreads();
  // the records of a bucket in list order: stable sort by bucket
  uint2 *sorted = wave_radix_sort(buf_a, buf_b, kept, hist, base);
  // per bucket: first position, offset into the shuffle's random sequence, first output position
  int32_t run_start = 0, run_off = 0, run_out = 0;
  for (int32_t b0 = 0; b0 < nb; b0 += 64) {
    const int32_t b = b0 + lane;
    const int32_t len = b < nb ? bstart[b + 1] : 0;
    const int32_t steps = len > 1 ? len - 1 : 0, take = len < max_features ? len : max_features;
    const int32_t i_len = wave_scan_i32(len), i_steps = wave_scan_i32(steps), i_take = wave_scan_i32(take);
    __syncthreads();
    if (b < nb) { bstart[b] = run_start + i_len - len; boff[b] = run_off + i_steps - steps; bout[b] = run_out + i_take - take; }
    run_start += __shfl(i_len, 63); run_off += __shfl(i_steps, 63); run_out += __shfl(i_take, 63);
  }
  __syncthreads();
  if (lane == 0) { bstart[nb] = run_start; m.out = run_out; if (out_count) out_count[p] = run_out; }
  if (run_out > out_cap || run_off > lfsr_n) {
    if (lane == 0) m.status = VH_VOTE_TRUNCATED;
    return;
  }
  // random_shuffle (matcher.cpp:126-138) of every bucket, one lane per bucket: the k-th draw of the whole call is lfsr[k]
  for (int32_t b0 = 0; b0 < nb; b0 += 64) {
    const int32_t b = b0 + lane;
    if (b >= nb) continue;
    const int32_t first = bstart[b], len = (b + 1 < nb ? bstart[b + 1] : run_start) - first;
    uint2 *v = sorted + first;
    const uint32_t *rnd = lfsr + boff[b];
    for (int32_t i = 1; i < len; i++) {
      const int32_t j = (int32_t)(rnd[i - 1] % (uint32_t)(i + 1));
      const uint32_t vi = v[i].y, vj = v[j].y;
      v[i].y = vj; v[j].y = vi;
    }
    const int32_t take = len < max_features ? len : max_features;
    vh_p_match *o = out + (int64_t)p * out_cap + bout[b];
    for (int32_t j = 0; j < take; j++) {
      const int4 *q = (const int4 *)(pm + v[j].y);
      int4 *d = (int4 *)(o + j);
      d[0] = q[0]; d[1] = q[1]; d[2] = q[2];
    }
  }
}

}  // namespace

// test hook (include/viso_hip.h): flip-stack slots the sweep uses, so that the refusal path can be driven with ordinary lists
static int32_t g_vote_pend_cap = VH_VOTE_PEND;
extern "C" int32_t vh_debug_vote_stack_slots(int32_t slots) {
  if (slots < 0 || slots > VH_VOTE_PEND + 1) return VH_ERR_INVALID_ARG;
  g_vote_pend_cap = slots == 0 ? VH_VOTE_PEND : slots - 1;  // one more entry lives in a register
  return VH_OK;
}

void vh_launch_vote_prep(const VhVote &vt, int32_t p0, int32_t S, const vh_p_match *src, int64_t src_stride, const int32_t *src_count, int32_t src_cap,
                         const int32_t *src_overflow, int32_t vote, hipStream_t st) {
  if (S < 1) return;
  const int32_t width = vt.cap < src_cap ? vt.cap : src_cap;
  hipLaunchKernelGGL(vote_prep_kernel, dim3((width + 255) / 256, S), dim3(256), 0, st, vt, p0, src, src_stride, src_count, src_cap, src_overflow, vote);
}

void vh_launch_vote(const VhVote &vt, int32_t lanes, int32_t max_features, float bw, float bh, const uint32_t *lfsr, int32_t lfsr_n, vh_p_match *out,
                    int32_t out_cap, int32_t *out_count, hipEvent_t *sweep_ev, hipStream_t st) {
  if (vt.P < 1) return;
  lanes = lanes < 1 ? 1 : (lanes > 64 ? 64 : lanes);
  {  // the lanes' hashes and flip stacks share 64 KB of LDS: fewer lists per wave for very long lists
    const int32_t fit = (int32_t)(65536 / (sizeof(int32_t) * (size_t)(vt.hsize + VH_VOTE_PEND)));
    if (lanes > fit) lanes = fit < 1 ? 1 : fit;
  }
  hipLaunchKernelGGL(vote_order_kernel, dim3(vt.P), dim3(64), 0, st, vt);
  if (sweep_ev) (void)hipEventRecord(sweep_ev[0], st);
  static const int32_t prio = [] { const char *e = getenv("VH_VOTE_PRIO"); return e ? atoi(e) : 3; }();
  hipLaunchKernelGGL(vote_sweep_kernel, dim3((vt.P + lanes - 1) / lanes), dim3(64), sizeof(int32_t) * (size_t)lanes * (vt.hsize + VH_VOTE_PEND), st, vt, lanes, prio, g_vote_pend_cap);
  if (sweep_ev) (void)hipEventRecord(sweep_ev[1], st);
  if (vt.cap <= 16384) hipLaunchKernelGGL(vote_tally_kernel, dim3(vt.P), dim3(1024), sizeof(int32_t) * (size_t)vt.cap, st, vt);
  else hipLaunchKernelGGL(vote_tally_global_kernel, dim3((2 * vt.cap + 255) / 256, vt.P), dim3(256), 0, st, vt);
  hipLaunchKernelGGL(vote_select_kernel, dim3(vt.P), dim3(64), 0, st, vt, max_features, bw, bh, lfsr, lfsr_n, out, out_cap, out_count);
}
