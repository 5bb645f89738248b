// vh_vote.h -- a batch of match lists on their way through the outlier vote and the bucketing
// (kernels_vote.hip); shared by the kernels and engine.hip.
#ifndef VH_VOTE_H
#define VH_VOTE_H

#include "vh_dev.h"
#define VH_SH_LINK16 1  // hull links in 16 bits: device lists hold at most 65 535 records (VH_VOTE_LIST_MAX)
#include "sweep_hull.h"

struct vh_p_match;

#define VH_VOTE_PEND 30   // flip-stack slots per list beyond the one in a register, in LDS (the reference has 13 in all)
#define VH_VOTE_HASH_MAX 256  // angular-hash slots per list, in LDS: lists of up to 65 536 records
#define VH_VOTE_LIST_MAX 65535  // records per list (16-bit hull links, sweep_hull.h)

// status of a list
#define VH_VOTE_OK 0           // to be voted on
#define VH_VOTE_SKIP 1         // n <= 3 (remove_outliers.cpp:6-7) or a stereo list (no flow to vote on): kept as it is
#define VH_VOTE_TRUNCATED 2    // the list did not fit (its slot, the matcher's capacities, or the output): VH_ERR_CAPACITY
#define VH_VOTE_UNSUPPORTED 3  // NaN / infinite / negative coordinates, or a bucket grid beyond the scratch: VH_ERR_UNSUPPORTED
#define VH_VOTE_STACK 4        // more than VH_VOTE_PEND + 1 flips pending: VH_ERR_UNSUPPORTED

struct VhVoteMeta {
  int32_t n, status;
  int32_t seeds[3];
  float span;
  int32_t ntri, kept, depth, out;
  int32_t pad[2];
};

// Per record slot of a list in flight: 48 (record) + 8 (point in visiting order) + 4 + 4 (votes, order) + 16 (hull node)
// + 96 (six half-edge records) = 176 bytes (round 4: 240 -- four half-edge slots per triangle, 32-byte nodes, points and
// flow vectors kept beside the records they come from).
struct VhVote {
  int32_t P, cap;          // lists of the batch, records per list
  int32_t hsize;           // hash_size(cap)
  vh_p_match *pm;          // [P][cap] the lists, compacted in place by vote_select
  float2 *spts;            // [P][cap] the points in visiting order (the sweep's numbering)
  int32_t *votes, *order;  // [P][cap]
  vh_sh::Node *node;       // [P][cap]
  // [P][6 cap] half-edge records, three per triangle.  The region is dead before the sweep and after the tally: until
  // the sweep it holds the sort's ping-pong buffers (2 x cap x 8 bytes) and, behind them, the points (u1c, v1c) in list
  // order (vote_prep -> vote_order); after the tally the bucketing's scratch.
  vh_sh::Half *half;
  VhVoteMeta *meta;        // [P]
  int32_t *bgrid;          // [P][3][nb_max + 1] bucketFeatures: per bucket first position / shuffle offset / output offset
  int32_t nb_max;          // buckets the grid of a list may have
};
__host__ __device__ inline vh_sh::Half *vote_half(const VhVote &vt, int64_t p) { return vt.half + p * 6 * vt.cap; }
__host__ __device__ inline uint2 *vote_sort_buf(const VhVote &vt, int64_t p) { return (uint2 *)vote_half(vt, p); }          // [2][cap]
__host__ __device__ inline float2 *vote_pts(const VhVote &vt, int64_t p) { return (float2 *)vote_half(vt, p) + 2 * vt.cap; }  // [cap]

// the lists of S streams (one step) enter the batch as its lists [p0, p0 + S); vote = 0: no vote (stereo lists)
void vh_launch_vote_prep(const VhVote &vt, int32_t p0, int32_t S, const vh_p_match *src, int64_t src_stride, const int32_t *src_count, int32_t src_cap,
                         const int32_t *src_overflow, int32_t vote, hipStream_t st);
// order -> sweep (`lanes` lists per wave) -> tally -> select (+ bucketing into out[P][out_cap] when max_features >= 1);
// sweep_ev: optional pair of events recorded around the sweep kernel
void vh_launch_vote(const VhVote &vt, int32_t lanes, int32_t max_features, float bw, float bh, const uint32_t *lfsr, int32_t lfsr_n, vh_p_match *out,
                    int32_t out_cap, int32_t *out_count, hipEvent_t *sweep_ev, hipStream_t st);

// Matcher::rand_number (matcher.cpp:113-124): LFSR, taps {32,22,2,1}, evaluated by the reference in double
// arithmetic; its int(floor(number/2^0)) term is an out-of-range double->int conversion for number >= 2^31,
// which on x86-64 (cvttsd2si) produces INT_MIN, i.e. a 0 low bit.
inline uint32_t vh_lfsr_next(uint32_t x) {
  uint32_t b = (x < 0x80000000u) ? (x & 1u) : 0u;
  b ^= (x >> 10) & 1u;
  b ^= (x >> 30) & 1u;
  b ^= (x >> 31) & 1u;
  return (x >> 1) + (b << 31);
}

// Host-side owner of the device memory behind a VhVote (one hipMalloc, carved) and of the bucketed output lists.
struct VhVoteBuffers {
  VhVote v{};
  uint8_t *block = nullptr;
  size_t bytes = 0;
  vh_p_match *out = nullptr;  // [P][out_cap]
  int32_t *out_count = nullptr;
  int32_t out_cap = 0;
  uint32_t *lfsr = nullptr;   // the shuffle's random sequence from seed 5 (matcher.cpp:130,160), lfsr_n draws
  int32_t lfsr_n = 0;

  // device bytes of a batch of P lists of `cap` record slots (what alloc() asks for): 176 bytes per slot, 48 per output
  // record, the LFSR table and the bucket grid (include/viso_hip.h: vh_group_post_device_config states the formula)
  static void layout(int32_t P, int32_t cap, int32_t out_cap_, int32_t nb_max, size_t off[10], size_t *total) {
    const auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t n = (size_t)P * (size_t)cap;
    const size_t sz[10] = {n * 48, n * 8, n * 4, n * 4, n * sizeof(vh_sh::Node), 6 * n * sizeof(vh_sh::Half), (size_t)P * sizeof(VhVoteMeta),
                           (size_t)P * (size_t)out_cap_ * 48, (size_t)P * 4 + (size_t)cap * 4, (size_t)P * 3 * ((size_t)nb_max + 1) * 4};
    size_t t = 0;
    for (int k = 0; k < 10; k++) { off[k] = t; t += up(sz[k]); }
    *total = t;
  }
  static void clamp_args(int32_t &P, int32_t &cap, int32_t &out_cap_, int32_t &nb_max) {
    if (nb_max < 1) nb_max = 1;
    if (P < 1) P = 1;
    if (cap < 4) cap = 4;
    if (out_cap_ < 1) out_cap_ = 1;
  }
  static size_t bytes_for(int32_t P, int32_t cap, int32_t out_cap_, int32_t nb_max) {
    clamp_args(P, cap, out_cap_, nb_max);
    size_t off[10], total;
    layout(P, cap, out_cap_, nb_max, off, &total);
    return total;
  }
  // nb_max: bound on the bucket grid of Matcher::bucketFeatures (columns x rows) for these lists
  hipError_t alloc(int32_t P, int32_t cap, int32_t out_cap_, int32_t nb_max) {
    release();
    clamp_args(P, cap, out_cap_, nb_max);
    static_assert(sizeof(vh_sh::Node) == 16 && sizeof(vh_sh::Half) == 16, "record sizes the footprint formula states");
    size_t off[10], total;
    layout(P, cap, out_cap_, nb_max, off, &total);
    const hipError_t e = hipMalloc((void **)&block, total);
    if (e != hipSuccess) { block = nullptr; return e; }
    bytes = total;
    v.P = P; v.cap = cap; v.hsize = vh_sh::hash_size(cap);
    v.pm = (vh_p_match *)(block + off[0]); v.spts = (float2 *)(block + off[1]);
    v.votes = (int32_t *)(block + off[2]); v.order = (int32_t *)(block + off[3]); v.node = (vh_sh::Node *)(block + off[4]);
    v.half = (vh_sh::Half *)(block + off[5]);
    v.meta = (VhVoteMeta *)(block + off[6]);
    out = (vh_p_match *)(block + off[7]); out_cap = out_cap_;
    out_count = (int32_t *)(block + off[8]); lfsr = (uint32_t *)(block + off[8]) + P; lfsr_n = cap;
    v.bgrid = (int32_t *)(block + off[9]); v.nb_max = nb_max;
    return hipSuccess;
  }
  // the table of draws; the copy is synchronous (pageable source)
  hipError_t upload_lfsr() {
    uint32_t *h = new uint32_t[(size_t)lfsr_n];
    uint32_t r = 5;
    for (int32_t k = 0; k < lfsr_n; k++) { h[k] = r; r = vh_lfsr_next(r); }
    const hipError_t e = hipMemcpy(lfsr, h, sizeof(uint32_t) * (size_t)lfsr_n, hipMemcpyHostToDevice);
    delete[] h;
    return e;
  }
  void release() {
    if (block) (void)hipFree(block);
    block = nullptr; bytes = 0; v = VhVote{}; out = nullptr; out_count = nullptr; lfsr = nullptr;
  }
};

#endif
