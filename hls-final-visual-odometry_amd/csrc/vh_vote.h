// vh_vote.h -- a batch of match lists on their way through the outlier vote and the bucketing
// (kernels_vote.hip); shared by the kernels and engine.hip.
#ifndef VH_VOTE_H
#define VH_VOTE_H

#include "vh_dev.h"
#include "sweep_hull.h"

struct vh_p_match;

#define VH_VOTE_PEND 30   // flip-stack slots per list beyond the one in a register, in LDS (the reference has 13 in all)
#define VH_VOTE_HASH_MAX 256  // angular-hash slots per list, in LDS: lists of up to 65 536 records

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

struct VhVote {
  int32_t P, cap;          // lists of the batch, records per list
  int32_t hsize;           // hash_size(cap)
  vh_p_match *pm;          // [P][cap] the lists, compacted in place by vote_select
  float2 *pts, *flow;      // [P][cap] (u1c, v1c) / (u1c - u1p, v1c - v1p)
  float2 *spts;            // [P][cap] the points in visiting order (the sweep's numbering)
  int32_t *votes, *order;  // [P][cap]
  vh_sh::Node *node;       // [P][cap]
  vh_sh::Half *half;       // [P][8 cap] half-edge records, four slots per triangle; before the sweep the sort's ping-pong buffers, after the tally the bucketing's scratch
  VhVoteMeta *meta;        // [P]
  int32_t *bgrid;          // [P][3][nb_max + 1] bucketFeatures: per bucket first position / shuffle offset / output offset
  int32_t nb_max;          // buckets the grid of a list may have
};

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

  // nb_max: bound on the bucket grid of Matcher::bucketFeatures (columns x rows) for these lists
  hipError_t alloc(int32_t P, int32_t cap, int32_t out_cap_, int32_t nb_max) {
    release();
    if (nb_max < 1) nb_max = 1;
    if (P < 1) P = 1;
    if (cap < 4) cap = 4;
    if (out_cap_ < 1) out_cap_ = 1;
    const int32_t hs = vh_sh::hash_size(cap);
    const auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t n = (size_t)P * (size_t)cap;
    const size_t sz[13] = {n * 48, n * 8, n * 8, n * 4, n * 4, n * sizeof(vh_sh::Node), 8 * n * sizeof(vh_sh::Half),
                           (size_t)P * sizeof(VhVoteMeta), (size_t)P * (size_t)out_cap_ * 48, (size_t)P * 4, (size_t)cap * 4, n * 8,
                           (size_t)P * 3 * ((size_t)nb_max + 1) * 4};
    size_t off[13], total = 0;
    for (int k = 0; k < 13; k++) { off[k] = total; total += up(sz[k]); }
    const hipError_t e = hipMalloc((void **)&block, total);
    if (e != hipSuccess) { block = nullptr; return e; }
    bytes = total;
    v.P = P; v.cap = cap; v.hsize = hs;
    v.pm = (vh_p_match *)(block + off[0]); v.pts = (float2 *)(block + off[1]); v.flow = (float2 *)(block + off[2]);
    v.votes = (int32_t *)(block + off[3]); v.order = (int32_t *)(block + off[4]); v.node = (vh_sh::Node *)(block + off[5]);
    v.half = (vh_sh::Half *)(block + off[6]);
    v.meta = (VhVoteMeta *)(block + off[7]);
    out = (vh_p_match *)(block + off[8]); out_count = (int32_t *)(block + off[9]); out_cap = out_cap_;
    lfsr = (uint32_t *)(block + off[10]); lfsr_n = cap;
    v.spts = (float2 *)(block + off[11]);
    v.bgrid = (int32_t *)(block + off[12]); v.nb_max = nb_max;
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
