// engine.hip -- host side of libviso_hip.so: device memory, launch sequencing
// and the extern "C" ABI declared in include/viso_hip.h.
//
// A vh_group owns S independent camera streams that are stepped together; a
// vh_matcher is a group of one.  The reference's Matcher state (ring buffer of
// two feature-set pairs, src/matcher.h:245-259) lives in HBM and rotates by
// moving the current/previous roles between the slots of a ring; nothing is copied on pushBack.
#include "vh_dev.h"
#include "../../include/viso_hip.h"
#include "vh_vote.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// Ring slots per stream.  Three are the minimum for detecting frame t+1 while frame t is matched
// against t-1 -- but then the detection of t+2 overwrites the slot of t-1 and has to wait for the
// emission of match t, and the search of t+2 for that detection: both chains idle ~6 % of a step
// (rocprofv3 timeline, KITTI, S = 256).  With four, detection runs a whole frame ahead and neither
// stream waits for the other.  VH_RING=3 rebuilds the old ring.
#ifndef VH_RING
#define VH_RING 4
#endif
static_assert(VH_RING >= 3 && VH_RING <= 8, "ring slots");

thread_local std::string t_last_error;

#define VH_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      t_last_error = std::string(#call) + ": " + hipGetErrorString(e_);                       \
      return VH_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

int32_t round_up(int32_t x, int32_t m) { return (x + m - 1) / m * m; }

struct ProfEntry {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double ms = 0;
  int64_t launches = 0;
};

uint32_t lfsr_next(uint32_t x);
// Matcher::bucketFeatures (matcher.cpp:140-187) on the records pm[0, n): the selected records are written to
// out (at most out_cap of them) in the reference's order; returns how many the reference would keep.
int32_t bucket_records(const vh_p_match *pm, int32_t n, int32_t max_features, float bw, float bh, vh_p_match *out, int32_t out_cap,
                       std::vector<int32_t> &work);

struct Group {
  vh_params p{};
  int32_t device = 0, S = 1;
  int32_t req_features = 0, req_matches = 0;
  // Internal streams: detection+indexing of frame t+1 overlaps the matching
  // of frame t (the ring has VH_RING slots for that).  `stream` is the detect
  // stream (also used by the stateless paths); a caller-owned stream, if set,
  // only orders our work after the caller's (image producers).
  // A third stream (default; VH_POST_STREAM=0: the match stream) takes the short,
  // latency-bound post-processing (chain + emission) of frame t, so that the flow
  // search of frame t+1 follows that of frame t back to back; the match tables are
  // double-buffered for that.
  hipStream_t own_stream = nullptr, stream = nullptr, match_stream = nullptr, post_stream = nullptr, user_stream = nullptr;
  hipEvent_t ev_tables[2] = {nullptr, nullptr};  // match tables of buffer b complete
  hipEvent_t ev_post[2] = {nullptr, nullptr};    // post-processing finished reading buffer b
  bool ev_post_valid[2] = {false, false};
  int64_t match_seq = 0;
  int32_t *d_mchunk2[2] = {nullptr, nullptr};  // [S][cap/256] survivors per emission chunk, one buffer per match-table buffer (each launch's emission zeroes the other one)
  // per stream {match count, overflow flag, queries searched again, queries searched} of the last launch on each
  // buffer, written by emit_matches into host-mapped page-locked memory: valid after ev_post[buf]
  int4 *h_out[2] = {nullptr, nullptr}, *d_out_mapped[2] = {nullptr, nullptr};
  // small groups (serial): the match records are written to host-mapped memory as well, so getMatches is an
  // event wait and a host copy instead of a device->host transfer of its own
  vh_p_match *h_matches = nullptr; void *d_matches_mapped = nullptr;
  int32_t *d_redo = nullptr;     // [2][S] queries the speculative searches had to search again, per table buffer (reset by emit_matches)
  // Loop policy of the searches (match()): speculative (no accept test in the loop, the winner
  // verified, failures searched again) or tested.  The speculative loop is ~12 % faster when
  // almost every query's best candidate lies inside its window (0.5 % re-searched on the
  // benchmark frames) and slower once more than ~6 % fail (noisy images full of features
  // without a partner; measured round 3 with grouped second searches: +10 % at 3.3 % re-searched,
  // +3 % at 4.8 %, -1 % at 6.6 %, -6 % at 8.9 %).  Every launch reports (re-searched, searched)
  // with a lag of one or two steps; above 6.5 % the tested loop takes over and the speculative
  // one is probed every 16th launch, below 5.5 % it comes back.  Results never depend on the choice.
  bool stats_pending[2] = {false, false}, stats_was_spec[2] = {false, false};
  int32_t stats_npass[2] = {0, 0};
  int32_t tiles_hint = 0;  // query tiles per (pass, stream) row seen by an earlier launch (0: none yet)
  int32_t probe_countdown = 0, force_mode = -1;
  bool spec_mode = true;
  double last_redo_rate = -1;
  hipEvent_t ev_det[VH_RING] = {};   // slot fully detected + indexed
  hipEvent_t ev_read[VH_RING] = {};  // last match that read the slot
  bool ev_read_valid[VH_RING] = {};
  hipEvent_t ev_user = nullptr;
  bool user_stream_set = false;  // handle 0 is a real stream (the legacy default stream): "unset" is a flag, not a value
  bool failed = false;           // the last push did not complete: no matching until the next successful one
  int32_t *d_overflow = nullptr; // [S] 1: a feature set of the stream's last match held more records than cap
  int32_t *h_overflow = nullptr; // page-locked mirror for the asynchronous download
  int32_t pair_prev = 1;
  bool serial = false, own_post = false;

  bool allocated = false;
  int32_t dims[3] = {0, 0, 0};
  VhGeom g{};
  VhSets sets{};
  int32_t cap = 0, mcap = 0;
  int32_t pair_cur = 0;
  int64_t frames = 0;

  // host-image staging, S images per camera, two slots: the upload of frame t+1
  // does not wait for the detection of frame t, only for that of frame t-1
  uint8_t *d_stage_buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  uint8_t *d_stage[2] = {nullptr, nullptr};  // the slot of the last push_host
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  bool ev_stage_valid[2] = {false, false};
  int32_t stage_slot = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_h2d = nullptr;
  size_t stage_bytes = 0;
  // asynchronous download of the match lists (vh_group_download_matches_async)
  hipStream_t down_stream = nullptr;
  hipEvent_t ev_down = nullptr;
  bool ev_down_valid = false;
  int32_t last_buf = 0;
  uint8_t *d_half = nullptr;                 // half-resolution images [S*2]
  uint64_t *d_rec = nullptr;
  int32_t *d_chunk_count = nullptr;
  int32_t *d_best = nullptr, *d_best2[2] = {nullptr, nullptr};
  int4 *d_chain = nullptr, *d_chain2[2] = {nullptr, nullptr};
  uint32_t *d_mask = nullptr;
  uint32_t epoch = 0;
  void *d_matches = nullptr;
  int32_t *d_match_count = nullptr;
  std::vector<void *> allocs;
  int64_t device_bytes = 0;

  int32_t last_method = -1;
  // streams whose current matches were post-processed on the host
  // (vh_remove_outliers / vh_bucket_features): served from here until the next step
  std::vector<std::vector<vh_p_match>> host_matches;
  std::vector<uint8_t> host_filtered;
  void drop_host_matches() { std::fill(host_filtered.begin(), host_filtered.end(), 0); }

  bool prof = false;
  std::map<std::string, ProfEntry> prof_entries;

  ~Group() {
    release();
    for (int k = 0; k < VH_RING; k++) { if (ev_det[k]) (void)hipEventDestroy(ev_det[k]); if (ev_read[k]) (void)hipEventDestroy(ev_read[k]); }
    if (ev_user) (void)hipEventDestroy(ev_user);
    for (int k = 0; k < 2; k++) if (ev_stage[k]) (void)hipEventDestroy(ev_stage[k]);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (ev_h2d) (void)hipEventDestroy(ev_h2d);
    if (ev_down) (void)hipEventDestroy(ev_down);
    for (auto &sl : post_slot) if (sl.ev) (void)hipEventDestroy(sl.ev);
    if (down_stream) (void)hipStreamDestroy(down_stream);
    for (int k = 0; k < kVoteStreams; k++) if (vote_stream[k]) (void)hipStreamDestroy(vote_stream[k]);
    for (int k = 0; k < 2; k++) { if (ev_tables[k]) (void)hipEventDestroy(ev_tables[k]); if (ev_post[k]) (void)hipEventDestroy(ev_post[k]); }
    if (post_stream && own_post) (void)hipStreamDestroy(post_stream);
    if (match_stream && !serial) (void)hipStreamDestroy(match_stream);
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }
  int32_t pairs() const { return pair_cur | (pair_prev << 8); }
  int32_t sync_all() {
    VH_HIP(hipStreamSynchronize(stream));
    VH_HIP(hipStreamSynchronize(match_stream));
    VH_HIP(hipStreamSynchronize(post_stream));
    VH_HIP(hipStreamSynchronize(down_stream));
    for (int k = 0; k < kVoteStreams; k++) if (vote_stream[k]) VH_HIP(hipStreamSynchronize(vote_stream[k]));
    return check_violation();
  }
  // -DVH_CHECK builds: the kernels verify the index invariants they otherwise trust (vh_dev.h,
  // VH_CHECK_RANGE) and record the first violation; the host aborts at the next point where it
  // waits for the device anyway.  The shipped build compiles this to nothing.
  int32_t check_violation() {
#ifdef VH_CHECK
    if (allocated && sets.check) {
      uint32_t c[4] = {0, 0, 0, 0};
      VH_HIP(hipDeviceSynchronize());
      VH_HIP(hipMemcpy(c, sets.check, sizeof(c), hipMemcpyDeviceToHost));
      if (c[0]) {
        fprintf(stderr, "VH_CHECK: %u index violations; first: code %u, value %d, bound %d (codes: vh_dev.h)\n", c[0], c[1], (int)c[2], (int)c[3]);
        fflush(stderr);
        abort();
      }
    }
#endif
    return VH_OK;
  }

  void release() {
    vote_release();
    for (void *q : allocs) (void)hipFree(q);
    allocs.clear();
    device_bytes = 0;
    d_stage[0] = d_stage[1] = nullptr; stage_bytes = 0; ev_down_valid = false;
    for (int k = 0; k < 2; k++) d_stage_buf[k][0] = d_stage_buf[k][1] = nullptr, ev_stage_valid[k] = false;
    d_half = nullptr; d_rec = nullptr; d_chunk_count = nullptr; d_best = nullptr; d_chain = nullptr; d_prior_tr = nullptr;
    d_best2[0] = d_best2[1] = nullptr; d_chain2[0] = d_chain2[1] = nullptr; d_mchunk2[0] = d_mchunk2[1] = nullptr; d_redo = nullptr;
    for (int k = 0; k < 2; k++) if (h_out[k]) { (void)hipHostFree(h_out[k]); h_out[k] = nullptr; d_out_mapped[k] = nullptr; }
    if (h_matches) { (void)hipHostFree(h_matches); h_matches = nullptr; d_matches_mapped = nullptr; }
    stats_pending[0] = stats_pending[1] = false; tiles_hint = 0;
    d_mask = nullptr; d_matches = nullptr; d_match_count = nullptr; d_overflow = nullptr;
    d_ego_rand = nullptr; d_ego_ok = nullptr; d_ego_xyz = nullptr; d_ego_tr = nullptr; ego_rand_n = 0;
    d_mono_scratch = nullptr; d_mono_rand = nullptr; mono_rand_n = 0; mono_scratch_iters = 0;
    d_bucket = nullptr; d_bcnt = nullptr; bcap = 0; d_post_rand = nullptr; post_rand_n = 0; d_post_mono = nullptr; post_mono_iters = 0; d_post_xyz = nullptr; d_post_tr = nullptr; d_post_ok = nullptr;
    if (h_bucket) { (void)hipHostFree(h_bucket); h_bucket = nullptr; }
    if (h_bcnt) { (void)hipHostFree(h_bcnt); h_bcnt = nullptr; }
    for (auto &sl : post_slot) {
      if (sl.h_pm) { (void)hipHostFree(sl.h_pm); sl.h_pm = nullptr; }
      if (sl.h_cnt) { (void)hipHostFree(sl.h_cnt); sl.h_cnt = nullptr; }
      sl.cap_ps = 0; sl.pending = false;
    }
    if (h_overflow) { (void)hipHostFree(h_overflow); h_overflow = nullptr; }
    if (h_prior_tr) { (void)hipHostFree(h_prior_tr); h_prior_tr = nullptr; }
    allocated = false;
  }

  template <class T> int32_t dmalloc(T **out, size_t count, bool zero) {
    void *q = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (fail_next_alloc) { fail_next_alloc = false; t_last_error = "allocation failure requested by vh_group_debug_fail_next_alloc"; return VH_ERR_HIP; }
    VH_HIP(hipMalloc(&q, bytes));
    allocs.push_back(q);
    device_bytes += (int64_t)bytes;
    if (zero) VH_HIP(hipMemsetAsync(q, 0, bytes, stream));
    else {
      // VH_POISON=1 (test aid): fill every buffer that is not zero-initialised with 0xA5, so that a
      // kernel consuming memory nobody wrote misbehaves the same way on every box
      static const bool poison = [] { const char *e = getenv("VH_POISON"); return e && e[0] == '1'; }();
      if (poison) { VH_HIP(hipMemset(q, 0xA5, bytes)); VH_HIP(hipDeviceSynchronize()); }  // (blocking: the buffer's first user may be any stream)
    }
    *out = (T *)q;
    return VH_OK;
  }

  // release one block of `allocs` early (the work that used it must have completed)
  void dfree(void *q) {
    if (!q) return;
    auto it = std::find(allocs.begin(), allocs.end(), q);
    if (it == allocs.end()) return;
    (void)hipStreamSynchronize(down_stream);
    (void)hipFree(q);
    allocs.erase(it);
  }

  // ---- geometry ----------------------------------------------------------
  static int32_t block_count(int32_t extent, int32_t n) {
    // for (i=n+margin; i<extent-n-margin; i+=n+1)   (matcher.cpp:381-382)
    const int32_t lo = n + VH_MARGIN, hi = extent - n - VH_MARGIN;
    return hi > lo ? (hi - lo + n) / (n + 1) : 0;
  }

  int32_t setup_geometry(const int32_t d[3]) {
    g = VhGeom{};
    g.W = d[0]; g.H = d[1]; g.bpl = d[2];
    if (p.half_resolution) {  // getHalfResolutionDimensions, matcher.cpp:566-570
      g.Wm = d[0] / 2; g.Hm = d[1] / 2;
      g.bplm = g.Wm > 0 ? g.Wm + 15 - (g.Wm - 1) % 16 : 16;
      g.scale = 2;
    } else {
      g.Wm = d[0]; g.Hm = d[1]; g.bplm = d[2]; g.scale = 1;
    }
    g.n = p.nms_n; g.tau = p.nms_tau;
    g.nbx = block_count(g.Wm, g.n); g.nby = block_count(g.Hm, g.n);
    if (g.nbx == 0 || g.nby == 0) g.nbx = g.nby = 0;
    g.nblocks = g.nbx * g.nby;
    g.nchunks = std::max(1, (g.nblocks + VH_CHUNK - 1) / VH_CHUNK);
    static const int32_t cand[][2] = {{32, 8}, {16, 8}, {16, 4}, {8, 4}, {4, 4}, {4, 2}, {2, 2}, {2, 1}, {1, 1}};
    for (auto &c : cand) {
      g.tbx = c[0]; g.tby = c[1];
      g.FW = g.tbx * (g.n + 1) + 2 * g.n; g.FH = g.tby * (g.n + 1) + 2 * g.n;
      g.IW = g.FW + 4; g.IH = g.FH + 4;
      g.IWp = round_up(g.IW, 4); g.FWp = g.FW;
      const size_t lds = (size_t)g.IH * g.IWp + 4 * (size_t)g.FH * g.FWp + 4 + 24 * (size_t)g.tbx * g.tby;
      if (lds <= 60 * 1024) return VH_OK;
    }
    return VH_ERR_UNSUPPORTED;
  }

  int32_t ensure(const int32_t d[3]) {
    if (allocated && d[0] == dims[0] && d[1] == dims[1] && d[2] == dims[2]) return VH_OK;
    if (allocated) { int32_t rs = sync_all(); if (rs) return rs; release(); }
    if (d[0] <= 0 || d[1] <= 0 || d[2] < d[0]) return VH_ERR_INVALID_ARG;
    if (d[0] > 16384 || d[1] > 16384) return VH_ERR_UNSUPPORTED;
    // emit_features addresses its patch rows with 24 x 24 -> 32-bit byte offsets from the image base
    if (d[2] >= (1 << 24) || (int64_t)d[2] * d[1] > (1ll << 28)) return VH_ERR_UNSUPPORTED;
    int32_t rc = setup_geometry(d);
    if (rc != VH_OK) return rc;
    dims[0] = d[0]; dims[1] = d[1]; dims[2] = d[2];
    int64_t c = req_features > 0 ? req_features : std::max<int64_t>(4 * (int64_t)g.nblocks, 64);
    if (c > (1 << VH_MASK_IDX_BITS) - 1) c = (1 << VH_MASK_IDX_BITS) - 1;  // feature indices are packed into VH_MASK_IDX_BITS bits (flow pixel mask)
    cap = (int32_t)c;
    mcap = req_matches > 0 ? req_matches : cap;

    sets = VhSets{};
    sets.cap = cap;
    sets.binsize = p.match_binsize;
    // exact for 0 <= x <= 32768 (every coordinate +- radius) when binsize <= 32768; a larger bin holds every x: quotient 0
    sets.inv_binsize = p.match_binsize > 32768 ? 0u : (uint32_t)(((1ull << 32) + p.match_binsize - 1) / p.match_binsize);
    sets.ubn = (dims[0] + p.match_binsize - 1) / p.match_binsize;  // ceil(W/binsize), matcher.cpp:282-283
    sets.vbn = (dims[1] + p.match_binsize - 1) / p.match_binsize;
    sets.nbins = 4 * sets.ubn * sets.vbn;
    sets.max_tiles = cap / VH_TILE_Q + 5;  // full tiles + one partial tile per class
    sets.W = dims[0]; sets.H = dims[1];
    {  // a bin of binsize px meets at most ceil(binsize/block)+1 NMS blocks per axis, one feature per class each
      const int32_t blk = g.scale * (g.n + 1);
      const int64_t per_axis = (p.match_binsize + blk - 1) / blk + 1;
      sets.stage_cap = (int32_t)std::min<int64_t>(per_axis * per_axis, cap);
    }
    const size_t ns = 2 * VH_RING * (size_t)S;  // ring slots x (left, right) per stream
    if ((rc = dmalloc(&sets.feat, ns * cap * 12, false))) return rc;
    if ((rc = dmalloc(&sets.f_uv, ns * cap, false))) return rc;
    if ((rc = dmalloc(&sets.s_uv, ns * cap, false))) return rc;
    if ((rc = dmalloc(&sets.s_idx, ns * cap, false))) return rc;
    if ((rc = dmalloc(&sets.s_desc, ns * cap * 8, false))) return rc;
    if ((rc = dmalloc(&sets.bin_start, ns * (sets.nbins + 1), true))) return rc;
    if ((rc = dmalloc(&sets.hist, ns * sets.nbins, true))) return rc;
    if ((rc = dmalloc(&sets.cursor, ns * sets.nbins, true))) return rc;
    if ((rc = dmalloc(&sets.tmp_idx, ns * cap, false))) return rc;
    if ((rc = dmalloc(&sets.stage, ns * (size_t)sets.nbins * sets.stage_cap, false))) return rc;  // (per-bin staging lists: 2 MB per set at KITTI size)
    if ((rc = dmalloc(&sets.count, ns, true))) return rc;
    const size_t nrow = 4 * (size_t)dims[1];
    if ((rc = dmalloc(&sets.row_start, ns * (nrow + 1), true))) return rc;
    if ((rc = dmalloc(&sets.row_hist, ns * nrow, true))) return rc;
    if ((rc = dmalloc(&sets.row_cursor, ns * nrow, true))) return rc;
    if ((rc = dmalloc(&sets.r_pos, ns * cap, false))) return rc;
    if ((rc = dmalloc(&sets.tiles, ns * sets.max_tiles, false))) return rc;
    if ((rc = dmalloc(&sets.tile_cnt, ns, true))) return rc;
    if ((rc = dmalloc(&sets.check, 4, true))) return rc;
    if ((rc = dmalloc(&d_rec, 2 * (size_t)S * std::max(g.nblocks, 1), false))) return rc;
    if ((rc = dmalloc(&d_chunk_count, 2 * (size_t)S * g.nchunks, true))) return rc;
    for (int k = 0; k < 2; k++) {
      if ((rc = dmalloc(&d_best2[k], 4 * (size_t)S * cap, false))) return rc;
      if ((rc = dmalloc(&d_chain2[k], 2 * (size_t)S * cap, false))) return rc;  // index tuple + coordinate tuple per driving feature
    }
    d_best = d_best2[0]; d_chain = d_chain2[0];
    for (int k = 0; k < 2; k++) {
      if ((rc = dmalloc(&d_mchunk2[k], (size_t)S * ((cap + 255) / 256), true))) return rc;
      VH_HIP(hipHostMalloc((void **)&h_out[k], sizeof(int4) * (size_t)S, hipHostMallocMapped));
      memset(h_out[k], 0, sizeof(int4) * (size_t)S);
      VH_HIP(hipHostGetDevicePointer((void **)&d_out_mapped[k], h_out[k], 0));
    }
    if ((rc = dmalloc(&d_redo, 2 * (size_t)S, true))) return rc;  // one set of counters per match-table buffer: the search of match n+1 runs beside the emission of match n
    if ((rc = dmalloc((uint8_t **)&d_matches, (size_t)S * mcap * sizeof(vh_p_match), false))) return rc;
    if ((rc = dmalloc(&d_match_count, (size_t)S, true))) return rc;
    if (serial && (size_t)S * mcap * sizeof(vh_p_match) <= (64u << 20)) {
      VH_HIP(hipHostMalloc((void **)&h_matches, (size_t)S * mcap * sizeof(vh_p_match), hipHostMallocMapped));
      VH_HIP(hipHostGetDevicePointer(&d_matches_mapped, h_matches, 0));
    }
    if ((rc = dmalloc(&d_overflow, (size_t)S, true))) return rc;
    VH_HIP(hipHostMalloc((void **)&h_overflow, sizeof(int32_t) * (size_t)S, hipHostMallocDefault));
    memset(h_overflow, 0, sizeof(int32_t) * (size_t)S);
    if (p.half_resolution)
      if ((rc = dmalloc(&d_half, 2 * (size_t)S * g.bplm * g.Hm, false))) return rc;
    allocated = true;
    pair_cur = 0; pair_prev = 1; frames = 0; epoch = 0; last_method = -1; failed = false;
    host_matches.assign((size_t)S, {}); host_filtered.assign((size_t)S, 0);
    for (int k = 0; k < VH_RING; k++) ev_read_valid[k] = false;
    ev_post_valid[0] = ev_post_valid[1] = false; match_seq = 0;
    // every slot starts "detected" (empty): matches may wait on any of them
    for (int k = 0; k < VH_RING; k++) VH_HIP(hipEventRecord(ev_det[k], stream));
    return VH_OK;
  }

  // ---- profiling ---------------------------------------------------------
  struct Scope {
    Group *gq; const char *name; hipStream_t st; hipEvent_t e0 = nullptr, e1 = nullptr;
    Scope(Group *gq_, const char *n, hipStream_t st_) : gq(gq_), name(n), st(st_) {
      if (gq->prof) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, st); }
    }
    ~Scope() {
      if (gq->prof) { (void)hipEventRecord(e1, st); gq->prof_entries[name].pending.emplace_back(e0, e1); }
    }
  };
  void prof_collect() {
    for (auto &kv : prof_entries) {
      for (auto &pr : kv.second.pending) {
        float ms = 0;
        (void)hipEventSynchronize(pr.second);
        (void)hipEventElapsedTime(&ms, pr.first, pr.second);
        kv.second.ms += ms; kv.second.launches++;
        (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second);
      }
      kv.second.pending.clear();
    }
  }

  // ---- detect + bin ------------------------------------------------------
  int32_t zero_bin_counters(int32_t set0, int32_t nsets, int32_t *extra = nullptr, int64_t n_extra = 0) {
    // one launch instead of a memset per array
    vh_launch_zero_counters(sets, set0, nsets, extra, n_extra, stream);
    VH_HIP(hipGetLastError());
    return VH_OK;
  }

  // staged: the histograms and per-bin member lists were already produced by
  // emit_features; otherwise (caller-supplied features) build them here.
  int32_t bin_sets(int32_t set0, int32_t nsets, bool staged) {
    if (!staged) { Scope sc(this, "bin_hist", stream); vh_launch_bin_hist(sets, set0, nsets, stream); }
    { Scope sc(this, "bin_scan", stream); vh_launch_bin_scan(sets, set0, nsets, stream); }
    if (!staged) { Scope sc(this, "bin_fill", stream); vh_launch_bin_fill(sets, set0, nsets, stream); }
    { Scope sc(this, "bin_sort", stream); vh_launch_bin_sort(sets, set0, nsets, staged ? 1 : 0, stream); }
    VH_HIP(hipGetLastError());
    return VH_OK;
  }

  // pushBack: a failure after the ring has rotated leaves the new slot half written;
  // the roles are put back and the handle refuses to match (VH_ERR_STATE) until a
  // later push has succeeded.
  int32_t push_device(const void *dI1, const void *dI2, int64_t stride, const int32_t d[3], int32_t replace) {
    if (!dI1 || !d) return VH_ERR_INVALID_ARG;
    int32_t rc = ensure(d);
    if (rc != VH_OK) return rc;
    const int32_t old_cur = pair_cur, old_prev = pair_prev;
    const int64_t old_frames = frames;
    rc = push_device_queued(dI1, dI2, stride, d, replace);
    if (rc != VH_OK) { pair_cur = old_cur; pair_prev = old_prev; frames = old_frames; failed = true; }
    else failed = false;
    return rc;
  }

  int32_t push_device_queued(const void *dI1, const void *dI2, int64_t stride, const int32_t d[3], int32_t replace) {
    int32_t rc = VH_OK;
    if (!replace && frames > 0) {  // ring buffer shift (matcher.cpp:64-79): prev <- cur, cur <- the slot used longest ago
      const int32_t fresh = (pair_cur + 1) % VH_RING;
      pair_prev = pair_cur;
      pair_cur = fresh;
    }
    frames++;
    drop_host_matches(); last_method = -1;
    const int32_t set0 = pair_cur * 2 * S, nsets = 2 * S;
    // order after the caller's stream (image producers) and after the last match
    // that still reads the slot we are about to overwrite
    if (user_stream_set) {
      VH_HIP(hipEventRecord(ev_user, user_stream));
      VH_HIP(hipStreamWaitEvent(stream, ev_user, 0));
    }
    if (ev_read_valid[pair_cur]) VH_HIP(hipStreamWaitEvent(stream, ev_read[pair_cur], 0));
    if ((rc = zero_bin_counters(set0, nsets, d_chunk_count, 2 * (int64_t)S * g.nchunks))) return rc;
    // The group is detected in up to four sub-batches of streams, one after the other on this
    // stream: the latency-bound kernels of a sub-batch (emit_features, bin_scan, bin_sort: < 45 %
    // of the VALU issue slots) then run beside the issue-bound ones of its neighbours and of the
    // previous frame's search instead of all at once.  Measured on MI355X, KITTI, S = 256 (with
    // the post stream): 1 / 2 / 4 / 8 / 16 sub-batches = 94.9 / 95.9 / 97.1 / 92.1 / 82.0 k pairs/s
    // -- below ~12 k detection workgroups per launch the launches themselves cost more.  (Round 1
    // measured the opposite, -9 % at n = 2: its kernels were not yet issue-bound.)  Sub-batches on
    // two alternating streams lose 8 %.  VH_SUBBATCH=n overrides.
    const int32_t ncam = dI2 ? 2 : 1;
    static const int subbatch_env = [] { const char *ev = getenv("VH_SUBBATCH"); return ev ? atoi(ev) : 0; }();
    const int64_t det_wgs = (int64_t)S * ncam * ((g.nblocks + 255) / 256);
    // (a mono push is half the detection work of a stereo one: 4 sub-batches of 64 KITTI images leave emit_features with two
    //  rounds of workgroups per launch -- mono flow, S = 256: 1 / 2 / 4 / 8 sub-batches = 118 / 115 / 109 / 101 k frames/s)
    const int32_t subbatch = subbatch_env > 0 ? subbatch_env : (serial ? 1 : (int32_t)std::min<int64_t>(4, det_wgs / (ncam == 2 ? 12000 : 40000)));
    const int32_t nsub = std::max(1, std::min(subbatch, S));
    const int32_t ssub = (S + nsub - 1) / nsub;
    for (int32_t s0 = 0; s0 < S; s0 += ssub) {
      const int32_t sn = std::min(ssub, S - s0);
      VhImages im{};
      im.base[0] = (const uint8_t *)dI1 + (int64_t)s0 * stride;
      im.base[1] = dI2 ? (const uint8_t *)dI2 + (int64_t)s0 * stride : nullptr;
      im.stride = stride; im.ncam = ncam; im.S = sn; im.S_total = S; im.s0 = s0; im.pair_cur = pair_cur;
      uint64_t *rec = d_rec + (size_t)s0 * ncam * std::max(g.nblocks, 1);
      int32_t *chunks = d_chunk_count + (size_t)s0 * ncam * g.nchunks;
      if (p.half_resolution) {
        const int64_t isz = (int64_t)g.bplm * g.Hm;
        uint8_t *half = d_half + (int64_t)s0 * ncam * isz;
        { Scope sc(this, "half_res", stream); vh_launch_half_res(im, half, g, stream); }
        // half images are stored by image id; present them as cameras with stride ncam*isz
        im.base[0] = half; im.base[1] = half + isz; im.stride = isz * ncam;
      }
      { Scope sc(this, "detect_nms", stream); vh_launch_detect_nms(im, g, rec, chunks, stream); }
      { Scope sc(this, "emit_features", stream); vh_launch_emit_features(im, g, rec, chunks, sets, stream); }
      VH_HIP(hipGetLastError());
      if ((rc = bin_sets(set0 + 2 * s0, 2 * sn, true))) return rc;
    }
    VH_HIP(hipEventRecord(ev_det[pair_cur], stream));
    return VH_OK;
  }

  int32_t push_host(const uint8_t *I1, const uint8_t *I2, int64_t stride, const int32_t d[3], int32_t replace) {
    if (!I1 || !d) return VH_ERR_INVALID_ARG;
    int32_t rc = ensure(d);
    if (rc != VH_OK) return rc;
    const size_t isz = (size_t)d[2] * d[1];
    if (stage_bytes < isz * S) {
      VH_HIP(hipStreamSynchronize(stream));
      for (int sl = 0; sl < 2; sl++)
        for (int k = 0; k < 2; k++) if ((rc = dmalloc(&d_stage_buf[sl][k], isz * S, false))) return rc;
      stage_bytes = isz * S;
      ev_stage_valid[0] = ev_stage_valid[1] = false;
    }
    const int32_t sl = stage_slot;
    stage_slot ^= 1;
    // the detection that last read this staging slot (two pushes ago) must be done
    hipStream_t cs = serial ? stream : copy_stream;  // (a small group runs everything on one stream: no hand-over between streams)
    if (ev_stage_valid[sl] && !serial) VH_HIP(hipStreamWaitEvent(cs, ev_stage[sl], 0));
    for (int k = 0; k < 2; k++) {
      const uint8_t *src = k ? I2 : I1;
      d_stage[k] = d_stage_buf[sl][k];
      if (!src) continue;
      if (stride == (int64_t)isz) {  // one transfer for all S images
        VH_HIP(hipMemcpyAsync(d_stage[k], src, isz * S, hipMemcpyHostToDevice, cs));
      } else {
        for (int32_t s = 0; s < S; s++)
          VH_HIP(hipMemcpyAsync(d_stage[k] + isz * s, src + stride * s, isz, hipMemcpyHostToDevice, cs));
      }
    }
    // The images are only borrowed for the duration of the call (demo.cpp:250-251): the host waits for the
    // copies -- but only after the detection has been queued behind them, so the first kernel starts
    // when the last byte lands instead of a host round trip later.
    if (!ev_h2d) VH_HIP(hipEventCreateWithFlags(&ev_h2d, hipEventDisableTiming));
    VH_HIP(hipEventRecord(ev_h2d, cs));
    if (!serial) VH_HIP(hipStreamWaitEvent(stream, ev_h2d, 0));
    rc = push_device(d_stage[0], I2 ? d_stage[1] : nullptr, (int64_t)isz, d, replace);
    VH_HIP(hipEventSynchronize(ev_h2d));
    if (rc == VH_OK) {
      VH_HIP(hipEventRecord(ev_stage[sl], stream));
      ev_stage_valid[sl] = true;
    }
    return rc;
  }

  // ---- match ---------------------------------------------------------------
  VhMatchArgs match_args(int32_t method) const {
    VhMatchArgs a{};
    a.S = S; a.pair_cur = pairs(); a.radius = p.match_radius; a.disp_tol = p.match_disp_tolerance;
    if (method == VH_METHOD_FLOW) {  // matcher.cpp:320-321
      a.npass = 2; a.pass[0] = {VH_SET_1C, VH_SET_1P, 1, 0}; a.pass[1] = {VH_SET_1P, VH_SET_1C, 1, 1};
    } else if (method == VH_METHOD_STEREO) {
      a.npass = 2; a.pass[0] = {VH_SET_1C, VH_SET_2C, 0, 0}; a.pass[1] = {VH_SET_2C, VH_SET_1C, 0, 1};
    } else {
      a.npass = 4;
      a.pass[0] = {VH_SET_1P, VH_SET_2P, 0, 0}; a.pass[1] = {VH_SET_2P, VH_SET_2C, 1, 1};
      a.pass[2] = {VH_SET_2C, VH_SET_1C, 0, 2}; a.pass[3] = {VH_SET_1C, VH_SET_1P, 1, 3};
    }
    return a;
  }

  // speculative or tested search loops for the next launch (see the members above)
  bool choose_loop() {
    for (int sl = 0; sl < 2; sl++) {
      if (!stats_pending[sl] || hipEventQuery(ev_post[sl]) != hipSuccess) continue;
      stats_pending[sl] = false;
      int64_t again = 0, searched = 0;
      int32_t nq_max = 0;
      for (int32_t s = 0; s < S; s++) { again += h_out[sl][s].z; searched += h_out[sl][s].w; nq_max = std::max(nq_max, h_out[sl][s].w); }
      // query tiles the fullest stream's sets held, per pass (the searches' grid is sized by it: vh_launch_match)
      if (stats_npass[sl] > 0) tiles_hint = nq_max / stats_npass[sl] / VH_TILE_Q + 4;
      if (!stats_was_spec[sl] || searched <= 0) continue;  // the tested loop reports nothing
      last_redo_rate = (double)again / (double)searched;
      if (spec_mode && last_redo_rate > 0.065) { spec_mode = false; probe_countdown = 16; }
      else if (!spec_mode && last_redo_rate < 0.055) spec_mode = true;
    }
    if (force_mode >= 0) return force_mode == 1;
    if (spec_mode) return true;
    if (--probe_countdown <= 0) { probe_countdown = 16; return true; }  // probe
    return false;
  }

  // A match launch that failed half way leaves the emission's chunk counters and the re-search counters of its table
  // buffer in an unknown state (each emission zeroes the OTHER buffer's counters for the next launch): the next
  // match() puts both buffers back to zero before it queues anything.
  bool match_dirty = false;
  bool fail_next_alloc = false;  // test hook (vh_group_debug_fail_next_alloc)

  int32_t match_recover() {
    VH_HIP(hipStreamSynchronize(match_stream));
    VH_HIP(hipStreamSynchronize(post_stream));
    for (int k = 0; k < 2; k++) VH_HIP(hipMemset(d_mchunk2[k], 0, sizeof(int32_t) * (size_t)S * ((cap + 255) / 256)));
    VH_HIP(hipMemset(d_redo, 0, sizeof(int32_t) * 2 * (size_t)S));
    stats_pending[0] = stats_pending[1] = false; tiles_hint = 0;
    last_method = -1;
    match_dirty = false;
    return VH_OK;
  }

  // tr16: null, or [S][16] row-major motion estimates for the quad method's prior (kernels_prior.hip); the intrinsics must be set
  double *d_prior_tr = nullptr;
  double *h_prior_tr = nullptr;  // page-locked staging, two slots (one per table buffer): the caller's array is only borrowed
  int32_t match(int32_t method, const double *tr16 = nullptr) {
    if (method < 0 || method > 2) return VH_ERR_INVALID_ARG;
    if (!allocated || failed) return VH_ERR_STATE;
    if (tr16 && method != VH_METHOD_QUAD) tr16 = nullptr;  // (stock libviso2 uses the prediction in the quad circle only)
    if (tr16 && !(p.f > 0 && p.base > 0)) return VH_ERR_STATE;  // setIntrinsics first
    if (match_dirty) { const int32_t rr = match_recover(); if (rr) return rr; }
    if (tr16 && !d_prior_tr) { const int32_t rt = dmalloc(&d_prior_tr, 16 * (size_t)S, false); if (rt) { d_prior_tr = nullptr; return rt; } }
    if (tr16 && !h_prior_tr) VH_HIP(hipHostMalloc((void **)&h_prior_tr, sizeof(double) * 2 * 16 * (size_t)S, hipHostMallocDefault));
    // everything that can fail without a kernel of the step in flight comes first
    bool fresh_mask = false;
    if (method == VH_METHOD_FLOW && !d_mask) {
      int32_t rc = dmalloc(&d_mask, (size_t)S * dims[0] * dims[1], false); if (rc) { d_mask = nullptr; return rc; }
      fresh_mask = true;
    }
    const int32_t rc = match_queued(method, fresh_mask, tr16);
    if (rc) match_dirty = true;
    return rc;
  }

  int32_t match_queued(int32_t method, bool fresh_mask, const double *tr16) {
    VhMatchArgs a = match_args(method);
    a.prior = tr16 ? 1 : 0;
    hipStream_t ms = match_stream, ps = post_stream;
    const int32_t buf = (int32_t)(match_seq++ & 1);
    // the current slot's detection+indexing must be complete (the previous
    // slot's finished earlier on the same stream), and the post-processing that
    // last read this table buffer (two matches ago) must be done with it
    VH_HIP(hipStreamWaitEvent(ms, ev_det[pair_cur], 0));
    VH_HIP(hipStreamWaitEvent(ms, ev_det[pair_prev], 0));
    if (ev_post_valid[buf]) VH_HIP(hipStreamWaitEvent(ms, ev_post[buf], 0));
    const bool spec = choose_loop();
    // A stepped group shares the chip with its own detection chain: a grid as tight as the tiles the sets really hold, and
    // the searches' workgroups padded to an LDS footprint that leaves the chain room on every CU (vh_launch_match has the
    // measurements).  How many LDS allocation units (1 280 bytes) is a property of what runs beside the searches: with
    // detect_nms<1|2> (14 units) and emit_features (21) six search workgroups of 21 units per CU are best (KITTI 107.5 ->
    // 111.5 k, 1080p 18.5 -> 19.2 k; 20 or 22-25 units lose 2-3 % against no hint at all); with detect_nms<3> (22 units,
    // 69 registers: the 4K configuration) five workgroups of 23-25 units (3.68 -> 4.08 k; 21 units: 3.67).  KITTI frames at
    // nms_n = 1 / 4 (detect_nms<1> 9 units, <4> 33 units): 21 units 86.0 -> 89.2 k / 106.1 -> 108.9 k, 24 units 87.7 / 105.6.
    // The generic detector (nms_n >= 5, unaligned strides) was not measured: no hint.  VH_MATCH_LDS_UNITS overrides (0: none).
    static const int units_env = [] { const char *ev = getenv("VH_MATCH_LDS_UNITS"); return ev ? atoi(ev) : -1; }();
    const int32_t units = units_env >= 0 ? units_env : (g.n == 3 ? 24 : (g.n <= 4 ? 21 : 0));
    const int32_t gx_hint = (!serial && units > 0 && tiles_hint > 0) ? (tiles_hint + 3) / 4 : 0;
    { Scope sc(this, "match", ms); vh_launch_match(sets, a, d_best2[buf], d_redo + (size_t)buf * S, spec ? 1 : 0, gx_hint, units * 1280, ms); }
    VH_HIP(hipGetLastError());
    if (tr16) {  // hop 2 of the circle, per driving feature, behind the 1p -> 2p table of the launch above
      double *ht = h_prior_tr + (size_t)buf * 16 * S;
      VH_HIP(hipEventSynchronize(ev_tables[buf]));  // (recorded behind the copy that last read this slot, two matches ago; at once if never recorded)
      memcpy(ht, tr16, sizeof(double) * 16 * (size_t)S);
      VH_HIP(hipMemcpyAsync(d_prior_tr, ht, sizeof(double) * 16 * (size_t)S, hipMemcpyHostToDevice, ms));
      { Scope sc(this, "quad_prior", ms); vh_launch_quad_prior(sets, a, d_prior_tr, p.f, p.cu, p.cv, p.base, d_best2[buf], ms); }
      VH_HIP(hipGetLastError());
    }
    VH_HIP(hipEventRecord(ev_tables[buf], ms));
    VH_HIP(hipStreamWaitEvent(ps, ev_tables[buf], 0));
    int32_t *d_mchunk = d_mchunk2[buf];  // zeroed by the previous launch's emission (at allocation for the first two)
    if (method == VH_METHOD_FLOW) {
      if (fresh_mask) {
        VH_HIP(hipMemsetAsync(d_mask, 0, sizeof(uint32_t) * (size_t)S * dims[0] * dims[1], ps));
        epoch = 0;
      }
      if (++epoch >= (1u << (32 - VH_MASK_IDX_BITS)) - 1) {
        VH_HIP(hipMemsetAsync(d_mask, 0, sizeof(uint32_t) * (size_t)S * dims[0] * dims[1], ps));
        epoch = 1;
      }
    }
    { Scope sc(this, "chain", ps); vh_launch_chain(sets, a, method, d_best2[buf], d_chain2[buf], d_mask, epoch, d_mchunk, ps); }
    // a download of the previous step's lists may still be reading d_matches
    if (ev_down_valid) VH_HIP(hipStreamWaitEvent(ps, ev_down, 0));
    { Scope sc(this, "emit_matches", ps); vh_launch_emit_matches(sets, a, method, d_chain2[buf], d_matches, mcap, d_match_count, d_overflow, d_mchunk, d_redo + (size_t)buf * S, d_mchunk2[buf ^ 1], d_out_mapped[buf], d_matches_mapped, ps); }
    VH_HIP(hipGetLastError());
    // (re-searched, searched) of this launch are read from h_out[buf] by a later choose_loop()
    stats_pending[buf] = true; stats_was_spec[buf] = spec; stats_npass[buf] = a.npass;
    VH_HIP(hipEventRecord(ev_post[buf], ps)); ev_post_valid[buf] = true;
    // both slots stay in use until this point of the post stream
    VH_HIP(hipEventRecord(ev_read[pair_cur], ps)); ev_read_valid[pair_cur] = true;
    VH_HIP(hipEventRecord(ev_read[pair_prev], ps)); ev_read_valid[pair_prev] = true;
    last_method = method; drop_host_matches(); last_buf = buf;
    return VH_OK;
  }

  // Start the device->host copy of every stream's first cap_per_stream match
  // records and of the S counts, ordered after the emission of the last step,
  // and return at once.  One strided transfer: the copy engine moves it while
  // the next step computes (its emit_matches waits for the download, above).
  int32_t download_async(vh_p_match *out, int32_t cap_per_stream, int32_t *counts) {
    if (!out || !counts || cap_per_stream < 1) return VH_ERR_INVALID_ARG;
    if (!allocated || last_method < 0) return VH_ERR_STATE;
    VH_HIP(hipStreamWaitEvent(down_stream, ev_post[last_buf], 0));
    const size_t width = sizeof(vh_p_match) * (size_t)std::min(cap_per_stream, mcap);
    VH_HIP(hipMemcpy2DAsync(out, sizeof(vh_p_match) * (size_t)cap_per_stream, d_matches, sizeof(vh_p_match) * (size_t)mcap,
                            width, (size_t)S, hipMemcpyDeviceToHost, down_stream));
    VH_HIP(hipMemcpyAsync(counts, d_match_count, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, down_stream));
    VH_HIP(hipMemcpyAsync(h_overflow, d_overflow, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, down_stream));
    VH_HIP(hipEventRecord(ev_down, down_stream));
    ev_down_valid = true;
    return VH_OK;
  }
  int32_t wait_download() {
    if (!ev_down_valid) return VH_OK;
    VH_HIP(hipEventSynchronize(ev_down));
    { const int32_t rv_ = check_violation(); if (rv_) return rv_; }
    for (int32_t s = 0; s < S; s++) if (h_overflow[s]) return VH_ERR_CAPACITY;
    return VH_OK;
  }

  int32_t get_matches(int32_t s, vh_p_match *out, int32_t capo, int32_t *n) {
    if (!n || s < 0 || s >= S || capo < 0 || (capo > 0 && !out)) return VH_ERR_INVALID_ARG;
    *n = 0;
    if (!allocated || last_method < 0) return VH_OK;
    if (host_filtered[s]) {
      *n = (int32_t)host_matches[s].size();
      const int32_t k = std::min(*n, capo);
      if (k) memcpy(out, host_matches[s].data(), sizeof(vh_p_match) * (size_t)k);
      return *n > capo ? VH_ERR_CAPACITY : VH_OK;
    }
    // count and overflow flag of the last launch: host-mapped memory, valid once its emission has run
    VH_HIP(hipEventSynchronize(ev_post[last_buf]));
    { const int32_t rv_ = check_violation(); if (rv_) return rv_; }
    const int32_t cnt = h_out[last_buf][s].x, ov = h_out[last_buf][s].y;
    *n = cnt;
    const int32_t k = std::min(std::min(cnt, mcap), capo);
    if (k > 0 && h_matches) {
      memcpy(out, h_matches + (size_t)s * mcap, sizeof(vh_p_match) * (size_t)k);
    } else if (k > 0) {
      VH_HIP(hipMemcpyAsync(out, (const uint8_t *)d_matches + (size_t)s * mcap * sizeof(vh_p_match),
                            sizeof(vh_p_match) * (size_t)k, hipMemcpyDeviceToHost, post_stream));
      VH_HIP(hipStreamSynchronize(post_stream));
    }
    // ov: a feature set of this match exceeded the feature capacity (the records beyond
    // it were dropped, so the list above comes from a truncated set)
    return (cnt > capo || cnt > mcap || ov) ? VH_ERR_CAPACITY : VH_OK;
  }

  int32_t get_features(int32_t s, int32_t which, int32_t *out12, int32_t capo, int32_t *n) {
    if (!n || s < 0 || s >= S || which < 0 || which > 3 || capo < 0 || (capo > 0 && !out12)) return VH_ERR_INVALID_ARG;
    *n = 0;
    if (!allocated) return VH_OK;
    const int32_t set = vh_role_set(S, pairs(), s, which);
    int32_t cnt = 0;
    VH_HIP(hipMemcpyAsync(&cnt, sets.count + set, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    VH_HIP(hipStreamSynchronize(stream));
    *n = cnt;
    const int32_t k = std::min(std::min(cnt, cap), capo);
    if (k > 0) {
      VH_HIP(hipMemcpyAsync(out12, sets.feat + (size_t)set * cap * 12, sizeof(int32_t) * 12 * (size_t)k,
                            hipMemcpyDeviceToHost, stream));
      VH_HIP(hipStreamSynchronize(stream));
    }
    return (cnt > capo || cnt > cap) ? VH_ERR_CAPACITY : VH_OK;
  }

  int32_t get_counts(int32_t *nf, int32_t *nm) {
    if (!allocated) return VH_ERR_STATE;
    if (nf) {
      std::vector<int32_t> all(2 * VH_RING * (size_t)S);
      VH_HIP(hipMemcpyAsync(all.data(), sets.count, sizeof(int32_t) * all.size(), hipMemcpyDeviceToHost, stream));
      VH_HIP(hipStreamSynchronize(stream));
      for (int32_t s = 0; s < S; s++)
        for (int32_t r = 0; r < 4; r++) nf[4 * s + r] = all[vh_role_set(S, pairs(), s, r)];
    }
    if (nm) {
      VH_HIP(hipMemcpyAsync(nm, d_match_count, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
      VH_HIP(hipStreamSynchronize(post_stream));
      for (int32_t s = 0; s < S; s++)
        if (host_filtered[s]) nm[s] = (int32_t)host_matches[s].size();
    }
    return VH_OK;
  }

  // All streams' matches in one go: counts first, then one transfer per stream
  // into out[s * cap_per_stream ...], a single wait at the end.
  int32_t get_matches_all(vh_p_match *out, int32_t cap_per_stream, int32_t *counts) {
    if (!out || !counts || cap_per_stream < 0) return VH_ERR_INVALID_ARG;
    for (int32_t s = 0; s < S; s++) counts[s] = 0;
    if (!allocated || last_method < 0) return VH_OK;
    VH_HIP(hipMemcpyAsync(counts, d_match_count, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipMemcpyAsync(h_overflow, d_overflow, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipStreamSynchronize(post_stream));
    { const int32_t rv_ = check_violation(); if (rv_) return rv_; }
    bool over = false;
    for (int32_t s = 0; s < S; s++) over = over || h_overflow[s] != 0;
    for (int32_t s = 0; s < S; s++) {
      if (host_filtered[s]) {
        counts[s] = (int32_t)host_matches[s].size();
        const int32_t k = std::min(counts[s], cap_per_stream);
        if (k) memcpy(out + (size_t)s * cap_per_stream, host_matches[s].data(), sizeof(vh_p_match) * (size_t)k);
        over = over || counts[s] > cap_per_stream;
        continue;
      }
      const int32_t k = std::min(std::min(counts[s], mcap), cap_per_stream);
      over = over || counts[s] > cap_per_stream || counts[s] > mcap;
      if (k > 0)
        VH_HIP(hipMemcpyAsync(out + (size_t)s * cap_per_stream, (const uint8_t *)d_matches + (size_t)s * mcap * sizeof(vh_p_match),
                              sizeof(vh_p_match) * (size_t)k, hipMemcpyDeviceToHost, post_stream));
    }
    VH_HIP(hipStreamSynchronize(post_stream));
    return over ? VH_ERR_CAPACITY : VH_OK;
  }

  // Bring stream s's current matches to the host (no-op if already there).
  int32_t fetch_matches(int32_t s) {
    if (s < 0 || s >= S) return VH_ERR_INVALID_ARG;
    if (!allocated || last_method < 0) return VH_ERR_STATE;
    if (host_filtered[s]) return VH_OK;
    int32_t n = 0;
    int32_t rc = get_matches(s, nullptr, 0, &n);
    if (rc != VH_OK && rc != VH_ERR_CAPACITY) return rc;
    if (n > mcap) return VH_ERR_CAPACITY;
    std::vector<vh_p_match> pm((size_t)n);
    if ((rc = get_matches(s, n ? pm.data() : nullptr, n, &n))) return rc;  // VH_ERR_CAPACITY: a feature set overflowed
    host_matches[s].swap(pm);
    host_filtered[s] = 1;
    return VH_OK;
  }

  // removeOutliers (remove_outliers.cpp:4-94) on streams [0, S): host work, one
  // stream per task, `threads` workers.  Stereo records carry no previous-frame
  // position (u1p = -1), so the flow vote only applies to flow and quad matches.
  int32_t remove_outliers(int32_t s_lo, int32_t s_hi, int32_t threads) {
    if (!allocated || last_method < 0) return VH_ERR_STATE;
    if (last_method == VH_METHOD_STEREO) return VH_OK;
    for (int32_t s = s_lo; s < s_hi; s++) {
      const int32_t rc = fetch_matches(s);
      if (rc) return rc;
    }
    std::atomic<int32_t> next_stream(s_lo), failed(0);
    const auto work = [&]() {
      for (int32_t s = next_stream++; s < s_hi; s = next_stream++) {
        std::vector<vh_p_match> &pm = host_matches[s];
        int32_t kept = 0;
        if (vh_remove_outliers_pm(pm.data(), (int32_t)pm.size(), &kept) != VH_OK) { failed = 1; continue; }
        pm.resize((size_t)kept);
      }
    };
    const int32_t nw = std::max(1, std::min(threads, s_hi - s_lo));
    std::vector<std::thread> pool;
    for (int32_t w = 1; w < nw; w++) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    return failed ? VH_ERR_INVALID_ARG : VH_OK;
  }

  // VisualOdometryStereo::estimateMotion on the device-resident quad match lists of every stream.
  // The scratch (random values, 3-d points, results) belongs to the group and only grows: a
  // hipFree per call would drain the detect/match/post pipeline (it synchronises the device).
  int32_t *d_ego_rand = nullptr, *d_ego_ok = nullptr;
  double *d_ego_xyz = nullptr, *d_ego_tr = nullptr;
  size_t ego_rand_n = 0;
  int32_t estimate_motion(const vh_ego_params *e, const int32_t *rand3, double *tr, int32_t *ok, int32_t *ninl) {
    if (!e || !rand3 || !tr || !ok || !ninl || e->ransac_iters < 1) return VH_ERR_INVALID_ARG;
    if (!allocated || last_method != VH_METHOD_QUAD) return VH_ERR_STATE;
    const size_t nr = (size_t)S * e->ransac_iters * 3;
    int32_t rc;
    if (!d_ego_xyz) {
      if ((rc = dmalloc(&d_ego_xyz, (size_t)S * mcap * 4, false))) return rc;
      if (!d_ego_tr) {
        if ((rc = dmalloc(&d_ego_tr, 6 * (size_t)S, false))) return rc;
        if ((rc = dmalloc(&d_ego_ok, 2 * (size_t)S, false))) return rc;
      }
    }
    if (ego_rand_n < nr) {  // (the old block stays in `allocs` until the group is released: a few KB per change of ransac_iters)
      if ((rc = dmalloc(&d_ego_rand, nr, false))) return rc;
      ego_rand_n = nr;
    }
    VH_HIP(hipMemcpyAsync(d_ego_rand, rand3, sizeof(int32_t) * nr, hipMemcpyHostToDevice, post_stream));
    vh_launch_ego(*e, S, (const vh_p_match *)d_matches, mcap, nullptr, d_match_count, mcap, d_ego_rand, d_ego_xyz, mcap, d_ego_tr, d_ego_ok,
                  d_ego_ok + S, nullptr, 0, post_stream);
    VH_HIP(hipGetLastError());
    VH_HIP(hipMemcpyAsync(tr, d_ego_tr, sizeof(double) * 6 * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipMemcpyAsync(ok, d_ego_ok, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipMemcpyAsync(ninl, d_ego_ok + S, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    // a truncated match list or feature set yields a pose of the truncated data: say so, as every get_matches path does
    std::vector<int32_t> cnt((size_t)S);
    VH_HIP(hipMemcpyAsync(cnt.data(), d_match_count, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipMemcpyAsync(h_overflow, d_overflow, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipStreamSynchronize(post_stream));
    for (int32_t s = 0; s < S; s++) if (cnt[s] > mcap || h_overflow[s]) return VH_ERR_CAPACITY;
    return VH_OK;
  }

  // VisualOdometryMono::estimateMotion on the device-resident match lists of every stream
  uint8_t *d_mono_scratch = nullptr;
  int32_t *d_mono_rand = nullptr;
  size_t mono_rand_n = 0;
  int32_t mono_scratch_iters = 0;
  int32_t estimate_motion_mono(const vh_mono_params *e, const int32_t *rand8, double *tr, int32_t *ok, int32_t *ninl) {
    if (!e || !rand8 || !tr || !ok || !ninl || e->ransac_iters < 1) return VH_ERR_INVALID_ARG;
    if (!allocated || (last_method != VH_METHOD_FLOW && last_method != VH_METHOD_QUAD)) return VH_ERR_STATE;
    const size_t nr = (size_t)S * e->ransac_iters * 8;
    int32_t rc;
    if ((int64_t)S * e->ransac_iters > (int64_t)1 << 31) return VH_ERR_UNSUPPORTED;
    if (!d_mono_scratch || mono_scratch_iters < e->ransac_iters) {
      if ((rc = dmalloc(&d_mono_scratch, (size_t)vh_mono_scratch_bytes(S, mcap, e->ransac_iters), false))) return rc;
      mono_scratch_iters = e->ransac_iters;
      if (!d_ego_tr) {
        if ((rc = dmalloc(&d_ego_tr, 6 * (size_t)S, false))) return rc;
        if ((rc = dmalloc(&d_ego_ok, 2 * (size_t)S, false))) return rc;
      }
    }
    if (mono_rand_n < nr) {
      if ((rc = dmalloc(&d_mono_rand, nr, false))) return rc;
      mono_rand_n = nr;
    }
    VH_HIP(hipMemcpyAsync(d_mono_rand, rand8, sizeof(int32_t) * nr, hipMemcpyHostToDevice, post_stream));
    vh_launch_mono(*e, S, (const vh_p_match *)d_matches, mcap, nullptr, d_match_count, mcap, d_mono_rand, d_mono_scratch, mcap, d_ego_tr,
                   d_ego_ok, d_ego_ok + S, nullptr, 0, post_stream);
    VH_HIP(hipGetLastError());
    VH_HIP(hipMemcpyAsync(tr, d_ego_tr, sizeof(double) * 6 * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipMemcpyAsync(ok, d_ego_ok, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipMemcpyAsync(ninl, d_ego_ok + S, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    std::vector<int32_t> cnt((size_t)S);
    VH_HIP(hipMemcpyAsync(cnt.data(), d_match_count, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipMemcpyAsync(h_overflow, d_overflow, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, post_stream));
    VH_HIP(hipStreamSynchronize(post_stream));
    for (int32_t s = 0; s < S; s++) if (cnt[s] > mcap || h_overflow[s]) return VH_ERR_CAPACITY;
    return VH_OK;
  }

  // ---- the steps after matching, pipelined -----------------------------------
  // What the reference's loop does after Matcher::matching -- removeOutliers (src/matcher.cpp:108),
  // bucketFeatures (src/viso_stereo.cpp:41-43 -> matcher.cpp:140-187), estimateMotion
  // (src/viso_stereo.cpp:49-51) -- for every stream of the group: post_begin() starts the download of the
  // step's match lists into one of two page-locked slots and returns; post_finish() runs the Delaunay
  // vote and the bucketing of a begun step on `threads` host threads (one stream per task), uploads the
  // bucketed lists (a few hundred records per stream) and runs the batched egomotion kernel on them.
  // A caller that issues step t+1 before finishing step t has the host work of t running beside the
  // GPU work of t+1.
  struct PostSlot {
    vh_p_match *h_pm = nullptr;   // page-locked [S][cap_ps]
    int32_t *h_cnt = nullptr;     // page-locked [S] (+ [S] overflow flags)
    int32_t cap_ps = 0, width = 0, method = -1;  // allocated / downloaded records per stream
    hipEvent_t ev = nullptr;
    bool pending = false;
  } post_slot[2];
  int64_t post_seq = 0;
  vh_p_match *h_bucket = nullptr, *d_bucket = nullptr;  // [S][bcap]
  int32_t *h_bcnt = nullptr, *d_bcnt = nullptr, bcap = 0;
  int32_t *d_post_rand = nullptr; size_t post_rand_n = 0;
  uint8_t *d_post_mono = nullptr; int32_t post_mono_iters = 0;
  double *d_post_xyz = nullptr, *d_post_tr = nullptr; int32_t *d_post_ok = nullptr;

  int32_t post_begin(int32_t cap_ps) {
    if (cap_ps < 1) return VH_ERR_INVALID_ARG;
    if (!allocated || last_method < 0) return VH_ERR_STATE;
    PostSlot &sl = post_slot[post_seq & 1];
    cap_ps = std::min(cap_ps, mcap);
    if (sl.cap_ps < cap_ps) {
      if (sl.h_pm) { VH_HIP(hipHostFree(sl.h_pm)); sl.h_pm = nullptr; }
      VH_HIP(hipHostMalloc((void **)&sl.h_pm, sizeof(vh_p_match) * (size_t)S * cap_ps, hipHostMallocDefault));
      sl.cap_ps = cap_ps;
    }
    if (!sl.h_cnt) VH_HIP(hipHostMalloc((void **)&sl.h_cnt, sizeof(int32_t) * 2 * (size_t)S, hipHostMallocDefault));
    if (!sl.ev) VH_HIP(hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    VH_HIP(hipStreamWaitEvent(down_stream, ev_post[last_buf], 0));
    sl.width = cap_ps;
    VH_HIP(hipMemcpy2DAsync(sl.h_pm, sizeof(vh_p_match) * (size_t)sl.cap_ps, d_matches, sizeof(vh_p_match) * (size_t)mcap,
                            sizeof(vh_p_match) * (size_t)cap_ps, (size_t)S, hipMemcpyDeviceToHost, down_stream));
    VH_HIP(hipMemcpyAsync(sl.h_cnt, d_match_count, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, down_stream));
    VH_HIP(hipMemcpyAsync(sl.h_cnt + S, d_overflow, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, down_stream));
    VH_HIP(hipEventRecord(sl.ev, down_stream));
    // the next step's emission must not overwrite the lists before they have left (as vh_group_download_matches_async)
    VH_HIP(hipEventRecord(ev_down, down_stream)); ev_down_valid = true;
    sl.pending = true; sl.method = last_method;
    post_seq++;
    return VH_OK;
  }

  // (e: the stereo estimator with rand3, or mono: the monocular one with rand8 -- at most one of them)
  int32_t post_finish(int32_t age, int32_t max_features, float bw, float bh, int32_t threads, const vh_ego_params *e, const int32_t *rand3,
                      const vh_mono_params *mono, const int32_t *rand8,
                      double *tr, int32_t *ok, int32_t *ninl, vh_p_match *out, int32_t out_cap, int32_t *out_counts, double *host_ms) {
    if (age < 0 || age > 1 || max_features < 1 || !(bw > 0) || !(bh > 0) || threads < 1 || (e && mono)) return VH_ERR_INVALID_ARG;
    if (e && (!rand3 || !tr || !ok || !ninl || e->ransac_iters < 1)) return VH_ERR_INVALID_ARG;
    if (mono && (!rand8 || !tr || !ok || !ninl || mono->ransac_iters < 1 || (int64_t)S * mono->ransac_iters > (int64_t)1 << 31)) return VH_ERR_INVALID_ARG;
    if (post_seq - 1 - age < 0) return VH_ERR_STATE;
    PostSlot &sl = post_slot[(post_seq - 1 - age) & 1];
    if (!sl.pending) return VH_ERR_STATE;
    if (e && sl.method != VH_METHOD_QUAD) return VH_ERR_STATE;       // the stereo estimator needs both cameras of both frames
    if (mono && sl.method == VH_METHOD_STEREO) return VH_ERR_STATE;  // the monocular one the left camera of both frames
    int64_t need = 0;
    { const int32_t rb = bucket_need(max_features, bw, bh, &need); if (rb) return rb; }  // (a bucket below one pixel is refused: the grid would not fit any index type)
    VH_HIP(hipEventSynchronize(sl.ev));
    sl.pending = false;
    for (int32_t s = 0; s < S; s++)
      if (sl.h_cnt[s] > sl.width || sl.h_cnt[S + s]) return VH_ERR_CAPACITY;  // a list longer than what was downloaded / a truncated feature set
    if (bcap < need) {
      for (void *old : {(void *)d_bucket, (void *)d_post_xyz}) dfree(old);  // (the superseded blocks: a caller raising max_features step by step must not pile them up)
      d_bucket = nullptr; d_post_xyz = nullptr;
      if (h_bucket) { VH_HIP(hipHostFree(h_bucket)); h_bucket = nullptr; }
      VH_HIP(hipHostMalloc((void **)&h_bucket, sizeof(vh_p_match) * (size_t)S * need, hipHostMallocDefault));
      if (!h_bcnt) VH_HIP(hipHostMalloc((void **)&h_bcnt, sizeof(int32_t) * (size_t)S, hipHostMallocDefault));
      int32_t rc;
      if ((rc = dmalloc((uint8_t **)&d_bucket, sizeof(vh_p_match) * (size_t)S * need, false))) return rc;
      if (!d_bcnt && (rc = dmalloc(&d_bcnt, (size_t)S, false))) return rc;
      if ((rc = dmalloc(&d_post_xyz, (size_t)S * need * 4, false))) return rc;
      if (!d_post_tr) { if ((rc = dmalloc(&d_post_tr, 6 * (size_t)S, false))) return rc; if ((rc = dmalloc(&d_post_ok, 2 * (size_t)S, false))) return rc; }
      bcap = (int32_t)need;
      post_mono_iters = 0;  // (the monocular scratch is sized by bcap as well)
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::atomic<int32_t> next_stream(0), failed(0);
    const bool vote = sl.method != VH_METHOD_STEREO;  // stereo records carry no previous-frame position (as remove_outliers())
    const auto work = [&]() {
      std::vector<int32_t> scratch;
      for (int32_t s = next_stream++; s < S; s = next_stream++) {
        vh_p_match *pm = sl.h_pm + (size_t)s * sl.cap_ps;
        int32_t n = sl.h_cnt[s];
        if (vote && vh_remove_outliers_pm(pm, n, &n) != VH_OK) { failed = 1; continue; }
        h_bcnt[s] = bucket_records(pm, n, max_features, bw, bh, h_bucket + (size_t)s * bcap, bcap, scratch);
      }
    };
    {
      const int32_t nw = std::max(1, std::min(threads, S));
      std::vector<std::thread> pool;
      for (int32_t w = 1; w < nw; w++) pool.emplace_back(work);
      work();
      for (auto &t : pool) t.join();
    }
    if (host_ms) *host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (failed) return VH_ERR_INVALID_ARG;
    for (int32_t s = 0; s < S; s++) if (h_bcnt[s] > bcap) return VH_ERR_CAPACITY;
    if (out_counts) for (int32_t s = 0; s < S; s++) out_counts[s] = h_bcnt[s];
    if (out) {
      for (int32_t s = 0; s < S; s++) {
        if (h_bcnt[s] > out_cap) return VH_ERR_CAPACITY;
        memcpy(out + (size_t)s * out_cap, h_bucket + (size_t)s * bcap, sizeof(vh_p_match) * (size_t)h_bcnt[s]);
      }
    }
    if (!e && !mono) return VH_OK;
    const size_t nr = e ? (size_t)S * e->ransac_iters * 3 : (size_t)S * mono->ransac_iters * 8;
    if (post_rand_n < nr) { int32_t rc = dmalloc(&d_post_rand, nr, false); if (rc) return rc; post_rand_n = nr; }
    if (mono && post_mono_iters < mono->ransac_iters) {
      int32_t rc = dmalloc(&d_post_mono, (size_t)vh_mono_scratch_bytes(S, bcap, mono->ransac_iters), false);
      if (rc) return rc;
      post_mono_iters = mono->ransac_iters;
    }
    // the bucketed lists go up as one block; everything on the download stream, beside the next step's kernels
    VH_HIP(hipMemcpyAsync(d_bucket, h_bucket, sizeof(vh_p_match) * (size_t)S * bcap, hipMemcpyHostToDevice, down_stream));
    VH_HIP(hipMemcpyAsync(d_bcnt, h_bcnt, sizeof(int32_t) * (size_t)S, hipMemcpyHostToDevice, down_stream));
    VH_HIP(hipMemcpyAsync(d_post_rand, e ? rand3 : rand8, sizeof(int32_t) * nr, hipMemcpyHostToDevice, down_stream));
    if (e) vh_launch_ego(*e, S, d_bucket, bcap, nullptr, d_bcnt, bcap, d_post_rand, d_post_xyz, bcap, d_post_tr, d_post_ok, d_post_ok + S, nullptr, 0, down_stream);
    else vh_launch_mono(*mono, S, d_bucket, bcap, nullptr, d_bcnt, bcap, d_post_rand, d_post_mono, bcap, d_post_tr, d_post_ok, d_post_ok + S, nullptr, 0, down_stream);
    VH_HIP(hipGetLastError());
    VH_HIP(hipMemcpyAsync(tr, d_post_tr, sizeof(double) * 6 * (size_t)S, hipMemcpyDeviceToHost, down_stream));
    VH_HIP(hipMemcpyAsync(ok, d_post_ok, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, down_stream));
    VH_HIP(hipMemcpyAsync(ninl, d_post_ok + S, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, down_stream));
    VH_HIP(hipStreamSynchronize(down_stream));
    static const bool timing = [] { const char *ev_ = getenv("VH_POST_TIMING"); return ev_ && ev_[0] == '1'; }();
    if (timing) fprintf(stderr, "post_finish: host %.2f ms, upload + ego + results %.2f ms\n",
                        host_ms ? *host_ms : -1.0, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - (host_ms ? *host_ms : 0.0));
    return VH_OK;
  }

  // ---- the steps after matching ON THE DEVICE ---------------------------------
  // removeOutliers -> bucketFeatures -> estimateMotion without the host: kernels_vote.hip.  The triangulation
  // under the vote is a sequential chain per list that takes tens of milliseconds as a GPU lane, so the
  // throughput comes from lists in flight: post_begin_device() moves the step's S lists into the current
  // BATCH (the matcher's buffer is free again at once); a batch of `vote_steps` steps is launched as one
  // kernel sequence over vote_steps * S lists on one of a few low-priority streams (their own hardware
  // queues: the long sweep kernel never stands in front of the matcher's kernels), and up to
  // `vote_batches` batches are in flight.  post_finish_device(age) hands out the results of the step begun
  // `age` begins ago, waiting for its batch if it has to -- a caller that stays vote_steps * (vote_batches - 1)
  // steps ahead never waits.
  struct VoteBatch {
    VhVoteBuffers vb;
    int32_t steps = 0;      // steps moved in so far
    bool launched = false;  // the kernel sequence has been queued
    bool busy = false;      // holds steps whose results have not all been handed out
    int32_t handed = 0;
    hipEvent_t ev_prep = nullptr, ev_done = nullptr;
    int32_t method = -1, max_features = 0; float bw = 0, bh = 0;
    bool has_ego = false, has_mono = false;
    vh_ego_params ego{}; vh_mono_params mono{};
    int32_t *d_rand = nullptr; size_t rand_per_step = 0;
    int32_t *h_rand = nullptr; size_t h_rand_ints = 0;  // page-locked staging of the steps' random draws (the caller's array is only borrowed)
    double *d_xyz = nullptr, *d_tr = nullptr; int32_t *d_ok = nullptr; uint8_t *d_mono = nullptr;
    uint8_t *block = nullptr;  // d_rand | d_xyz | d_tr | d_ok | d_mono
    size_t block_bytes = 0;
    // page-locked results
    double *h_tr = nullptr; int32_t *h_ok = nullptr; int32_t *h_cnt = nullptr; VhVoteMeta *h_meta = nullptr; vh_p_match *h_out = nullptr;
    int32_t h_lists = 0, h_out_cap = 0;
    bool want_lists = false;
    void free_all() {
      vb.release();
      if (block) (void)hipFree(block);
      block = nullptr; block_bytes = 0; d_rand = nullptr; d_xyz = nullptr; d_tr = nullptr; d_ok = nullptr; d_mono = nullptr;
      for (void *q : {(void *)h_tr, (void *)h_ok, (void *)h_cnt, (void *)h_meta, (void *)h_out, (void *)h_rand}) if (q) (void)hipHostFree(q);
      h_tr = nullptr; h_ok = nullptr; h_cnt = nullptr; h_meta = nullptr; h_out = nullptr; h_lists = 0; h_out_cap = 0; h_rand = nullptr; h_rand_ints = 0;
      steps = 0; launched = false; busy = false; handed = 0;
    }
  };
  static constexpr int32_t kVoteStreams = 4;
  std::vector<VoteBatch> vbatch;
  hipStream_t vote_stream[kVoteStreams] = {};
  // Measured on MI355X, KITTI, S = 256 (bench.py e2e_matchfeatures, k pairs/s; steps per batch x batches, 64 lists per wave):
  // 4x5 10.4, 8x5 11.7, 16x5 19.9, 32x3 25.3, 32x4 30.4, 48x3 30.2, 64x3 33.9; 1 list per wave, 4x5: 17.9.  A batch takes
  // 0.4-0.5 s from launch to results whatever its size (profiles/r04_vote_trace.txt): the rate is the number of steps in
  // flight over that latency, and a wave of 64 lists costs the chip 1/14 of what 64 single-list waves cost.
  int32_t vote_steps = 64, vote_batches = 3, vote_lanes = 16;
  int64_t post_dev_seq = 0;   // steps begun
  int32_t vote_cur = 0;       // batch receiving steps
  struct VoteStep { int32_t batch = -1, pos = 0; bool open = false; };
  std::vector<VoteStep> vstep;  // ring over the steps begun, indexed by sequence number

  void vote_release() {
    for (auto &b : vbatch) {
      if (b.ev_done && b.launched) (void)hipEventSynchronize(b.ev_done);
      b.free_all();
      if (b.ev_prep) (void)hipEventDestroy(b.ev_prep);
      if (b.ev_done) (void)hipEventDestroy(b.ev_done);
    }
    vbatch.clear(); vstep.clear(); post_dev_seq = 0; vote_cur = 0;
  }

  int32_t post_device_config(int32_t steps_per_batch, int32_t batches, int32_t lanes) {
    if (steps_per_batch < 1 || steps_per_batch > 256 || batches < 1 || batches > 64 || lanes < 1 || lanes > 64) return VH_ERR_INVALID_ARG;
    for (auto &b : vbatch) if (b.busy) return VH_ERR_STATE;  // steps begun whose results have not been handed out
    vote_release();
    vote_steps = steps_per_batch; vote_batches = batches; vote_lanes = lanes;
    return VH_OK;
  }

  // bucket grid of Matcher::bucketFeatures on this group's images: floor(u_max / bw) + 1 columns, floor(v_max / bh) + 1 rows (matcher.cpp:150-151)
  int32_t bucket_need(int32_t max_features, float bw, float bh, int64_t *need, int64_t *grid = nullptr) const {
    if (max_features < 1 || !(bw >= 1) || !(bh >= 1)) return VH_ERR_INVALID_ARG;
    const int64_t cols = (int64_t)floorf((float)(dims[0] - 1) / bw) + 1, rows = (int64_t)floorf((float)(dims[1] - 1) / bh) + 1;
    if (cols * rows > (1 << 20)) return VH_ERR_UNSUPPORTED;
    *need = std::min<int64_t>(cols * rows * max_features, mcap);
    if (grid) *grid = cols * rows;
    return VH_OK;
  }

  int32_t vote_launch(VoteBatch &b, int32_t index) {
    if (b.launched || b.steps == 0) return VH_OK;
    hipStream_t vs = vote_stream[index % kVoteStreams];
    VhVote v = b.vb.v;
    v.P = b.steps * S;
    VH_HIP(hipStreamWaitEvent(vs, b.ev_prep, 0));
    vh_launch_vote(v, vote_lanes, b.max_features, b.bw, b.bh, b.vb.lfsr, b.vb.lfsr_n, b.vb.out, b.vb.out_cap, b.vb.out_count, nullptr, vs);
    VH_HIP(hipGetLastError());
    if (b.has_ego) vh_launch_ego(b.ego, v.P, b.vb.out, b.vb.out_cap, nullptr, b.vb.out_count, b.vb.out_cap, b.d_rand, b.d_xyz, b.vb.out_cap, b.d_tr, b.d_ok, b.d_ok + v.P, nullptr, 0, vs);
    else if (b.has_mono) vh_launch_mono(b.mono, v.P, b.vb.out, b.vb.out_cap, nullptr, b.vb.out_count, b.vb.out_cap, b.d_rand, b.d_mono, b.vb.out_cap, b.d_tr, b.d_ok, b.d_ok + v.P, nullptr, 0, vs);
    VH_HIP(hipGetLastError());
    if (b.has_ego || b.has_mono) {
      VH_HIP(hipMemcpyAsync(b.h_tr, b.d_tr, sizeof(double) * 6 * (size_t)v.P, hipMemcpyDeviceToHost, vs));
      VH_HIP(hipMemcpyAsync(b.h_ok, b.d_ok, sizeof(int32_t) * 2 * (size_t)v.P, hipMemcpyDeviceToHost, vs));
    }
    VH_HIP(hipMemcpyAsync(b.h_cnt, b.vb.out_count, sizeof(int32_t) * (size_t)v.P, hipMemcpyDeviceToHost, vs));
    VH_HIP(hipMemcpyAsync(b.h_meta, b.vb.v.meta, sizeof(VhVoteMeta) * (size_t)v.P, hipMemcpyDeviceToHost, vs));
    if (b.want_lists) VH_HIP(hipMemcpyAsync(b.h_out, b.vb.out, sizeof(vh_p_match) * (size_t)v.P * b.vb.out_cap, hipMemcpyDeviceToHost, vs));
    VH_HIP(hipEventRecord(b.ev_done, vs));
    static const bool serial_vote = [] { const char *ev = getenv("VH_VOTE_SERIAL"); return ev && ev[0] == '1'; }();
    if (serial_vote) VH_HIP(hipStreamWaitEvent(stream, b.ev_done, 0));  // experiment: the matcher's next step waits for this batch
    b.launched = true;
    return VH_OK;
  }

  int32_t post_begin_device(int32_t cap_ps, int32_t max_features, float bw, float bh, const vh_ego_params *e, const int32_t *rand3,
                            const vh_mono_params *mono, const int32_t *rand8, int32_t want_lists) {
    if (cap_ps < 1 || (e && mono)) return VH_ERR_INVALID_ARG;
    if (e && (!rand3 || e->ransac_iters < 1)) return VH_ERR_INVALID_ARG;
    if (mono && (!rand8 || mono->ransac_iters < 1 || (int64_t)S * vote_steps * mono->ransac_iters > (int64_t)1 << 31)) return VH_ERR_INVALID_ARG;
    if ((int64_t)S * vote_steps > 65535) return VH_ERR_UNSUPPORTED;  // (the tally and the monocular kernels put the list on grid.y: fewer steps per batch)
    if (!allocated || last_method < 0) return VH_ERR_STATE;
    if (e && last_method != VH_METHOD_QUAD) return VH_ERR_STATE;        // the stereo estimator needs both cameras of both frames
    if (mono && last_method == VH_METHOD_STEREO) return VH_ERR_STATE;   // the monocular one the left camera of both frames
    int64_t need = 0, grid = 0;
    int32_t rc = bucket_need(max_features, bw, bh, &need, &grid);
    if (rc) return rc;
    cap_ps = std::min(cap_ps, mcap);
    if (cap_ps > VH_VOTE_LIST_MAX) return VH_ERR_UNSUPPORTED;  // (16-bit hull links; the sweep's angular hash has VH_VOTE_HASH_MAX slots in LDS)
    if (vbatch.empty()) {
      // The ring is allocated batch by batch on first use: it must fit the device NOW, or a later begin call -- with steps
      // already moved -- fails in hipMalloc.  Steps per batch are halved until vote_batches batches fit 80 % of the free
      // memory; if a single step per batch does not fit, nothing has moved yet and the caller is told so.
      size_t free_b = 0, total_b = 0;
      VH_HIP(hipMemGetInfo(&free_b, &total_b));
      const auto ring_bytes = [&](int32_t steps) {
        const int32_t P = steps * S;
        const size_t post = (e ? sizeof(double) * 4 * (size_t)P * (size_t)need : 0) + (mono ? (size_t)vh_mono_scratch_bytes(P, (int32_t)need, mono->ransac_iters) : 0) +
                            sizeof(double) * 6 * (size_t)P + sizeof(int32_t) * 2 * (size_t)P +
                            sizeof(int32_t) * (size_t)steps * (e ? (size_t)S * e->ransac_iters * 3 : (mono ? (size_t)S * mono->ransac_iters * 8 : 0));
        return (double)vote_batches * (double)(VhVoteBuffers::bytes_for(P, cap_ps, (int32_t)need, (int32_t)grid) + post);
      };
      int32_t steps = vote_steps;
      while (steps > 1 && ring_bytes(steps) > 0.8 * (double)free_b) steps = (steps + 1) / 2;
      if (ring_bytes(steps) > 0.8 * (double)free_b) {
        t_last_error = "the post stage's ring of batches does not fit the device's free memory even at one step per batch";
        return VH_ERR_CAPACITY;
      }
      vote_steps = steps;
      vbatch.resize((size_t)vote_batches);
      vstep.assign((size_t)vote_steps * vote_batches, VoteStep{});
      int prio_lo = 0, prio_hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
      static const int prio_env = [] { const char *ev = getenv("VH_VOTE_STREAM_PRIO"); return ev ? atoi(ev) : 1; }();  // 1: lowest, 0: normal, -1: highest
      const int prio = prio_env > 0 ? prio_lo : (prio_env < 0 ? prio_hi : 0);
      for (int k = 0; k < kVoteStreams; k++)
        if (!vote_stream[k]) VH_HIP(hipStreamCreateWithPriority(&vote_stream[k], hipStreamNonBlocking, prio));
      for (auto &b : vbatch) {
        VH_HIP(hipEventCreateWithFlags(&b.ev_prep, hipEventDisableTiming));
        VH_HIP(hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming));
      }
    }
    VoteBatch *b = &vbatch[(size_t)vote_cur];
    const size_t rand_per_step = e ? (size_t)S * e->ransac_iters * 3 : (mono ? (size_t)S * mono->ransac_iters * 8 : 0);
    const auto same = [&](const VoteBatch &q) {
      return q.method == last_method && q.max_features == max_features && q.bw == bw && q.bh == bh && q.has_ego == (e != nullptr) &&
             q.has_mono == (mono != nullptr) && (!e || memcmp(&q.ego, e, sizeof(*e)) == 0) && (!mono || memcmp(&q.mono, mono, sizeof(*mono)) == 0) &&
             q.vb.v.cap >= cap_ps && q.want_lists == (want_lists != 0);
    };
    if (b->steps > 0 && (b->launched || b->steps >= vote_steps || !same(*b))) {  // the batch is closed (full, flushed, or configured differently): next one
      if ((rc = vote_launch(*b, vote_cur))) return rc;
      const int32_t next = (vote_cur + 1) % vote_batches;
      // the ring has come round: the next batch's results must have been handed out (checked before anything moves, so
      // that the caller can finish those steps and begin this one again)
      if (vbatch[(size_t)next].launched && vbatch[(size_t)next].busy && vbatch[(size_t)next].handed < vbatch[(size_t)next].steps) return VH_ERR_STATE;
      vote_cur = next;
      b = &vbatch[(size_t)vote_cur];
    }
    if (b->steps == 0 || b->launched) {  // start the batch
      if (b->launched) {  // a batch of the previous round: its kernels must be done
        if (b->busy && b->handed < b->steps) return VH_ERR_STATE;
        VH_HIP(hipEventSynchronize(b->ev_done));
      }
      b->steps = 0; b->launched = false; b->busy = false; b->handed = 0;
      const int32_t P = vote_steps * S;
      if (b->vb.v.cap < cap_ps || b->vb.out_cap < need || b->vb.v.P < P || b->vb.v.nb_max < grid) {
        if (b->vb.block) {  // a batch grows (longer lists than the ring was sized for): only if the difference fits
          size_t free_b = 0, total_b = 0;
          VH_HIP(hipMemGetInfo(&free_b, &total_b));
          const size_t want = VhVoteBuffers::bytes_for(P, cap_ps, (int32_t)need, (int32_t)grid);
          if (want > b->vb.bytes && want - b->vb.bytes > free_b) { t_last_error = "the post stage's batch cannot grow: device memory exhausted"; return VH_ERR_CAPACITY; }
        }
        b->vb.release();
        VH_HIP(b->vb.alloc(P, cap_ps, (int32_t)need, (int32_t)grid));
        VH_HIP(b->vb.upload_lfsr());
      }
      const int32_t ocap = b->vb.out_cap;
      const auto up = [](size_t x) { return (x + 255) / 256 * 256; };
      const size_t b_rand = up(sizeof(int32_t) * rand_per_step * vote_steps), b_xyz = up(e ? sizeof(double) * 4 * (size_t)P * ocap : 0), b_tr = up(sizeof(double) * 6 * (size_t)P),
                   b_ok = up(sizeof(int32_t) * 2 * (size_t)P), b_mono = mono ? (size_t)vh_mono_scratch_bytes(P, ocap, mono->ransac_iters) : 0;
      if (b->block_bytes < b_rand + b_xyz + b_tr + b_ok + b_mono || b->rand_per_step != rand_per_step) {
        if (b->block) (void)hipFree(b->block);
        b->block = nullptr; b->block_bytes = 0;
        VH_HIP(hipMalloc((void **)&b->block, b_rand + b_xyz + b_tr + b_ok + b_mono + 256));
        b->block_bytes = b_rand + b_xyz + b_tr + b_ok + b_mono;
      }
      b->d_rand = (int32_t *)b->block; b->d_xyz = (double *)(b->block + b_rand); b->d_tr = (double *)(b->block + b_rand + b_xyz);
      b->d_ok = (int32_t *)(b->block + b_rand + b_xyz + b_tr); b->d_mono = b->block + b_rand + b_xyz + b_tr + b_ok;
      b->rand_per_step = rand_per_step;
      if (b->h_rand_ints < rand_per_step * (size_t)vote_steps) {
        if (b->h_rand) (void)hipHostFree(b->h_rand);
        b->h_rand = nullptr; b->h_rand_ints = 0;
        VH_HIP(hipHostMalloc((void **)&b->h_rand, sizeof(int32_t) * rand_per_step * (size_t)vote_steps, hipHostMallocDefault));
        b->h_rand_ints = rand_per_step * (size_t)vote_steps;
      }
      if (b->h_lists < P) {
        for (void *q : {(void *)b->h_tr, (void *)b->h_ok, (void *)b->h_cnt, (void *)b->h_meta}) if (q) (void)hipHostFree(q);
        b->h_tr = nullptr; b->h_ok = nullptr; b->h_cnt = nullptr; b->h_meta = nullptr;
        VH_HIP(hipHostMalloc((void **)&b->h_tr, sizeof(double) * 6 * (size_t)P, hipHostMallocDefault));
        VH_HIP(hipHostMalloc((void **)&b->h_ok, sizeof(int32_t) * 2 * (size_t)P, hipHostMallocDefault));
        VH_HIP(hipHostMalloc((void **)&b->h_cnt, sizeof(int32_t) * (size_t)P, hipHostMallocDefault));
        VH_HIP(hipHostMalloc((void **)&b->h_meta, sizeof(VhVoteMeta) * (size_t)P, hipHostMallocDefault));
        b->h_lists = P;
      }
      if (want_lists && (!b->h_out || b->h_out_cap < ocap)) {
        if (b->h_out) (void)hipHostFree(b->h_out);
        b->h_out = nullptr;
        VH_HIP(hipHostMalloc((void **)&b->h_out, sizeof(vh_p_match) * (size_t)P * ocap, hipHostMallocDefault));
        b->h_out_cap = ocap;
      }
      b->method = last_method; b->max_features = max_features; b->bw = bw; b->bh = bh; b->has_ego = e != nullptr; b->has_mono = mono != nullptr;
      if (e) b->ego = *e;
      if (mono) b->mono = *mono;
      b->want_lists = want_lists != 0;
    }
    // the step's lists leave the matcher's buffer behind the emission that wrote them; the next emission waits for that (ev_down)
    VH_HIP(hipStreamWaitEvent(down_stream, ev_post[last_buf], 0));
    vh_launch_vote_prep(b->vb.v, b->steps * S, S, (const vh_p_match *)d_matches, mcap, d_match_count, mcap, d_overflow, last_method != VH_METHOD_STEREO ? 1 : 0, down_stream);
    VH_HIP(hipGetLastError());
    if (rand_per_step) {
      // through the batch's page-locked slot of this step: an asynchronous copy from the caller's pageable array would
      // make the host wait until the stream reaches it (behind the step's emission), and the array is only borrowed
      int32_t *hr = b->h_rand + rand_per_step * (size_t)b->steps;
      memcpy(hr, e ? rand3 : rand8, sizeof(int32_t) * rand_per_step);
      VH_HIP(hipMemcpyAsync(b->d_rand + rand_per_step * (size_t)b->steps, hr, sizeof(int32_t) * rand_per_step, hipMemcpyHostToDevice, down_stream));
    }
    VH_HIP(hipEventRecord(b->ev_prep, down_stream));
    VH_HIP(hipEventRecord(ev_down, down_stream)); ev_down_valid = true;
    VoteStep &st = vstep[(size_t)(post_dev_seq % (int64_t)vstep.size())];
    st.batch = vote_cur; st.pos = b->steps; st.open = true;
    b->steps++; b->busy = true;
    post_dev_seq++;
    if (b->steps >= vote_steps) return vote_launch(*b, vote_cur);
    return VH_OK;
  }

  int32_t post_finish_device(int32_t age, double *tr, int32_t *ok, int32_t *ninl, vh_p_match *out, int32_t out_cap, int32_t *out_counts) {
    if (age < 0 || (out && out_cap < 1)) return VH_ERR_INVALID_ARG;
    if (vstep.empty() || post_dev_seq - 1 - age < 0 || age >= (int64_t)vstep.size()) return VH_ERR_STATE;
    VoteStep &st = vstep[(size_t)((post_dev_seq - 1 - age) % (int64_t)vstep.size())];
    if (!st.open) return VH_ERR_STATE;
    VoteBatch &b = vbatch[(size_t)st.batch];
    int32_t rc = vote_launch(b, st.batch);  // (a batch that is not full yet is closed and launched now)
    if (rc) return rc;
    VH_HIP(hipEventSynchronize(b.ev_done));
    st.open = false;
    b.handed++;
    if (b.handed >= b.steps) b.busy = false;
    const size_t p0 = (size_t)st.pos * S, P = (size_t)b.steps * S;
    if ((b.has_ego || b.has_mono) && (!tr || !ok || !ninl)) return VH_ERR_INVALID_ARG;
    if (out && !b.want_lists) return VH_ERR_STATE;
    if (b.has_ego || b.has_mono) {
      memcpy(tr, b.h_tr + 6 * p0, sizeof(double) * 6 * (size_t)S);
      memcpy(ok, b.h_ok + p0, sizeof(int32_t) * (size_t)S);
      memcpy(ninl, b.h_ok + P + p0, sizeof(int32_t) * (size_t)S);
    }
    if (out_counts) memcpy(out_counts, b.h_cnt + p0, sizeof(int32_t) * (size_t)S);
    // One refused list does not void the step: the healthy streams are delivered, a refused stream reports
    // ok = 0, n_inliers = 0, tr = 0, counts = -1, and the call returns the error (capacity before unsupported).
    int32_t ret = VH_OK;
    for (int32_t s = 0; s < S; s++) {
      const VhVoteMeta &m = b.h_meta[p0 + s];
      const bool bad = m.status != VH_VOTE_OK && m.status != VH_VOTE_SKIP;
      if (m.status == VH_VOTE_TRUNCATED) ret = VH_ERR_CAPACITY;
      else if (bad && ret == VH_OK) ret = VH_ERR_UNSUPPORTED;
      if (bad) {
        if (b.has_ego || b.has_mono) { for (int k = 0; k < 6; k++) tr[6 * (size_t)s + k] = 0.0; ok[s] = 0; ninl[s] = 0; }
        if (out_counts) out_counts[s] = -1;
        continue;
      }
      if (out) {
        const int32_t k = b.h_cnt[p0 + s];
        if (k > out_cap) { ret = VH_ERR_CAPACITY; if (out_counts) out_counts[s] = -1; continue; }
        memcpy(out + (size_t)s * out_cap, b.h_out + (p0 + s) * (size_t)b.vb.out_cap, sizeof(vh_p_match) * (size_t)k);
      }
    }
    return ret;
  }

  // Load caller-supplied feature records into a role's set and index it.
  int32_t load_features(int32_t role, const int32_t *m, int32_t n) {
    if (n < 0 || (n > 0 && !m)) return VH_ERR_INVALID_ARG;
    if (n > cap) return VH_ERR_CAPACITY;
    for (int32_t i = 0; i < n; i++) {
      const int32_t *f = m + 12 * (size_t)i;
      if (f[0] < 0 || f[0] >= dims[0] || f[1] < 0 || f[1] >= dims[1] || f[3] < 0 || f[3] > 3) return VH_ERR_INVALID_ARG;
    }
    const int32_t set = vh_role_set(S, pairs(), 0, role);
    const int32_t slot = (role >= 2) ? pair_cur : pair_prev;
    if (ev_read_valid[slot]) VH_HIP(hipStreamWaitEvent(stream, ev_read[slot], 0));
    { int32_t rz = zero_bin_counters(set, 1); if (rz) return rz; }  // also clears the count, set right below
    if (n) VH_HIP(hipMemcpyAsync(sets.feat + (size_t)set * cap * 12, m, sizeof(int32_t) * 12 * (size_t)n, hipMemcpyHostToDevice, stream));
    VH_HIP(hipMemcpyAsync(sets.count + set, &n, sizeof(int32_t), hipMemcpyHostToDevice, stream));
    VH_HIP(hipStreamSynchronize(stream));
    int32_t rc = bin_sets(set, 1, false);
    if (rc) return rc;
    VH_HIP(hipEventRecord(ev_det[slot], stream));
    return VH_OK;
  }
};

uint32_t lfsr_next(uint32_t x) { return vh_lfsr_next(x); }  // (vh_vote.h: shared with the device form of the shuffle)

// Matcher::bucketFeatures (matcher.cpp:140-187) without the fixed
// buckets[126][256] capacity.
void bucket_host(std::vector<vh_p_match> &pm, int32_t max_features, float bw, float bh) {
  float u_max = 0, v_max = 0;
  for (auto &m : pm) { if (m.u1c > u_max) u_max = m.u1c; if (m.v1c > v_max) v_max = m.v1c; }
  const int32_t cols = (int32_t)floorf(u_max / bw) + 1, rows = (int32_t)floorf(v_max / bh) + 1;
  std::vector<std::vector<vh_p_match>> buckets((size_t)cols * rows);
  for (auto &m : pm) {
    const int32_t u = (int32_t)floorf(m.u1c / bw), v = (int32_t)floorf(m.v1c / bh);
    buckets[(size_t)v * cols + u].push_back(m);
  }
  pm.clear();
  uint32_t rnd = 5;
  for (auto &b : buckets) {
    const int32_t len = (int32_t)b.size();
    for (int32_t i = 1; i < len; i++) {  // random_shuffle, matcher.cpp:126-138
      const int32_t j = (int32_t)(rnd % (uint32_t)(i + 1));
      rnd = lfsr_next(rnd);
      std::swap(b[i], b[j]);
    }
    for (int32_t j = 0, k = 0; j < len; j++) { pm.push_back(b[j]); if (++k >= max_features) break; }
  }
}

int32_t bucket_records(const vh_p_match *pm, int32_t n, int32_t max_features, float bw, float bh, vh_p_match *out, int32_t out_cap,
                       std::vector<int32_t> &work) {
  float u_max = 0, v_max = 0;
  for (int32_t i = 0; i < n; i++) { if (pm[i].u1c > u_max) u_max = pm[i].u1c; if (pm[i].v1c > v_max) v_max = pm[i].v1c; }
  const int32_t cols = (int32_t)floorf(u_max / bw) + 1, rows = (int32_t)floorf(v_max / bh) + 1, nb = cols * rows;
  // counting sort of the record indices by bucket (row-major), stable: the reference appends in list order
  work.assign((size_t)nb + 1 + (size_t)n, 0);
  int32_t *start = work.data(), *idx = work.data() + nb + 1;
  const auto bucket_of = [&](const vh_p_match &m) { return (int32_t)floorf(m.v1c / bh) * cols + (int32_t)floorf(m.u1c / bw); };
  for (int32_t i = 0; i < n; i++) start[bucket_of(pm[i]) + 1]++;
  for (int32_t b = 0; b < nb; b++) start[b + 1] += start[b];
  {
    std::vector<int32_t> cur(start, start + nb);
    for (int32_t i = 0; i < n; i++) idx[cur[bucket_of(pm[i])]++] = i;
  }
  uint32_t rnd = 5;
  int32_t kept = 0;
  for (int32_t b = 0; b < nb; b++) {
    int32_t *v = idx + start[b];
    const int32_t len = start[b + 1] - start[b];
    for (int32_t i = 1; i < len; i++) {  // random_shuffle, matcher.cpp:126-138
      const int32_t j = (int32_t)(rnd % (uint32_t)(i + 1));
      rnd = lfsr_next(rnd);
      std::swap(v[i], v[j]);
    }
    for (int32_t j = 0, k = 0; j < len; j++) { if (kept < out_cap) out[kept] = pm[v[j]]; kept++; if (++k >= max_features) break; }
  }
  return kept;
}

int32_t check_params(const vh_params *p) {
  if (!p) return VH_ERR_INVALID_ARG;
  if (p->nms_n < 1 || p->nms_n > 32 || p->match_binsize < 1 || p->match_radius < 0 || p->match_disp_tolerance < 0 ||
      p->match_radius > 16384 || p->match_disp_tolerance > 16384 || p->nms_tau < 0)  // the accept test packs 2*tolerance into 16 bits
    return VH_ERR_UNSUPPORTED;
  return VH_OK;
}

int32_t select_device(int32_t device) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) { t_last_error = "no HIP device visible"; return VH_ERR_NO_DEVICE; }
  if (device < 0 || device >= cnt) return VH_ERR_INVALID_ARG;
  VH_HIP(hipSetDevice(device));
  return VH_OK;
}

int32_t group_new(const vh_params *p, int32_t device, int32_t S, int32_t mf, int32_t mm, Group **out) {
  if (!out) return VH_ERR_INVALID_ARG;
  *out = nullptr;
  int32_t rc = check_params(p);
  if (rc) return rc;
  if (S < 1 || S > 65535 / 4 || mf < 0 || mm < 0) return VH_ERR_INVALID_ARG;
  if ((rc = select_device(device))) return rc;
  Group *gq = new Group();
  gq->p = *p; gq->device = device; gq->S = S; gq->req_features = mf; gq->req_matches = mm;
  // Both streams at the default priority: raising the detect stream's priority
  // (so that detection finishes inside the shadow of the flow search) was
  // measured and lost ~3 % -- the single-workgroup-per-set kernels then wait for
  // the starved low-priority stream instead (profiles/, round 1).
  int prio_lo = 0, prio_hi = 0;
  if (getenv("VH_DET_PRIORITY")) (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (hipStreamCreateWithPriority(&gq->own_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) {
    t_last_error = "hipStreamCreateWithFlags failed";
    gq->own_stream = nullptr; delete gq;
    return VH_ERR_HIP;
  }
  gq->stream = gq->own_stream;
  // VH_SERIAL=1 (profiling aid): run matching on the detect stream, i.e. no
  // overlap, so that per-kernel timings are exclusive
  // A small group (one or two cameras: the drop-in Matcher) also runs on one stream: it fills a few percent of
  // the chip, detection of frame t+1 has nothing to hide behind, and every hand-over between streams is a
  // ~14 us bubble on the path pushBack -> matchFeatures -> getMatches (VH_SERIAL=0 keeps the three streams).
  const char *serial = getenv("VH_SERIAL");
  bool ok = true;
  if (serial ? serial[0] == '1' : S <= 2) gq->match_stream = gq->post_stream = gq->own_stream, gq->serial = true;
  else {
    ok = hipStreamCreateWithPriority(&gq->match_stream, hipStreamNonBlocking, prio_lo) == hipSuccess;
    // The chain/emission step runs on a stream of its own so that consecutive searches run back
    // to back and the two short, latency-bound kernels hide beside the next search: +2.8 % on
    // MI355X (KITTI, S = 256).  (Round 1 measured -17 % for the same switch, when the searches
    // did not yet fill the chip's issue slots.)  VH_POST_STREAM=0: same stream as the search.
    const char *pse = getenv("VH_POST_STREAM");
    if (ok && !(pse && pse[0] == '0')) ok = hipStreamCreateWithPriority(&gq->post_stream, hipStreamNonBlocking, prio_lo) == hipSuccess, gq->own_post = true;
    else gq->post_stream = gq->match_stream;
  }
  for (int k = 0; k < 2 && ok; k++)
    ok = hipEventCreateWithFlags(&gq->ev_tables[k], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&gq->ev_post[k], hipEventDisableTiming) == hipSuccess;
  for (int k = 0; k < VH_RING && ok; k++)
    ok = hipEventCreateWithFlags(&gq->ev_det[k], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&gq->ev_read[k], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&gq->ev_user, hipEventDisableTiming) == hipSuccess;
  // VH_FLOW_TESTED=1 / =0: always the tested / always the speculative loops (default: adaptive)
  if (const char *ft = getenv("VH_FLOW_TESTED")) gq->force_mode = atoi(ft) ? 0 : 1;
  for (int k = 0; k < 2 && ok; k++) ok = hipEventCreateWithFlags(&gq->ev_stage[k], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&gq->copy_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&gq->down_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&gq->ev_down, hipEventDisableTiming) == hipSuccess;
  if (!ok) { t_last_error = "stream/event creation failed"; delete gq; return VH_ERR_HIP; }
  *out = gq;
  return VH_OK;
}

#define ENTER(gq)                                   \
  if (!(gq)) return VH_ERR_INVALID_ARG;             \
  { hipError_t e_ = hipSetDevice((gq)->device);     \
    if (e_ != hipSuccess) { t_last_error = hipGetErrorString(e_); return VH_ERR_HIP; } }

struct Temp {  // transient one-stream group for the stateless entry points
  Group *gq = nullptr;
  ~Temp() { if (gq) { (void)gq->sync_all(); delete gq; } }
};

}  // namespace

// vh_group / vh_matcher are opaque aliases of Group (a matcher is a group of one stream).

extern "C" {

int32_t vh_abi_version(void) { return VH_ABI_VERSION; }

int32_t vh_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return VH_ERR_NO_DEVICE;
  return cnt;
}

const char *vh_error_string(int32_t code) {
  switch (code) {
    case VH_OK: return "ok";
    case VH_ERR_INVALID_ARG: return "invalid argument (image dimension mismatch / null pointer)";
    case VH_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
    case VH_ERR_HIP: return "HIP runtime error";
    case VH_ERR_CAPACITY: return "capacity exceeded";
    case VH_ERR_UNSUPPORTED: return "parameter outside the supported envelope";
    case VH_ERR_STATE: return "call sequence error";
    default: return "unknown error";
  }
}

const char *vh_last_error(void) { return t_last_error.c_str(); }

void vh_default_params(vh_params *p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->nms_n = 2; p->nms_tau = 50; p->match_binsize = 50; p->match_radius = 200;
  p->match_disp_tolerance = 2; p->outlier_disp_tolerance = 5; p->outlier_flow_tolerance = 5;
}

// ---- group -----------------------------------------------------------------
int32_t vh_group_create(const vh_params *p, int32_t device, int32_t n_streams, int32_t max_features,
                        int32_t max_matches, vh_group **out) {
  return group_new(p, device, n_streams, max_features, max_matches, (Group **)out);
}
void vh_group_destroy(vh_group *g) {
  if (!g) return;
  Group *gq = (Group *)g;
  (void)hipSetDevice(gq->device);
  (void)gq->sync_all();
  gq->prof_collect();
  delete gq;
}
int32_t vh_group_streams(const vh_group *g) { return g ? ((const Group *)g)->S : VH_ERR_INVALID_ARG; }
int64_t vh_group_device_bytes(const vh_group *g) {  // (the matcher's arrays and, once begun, the post stage's ring of batches)
  if (!g) return (int64_t)VH_ERR_INVALID_ARG;
  const Group *gq = (const Group *)g;
  int64_t b = (int64_t)gq->device_bytes;
  for (const auto &vb : gq->vbatch) b += (int64_t)vb.vb.bytes + (int64_t)vb.block_bytes;
  return b;
}
int32_t vh_group_push_back_device(vh_group *g, const void *dI1, const void *dI2, int64_t stride_bytes,
                                  const int32_t dims[3], int32_t replace) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->push_device(dI1, dI2, stride_bytes, dims, replace);
}
int32_t vh_group_push_back(vh_group *g, const uint8_t *I1, const uint8_t *I2, int64_t stride_bytes,
                           const int32_t dims[3], int32_t replace) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->push_host(I1, I2, stride_bytes, dims, replace);
}
int32_t vh_group_match_features(vh_group *g, int32_t method) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->match(method);
}
int32_t vh_group_match_features_prior(vh_group *g, int32_t method, const double *Tr_delta16) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->match(method, Tr_delta16);
}
int32_t vh_group_get_matches(vh_group *g, int32_t stream, vh_p_match *out, int32_t cap, int32_t *n) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->get_matches(stream, out, cap, n);
}
int32_t vh_group_get_matches_all(vh_group *g, vh_p_match *out, int32_t cap_per_stream, int32_t *counts) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->get_matches_all(out, cap_per_stream, counts);
}
int32_t vh_group_download_matches_async(vh_group *g, vh_p_match *out, int32_t cap_per_stream, int32_t *counts) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->download_async(out, cap_per_stream, counts);
}
int32_t vh_group_wait_download(vh_group *g) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->wait_download();
}
int32_t vh_group_get_features(vh_group *g, int32_t stream, int32_t which, int32_t *out12, int32_t cap, int32_t *n) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->get_features(stream, which, out12, cap, n);
}
int32_t vh_group_get_counts(vh_group *g, int32_t *n_features, int32_t *n_matches) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->get_counts(n_features, n_matches);
}
int32_t vh_group_synchronize(vh_group *g) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->sync_all();
}
int32_t vh_group_set_stream(vh_group *g, void *hip_stream) {
  Group *gq = (Group *)g; ENTER(gq);
  int32_t rc = gq->sync_all();
  if (rc) return rc;
  gq->user_stream = (hipStream_t)hip_stream;  // handle 0 is the legacy default stream, a stream like any other
  gq->user_stream_set = true;
  return VH_OK;
}
int32_t vh_group_clear_stream(vh_group *g) {
  Group *gq = (Group *)g; ENTER(gq);
  int32_t rc = gq->sync_all();
  if (rc) return rc;
  gq->user_stream = nullptr; gq->user_stream_set = false;
  return VH_OK;
}
int32_t vh_group_stream_wait_images(vh_group *g, void *hip_stream) {
  Group *gq = (Group *)g; ENTER(gq);
  if (!gq->allocated) return VH_OK;  // nothing pushed yet: nothing reads any image
  // the detection (and indexing) of the last pushed frame is the last reader of its images
  VH_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, gq->ev_det[gq->pair_cur], 0));
  return VH_OK;
}
int32_t vh_group_search_stats(vh_group *g, int32_t *speculative, double *research_rate) {
  Group *gq = (Group *)g; ENTER(gq);
  if (speculative) *speculative = gq->force_mode >= 0 ? gq->force_mode : (gq->spec_mode ? 1 : 0);
  if (research_rate) *research_rate = gq->last_redo_rate;
  return VH_OK;
}
int32_t vh_group_debug_fail_next_alloc(vh_group *g) {
  Group *gq = (Group *)g; ENTER(gq);
  gq->fail_next_alloc = true;
  return VH_OK;
}
int32_t vh_group_profile_enable(vh_group *g, int32_t on) {
  Group *gq = (Group *)g; ENTER(gq);
  gq->prof = on != 0;
  return VH_OK;
}
int32_t vh_group_profile_read(vh_group *g, const char *name, double *ms, int64_t *launches) {
  Group *gq = (Group *)g; ENTER(gq);
  if (!name) return VH_ERR_INVALID_ARG;
  gq->prof_collect();
  auto it = gq->prof_entries.find(name);
  if (ms) *ms = it == gq->prof_entries.end() ? 0.0 : it->second.ms;
  if (launches) *launches = it == gq->prof_entries.end() ? 0 : it->second.launches;
  return VH_OK;
}
int32_t vh_group_profile_reset(vh_group *g) {
  Group *gq = (Group *)g; ENTER(gq);
  gq->prof_collect();
  gq->prof_entries.clear();
  return VH_OK;
}

// ---- one stream ------------------------------------------------------------
int32_t vh_create_ex(const vh_params *p, int32_t device, int32_t max_features, int32_t max_matches,
                     vh_matcher **out) {
  return group_new(p, device, 1, max_features, max_matches, (Group **)out);
}
int32_t vh_create(const vh_params *p, int32_t device, vh_matcher **out) { return vh_create_ex(p, device, 0, 0, out); }
void vh_destroy(vh_matcher *m) { vh_group_destroy((vh_group *)m); }
int32_t vh_set_intrinsics(vh_matcher *m, double f, double cu, double cv, double base) {
  if (!m) return VH_ERR_INVALID_ARG;
  Group *gq = (Group *)m;
  gq->p.f = f; gq->p.cu = cu; gq->p.cv = cv; gq->p.base = base;
  return VH_OK;
}
int32_t vh_push_back(vh_matcher *m, const uint8_t *I1, const uint8_t *I2, const int32_t dims[3], int32_t replace) {
  Group *gq = (Group *)m; ENTER(gq);
  return gq->push_host(I1, I2, 0, dims, replace);
}
int32_t vh_push_back_device(vh_matcher *m, const void *dI1, const void *dI2, const int32_t dims[3], int32_t replace) {
  Group *gq = (Group *)m; ENTER(gq);
  return gq->push_device(dI1, dI2, 0, dims, replace);
}
int32_t vh_match_features(vh_matcher *m, int32_t method, const double *Tr_delta16) {
  Group *gq = (Group *)m; ENTER(gq);
  return gq->match(method, Tr_delta16);  // (null: as the reference's Matcher::matchFeatures, which ignores its Tr_delta, matcher.cpp:93-111)
}
int32_t vh_bucket_features(vh_matcher *m, int32_t max_features, float bucket_width, float bucket_height) {
  Group *gq = (Group *)m; ENTER(gq);
  if (max_features < 1 || !(bucket_width > 0) || !(bucket_height > 0)) return VH_ERR_INVALID_ARG;
  // (a bucket grid beyond 2^24 cells -- bucket sides of a fraction of a pixel -- would overflow the reference's int arithmetic too)
  if (gq->allocated && ((double)gq->dims[0] / bucket_width + 1) * ((double)gq->dims[1] / bucket_height + 1) > (double)(1 << 24)) return VH_ERR_INVALID_ARG;
  const int32_t rc = gq->fetch_matches(0);
  if (rc) return rc == VH_ERR_STATE ? VH_OK : rc;  // nothing matched yet: nothing to bucket
  bucket_host(gq->host_matches[0], max_features, bucket_width, bucket_height);
  return VH_OK;
}
int32_t vh_remove_outliers(vh_matcher *m) {
  Group *gq = (Group *)m; ENTER(gq);
  return gq->remove_outliers(0, 1, 1);
}
int32_t vh_group_remove_outliers(vh_group *g, int32_t host_threads) {
  Group *gq = (Group *)g; ENTER(gq);
  if (host_threads < 1) host_threads = (int32_t)std::max(1u, std::thread::hardware_concurrency());
  return gq->remove_outliers(0, gq->S, host_threads);
}
int32_t vh_get_matches(vh_matcher *m, vh_p_match *out, int32_t cap, int32_t *n) {
  Group *gq = (Group *)m; ENTER(gq);
  return gq->get_matches(0, out, cap, n);
}
int32_t vh_get_features(vh_matcher *m, int32_t which, int32_t *out12, int32_t cap, int32_t *n) {
  Group *gq = (Group *)m; ENTER(gq);
  return gq->get_features(0, which, out12, cap, n);
}
int32_t vh_host_alloc(int32_t device, size_t bytes, void **out) {
  if (!out || bytes == 0) return VH_ERR_INVALID_ARG;
  *out = nullptr;
  const int32_t rc = select_device(device);
  if (rc) return rc;
  VH_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return VH_OK;
}
int32_t vh_host_free(void *ptr) {
  if (!ptr) return VH_OK;
  VH_HIP(hipHostFree(ptr));
  return VH_OK;
}
int32_t vh_synchronize(vh_matcher *m) { return vh_group_synchronize((vh_group *)m); }
int32_t vh_set_stream(vh_matcher *m, void *hip_stream) { return vh_group_set_stream((vh_group *)m, hip_stream); }
int32_t vh_clear_stream(vh_matcher *m) { return vh_group_clear_stream((vh_group *)m); }
int32_t vh_stream_wait_images(vh_matcher *m, void *hip_stream) { return vh_group_stream_wait_images((vh_group *)m, hip_stream); }

#ifdef VH_DEBUG_ROWS
// debug build only: the (class, v) row index of one feature set
int32_t vh_debug_rows(vh_matcher *m, int32_t which, int32_t *row_start, int32_t *r_pos, int32_t *bin_start) {
  Group *gq = (Group *)m; ENTER(gq);
  const int32_t set = vh_role_set(gq->S, gq->pairs(), 0, which);
  const size_t nrow = 4 * (size_t)gq->dims[1];
  VH_HIP(hipDeviceSynchronize());
  VH_HIP(hipMemcpy(row_start, gq->sets.row_start + (size_t)set * (nrow + 1), sizeof(int32_t) * (nrow + 1), hipMemcpyDeviceToHost));
  VH_HIP(hipMemcpy(r_pos, gq->sets.r_pos + (size_t)set * gq->cap, sizeof(int32_t) * gq->cap, hipMemcpyDeviceToHost));
  VH_HIP(hipMemcpy(bin_start, gq->sets.bin_start + (size_t)set * (gq->sets.nbins + 1), sizeof(int32_t) * (gq->sets.nbins + 1), hipMemcpyDeviceToHost));
  return gq->cap;
}
#endif

// ---- stereo egomotion ------------------------------------------
void vh_default_ego_params(vh_ego_params *e) {
  if (!e) return;
  memset(e, 0, sizeof(*e));
  e->ransac_iters = 200; e->reweighting = 1; e->inlier_threshold = 2.0;  // src/viso_stereo.h:39-41
  e->f = 1; e->cu = 0; e->cv = 0; e->base = 1;                            // src/viso.h:46-48, src/viso_stereo.h:38
}
int32_t vh_group_estimate_motion(vh_group *g, const vh_ego_params *e, const int32_t *rand3, double *tr, int32_t *ok,
                                 int32_t *n_inliers) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->estimate_motion(e, rand3, tr, ok, n_inliers);
}
// Device work buffer of the stateless estimators: one per device, grow-only, kept between calls (an allocation and its
// release cost more than the kernels of a bucketed batch).  Requests above 1 GiB are not kept.  The lock is held for the
// whole call: stateless estimates on one device run one at a time.
struct EgoWork {
  std::mutex mu;
  uint8_t *buf[16] = {};
  size_t bytes[16] = {};
};
static EgoWork g_ego_work;
struct EgoWorkLease {
  std::unique_lock<std::mutex> lock;
  uint8_t *d = nullptr;
  bool kept = false;
  hipError_t take(int32_t device, size_t need) {
    lock = std::unique_lock<std::mutex>(g_ego_work.mu);
    if (device >= 0 && device < 16 && need <= ((size_t)1 << 30)) {
      kept = true;
      if (g_ego_work.bytes[device] < need) {
        if (g_ego_work.buf[device]) (void)hipFree(g_ego_work.buf[device]);
        g_ego_work.buf[device] = nullptr; g_ego_work.bytes[device] = 0;
        const size_t want = need + need / 4;
        const hipError_t er = hipMalloc((void **)&g_ego_work.buf[device], want);
        if (er != hipSuccess) return er;
        g_ego_work.bytes[device] = want;
      }
      d = g_ego_work.buf[device];
      return hipSuccess;
    }
    return hipMalloc((void **)&d, need);
  }
  ~EgoWorkLease() { if (d && !kept) (void)hipFree(d); }
};

int32_t vh_estimate_motion_stereo(const vh_ego_params *e, int32_t device, int32_t n_sets, const vh_p_match *pm,
                                  const int32_t *offsets, const int32_t *rand3, double *tr, int32_t *ok,
                                  int32_t *n_inliers, int32_t *inliers) {
  if (!e || n_sets < 1 || !offsets || !rand3 || !tr || !ok || !n_inliers || e->ransac_iters < 1) return VH_ERR_INVALID_ARG;
  int64_t nmax = 0;
  if (offsets[0] < 0) return VH_ERR_INVALID_ARG;
  for (int32_t s = 0; s < n_sets; s++) {
    if (offsets[s + 1] < offsets[s]) return VH_ERR_INVALID_ARG;
    nmax = std::max<int64_t>(nmax, offsets[s + 1] - offsets[s]);
  }
  const int64_t total = offsets[n_sets];
  if (total > 0 && !pm) return VH_ERR_INVALID_ARG;
  const int32_t rc = select_device(device);
  if (rc) return rc;
  const size_t nr = (size_t)n_sets * e->ransac_iters * 3;
  uint8_t *d = nullptr;
  // one allocation: matches | offsets | rand3 | ok,ninl | inliers | tr | xyz+flags
  const size_t b_pm = sizeof(vh_p_match) * (size_t)std::max<int64_t>(total, 1), b_off = sizeof(int32_t) * ((size_t)n_sets + 1);
  const size_t b_r = sizeof(int32_t) * nr, b_ok = sizeof(int32_t) * 2 * (size_t)n_sets, b_inl = sizeof(int32_t) * (size_t)std::max<int64_t>(total, 1);
  const size_t b_tr = sizeof(double) * 6 * (size_t)n_sets, b_xyz = sizeof(double) * 4 * (size_t)n_sets * (size_t)std::max<int64_t>(nmax, 1);
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t o_off = up(b_pm), o_r = o_off + up(b_off), o_ok = o_r + up(b_r), o_inl = o_ok + up(b_ok), o_tr = o_inl + up(b_inl), o_xyz = o_tr + up(b_tr);
  EgoWorkLease lease;
  VH_HIP(lease.take(device, o_xyz + b_xyz));
  d = lease.d;
  hipError_t er = hipSuccess;
  if (total) er = hipMemcpy(d, pm, sizeof(vh_p_match) * (size_t)total, hipMemcpyHostToDevice);
  if (er == hipSuccess) er = hipMemcpy(d + o_off, offsets, b_off, hipMemcpyHostToDevice);
  if (er == hipSuccess) er = hipMemcpy(d + o_r, rand3, b_r, hipMemcpyHostToDevice);
  if (er == hipSuccess) {
    vh_launch_ego(*e, n_sets, (const vh_p_match *)d, 0, (const int32_t *)(d + o_off), nullptr, 0, (const int32_t *)(d + o_r),
                  (double *)(d + o_xyz), std::max<int64_t>(nmax, 1), (double *)(d + o_tr), (int32_t *)(d + o_ok), (int32_t *)(d + o_ok) + n_sets,
                  (int32_t *)(d + o_inl), 0, nullptr);
    er = hipDeviceSynchronize();
  }
  if (er == hipSuccess) er = hipMemcpy(tr, d + o_tr, b_tr, hipMemcpyDeviceToHost);
  if (er == hipSuccess) er = hipMemcpy(ok, d + o_ok, sizeof(int32_t) * (size_t)n_sets, hipMemcpyDeviceToHost);
  if (er == hipSuccess) er = hipMemcpy(n_inliers, d + o_ok + sizeof(int32_t) * (size_t)n_sets, sizeof(int32_t) * (size_t)n_sets, hipMemcpyDeviceToHost);
  if (er == hipSuccess && inliers && total) er = hipMemcpy(inliers, d + o_inl, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost);
  if (er != hipSuccess) { t_last_error = hipGetErrorString(er); return VH_ERR_HIP; }
  return VH_OK;
}

int32_t vh_group_post_begin(vh_group *g, int32_t cap_per_stream) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->post_begin(cap_per_stream);
}
int32_t vh_group_post_finish(vh_group *g, int32_t age, int32_t max_features, float bucket_width, float bucket_height, int32_t host_threads,
                             const vh_ego_params *e, const int32_t *rand3, double *tr, int32_t *ok, int32_t *n_inliers,
                             vh_p_match *bucketed, int32_t cap_per_stream, int32_t *counts, double *host_ms) {
  Group *gq = (Group *)g; ENTER(gq);
  if (host_threads < 1) host_threads = (int32_t)std::max(1u, std::thread::hardware_concurrency());
  return gq->post_finish(age, max_features, bucket_width, bucket_height, host_threads, e, rand3, nullptr, nullptr, tr, ok, n_inliers, bucketed, cap_per_stream, counts, host_ms);
}
int32_t vh_group_post_finish_mono(vh_group *g, int32_t age, int32_t max_features, float bucket_width, float bucket_height, int32_t host_threads,
                                  const vh_mono_params *e, const int32_t *rand8, double *tr, int32_t *ok, int32_t *n_inliers,
                                  vh_p_match *bucketed, int32_t cap_per_stream, int32_t *counts, double *host_ms) {
  Group *gq = (Group *)g; ENTER(gq);
  if (host_threads < 1) host_threads = (int32_t)std::max(1u, std::thread::hardware_concurrency());
  return gq->post_finish(age, max_features, bucket_width, bucket_height, host_threads, nullptr, nullptr, e, rand8, tr, ok, n_inliers, bucketed, cap_per_stream, counts, host_ms);
}

int32_t vh_group_post_device_config(vh_group *g, int32_t steps_per_batch, int32_t batches, int32_t lanes_per_wave) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->post_device_config(steps_per_batch, batches, lanes_per_wave);
}
int32_t vh_group_post_begin_device(vh_group *g, int32_t cap_per_stream, int32_t max_features, float bucket_width, float bucket_height,
                                   const vh_ego_params *e, const int32_t *rand3, const vh_mono_params *mono, const int32_t *rand8, int32_t want_lists) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->post_begin_device(cap_per_stream, max_features, bucket_width, bucket_height, e, rand3, mono, rand8, want_lists);
}
int32_t vh_group_post_finish_device(vh_group *g, int32_t age, double *tr, int32_t *ok, int32_t *n_inliers, vh_p_match *bucketed, int32_t cap_per_stream,
                                    int32_t *counts) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->post_finish_device(age, tr, ok, n_inliers, bucketed, cap_per_stream, counts);
}

// ---- monocular egomotion -----------------------------------------
void vh_default_mono_params(vh_mono_params *e) {
  if (!e) return;
  memset(e, 0, sizeof(*e));
  e->ransac_iters = 2000; e->inlier_threshold = 0.00001; e->motion_threshold = 100.0;  // src/viso_mono.h:39-45
  e->height = 1.0; e->pitch = 0.0; e->f = 1; e->cu = 0; e->cv = 0;                      // src/viso.h:46-48
}
int32_t vh_group_estimate_motion_mono(vh_group *g, const vh_mono_params *e, const int32_t *rand8, double *tr, int32_t *ok,
                                      int32_t *n_inliers) {
  Group *gq = (Group *)g; ENTER(gq);
  return gq->estimate_motion_mono(e, rand8, tr, ok, n_inliers);
}
int32_t vh_estimate_motion_mono(const vh_mono_params *e, int32_t device, int32_t n_sets, const vh_p_match *pm,
                                const int32_t *offsets, const int32_t *rand8, double *tr, int32_t *ok, int32_t *n_inliers,
                                int32_t *inliers) {
  if (!e || n_sets < 1 || !offsets || !rand8 || !tr || !ok || !n_inliers || e->ransac_iters < 1) return VH_ERR_INVALID_ARG;
  if (offsets[0] < 0) return VH_ERR_INVALID_ARG;
  int64_t nmax = 0;
  for (int32_t s = 0; s < n_sets; s++) {
    if (offsets[s + 1] < offsets[s]) return VH_ERR_INVALID_ARG;
    nmax = std::max<int64_t>(nmax, offsets[s + 1] - offsets[s]);
  }
  const int64_t total = offsets[n_sets], cap = std::max<int64_t>(nmax, 1);
  if (total > 0 && !pm) return VH_ERR_INVALID_ARG;
  if ((int64_t)n_sets * e->ransac_iters > (int64_t)1 << 31) return VH_ERR_UNSUPPORTED;
  if (n_sets > 65535) return VH_ERR_UNSUPPORTED;  // (the hypothesis and triangulation kernels put the list on grid.y)
  const int32_t rc = select_device(device);
  if (rc) return rc;
  const size_t nr = (size_t)n_sets * e->ransac_iters * 8;
  uint8_t *d = nullptr;
  // one allocation: matches | offsets | rand8 | ok,ninl | inliers | tr | per-list scratch
  const size_t b_pm = sizeof(vh_p_match) * (size_t)std::max<int64_t>(total, 1), b_off = sizeof(int32_t) * ((size_t)n_sets + 1);
  const size_t b_r = sizeof(int32_t) * nr, b_ok = sizeof(int32_t) * 2 * (size_t)n_sets, b_inl = sizeof(int32_t) * (size_t)std::max<int64_t>(total, 1);
  const size_t b_tr = sizeof(double) * 6 * (size_t)n_sets, b_scr = (size_t)vh_mono_scratch_bytes(n_sets, cap, e->ransac_iters);
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t o_off = up(b_pm), o_r = o_off + up(b_off), o_ok = o_r + up(b_r), o_inl = o_ok + up(b_ok), o_tr = o_inl + up(b_inl), o_scr = o_tr + up(b_tr);
  EgoWorkLease lease;
  VH_HIP(lease.take(device, o_scr + b_scr));
  d = lease.d;
  hipError_t er = hipSuccess;
  if (total) er = hipMemcpy(d, pm, sizeof(vh_p_match) * (size_t)total, hipMemcpyHostToDevice);
  if (er == hipSuccess) er = hipMemcpy(d + o_off, offsets, b_off, hipMemcpyHostToDevice);
  if (er == hipSuccess) er = hipMemcpy(d + o_r, rand8, b_r, hipMemcpyHostToDevice);
  if (er == hipSuccess) {
    vh_launch_mono(*e, n_sets, (const vh_p_match *)d, 0, (const int32_t *)(d + o_off), nullptr, 0, (const int32_t *)(d + o_r), d + o_scr, cap,
                   (double *)(d + o_tr), (int32_t *)(d + o_ok), (int32_t *)(d + o_ok) + n_sets, (int32_t *)(d + o_inl), 0, nullptr);
    er = hipGetLastError();  // (a rejected launch is not reported by the synchronisation)
    if (er == hipSuccess) er = hipDeviceSynchronize();
  }
  if (er == hipSuccess) er = hipMemcpy(tr, d + o_tr, b_tr, hipMemcpyDeviceToHost);
  if (er == hipSuccess) er = hipMemcpy(ok, d + o_ok, sizeof(int32_t) * (size_t)n_sets, hipMemcpyDeviceToHost);
  if (er == hipSuccess) er = hipMemcpy(n_inliers, d + o_ok + sizeof(int32_t) * (size_t)n_sets, sizeof(int32_t) * (size_t)n_sets, hipMemcpyDeviceToHost);
  if (er == hipSuccess && inliers && total) er = hipMemcpy(inliers, d + o_inl, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost);
  if (er != hipSuccess) { t_last_error = hipGetErrorString(er); return VH_ERR_HIP; }
  return VH_OK;
}

// ---- stateless primitives ----------------------------------------------------
int32_t vh_filters(int32_t device, const uint8_t *I, int32_t bpl, int32_t H, uint8_t *du, uint8_t *dv,
                   int16_t *f1, int16_t *f2) {
  if (!I || bpl < 5 || H < 5) return VH_ERR_INVALID_ARG;
  int32_t rc = select_device(device);
  if (rc) return rc;
  const size_t n = (size_t)bpl * H;
  uint8_t *d = nullptr;
  VH_HIP(hipMalloc((void **)&d, n * 7));
  uint8_t *dI = d, *ddu = d + n, *ddv = d + 2 * n;
  int16_t *df1 = (int16_t *)(d + 3 * n), *df2 = (int16_t *)(d + 5 * n);
  hipError_t e = hipMemcpy(dI, I, n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    vh_launch_planes(dI, bpl, H, ddu, ddv, df1, df2, nullptr);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess && du) e = hipMemcpy(du, ddu, n, hipMemcpyDeviceToHost);
  if (e == hipSuccess && dv) e = hipMemcpy(dv, ddv, n, hipMemcpyDeviceToHost);
  if (e == hipSuccess && f1) e = hipMemcpy(f1, df1, 2 * n, hipMemcpyDeviceToHost);
  if (e == hipSuccess && f2) e = hipMemcpy(f2, df2, 2 * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { t_last_error = hipGetErrorString(e); return VH_ERR_HIP; }
  return VH_OK;
}

int32_t vh_compute_features(const vh_params *p, int32_t device, const uint8_t *I, const int32_t dims[3],
                            int32_t *max1, int32_t cap1, int32_t *num1, int32_t *max2, int32_t cap2,
                            int32_t *num2, uint8_t *du, uint8_t *dv) {
  if (!p || !I || !dims) return VH_ERR_INVALID_ARG;
  if (num1) *num1 = 0;
  if (num2) *num2 = 0;
  int32_t rc, overflow = VH_OK;
  {  // dense set (matcher.cpp:634-635)
    Temp t;
    if ((rc = group_new(p, device, 1, 0, 0, &t.gq))) return rc;
    if ((rc = t.gq->push_host(I, nullptr, 0, dims, 0))) return rc;
    int32_t n = 0;
    rc = t.gq->get_features(0, VH_SET_1C, max2, max2 ? cap2 : 0, &n);
    if (num2) *num2 = n;
    if (rc == VH_ERR_CAPACITY) overflow = rc; else if (rc) return rc;
    if (du || dv) {  // I_du / I_dv at matching resolution (matcher.cpp:596-600, :606-612)
      const VhGeom &g = t.gq->g;
      const size_t np = (size_t)g.bplm * g.Hm;
      uint8_t *d = nullptr;
      VH_HIP(hipMalloc((void **)&d, 2 * np));
      const uint8_t *src = p->half_resolution ? t.gq->d_half : t.gq->d_stage[0];
      vh_launch_planes(src, g.bplm, g.Hm, d, d + np, nullptr, nullptr, t.gq->stream);
      hipError_t e = hipStreamSynchronize(t.gq->stream);
      if (e == hipSuccess && du) e = hipMemcpy(du, d, np, hipMemcpyDeviceToHost);
      if (e == hipSuccess && dv) e = hipMemcpy(dv, d + np, np, hipMemcpyDeviceToHost);
      (void)hipFree(d);
      if (e != hipSuccess) { t_last_error = hipGetErrorString(e); return VH_ERR_HIP; }
    }
  }
  if (p->multi_stage) {  // sparse set (matcher.cpp:621-628)
    vh_params ps = *p;
    int32_t ns = p->nms_n * 4;
    if (ns > 10) ns = std::max(p->nms_n, 10);
    ps.nms_n = ns;
    Temp t;
    if ((rc = group_new(&ps, device, 1, 0, 0, &t.gq))) return rc;
    if ((rc = t.gq->push_host(I, nullptr, 0, dims, 0))) return rc;
    int32_t n = 0;
    rc = t.gq->get_features(0, VH_SET_1C, max1, max1 ? cap1 : 0, &n);
    if (num1) *num1 = n;
    if (rc == VH_ERR_CAPACITY) overflow = rc; else if (rc) return rc;
  }
  return overflow;
}

int32_t vh_create_index(const vh_params *p, int32_t device, const int32_t dims[3], const int32_t *m,
                        int32_t n, int32_t *bin_start, int32_t *list) {
  if (!p || !dims || !bin_start || (n > 0 && !list)) return VH_ERR_INVALID_ARG;
  Temp t;
  int32_t rc;
  if ((rc = group_new(p, device, 1, std::max(n, 64), 1, &t.gq))) return rc;
  const int32_t d[3] = {dims[0], dims[1], std::max(dims[2], dims[0])};
  if ((rc = t.gq->ensure(d))) return rc;
  if ((rc = t.gq->load_features(VH_SET_1C, m, n))) return rc;
  Group *gq = t.gq;
  int32_t *d_bs = nullptr, *d_list = nullptr;
  if ((rc = gq->dmalloc(&d_bs, (size_t)gq->sets.nbins + 1, false))) return rc;
  if ((rc = gq->dmalloc(&d_list, (size_t)std::max(n, 1), false))) return rc;
  vh_launch_ref_index(gq->sets, vh_role_set(1, gq->pairs(), 0, VH_SET_1C), d_bs, d_list, gq->stream);
  VH_HIP(hipMemcpyAsync(bin_start, d_bs, sizeof(int32_t) * ((size_t)gq->sets.nbins + 1), hipMemcpyDeviceToHost, gq->stream));
  if (n) VH_HIP(hipMemcpyAsync(list, d_list, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, gq->stream));
  VH_HIP(hipStreamSynchronize(gq->stream));
  return VH_OK;
}

int32_t vh_match_all(const vh_params *p, int32_t device, const int32_t dims[3], const int32_t *m1, int32_t n1,
                     const int32_t *m2, int32_t n2, int32_t flow, int32_t *best) {
  if (!p || !dims || (n1 > 0 && !best)) return VH_ERR_INVALID_ARG;
  Temp t;
  int32_t rc;
  if ((rc = group_new(p, device, 1, std::max(std::max(n1, n2), 64), 1, &t.gq))) return rc;
  const int32_t d[3] = {dims[0], dims[1], std::max(dims[2], dims[0])};
  Group *gq = t.gq;
  if ((rc = gq->ensure(d))) return rc;
  if ((rc = gq->load_features(VH_SET_1C, m1, n1))) return rc;
  if ((rc = gq->load_features(VH_SET_1P, m2, n2))) return rc;
  VhMatchArgs a = gq->match_args(VH_METHOD_FLOW);
  a.npass = 1; a.pass[0] = {VH_SET_1C, VH_SET_1P, flow ? 1 : 0, 0};
  vh_launch_match(gq->sets, a, gq->d_best, gq->d_redo, gq->force_mode == 0 ? 0 : 1, 0, 0, gq->stream);
  VH_HIP(hipGetLastError());
  if (n1) VH_HIP(hipMemcpyAsync(best, gq->d_best, sizeof(int32_t) * (size_t)n1, hipMemcpyDeviceToHost, gq->stream));
  VH_HIP(hipStreamSynchronize(gq->stream));
  return VH_OK;
}

int32_t vh_match_all_prior(const vh_params *p, int32_t device, const int32_t dims[3], const int32_t *m1, int32_t n1,
                           const int32_t *m2, int32_t n2, int32_t flow, double u_, double v_, int32_t *best) {
  if (!p || !dims || (n1 > 0 && !best)) return VH_ERR_INVALID_ARG;
  Temp t;
  int32_t rc;
  if ((rc = group_new(p, device, 1, std::max(std::max(n1, n2), 64), 1, &t.gq))) return rc;
  const int32_t d[3] = {dims[0], dims[1], std::max(dims[2], dims[0])};
  Group *gq = t.gq;
  if ((rc = gq->ensure(d))) return rc;
  if ((rc = gq->load_features(VH_SET_1C, m1, n1))) return rc;
  if ((rc = gq->load_features(VH_SET_1P, m2, n2))) return rc;
  VhMatchArgs a = gq->match_args(VH_METHOD_FLOW);
  a.npass = 1; a.pass[0] = {VH_SET_1C, VH_SET_1P, flow ? 1 : 0, 0};
  vh_launch_match_prior(gq->sets, a, u_, v_, gq->d_best, gq->stream);
  VH_HIP(hipGetLastError());
  if (n1) VH_HIP(hipMemcpyAsync(best, gq->d_best, sizeof(int32_t) * (size_t)n1, hipMemcpyDeviceToHost, gq->stream));
  VH_HIP(hipStreamSynchronize(gq->stream));
  return VH_OK;
}

int32_t vh_match(const vh_params *p, int32_t device, const int32_t dims[3], int32_t method, const int32_t *m1p,
                 int32_t n1p, const int32_t *m2p, int32_t n2p, const int32_t *m1c, int32_t n1c,
                 const int32_t *m2c, int32_t n2c, vh_p_match *out, int32_t cap, int32_t *n) {
  if (!p || !dims || !n) return VH_ERR_INVALID_ARG;
  if (method < 0 || method > 2) return VH_ERR_INVALID_ARG;
  Temp t;
  int32_t rc;
  const int32_t nmax = std::max(std::max(n1p, n2p), std::max(n1c, n2c));
  if ((rc = group_new(p, device, 1, std::max(nmax, 64), std::max(nmax, 64), &t.gq))) return rc;
  const int32_t d[3] = {dims[0], dims[1], std::max(dims[2], dims[0])};
  Group *gq = t.gq;
  if ((rc = gq->ensure(d))) return rc;
  if ((rc = gq->load_features(VH_SET_1P, m1p, n1p))) return rc;
  if ((rc = gq->load_features(VH_SET_2P, m2p, n2p))) return rc;
  if ((rc = gq->load_features(VH_SET_1C, m1c, n1c))) return rc;
  if ((rc = gq->load_features(VH_SET_2C, m2c, n2c))) return rc;
  if ((rc = gq->match(method))) return rc;
  return gq->get_matches(0, out, cap, n);
}

// ---- removeOutliers (+ bucketFeatures) on the device, stateless form ------------
// n_lists match lists, list l = pm[l * stride .. + counts[l]).  max_features < 1: the vote only, out[l * out_cap ..] receives
// the survivors; otherwise the survivors are bucketed as Matcher::bucketFeatures(max_features, bw, bh) does and out receives
// the bucketed lists.  The group form (vh_group_post_begin_device) is the throughput path; this one exists for tests and timing.
int32_t vh_remove_outliers_device(int32_t device, int32_t n_lists, const vh_p_match *pm, int64_t stride, const int32_t *counts, int32_t lanes_per_wave,
                                  int32_t max_features, float bw, float bh, vh_p_match *out, int32_t out_cap, int32_t *out_counts,
                                  int32_t *n_triangles, float *sweep_ms) {
  if (n_lists < 1 || !counts || !out_counts || out_cap < 0 || (out_cap > 0 && !out) || stride < 0) return VH_ERR_INVALID_ARG;
  if (max_features >= 1 && (!(bw >= 1) || !(bh >= 1))) return VH_ERR_INVALID_ARG;
  int32_t cap = 4;
  for (int32_t l = 0; l < n_lists; l++) {
    if (counts[l] < 0 || counts[l] > stride) return VH_ERR_INVALID_ARG;
    cap = std::max(cap, counts[l]);
  }
  if (cap > 1 && !pm) return VH_ERR_INVALID_ARG;
  if (cap > VH_VOTE_LIST_MAX) return VH_ERR_UNSUPPORTED;  // (16-bit hull links; the sweep's angular hash has VH_VOTE_HASH_MAX slots in LDS)
  const int32_t rc = select_device(device);
  if (rc) return rc;
  VhVoteBuffers vb;
  struct Guard { VhVoteBuffers &b; vh_p_match *src = nullptr; int32_t *cnt = nullptr; hipEvent_t ev[2] = {nullptr, nullptr};
                 ~Guard() { b.release(); if (src) (void)hipFree(src); if (cnt) (void)hipFree(cnt); for (auto e : ev) if (e) (void)hipEventDestroy(e); } } gd{vb};
  // the bucket grid these lists can need (matcher.cpp:150-151: floor(u_max / bw) + 1 columns, floor(v_max / bh) + 1 rows)
  int64_t grid = 1;
  if (max_features >= 1) {
    float u_max = 0, v_max = 0;
    for (int32_t l = 0; l < n_lists; l++)
      for (int32_t i = 0; i < counts[l]; i++) {
        const vh_p_match &q = pm[(size_t)l * stride + i];
        if (q.u1c > u_max) u_max = q.u1c;
        if (q.v1c > v_max) v_max = q.v1c;
      }
    grid = ((int64_t)floorf(u_max / bw) + 1) * ((int64_t)floorf(v_max / bh) + 1);
    if (!(grid >= 1) || grid > (1 << 20)) return VH_ERR_UNSUPPORTED;
  }
  VH_HIP(vb.alloc(n_lists, cap, std::max(out_cap, 1), (int32_t)grid));
  VH_HIP(vb.upload_lfsr());
  VH_HIP(hipMalloc((void **)&gd.src, sizeof(vh_p_match) * (size_t)n_lists * cap));
  VH_HIP(hipMalloc((void **)&gd.cnt, sizeof(int32_t) * (size_t)n_lists));
  for (int32_t l = 0; l < n_lists; l++)
    if (counts[l]) VH_HIP(hipMemcpy(gd.src + (size_t)l * cap, pm + (size_t)l * stride, sizeof(vh_p_match) * (size_t)counts[l], hipMemcpyHostToDevice));
  VH_HIP(hipMemcpy(gd.cnt, counts, sizeof(int32_t) * (size_t)n_lists, hipMemcpyHostToDevice));
  VH_HIP(hipEventCreate(&gd.ev[0])); VH_HIP(hipEventCreate(&gd.ev[1]));
  vh_launch_vote_prep(vb.v, 0, n_lists, gd.src, cap, gd.cnt, cap, nullptr, 1, nullptr);
  vh_launch_vote(vb.v, lanes_per_wave, max_features, bw, bh, vb.lfsr, vb.lfsr_n, vb.out, vb.out_cap, vb.out_count, gd.ev, nullptr);
  VH_HIP(hipGetLastError());
  VH_HIP(hipDeviceSynchronize());
  if (sweep_ms) VH_HIP(hipEventElapsedTime(sweep_ms, gd.ev[0], gd.ev[1]));
  std::vector<VhVoteMeta> meta((size_t)n_lists);
  VH_HIP(hipMemcpy(meta.data(), vb.v.meta, sizeof(VhVoteMeta) * (size_t)n_lists, hipMemcpyDeviceToHost));
  int32_t ret = VH_OK;
  for (int32_t l = 0; l < n_lists; l++) {
    const VhVoteMeta &m = meta[(size_t)l];
    if (n_triangles) n_triangles[l] = m.ntri;
    if (m.status == VH_VOTE_TRUNCATED) { out_counts[l] = max_features >= 1 ? m.out : m.kept; ret = VH_ERR_CAPACITY; continue; }
    if (m.status != VH_VOTE_OK && m.status != VH_VOTE_SKIP) { out_counts[l] = 0; if (ret == VH_OK) ret = VH_ERR_UNSUPPORTED; continue; }
    const int32_t k = max_features >= 1 ? m.out : m.kept;
    out_counts[l] = k;
    if (k > out_cap) { ret = VH_ERR_CAPACITY; continue; }
    if (k > 0) VH_HIP(hipMemcpy(out + (size_t)l * out_cap, max_features >= 1 ? vb.out + (size_t)l * vb.out_cap : vb.v.pm + (size_t)l * cap,
                                sizeof(vh_p_match) * (size_t)k, hipMemcpyDeviceToHost));
  }
  return ret;
}

}  // extern "C"
