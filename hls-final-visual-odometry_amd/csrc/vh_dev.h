// vh_dev.h -- shared host/device declarations of the gfx950 detect+match engine.
//
// Data layout in HBM (all arrays are flat, one allocation per kind, indexed by
// a *set id* or an *image id* so that one kernel launch covers every camera
// stream of a group):
//
//   image id  = stream*2 + cam                      (cam 0 = left, 1 = right)
//   set id    = pair*(2*S) + stream*2 + cam         (pair = ring slot 0..2; the
//               current/previous roles move from slot to slot, never by copy:
//               reference ring buffer src/matcher.cpp:64-79.  Three slots, so
//               that detection of frame t+1 can run while frame t is matched)
//
//   feat      [set][cap][12] int32   the reference's packed record
//                                    {u,v,0,c,d1..d8} (src/matcher.cpp:663-671)
//   f_uv      [set][cap]     uint32  u | v<<16, in reference order (compact copy of feat's first two words)
//   s_uv      [set][cap]     uint32  u | v<<16, in *bin order*
//   s_idx     [set][cap]     int32   original feature index of that position
//   s_desc    [set][cap][8]  uint32  32-byte descriptor, in bin order
//   bin_start [set][nbins+1] int32   CSR over bins, bin = (c*ubn+ub)*vbn+vb
//                                    (u-bin major: the iteration order of
//                                    Matcher::findMatch, src/matcher.cpp:243-246)
//   row_start [set][4*H+1]   int32   CSR over (class, v) rows: the stereo search
//                                    (v window of +-disp_tolerance) walks one
//                                    contiguous row range per query
//   r_pos     [set][cap]     int32   row order -> bin-order position
//   rec       [image][nblocks] u64   per NMS block 4 x u16 position codes
//   best      [stream][pass][cap] int32  findMatch result for every query
//   matches   [stream][mcap] p_match (48 B, src/matcher.h:89-104)
#ifndef VH_DEV_H
#define VH_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define VH_MARGIN 7          // src/matcher.cpp:38
#ifndef VH_CHUNK
#define VH_CHUNK 1024        // NMS blocks per emit workgroup (a multiple of 256)
#endif
#define VH_WAVE 64
// Flow search (kernels_match.hip): a wave is VH_FLOW_P phases of 64/VH_FLOW_P lanes, VH_FLOW_Q
// queries per lane; the phases share one candidate stream.
#ifndef VH_FLOW_Q
#define VH_FLOW_Q 2
#endif
#ifndef VH_FLOW_P
#define VH_FLOW_P 4
#endif
#define VH_TILE_Q (64 * VH_FLOW_Q / VH_FLOW_P)  // queries per flow-search tile
#define VH_NO_CODE 0xFFFFu
// Wave priority of the detection chain's kernels (detect_nms, emit_features, bin_scan, bin_sort): their instructions
// issue ahead of the searches' on a shared SIMD.  The chain is the step's critical path (4 sub-batches x 4 dependent
// kernels, each slowed 2-3x by the searches beside it) while the searches only need their sum of issue slots.
// MI355X, KITTI, S = 256, k pairs/s: priority 0 / 1 / 2 / 3 = 104.8 / 107.2 / 107.3 / 107.2 (round 5, after the
// searches lost 6 % of their instructions; in round 3, with the heavier searches, the same switch was +-0).
#ifndef VH_DET_PRIO
#define VH_DET_PRIO 1
#endif
#define VH_DET_SETPRIO() do { if (VH_DET_PRIO) __builtin_amdgcn_s_setprio(VH_DET_PRIO); } while (0)
// 32-bit match keys SAD << 19 | (bin-order position - first position of the class) serve classes
// of up to 2^19 - 64 features per set (the staged chunks of the searches repeat the last candidate
// of a run under positions up to 63 past it); larger classes take 64-bit keys (kernels_match.hip).
// The detector yields at most one feature per class and NMS block, so only images of more than
// 524 224 blocks can need those.
#define VH_CLASS_POS_BITS 19
#define VH_CLASS_POS_MAX ((1 << VH_CLASS_POS_BITS) - 64)
// first-writer pixel mask of the flow method: epoch << 24 | (2^24 - 1 - i1c); also the
// largest feature capacity per image (2^24 - 1)
#define VH_MASK_IDX_BITS 24

struct VhGeom {
  // full-resolution image
  int32_t W, H, bpl;
  // matching-resolution image (== full unless half_resolution)
  int32_t Wm, Hm, bplm;
  int32_t scale;            // 1, or 2 with half_resolution (src/matcher.cpp:647-649)
  int32_t n, tau;           // nms_n, nms_tau
  int32_t nbx, nby, nblocks, nchunks;
  int32_t tbx, tby;         // NMS blocks per detect tile
  int32_t FW, FH, IW, IH;   // filter-response tile / image tile extents (pixels)
  int32_t IWp, FWp;         // padded LDS row pitches
};

struct VhSets {
  int32_t *feat;
  uint32_t *f_uv;      // [set][cap] u | v << 16 in reference (feature index) order: what the chains and the emission gather
  uint32_t *s_uv;
  int32_t *s_idx;
  uint32_t *s_desc;
  int32_t *bin_start;
  int32_t *hist;
  int32_t *cursor;
  int32_t *tmp_idx;
  int2 *stage;         // [set][nbins][stage_cap] {feature index, rank in its (class, v) row} appended by emit_features (arbitrary order)
  int32_t *count;
  // row index (stereo search): the same records ordered by (class, v)
  int32_t *row_start;  // [set][4*H+1]
  int32_t *row_hist;   // [set][4*H]
  int32_t *row_cursor; // [set][4*H]
  int32_t *r_pos;      // [set][cap] bin-order position of the feature at each row-order position
  int4 *tiles;       // [set][max_tiles] {first snake index, end, class, column of the first} (kernels_bin.hip: make_tiles)
  int32_t *tile_cnt; // [set]
  int32_t cap, nbins, ubn, vbn, binsize, max_tiles;
  uint32_t inv_binsize;  // ceil(2^32 / binsize): x / binsize == __umulhi(x, inv_binsize) for 0 <= x < 2^18
  int32_t W, H;      // dims_c of the matcher (full resolution)
  int32_t stage_cap;   // max features of one class in one bin for features of this detector (geometry bound)
  uint32_t *check;   // [4] -DVH_CHECK builds: {violations, code of the first, its value, its bound}; unused otherwise
};

// Index invariants of the device-side indices (row index -> bin position -> record).  The shipped
// build trusts them (no clamps: DESIGN.md section 2a says why each holds); a -DVH_CHECK build
// verifies every one on the device, records the first violation in VhSets::check and continues with
// the index forced into range, and the host side aborts the process at the next synchronisation
// (engine.hip: check_violation) -- loud, but without a wild access on a shared GPU.
//   codes: 1 rows_tile query position, 2 rows_tile candidate position, 3 row re-search candidate
//          position, 4 bin_sort row slot, 5 emit_features stage slot, 6 bin_sort staged bin length,
//          7 winner position of a search
#ifdef VH_CHECK
#define VH_CHECK_RANGE(s_, code_, x_, lo_, hi_)                                          \
  do {                                                                                   \
    if ((x_) < (lo_) || (x_) >= (hi_)) {                                                 \
      if (atomicAdd(&(s_).check[0], 1u) == 0u) {                                         \
        (s_).check[1] = (uint32_t)(code_); (s_).check[2] = (uint32_t)(x_); (s_).check[3] = (uint32_t)(hi_); \
      }                                                                                  \
      (x_) = (lo_) < (hi_) ? (((x_) < (lo_)) ? (lo_) : (hi_) - 1) : (lo_);                \
    }                                                                                    \
  } while (0)
#else
#define VH_CHECK_RANGE(s_, code_, x_, lo_, hi_) do { } while (0)
#endif

struct VhPass {
  int32_t qset;  // role (VH_SET_*) providing the queries
  int32_t cset;  // role providing the candidates
  int32_t flow;  // 1: +-radius in v; 0: +-disp_tolerance (stereo search)
  int32_t slot;  // which of the 4 result tables of the stream receives this pass
};

struct VhMatchArgs {
  VhPass pass[4];
  int32_t npass;
  int32_t pair_cur;  // ring slots: current frame | previous frame << 8
  int32_t S;
  int32_t radius, disp_tol;
  int32_t wide_keys;  // test hook (VH_FLOW_WIDE_KEYS): 1 = never the 16-bit position keys, 2 = always the 64-bit keys
  int32_t prior;      // quad with a motion prior (kernels_prior.hip): pass 1 is not searched by match_kernel, and its table is
                      // indexed by the driving feature i1p instead of by the query i2p
};

__host__ __device__ inline int32_t vh_set_id(int32_t S, int32_t pair, int32_t stream, int32_t cam) {
  return pair * (2 * S) + stream * 2 + cam;
}
// role: 0=1p 1=2p 2=1c 3=2c
__host__ __device__ inline int32_t vh_role_set(int32_t S, int32_t pairs, int32_t stream, int32_t role) {
  const int32_t pair = (role >= 2) ? (pairs & 0xFF) : (pairs >> 8);
  return vh_set_id(S, pair, stream, role & 1);
}

struct VhImages {
  const uint8_t *base[2];  // left / right image of stream 0
  int64_t stride;          // bytes between consecutive streams
  int32_t ncam;            // 1 (mono / flow) or 2 (stereo)
  int32_t S;               // streams covered by this launch (a sub-batch of the group)
  int32_t pair_cur;        // ring slot the new features are written to
  int32_t S_total, s0;     // streams of the group / first stream of this launch (base[] already point at it)
};
// image id = stream*ncam + cam
__host__ __device__ inline const uint8_t *vh_image_ptr(const VhImages &im, int32_t id) {
  const int32_t s = id / im.ncam, cam = id % im.ncam;
  return im.base[cam] + (int64_t)s * im.stride;
}
__host__ __device__ inline int32_t vh_image_set(const VhImages &im, int32_t id) {
  return vh_set_id(im.S_total, im.pair_cur, im.s0 + id / im.ncam, id % im.ncam);
}

// ---- launchers (defined in the kernels_*.hip files) ------------------------
void vh_launch_half_res(const VhImages &src, uint8_t *dst, const VhGeom &g, hipStream_t st);
void vh_launch_detect_nms(const VhImages &im, const VhGeom &g, uint64_t *rec, int32_t *chunk_count,
                          hipStream_t st);
void vh_launch_emit_features(const VhImages &im, const VhGeom &g, const uint64_t *rec,
                             const int32_t *chunk_count, const VhSets &s, hipStream_t st);
void vh_launch_planes(const uint8_t *img, int32_t bpl, int32_t H, uint8_t *du, uint8_t *dv,
                      int16_t *f1, int16_t *f2, hipStream_t st);

void vh_launch_zero_counters(const VhSets &s, int32_t set0, int32_t nsets, int32_t *extra, int64_t n_extra, hipStream_t st);
void vh_launch_bin_hist(const VhSets &s, int32_t set0, int32_t nsets, hipStream_t st);
void vh_launch_bin_scan(const VhSets &s, int32_t set0, int32_t nsets, hipStream_t st);
void vh_launch_bin_fill(const VhSets &s, int32_t set0, int32_t nsets, hipStream_t st);
void vh_launch_bin_sort(const VhSets &s, int32_t set0, int32_t nsets, int32_t staged, hipStream_t st);
void vh_launch_ref_index(const VhSets &s, int32_t set, int32_t *bin_start_ref, int32_t *list_ref,
                         hipStream_t st);

void vh_launch_match_prior(const VhSets &s, const VhMatchArgs &a, double u_, double v_, int32_t *best,
                           hipStream_t st);
// grid_x: workgroups per (pass, stream) row (<= 0: one per 4 tiles of the capacity-sized tile list); lds_bytes: LDS a workgroup
// is to occupy, static part included (<= 0: the static part only) -- see the launcher for what both are for
void vh_launch_match(const VhSets &s, const VhMatchArgs &a, int32_t *best, int32_t *redo, int32_t speculative, int32_t grid_x,
                     int32_t lds_bytes, hipStream_t st);
void vh_launch_quad_prior(const VhSets &s, const VhMatchArgs &a, const double *tr, double f, double cu, double cv, double base, int32_t *best,
                          hipStream_t st);
// chain: [stream][cap][2] int4 = {i1p,i2p,i1c,i2c} (z = -2: no match), {uv1p,uv2p,uv1c,uv2c}
void vh_launch_chain(const VhSets &s, const VhMatchArgs &a, int32_t method, const int32_t *best,
                     int4 *chain, uint32_t *mask, uint32_t epoch, int32_t *mchunk, hipStream_t st);
void vh_launch_emit_matches(const VhSets &s, const VhMatchArgs &a, int32_t method, const int4 *chain,
                            void *matches, int32_t mcap, int32_t *match_count, int32_t *overflow,
                            const int32_t *mchunk, int32_t *redo, int32_t *mchunk_next, void *host_out, void *host_matches,
                            hipStream_t st);

struct vh_ego_params;
struct vh_p_match;
void vh_launch_ego(const vh_ego_params &e, int32_t n_sets, const vh_p_match *pm, int64_t pm_stride, const int32_t *offsets,
                   const int32_t *counts, int32_t count_cap, const int32_t *rand3, double *xyz, int64_t xyz_stride, double *tr,
                   int32_t *ok, int32_t *ninl, int32_t *inl, int64_t inl_stride, hipStream_t st);

struct vh_mono_params;
int64_t vh_mono_scratch_bytes(int32_t n_sets, int64_t cap, int32_t ransac_iters);
void vh_launch_mono(const vh_mono_params &e, int32_t n_sets, const vh_p_match *pm, int64_t pm_stride, const int32_t *offsets,
                    const int32_t *counts, int32_t count_cap, const int32_t *rand8, uint8_t *scratch, int64_t cap, double *tr,
                    int32_t *ok, int32_t *ninl, int32_t *inl, int64_t inl_stride, hipStream_t st);

#endif
