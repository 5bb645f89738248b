/*
 * viso_hip.h -- C ABI of libviso_hip.so: the MI355X (gfx950) implementation of
 * the libviso2-style feature detection + matching hot path of
 * Chang-Tun-Yu/HLS-final-Visual-Odometry.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types.  Each entry point names the reference interface it replaces
 * (file:line relative to the reference tree).  include/viso_hip_matcher.hpp
 * wraps these calls in a C++ `Matcher` class with the reference's public
 * method set, so src/viso_stereo.cpp / src/viso_mono.cpp compile against it
 * unchanged.
 *
 * All compute runs in hand-written HIP kernels; there is no CPU fallback.
 * Every call returns VH_OK (0) or a negative VH_ERR_* code; nothing throws.
 *
 * Threading: a handle is not re-entrant (neither is the reference's Matcher);
 * use one handle per camera stream and one host thread per handle.
 */
#ifndef VISO_HIP_H
#define VISO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VH_ABI_VERSION 1

/* ---- error codes ------------------------------------------------------- */
#define VH_OK 0
#define VH_ERR_INVALID_ARG (-1)  /* null pointer, bad dims (reference: cerr "Image dimension mismatch", src/matcher.cpp:59-62) */
#define VH_ERR_NO_DEVICE (-2)    /* no HIP device / kernel image for this GPU: the product path never falls back to the CPU */
#define VH_ERR_HIP (-3)          /* a HIP runtime call failed; see vh_last_error() */
#define VH_ERR_CAPACITY (-4)     /* more features/matches than the capacity given (the reference overruns POINT_L silently, src/matcher.cpp:332) */
#define VH_ERR_UNSUPPORTED (-5)  /* parameter outside the supported envelope (see vh_create) */
#define VH_ERR_STATE (-6)        /* e.g. match before two frames were pushed */

/* ---- types --------------------------------------------------------------- */

/* POD mirror of Matcher::parameters, field for field (src/matcher.h:45-72). */
typedef struct vh_params {
  int32_t nms_n;                  /* non-max-suppression: min. distance between maxima (pixels) */
  int32_t nms_tau;                /* non-max-suppression: interest point peakiness threshold */
  int32_t match_binsize;          /* matching bin width/height */
  int32_t match_radius;           /* matching radius (du/dv in pixels) */
  int32_t match_disp_tolerance;   /* dv tolerance for stereo matches (pixels) */
  int32_t outlier_disp_tolerance; /* accepted, unused by this path (as in the reference) */
  int32_t outlier_flow_tolerance; /* accepted, unused by this path */
  int32_t multi_stage;            /* 1 = also extract the sparse feature set (max1) */
  int32_t half_resolution;        /* 1 = detect at half resolution, coordinates x2 */
  int32_t refinement;             /* accepted, unused (absent from the reference) */
  double f, cu, cv, base;         /* calibration (only for match prediction; unused) */
} vh_params;

/* Mirror of Matcher::p_match (src/matcher.h:89-104): 48 bytes, unused slots = -1. */
typedef struct vh_p_match {
  float u1p, v1p; int32_t i1p; /* previous left  */
  float u2p, v2p; int32_t i2p; /* previous right */
  float u1c, v1c; int32_t i1c; /* current  left  */
  float u2c, v2c; int32_t i2c; /* current  right */
} vh_p_match;

/* Feature records are int32[12] = {u, v, 0, class, d1..d8} exactly as
 * Matcher::computeFeatures packs them (src/matcher.cpp:663-671). */
#define VH_FEATURE_WORDS 12

/* Which ring-buffer feature set (vh_get_features). */
#define VH_SET_1P 0 /* previous left  */
#define VH_SET_2P 1 /* previous right */
#define VH_SET_1C 2 /* current  left  */
#define VH_SET_2C 3 /* current  right */

/* Matching method (Matcher::matchFeatures, src/matcher.h:124-128). */
#define VH_METHOD_FLOW 0
#define VH_METHOD_STEREO 1
#define VH_METHOD_QUAD 2

typedef struct vh_matcher vh_matcher; /* one camera stream (== one Matcher)      */
typedef struct vh_group vh_group;     /* S independent streams stepped together  */

/* ---- library ------------------------------------------------------------- */
int32_t vh_abi_version(void);
/* Number of visible HIP devices, or VH_ERR_NO_DEVICE. */
int32_t vh_device_count(void);
const char *vh_error_string(int32_t code);
/* Text of the last failing HIP call on this thread ("" if none). */
const char *vh_last_error(void);
/* Matcher::parameters() defaults (src/matcher.h:60-71). */
void vh_default_params(vh_params *p);

/* ---- one stream: the Matcher surface ----------------------------------- */

/* Matcher::Matcher(parameters) (src/matcher.cpp:32-41) on HIP device `device`.
 * Envelope (VH_ERR_UNSUPPORTED outside): 1 <= nms_n <= 32, match_binsize >= 1,
 * 0 <= match_radius <= 16384, 0 <= match_disp_tolerance <= 16384, nms_tau >= 0,
 * images up to 16384 x 16384 with a row pitch dims[2] < 2^24 bytes and dims[2]*dims[1] <= 2^28
 * bytes.  max_features/max_matches = 0 select the
 * worst-case capacity for the pushed image size (4 per NMS block), clamped to
 * 16 777 215 features per image.
 * When a pushed image yields more features than the capacity, the records
 * beyond it are dropped, matching runs on the truncated sets, and
 * vh_get_matches / vh_group_get_matches(_all) / vh_group_wait_download return
 * VH_ERR_CAPACITY (vh_get_features reports the true count). */
int32_t vh_create(const vh_params *p, int32_t device, vh_matcher **out);
int32_t vh_create_ex(const vh_params *p, int32_t device, int32_t max_features,
                     int32_t max_matches, vh_matcher **out);
/* Matcher::~Matcher (src/matcher.cpp:44-49). */
void vh_destroy(vh_matcher *m);
/* Matcher::setIntrinsics (src/matcher.h:81-86). */
int32_t vh_set_intrinsics(vh_matcher *m, double f, double cu, double cv, double base);

/* Matcher::pushBack(I1,I2,dims,replace) (src/matcher.h:116, src/matcher.cpp:51-91)
 * with the stock computeFeatures behind it (src/matcher.cpp:585-672).
 * I1/I2: host images, row-major u8, stride dims[2] >= dims[0]; I2 may be NULL
 * (mono/flow).  The images are borrowed for the duration of the call.
 * A call whose dims differ from the previous one's starts a new sequence: the
 * ring buffer is emptied (the reference keeps the old pair and would match
 * across image sizes, src/matcher.cpp:64-84; no caller does that). */
int32_t vh_push_back(vh_matcher *m, const uint8_t *I1, const uint8_t *I2,
                     const int32_t dims[3], int32_t replace);
/* Same, images already resident in device memory (e.g. a torch tensor's
 * data_ptr); asynchronous (the images must stay valid until vh_synchronize or
 * the next vh_get_*). */
int32_t vh_push_back_device(vh_matcher *m, const void *dI1, const void *dI2,
                            const int32_t dims[3], int32_t replace);

/* Page-locked host memory for image buffers handed to vh_push_back /
 * vh_group_push_back.  The reference's callers read frames into malloc'd
 * buffers (src/demo.cpp:107-110) and lend them to pushBack for the call
 * (src/matcher.cpp:51-91); any host pointer works here too, but uploads from
 * page-locked memory run at PCIe rate instead of through the driver's bounce
 * buffer.  Needs a device (VH_ERR_NO_DEVICE otherwise). */
int32_t vh_host_alloc(int32_t device, size_t bytes, void **out);
int32_t vh_host_free(void *ptr);

/* Matcher::matchFeatures(method, Tr_delta) (src/matcher.h:128,
 * src/matcher.cpp:93-111) with the stock Matcher::matching behind it
 * (src/matcher.cpp:274-344).  Tr_delta16 = NULL: no motion prior -- what the
 * reference's matchFeatures does with ANY Tr_delta (it ignores the argument),
 * and what the C++ shim passes.  Tr_delta16 != NULL (row-major 4x4) with
 * method 2 after vh_set_intrinsics: stock libviso2's prior
 * [upstream-recollection; absent from the reference tree] -- the hop previous
 * right -> current right of the quad circle is searched with findMatch's
 * prediction term (src/matcher.cpp:257-262, pinned) around the position that
 * Tr_delta predicts for the 3-d point of the (1p, 2p) pair
 * (csrc/kernels_prior.hip).  VH_ERR_STATE without intrinsics (f, base > 0).  The reference's matchFeatures goes on to
 * call removeOutliers (src/matcher.cpp:108); here that is the separate
 * vh_remove_outliers below, which the C++ shim calls for you. */
int32_t vh_match_features(vh_matcher *m, int32_t method, const double *Tr_delta16);

/* removeOutliers (src/remove_outliers.cpp:4-94 over src/delaunator.cpp:183-407):
 * Delaunay-neighbour flow-consistency vote on the current matches, host side
 * (a sequential float triangulation whose result depends on its
 * visiting order -- see csrc/outliers.cpp).  Filters flow and quad matches;
 * stereo matches (no previous-frame position) are left as they are.
 * VH_ERR_STATE before the first vh_match_features. */
int32_t vh_remove_outliers(vh_matcher *m);
/* The same on caller-owned records, in place, order preserved; *n_out = count
 * kept.  Pure host function: needs no device. */
int32_t vh_remove_outliers_pm(vh_p_match *pm, int32_t n, int32_t *n_out);
/* The same on the DEVICE for n_lists lists at once (csrc/kernels_vote.hip): list l = pm[l * stride .. + counts[l]).
 * The triangulation under the vote stays the sequential chain it is (csrc/sweep_hull.h, the code the host form
 * runs): one list per GPU lane, lanes_per_wave (1..64) lists per wavefront -- latency per list is tens of
 * milliseconds, the throughput comes from the number of lists in flight (see vh_group_post_begin_device).
 * max_features < 1: removeOutliers only, out[l * out_cap ..] receives list l's survivors in order;
 * max_features >= 1: followed by Matcher::bucketFeatures(max_features, bucket_width, bucket_height)
 * (src/matcher.cpp:140-187), out receives the bucketed lists.  out_counts[l]: records of list l in out;
 * n_triangles (nullable): triangles of each list's triangulation; sweep_ms (nullable): device time of the
 * sweep kernel.  VH_ERR_CAPACITY if a list does not fit out_cap; VH_ERR_UNSUPPORTED for lists the sweep
 * refuses (NaN / infinite / negative coordinates, more than 31 flips pending in one legalisation -- the reference's
 * stack has 13 slots and is undefined beyond; such a list's sweep stops at the flip that does not fit, nothing of it is
 * delivered, out_counts[l] = 0, and the other lists of the call are delivered as usual). */
int32_t vh_remove_outliers_device(int32_t device, int32_t n_lists, const vh_p_match *pm, int64_t stride, const int32_t *counts,
                                  int32_t lanes_per_wave, int32_t max_features, float bucket_width, float bucket_height,
                                  vh_p_match *out, int32_t out_cap, int32_t *out_counts, int32_t *n_triangles, float *sweep_ms);

/* Matcher::bucketFeatures (src/matcher.h:132, src/matcher.cpp:140-187):
 * host-side post-processing of the current matches, LFSR shuffle included. */
int32_t vh_bucket_features(vh_matcher *m, int32_t max_features, float bucket_width,
                           float bucket_height);

/* Matcher::getMatches (src/matcher.h:138-143).  *n receives the true count;
 * at most cap records are written (VH_ERR_CAPACITY if n > cap). */
int32_t vh_get_matches(vh_matcher *m, vh_p_match *out, int32_t cap, int32_t *n);
/* The ring buffer's feature records (max2p/max2c in the reference,
 * src/matcher.h:252): the parity contract includes descriptors. */
int32_t vh_get_features(vh_matcher *m, int32_t which, int32_t *out12, int32_t cap, int32_t *n);
/* Block until everything queued on the handle's stream has finished. */
int32_t vh_synchronize(vh_matcher *m);
/* Order this handle's work after a caller-owned hipStream_t (e.g. the stream
 * that produces the device images, torch's current stream): every pushBack
 * first waits for what that stream has been given so far.  The work itself runs
 * on the handle's internal (non-blocking) streams -- detection of frame t+1
 * overlaps matching of frame t -- and results are complete after
 * vh_synchronize / vh_get_*.  The handle 0 (NULL) is the legacy default
 * stream and is ordered after like any other stream; without a set stream
 * (the initial state, or after vh_clear_stream) a pushBack is ordered after
 * nothing, and images produced on ANY stream, the default one included, must
 * be complete (hipStreamSynchronize) before vh_push_back_device. */
int32_t vh_set_stream(vh_matcher *m, void *hip_stream);
int32_t vh_clear_stream(vh_matcher *m);
/* The reverse ordering: make `hip_stream` wait (on the device, without
 * blocking the host) until the images handed to the last vh_push_back_device
 * have been consumed, so that work queued on it afterwards may overwrite them. */
int32_t vh_stream_wait_images(vh_matcher *m, void *hip_stream);

/* ---- stateless primitives (private members of the reference's Matcher) -- */

/* Matcher::computeFeatures (src/matcher.h:209, src/matcher.cpp:585-672).
 * max1/num1 (sparse set, multi_stage only), du/dv (matching-resolution
 * gradient planes, stride dims_matching[2]) may be NULL.  num1/num2 return
 * the true counts. */
int32_t vh_compute_features(const vh_params *p, int32_t device, const uint8_t *I,
                            const int32_t dims[3], int32_t *max1, int32_t cap1,
                            int32_t *num1, int32_t *max2, int32_t cap2, int32_t *num2,
                            uint8_t *du, uint8_t *dv);
/* filter::sobel5x5 / blob5x5 / checkerboard5x5 (src/filter.h:80-96) on the
 * valid interior; pixels outside it are 0.  Any output may be NULL. */
int32_t vh_filters(int32_t device, const uint8_t *I, int32_t bpl, int32_t H, uint8_t *du,
                   uint8_t *dv, int16_t *f1, int16_t *f2);
/* Matcher::createIndexVector (src/matcher.cpp:194-214) flattened to CSR in the
 * reference's bin numbering (c*v_bin_num+v_bin)*u_bin_num+u_bin:
 * bin_start[4*ubn*vbn+1], list[n]. */
int32_t vh_create_index(const vh_params *p, int32_t device, const int32_t dims[3],
                        const int32_t *m, int32_t n, int32_t *bin_start, int32_t *list);
/* Matcher::findMatch (src/matcher.cpp:216-272) for every query of set 1
 * against set 2: best[i1] = min_ind.  flow=0 narrows v to
 * +-match_disp_tolerance (stock libviso2 stereo search). */
int32_t vh_match_all(const vh_params *p, int32_t device, const int32_t dims[3],
                     const int32_t *m1, int32_t n1, const int32_t *m2, int32_t n2,
                     int32_t flow, int32_t *best);
/* Matcher::findMatch with its optional match-prediction term
 * (src/matcher.cpp:257-262): cost = SAD + 4*||(u2,v2)-(u_,v_)|| evaluated and
 * compared in double, as the reference does.  No caller in the reference passes
 * u_,v_ (they default to -1 = off); provided so that the whole primitive is
 * covered.  Not on the throughput path (one lane per query, double math). */
int32_t vh_match_all_prior(const vh_params *p, int32_t device, const int32_t dims[3],
                           const int32_t *m1, int32_t n1, const int32_t *m2, int32_t n2,
                           int32_t flow, double u_, double v_, int32_t *best);
/* Matcher::matching (src/matcher.h:218, src/matcher.cpp:274-344) on
 * caller-supplied feature arrays. Unused sets: NULL/0. */
int32_t vh_match(const vh_params *p, int32_t device, const int32_t dims[3], int32_t method,
                 const int32_t *m1p, int32_t n1p, const int32_t *m2p, int32_t n2p,
                 const int32_t *m1c, int32_t n1c, const int32_t *m2c, int32_t n2c,
                 vh_p_match *out, int32_t cap, int32_t *n);

/* ---- S independent camera streams stepped together ---------------------- */
/* The multi-stream configuration (one sequence per stream, no exchange
 * between streams): every kernel launch covers all S streams, which is what
 * fills an MI355X at KITTI image size.  Stream s of a group behaves exactly
 * like its own vh_matcher. */
int32_t vh_group_create(const vh_params *p, int32_t device, int32_t n_streams,
                        int32_t max_features, int32_t max_matches, vh_group **out);
void vh_group_destroy(vh_group *g);
int32_t vh_group_streams(const vh_group *g);
/* Device memory the group currently holds (allocated at the first push_back for
 * the image size and capacities in use): for sizing S against the HBM of a GPU. */
int64_t vh_group_device_bytes(const vh_group *g);
/* Device-resident images: stream s reads dI1 + s*stride_bytes (and dI2 + ...;
 * dI2 may be NULL).  Asynchronous: returns once the work is queued; the images
 * must stay valid until the detection has run (vh_group_synchronize, or the
 * next vh_group_get_*). */
int32_t vh_group_push_back_device(vh_group *g, const void *dI1, const void *dI2,
                                  int64_t stride_bytes, const int32_t dims[3],
                                  int32_t replace);
/* Host images, same addressing. */
int32_t vh_group_push_back(vh_group *g, const uint8_t *I1, const uint8_t *I2,
                           int64_t stride_bytes, const int32_t dims[3], int32_t replace);
int32_t vh_group_match_features(vh_group *g, int32_t method);
/* ... with a motion prior per stream, Tr_delta16[S][16] (see vh_match_features); NULL: none */
int32_t vh_group_match_features_prior(vh_group *g, int32_t method, const double *Tr_delta16);
/* vh_remove_outliers for every stream of the group, `host_threads` workers
 * (<= 0: one per hardware thread).  Host-bound: a few ms per stream. */
int32_t vh_group_remove_outliers(vh_group *g, int32_t host_threads);
int32_t vh_group_get_matches(vh_group *g, int32_t stream, vh_p_match *out, int32_t cap,
                             int32_t *n);
/* Every stream's matches with one wait: stream s's records go to
 * out[s * cap_per_stream ...], its true count to counts[s] (VH_ERR_CAPACITY if
 * any count exceeds cap_per_stream; the records that fit are still written).
 * `out` in page-locked memory (vh_host_alloc) makes the transfers run at PCIe rate. */
int32_t vh_group_get_matches_all(vh_group *g, vh_p_match *out, int32_t cap_per_stream, int32_t *counts);
/* The same without waiting: starts one strided device->host transfer of the first
 * cap_per_stream records of every stream (whatever their counts; records beyond
 * counts[s] are stale) plus the S counts, ordered after the last
 * vh_group_match_features, and returns.  The next step can be issued at once; its
 * emission waits for this download on the device.  `out` and `counts` must be
 * page-locked (vh_host_alloc) and stay untouched until vh_group_wait_download
 * (or vh_group_synchronize) returns.  Host-side post-processing
 * (vh_group_remove_outliers) is not reflected: these are the device lists. */
int32_t vh_group_download_matches_async(vh_group *g, vh_p_match *out, int32_t cap_per_stream, int32_t *counts);
int32_t vh_group_wait_download(vh_group *g);
int32_t vh_group_get_features(vh_group *g, int32_t stream, int32_t which, int32_t *out12,
                              int32_t cap, int32_t *n);
/* Per-stream counts of the last step without copying records:
 * n_features[4*S] (1p,2p,1c,2c per stream), n_matches[S]. Either may be NULL. */
int32_t vh_group_get_counts(vh_group *g, int32_t *n_features, int32_t *n_matches);
int32_t vh_group_synchronize(vh_group *g);
int32_t vh_group_set_stream(vh_group *g, void *hip_stream);
int32_t vh_group_clear_stream(vh_group *g);
int32_t vh_group_stream_wait_images(vh_group *g, void *hip_stream);

/* ---- stereo egomotion ------------------------------------- */

/* VisualOdometryStereo::parameters and the calibration it reads
 * (src/viso_stereo.h:31-43, src/viso.h:41-50). */
typedef struct vh_ego_params {
  int32_t ransac_iters;     /* number of RANSAC iterations (200) */
  int32_t reweighting;      /* 1 = lower border weights (src/viso_stereo.cpp:280-282) */
  double inlier_threshold;  /* reprojection error bound in pixels (2.0) */
  double f, cu, cv, base;   /* focal length, principal point (pixels), baseline (meters) */
} vh_ego_params;
/* VisualOdometryStereo::parameters() defaults; f = 1, cu = cv = 0, base = 1 as VisualOdometry::calibration(). */
void vh_default_ego_params(vh_ego_params *e);

/* VisualOdometryStereo::estimateMotion (src/viso_stereo.cpp:54-157) for n_sets independent
 * match lists in one launch (one workgroup per list; RANSAC hypotheses in parallel, double
 * precision): pm = the lists back to back, list s = pm[offsets[s] .. offsets[s+1]).
 * rand3[n_sets][ransac_iters][3] = the values rand() returns while
 * VisualOdometry::getRandomSample(N,3) draws each hypothesis' sample (src/viso.cpp:86-106;
 * the reference seeds srand(0) in its constructor, src/viso.cpp:35) -- the caller owns the
 * random stream, so the result is a function of the inputs.
 * Outputs per list: tr[6] = (rx,ry,rz,tx,ty,tz) and ok = 1, or ok = 0 (and tr = 0) where the
 * reference returns an empty vector (fewer than 6 matches / inliers, refinement not
 * converged); n_inliers and, if inliers != NULL, the ascending inlier indices of the best
 * hypothesis at inliers[offsets[s] ..] (VisualOdometry::getInlierIndices).
 * Each hypothesis follows the reference's operation order exactly; the refinement sums the
 * normal equations in parallel, so tr agrees with the reference to rounding (1e-9 relative
 * is what tests/ assert), the inlier sets exactly.
 * The two stateless estimators (this and vh_estimate_motion_mono) keep one device work buffer per
 * device between calls (grow-only, requests above 1 GiB are not kept, released with the process)
 * and run one at a time per process. */
int32_t vh_estimate_motion_stereo(const vh_ego_params *e, int32_t device, int32_t n_sets, const vh_p_match *pm,
                                  const int32_t *offsets, const int32_t *rand3, double *tr, int32_t *ok,
                                  int32_t *n_inliers, int32_t *inliers);
/* The same on the device-resident match lists of the group's last vh_group_match_features
 * (flow matches carry no disparity: VH_METHOD_QUAD only, VH_ERR_STATE otherwise): nothing
 * but rand3 [S][ransac_iters][3] goes up and S x (tr, ok, n_inliers) comes back. */
int32_t vh_group_estimate_motion(vh_group *g, const vh_ego_params *e, const int32_t *rand3, double *tr, int32_t *ok,
                                 int32_t *n_inliers);

/* ---- the steps after matching, pipelined ------------------------------------------------------------- */
/* What the reference's loop runs between Matcher::matching and the pose -- removeOutliers (the tail of
 * matchFeatures, src/matcher.cpp:108), bucketFeatures (src/viso_stereo.cpp:41-43 -> src/matcher.cpp:140-187)
 * and VisualOdometryStereo::estimateMotion (src/viso_stereo.cpp:49-51) -- for every stream of a group,
 * arranged so that the host part of step t runs beside the GPU work of step t+1:
 *   vh_group_post_begin   after vh_group_match_features: starts the download of the step's match lists (the
 *                         first cap_per_stream records of every stream) into one of two internal page-locked
 *                         slots and returns at once.
 *   vh_group_post_finish  age 0: the step begun last, 1: the one before.  Waits for that download, runs the
 *                         Delaunay vote (flow and quad lists) and the bucketing of every stream on host_threads
 *                         host threads (<= 0: one per hardware thread), copies the bucketed lists to
 *                         bucketed[s * cap_per_stream ..] / counts[s] (either may be NULL), and -- if e != NULL
 *                         (quad lists only) -- uploads them (a few hundred records per stream) and runs the
 *                         batched egomotion kernel: tr[S][6], ok[S], n_inliers[S] as vh_group_estimate_motion,
 *                         rand3[S][ransac_iters][3].  *host_ms (nullable) = wall time of the host part.
 * VH_ERR_CAPACITY when a list was longer than the slot or a feature set was truncated.  The lists
 * vh_group_get_matches returns are not changed by these calls. */
int32_t vh_group_post_begin(vh_group *g, int32_t cap_per_stream);
int32_t vh_group_post_finish(vh_group *g, int32_t age, int32_t max_features, float bucket_width, float bucket_height,
                             int32_t host_threads, const vh_ego_params *e, const int32_t *rand3, double *tr, int32_t *ok,
                             int32_t *n_inliers, vh_p_match *bucketed, int32_t cap_per_stream, int32_t *counts,
                             double *host_ms);

/* ---- monocular egomotion (mono half) ---------------------------- */

/* VisualOdometryMono::parameters and the calibration it reads (src/viso_mono.h:32-46, src/viso.h:41-50). */
typedef struct vh_mono_params {
  int32_t ransac_iters;     /* number of RANSAC iterations (2000) */
  int32_t reserved_;        /* 0 */
  double inlier_threshold;  /* fundamental-matrix (Sampson distance) inlier threshold (0.00001) */
  double motion_threshold;  /* median depth above which the motion counts as too small (100.0) */
  double height, pitch;     /* camera height above ground (m), pitch (rad, negative = pointing down) */
  double f, cu, cv;         /* focal length and principal point (pixels) */
} vh_mono_params;
/* VisualOdometryMono::parameters() defaults; f = 1, cu = cv = 0 as VisualOdometry::calibration(). */
void vh_default_mono_params(vh_mono_params *e);

/* VisualOdometryMono::estimateMotion (src/viso_mono.cpp:41-160) for n_sets independent lists of flow
 * matches (u1p,v1p -> u1c,v1c; the other fields are not read): 8-point RANSAC on normalised points,
 * F from all inliers of the best hypothesis, E, the four (R,t) candidates with the chirality vote,
 * scale from the ground plane.  Same layout as vh_estimate_motion_stereo; rand8[n_sets][ransac_iters][8]
 * = the values rand() returns while getRandomSample(N,8) draws each hypothesis' sample
 * (src/viso.cpp:86-106).  ok = 0 (tr = 0) where the reference returns an empty vector (fewer than
 * 10 matches / inliers / points in front of the cameras, median depth above motion_threshold) --
 * and where it would call exit(0) (no chirality solution, division by a vanishing plane distance).
 * Every hypothesis, the refit and the triangulation follow the reference's operation order exactly
 * (Matrix::svd restated, src/matrix.cpp:579-802), so the inlier sets are equal to the reference's;
 * the ground-plane vote uses the device's exp and the angles its asin/cos, so tr agrees to rounding
 * (tests assert 1e-9 relative). */
int32_t vh_estimate_motion_mono(const vh_mono_params *e, int32_t device, int32_t n_sets, const vh_p_match *pm,
                                const int32_t *offsets, const int32_t *rand8, double *tr, int32_t *ok,
                                int32_t *n_inliers, int32_t *inliers);
/* The same on the device-resident match lists of the group's last vh_group_match_features
 * (VH_METHOD_FLOW or VH_METHOD_QUAD: both carry the left camera's flow). */
int32_t vh_group_estimate_motion_mono(vh_group *g, const vh_mono_params *e, const int32_t *rand8, double *tr,
                                      int32_t *ok, int32_t *n_inliers);
/* ---- the same chain ON THE DEVICE: no host work between matching and the pose (csrc/kernels_vote.hip) ----
 * removeOutliers' Delaunay triangulation is a sequential chain per match list (csrc/sweep_hull.h); on the GPU a list
 * takes tens of milliseconds as one lane, and the throughput comes from the lists in flight:
 *   vh_group_post_device_config  steps_per_batch (1..256) steps are voted on by one kernel sequence (steps_per_batch * S
 *                         lists, at most 65 535: beyond that vh_group_post_begin_device returns VH_ERR_UNSUPPORTED), up to `batches` (1..64) such batches are in flight on low-priority streams beside the
 *                         matcher's own kernels; lanes_per_wave (1..64) lists share a wavefront.  Default 64, 3, 16 (a batch takes
 *                         0.4-0.5 s whatever its size: the rate is steps in flight over that latency).
 *                         VH_ERR_STATE while steps are in flight.
 *                         Device memory of the ring: batches * steps_per_batch * S lists, each 176 bytes per record slot
 *                         (cap_per_stream slots: the record, its point, votes, order, a 16-byte hull node, six 16-byte
 *                         half-edge records) + 48 bytes per bucketed output record + the estimator's scratch -- 2.0 MB per
 *                         KITTI list of 11 363 slots.  The first vh_group_post_begin_device sizes the ring against the
 *                         device's free memory: steps_per_batch is halved until the ring fits 80 % of it, and
 *                         VH_ERR_CAPACITY is returned -- before anything has moved -- if one step per batch does not.
 *   vh_group_post_begin_device   after vh_group_match_features: the step's S lists leave the matcher's buffer for the
 *                         current batch (a device-to-device move; the matcher can go on at once); a full batch is launched:
 *                         vote -> Matcher::bucketFeatures(max_features, bucket_width, bucket_height) (bucket sides >= 1 px)
 *                         -> the stereo estimator (e, rand3[S][ransac_iters][3]) or the monocular one (mono,
 *                         rand8[S][ransac_iters][8]) or neither.  want_lists: keep the bucketed lists for the finish call.
 *                         All steps of a batch share one configuration (a different one closes the batch early).
 *   vh_group_post_finish_device  the step begun `age` begins ago (0: the last): waits for its batch (launching it first if it
 *                         is not full yet), then tr[S][6], ok[S], n_inliers[S] (with an estimator), counts[S] (nullable) and
 *                         bucketed[S][cap_per_stream] (nullable; needs want_lists) as vh_group_post_finish.  A caller that
 *                         finishes step t - steps_per_batch * (batches - 1) after beginning step t never waits for the vote.
 *                         Every step begun must be finished before the ring of steps_per_batch * batches steps comes
 *                         round (VH_ERR_STATE from the begin call otherwise).  A stream whose list was refused
 *                         (truncated: VH_ERR_CAPACITY; NaN / negative coordinates or an exhausted flip stack:
 *                         VH_ERR_UNSUPPORTED) reports ok = 0, n_inliers = 0, tr = 0, counts = -1; the other streams of
 *                         the step are delivered as usual and the call returns the error code.
 * Results per stream are those of vh_group_post_finish: lists and flags bit for bit, tr to rounding. */
int32_t vh_group_post_device_config(vh_group *g, int32_t steps_per_batch, int32_t batches, int32_t lanes_per_wave);
int32_t vh_group_post_begin_device(vh_group *g, int32_t cap_per_stream, int32_t max_features, float bucket_width, float bucket_height,
                                   const vh_ego_params *e, const int32_t *rand3, const vh_mono_params *mono, const int32_t *rand8,
                                   int32_t want_lists);
int32_t vh_group_post_finish_device(vh_group *g, int32_t age, double *tr, int32_t *ok, int32_t *n_inliers, vh_p_match *bucketed,
                                    int32_t cap_per_stream, int32_t *counts);

/* vh_group_post_finish with the MONOCULAR estimator as its last stage: what VisualOdometryMono::process runs
 * after the matching (src/viso_mono.cpp:34-37: bucketFeatures, then estimateMotion on the bucketed list; the
 * Delaunay vote before them is the tail of matchFeatures) for every stream of a group, flow or quad lists,
 * rand8[S][ransac_iters][8].  Everything else as vh_group_post_finish. */
int32_t vh_group_post_finish_mono(vh_group *g, int32_t age, int32_t max_features, float bucket_width, float bucket_height,
                                  int32_t host_threads, const vh_mono_params *e, const int32_t *rand8, double *tr,
                                  int32_t *ok, int32_t *n_inliers, vh_p_match *bucketed, int32_t cap_per_stream,
                                  int32_t *counts, double *host_ms);

/* Which form of the search loops the group currently runs and the last observed share of
 * queries the speculative form had to search again (-1 before the first report).  The
 * searches are exact either way; the library switches between a speculative loop (no accept
 * test per candidate, the winner verified afterwards) and the literal tested loop on that share
 * (DESIGN.md section 4.1).  VH_FLOW_TESTED=1 / 0 in the environment pins the choice. */
int32_t vh_group_search_stats(vh_group *g, int32_t *speculative, double *research_rate);

/* Test hook: the next device allocation the group makes fails (VH_ERR_HIP), once.  Lets the suite drive the error
 * paths of lazily allocated buffers (the flow method's pixel mask). */
int32_t vh_group_debug_fail_next_alloc(vh_group *g);
/* Test hook: flip-stack entries the device vote's sweep may hold per list (1..31; 0 restores the default, 31), process-wide.
 * With a small value ordinary match lists take the refusal path (VH_ERR_UNSUPPORTED for that list, see
 * vh_remove_outliers_device). */
int32_t vh_debug_vote_stack_slots(int32_t slots);
/* Kernel timing (HIP events recorded on the group's stream around every
 * kernel launch while enabled).  vh_group_profile_read returns the
 * accumulated milliseconds and launch count of kernel `name`
 * ("detect_nms", "emit_features", "bin_hist", "bin_scan", "bin_fill",
 *  "bin_sort", "match", "chain", "emit_matches") since the last reset. */
int32_t vh_group_profile_enable(vh_group *g, int32_t on);
int32_t vh_group_profile_read(vh_group *g, const char *name, double *ms, int64_t *launches);
int32_t vh_group_profile_reset(vh_group *g);

#ifdef __cplusplus
}
#endif
#endif /* VISO_HIP_H */
