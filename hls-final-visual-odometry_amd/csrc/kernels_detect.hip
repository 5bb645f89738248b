// kernels_detect.hip -- feature detection on gfx950.
//
// Replaces, for every image of every stream in one launch:
//   filter::blob5x5 / checkerboard5x5     (reference src/filter.cpp:445-467, :433-438)
//   Matcher::nonMaximumSuppression        (reference src/matcher.cpp:366-468)
//   filter::sobel5x5 + computeDescriptor  (reference src/filter.cpp:418-426,
//                                          src/matcher.cpp:470-514)
//   the packing loop of computeFeatures   (reference src/matcher.cpp:663-671)
//
// Design (integer work, no MFMA; see DESIGN.md section 4 for what bounds each kernel):
//   detect_nms    one workgroup per tile of 32x8 NMS blocks: the image tile
//                 (+halo) is staged in LDS once (all loads of a lane in one
//                 batch), the blob and checkerboard responses are produced into
//                 LDS and never touch HBM; NMS runs on the LDS tile: block
//                 extrema and threshold per lane, then, for the queued few
//                 that passed it, "candidate == minimum of its clipped
//                 (2n+1)^2 window" -- equivalent to the reference's
//                 per-candidate dominance scan -- and leaves 8 bytes per block
//                 (4 x u16 position codes).
//                 detect_nms_fast<N> (nms_n 1..4, 4-byte aligned rows) is the
//                 compile-time specialised form: filters on packed 16-bit row
//                 sums, 2 or 4 columns per lane, responses stored + 8192; block
//                 extrema by packed min/max when nms_n is odd.
//                 detect_nms_kernel is the generic one (any nms_n, any stride)
//                 with one column per lane and the literal scan.
//   emit_features ordered compaction of those codes (reference output order:
//                 block row-major, class ascending), bin slots and (class, v)
//                 row ranks (first half of createIndexVector), then the
//                 32-byte descriptor of each survivor is computed from its
//                 15x15 image patch (fetched once, shared through LDS; 16 lanes
//                 per feature, one sample point each): the Sobel planes the
//                 reference materialises are never written.
//   Debug builds: -DVH_EMIT_TIMING[=2] puts phase clocks into emit_features
//                 [detect_nms_fast] (tools/emit_timing.py).
#include "vh_dev.h"
#include <type_traits>

namespace {

// ---------------------------------------------------------------- half-res
// Matcher::createHalfResolutionImage (reference src/matcher.cpp:572-583);
// padding columns [Wm,bplm) are written as zero.
__global__ void half_res_kernel(VhImages src, uint8_t *__restrict__ dst, VhGeom g) {
  const int32_t id = blockIdx.z;
  const uint8_t *__restrict__ I = vh_image_ptr(src, id);
  uint8_t *__restrict__ O = dst + (int64_t)id * g.bplm * g.Hm;
  const int32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  const int32_t v = blockIdx.y;
  if (u >= g.bplm) return;
  uint32_t val = 0;
  if (u < g.Wm) {
    const int64_t r0 = (int64_t)(2 * v) * g.bpl + 2 * u, r1 = r0 + g.bpl;
    val = ((uint32_t)I[r0] + I[r0 + 1] + I[r1] + I[r1 + 1]) >> 2;
  }
  O[(int64_t)v * g.bplm + u] = (uint8_t)val;
}

// --------------------------------------------------------------- detect_nms
// LDS: sI[IH][IWp] u8 | sF1[FH][FWp] i16 | sF2[FH][FWp] i16 | work-list count, entries | codes
// (position code of a survivor = row-in-block << 6 | column-in-block)
//
// Tile geometry (n = nms_n): the tile owns tbx x tby NMS blocks whose first
// pixel is (n+7 + bx0*(n+1), n+7 + by0*(n+1)); responses are needed n pixels
// around the blocks and the image 2 pixels around the responses.
__global__ void __launch_bounds__(256)
detect_nms_kernel(VhImages im, VhGeom g, uint64_t *__restrict__ rec, int32_t *__restrict__ chunk_count) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t *sI = smem;
  int16_t *sF1 = (int16_t *)(smem + (size_t)g.IH * g.IWp);
  int16_t *sF2 = sF1 + (size_t)g.FH * g.FWp;

  const int32_t id = blockIdx.z;
  const uint8_t *__restrict__ I = vh_image_ptr(im, id);
  const int32_t n = g.n, n1 = n + 1;
  const int32_t bx0 = blockIdx.x * g.tbx, by0 = blockIdx.y * g.tby;
  const int32_t fx0 = VH_MARGIN + bx0 * n1, fy0 = VH_MARGIN + by0 * n1;  // response tile origin
  const int32_t ix0 = fx0 - 2, iy0 = fy0 - 2;                            // image tile origin
  const int32_t tid = threadIdx.x;

  // 1. stage the image tile (rows/cols outside the image read as 0; they only
  //    feed responses that NMS never looks at)
  for (int32_t k = tid; k < g.IH * g.IWp; k += 256) {
    const int32_t r = k / g.IWp, c = k - r * g.IWp;
    const int32_t gx = ix0 + c, gy = iy0 + r;
    uint8_t v = 0;
    if (c < g.IW && gx < g.bplm && gy < g.Hm) v = I[(int64_t)gy * g.bplm + gx];
    sI[k] = v;
  }
  __syncthreads();

  // 2. blob (f1) and checkerboard (f2) responses, separable sliding window
  //    down each column.  Row sums of image-tile row y around centre column cx+2:
  //      h5 = a+b+c+d+e, h3 = b+c+d, hc = a+b-d-e
  //    f2(y) = hc(y-2)+hc(y-1)-hc(y+1)-hc(y+2)              (filter.cpp:339-347,:365-367)
  //    f1(y) = -sum5(h5) + 2*sum3(h3) + 7*c(y)               (filter.cpp:461-463)
  {
    const int32_t nseg = max(1, 256 / g.FW);
    const int32_t rows_per = (g.FH + nseg - 1) / nseg;
    for (int32_t task = tid; task < g.FW * nseg; task += 256) {
      const int32_t seg = task / g.FW, cx = task - seg * g.FW;
      const int32_t r0 = seg * rows_per, r1 = min(g.FH, r0 + rows_per);
      if (r0 >= r1) continue;
      int32_t h5[5], h3[5], hc[5], cc[5];
      // response row fr uses image-tile rows fr .. fr+4 (centre fr+2)
#pragma unroll
      for (int32_t k = 0; k < 4; k++) {
        const uint8_t *p = sI + (size_t)(r0 + k) * g.IWp + cx;
        const int32_t a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
        h5[k + 1] = a + b + c + d + e; h3[k + 1] = b + c + d; hc[k + 1] = a + b - d - e; cc[k + 1] = c;
      }
      for (int32_t fr = r0; fr < r1; fr++) {
#pragma unroll
        for (int32_t k = 0; k < 4; k++) { h5[k] = h5[k + 1]; h3[k] = h3[k + 1]; hc[k] = hc[k + 1]; cc[k] = cc[k + 1]; }
        const uint8_t *p = sI + (size_t)(fr + 4) * g.IWp + cx;
        const int32_t a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
        h5[4] = a + b + c + d + e; h3[4] = b + c + d; hc[4] = a + b - d - e; cc[4] = c;
        const int32_t f2 = hc[0] + hc[1] - hc[3] - hc[4];
        const int32_t f1 = -(h5[0] + h5[1] + h5[2] + h5[3] + h5[4]) + 2 * (h3[1] + h3[2] + h3[3]) + 7 * cc[2];
        sF1[(size_t)fr * g.FWp + cx] = (int16_t)f1;
        sF2[(size_t)fr * g.FWp + cx] = (int16_t)f2;
      }
    }
  }
  __syncthreads();

  // 3. NMS (Neubeck/Van Gool alg. 4, matcher.cpp:381-466) in two steps so that
  //    lanes stay busy: (a) one lane per block finds the block's four extrema and
  //    applies the threshold -- ~95 % of (block, class) candidates die here --
  //    and appends survivors to an LDS work list; (b) the list is processed
  //    densely, one lane per surviving candidate, for the (2n+1)^2 dominance test.
  int32_t *sCnt = (int32_t *)(sF2 + (size_t)g.FH * g.FWp);      // work-list length
  uint32_t *sWork = (uint32_t *)(sCnt + 1);                      // [4*tbx*tby] ex | ey<<10 | c<<20 | lane<<22
  uint16_t *sCode = (uint16_t *)(sWork + 4 * g.tbx * g.tby);     // [tbx*tby][4] position codes
  const int32_t nblk_tile = g.tbx * g.tby;
  if (tid == 0) *sCnt = 0;
  for (int32_t k = tid; k < 4 * nblk_tile; k += 256) sCode[k] = VH_NO_CODE;
  __syncthreads();
  // clip limits W-1-margin / H-1-margin in response-tile coords (matcher.cpp:420-421)
  const int32_t xlim = (g.Wm - 1 - VH_MARGIN) - fx0, ylim = (g.Hm - 1 - VH_MARGIN) - fy0;
  const int32_t lby = tid / g.tbx, lbx = tid - lby * g.tbx;
  const int32_t bx = bx0 + lbx, by = by0 + lby;
  const bool have_block = tid < nblk_tile && bx < g.nbx && by < g.nby;
  if (have_block) {
    const int32_t fx = n + lbx * n1, fy = n + lby * n1;  // block origin in response-tile coords
    int32_t ex[4], ey[4], ev[4];
    ex[0] = ex[1] = ex[2] = ex[3] = fx;
    ey[0] = ey[1] = ey[2] = ey[3] = fy;
    ev[0] = ev[1] = sF1[(size_t)fy * g.FWp + fx];
    ev[2] = ev[3] = sF2[(size_t)fy * g.FWp + fx];
    for (int32_t j2 = fy; j2 <= fy + n; j2++) {
      for (int32_t i2 = fx; i2 <= fx + n; i2++) {
        int32_t cur = sF1[(size_t)j2 * g.FWp + i2];
        if (cur < ev[0]) { ex[0] = i2; ey[0] = j2; ev[0] = cur; }       // first extremum in scan order wins
        else if (cur > ev[1]) { ex[1] = i2; ey[1] = j2; ev[1] = cur; }  // (matcher.cpp:397-405)
        cur = sF2[(size_t)j2 * g.FWp + i2];
        if (cur < ev[2]) { ex[2] = i2; ey[2] = j2; ev[2] = cur; }
        else if (cur > ev[3]) { ex[3] = i2; ey[3] = j2; ev[3] = cur; }
      }
    }
#pragma unroll
    for (int32_t c = 0; c < 4; c++) {
      // threshold (matcher.cpp:427,439,451,463); order vs the dominance test is immaterial
      const bool pass = (c & 1) ? (ev[c] >= g.tau) : (ev[c] <= -g.tau);
      if (pass) sWork[atomicAdd(sCnt, 1)] = (uint32_t)ex[c] | ((uint32_t)ey[c] << 10) | ((uint32_t)c << 20) | ((uint32_t)tid << 22);
    }
  }
  __syncthreads();
  const int32_t nwork = *sCnt;
  for (int32_t w = tid; w < nwork; w += 256) {
    const uint32_t e = sWork[w];
    const int32_t cx = e & 1023, cy = (e >> 10) & 1023, c = (e >> 20) & 3, owner = e >> 22;
    const int32_t oby = owner / g.tbx, obx = owner - oby * g.tbx;
    const int32_t fx = n + obx * n1, fy = n + oby * n1;
    const int16_t *F = (c < 2) ? sF1 : sF2;
    const bool is_min = (c & 1) == 0;
    const int32_t val = F[(size_t)cy * g.FWp + cx];
    const int32_t jhi = min(cy + n, ylim), ihi = min(cx + n, xlim);
    bool ok = true;
    for (int32_t j2 = cy - n; j2 <= jhi && ok; j2++) {
      for (int32_t i2 = cx - n; i2 <= ihi; i2++) {
        const int32_t cur = F[(size_t)j2 * g.FWp + i2];
        const bool outside = (i2 < fx) | (i2 > fx + n) | (j2 < fy) | (j2 > fy + n);
        if (outside && (is_min ? (cur < val) : (cur > val))) { ok = false; break; }
      }
    }
    if (ok) sCode[owner * 4 + c] = (uint16_t)(((cy - fy) << 6) | (cx - fx));
  }
  __syncthreads();
  if (have_block) {
    uint64_t code = 0;
    int32_t cnt = 0;
#pragma unroll
    for (int32_t c = 0; c < 4; c++) {
      const uint32_t pc = sCode[tid * 4 + c];
      code |= (uint64_t)pc << (16 * c);
      cnt += pc != VH_NO_CODE ? 1 : 0;
    }
    const int32_t blk = by * g.nbx + bx;
    rec[(int64_t)id * g.nblocks + blk] = code;
    if (cnt) atomicAdd(&chunk_count[(int64_t)id * g.nchunks + blk / VH_CHUNK], cnt);
  }
}

// ---------------------------------------------------------- detect_nms (fast)
// Same algorithm with nms_n as a compile-time constant (the values that occur in
// practice: 1..4) so that every LDS offset is an immediate and every loop is
// unrolled, and with the image tile staged by aligned 4-byte loads.  Requires
// 4-byte aligned rows (base, bpl and stream stride multiples of 4); anything
// else takes the generic kernel above.
#ifdef VH_EMIT_TIMING
// debug build only (EXTRA=-DVH_EMIT_TIMING; -DVH_EMIT_TIMING=2 puts the clocks into detect_nms_fast instead): s_memtime
// ticks the workgroups spent per phase, summed over workgroups.  emit_features: [0] prefix [1] phase A [2] A2 row ranks
// [3] bin slots + staging [4] phase B; detect_nms_fast: [0] image tile [1] filters [2] block extrema [3] window checks
// [4] records; [5] workgroups;
// [6] earliest start, [7] latest end (one launch between two resets)
__device__ unsigned long long g_emit_t[8];
extern "C" int32_t vh_debug_emit_timing(unsigned long long *out, int32_t reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_emit_t), sizeof(g_emit_t)) != hipSuccess) return -3;
  if (reset) { unsigned long long z[8] = {0}; z[6] = ~0ull; if (hipMemcpyToSymbol(HIP_SYMBOL(g_emit_t), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
// (one workgroup in 61 carries the clocks: same-address atomics from every workgroup would be the longest phase)
#define VH_TICK_IMPL(k) do { __syncthreads(); if (t_on_) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&g_emit_t[k], now_ - t_prev_); t_prev_ = now_; if ((k) == 4) atomicMax(&g_emit_t[7], now_); } } while (0)
#define VH_TICK_INIT_IMPL const bool t_on_ = threadIdx.x == 0 && (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) % 61 == 0; unsigned long long t_prev_ = __builtin_readcyclecounter(); if (t_on_) { atomicAdd(&g_emit_t[5], 1ull); atomicMin(&g_emit_t[6], t_prev_); }
#endif
#if defined(VH_EMIT_TIMING) && VH_EMIT_TIMING == 2
#define VH_DTICK(k) VH_TICK_IMPL(k)
#define VH_DTICK_INIT VH_TICK_INIT_IMPL
#else
#define VH_DTICK(k) do { } while (0)
#define VH_DTICK_INIT do { } while (0)
#endif
#if defined(VH_EMIT_TIMING) && VH_EMIT_TIMING != 2
#define VH_ETICK(k) VH_TICK_IMPL(k)
#define VH_ETICK_INIT VH_TICK_INIT_IMPL
#else
#define VH_ETICK(k) do { } while (0)
#define VH_ETICK_INIT do { } while (0)
#endif

template <int N> struct DetTile {
  static constexpr int N1 = N + 1, TBX = 32, TBY = 8;
  static constexpr int FW = TBX * N1 + 2 * N, FH = TBY * N1 + 2 * N;
  static constexpr int IW = FW + 4, IH = FH + 4;
  // Tile origins are 32*N1 columns apart, so response column 0 always sits at byte OFF of an aligned dword
  static constexpr int OFF = (VH_MARGIN - 2) & 3;
  static constexpr int DW = (IW + 3 + 3) / 4;  // dwords per staged row (room for the alignment offset)
  static constexpr int IP = DW * 4;
  // The filter pass works on groups of PX = 2 or 4 adjacent response columns, two per register: column
  // cx is column OFF + cx of the staged row and of the response rows (pitch FP, int16).  A lane
  // filters one group over ROWS consecutive rows after 4 rows of run-in, SEG = number of row
  // segments.  PX and SEG are picked for the shortest pass PER WAVE (then the fewest waves): a
  // workgroup's four waves sit on the four SIMDs of its CU, and a pass that keeps only two of them
  // busy loads two SIMDs with everybody's filter pass -- at nms_n = 2 the fewest wave instructions
  // in total (PX 4, 3 segments, 2 waves) measured 410 us per S = 256 launch, the balanced PX 4
  // layout (9 segments) 387 us.  Instruction costs per row and lane: row pass 28 (PX 4: 7 permutes,
  // 14 adds) or 18 (PX 2: 5 permutes, 8 adds), column pass 26 or 13.
  static constexpr int groups(int px) { return (FW + OFF + px - 1) / px; }
  static constexpr int seg_cost(int px, int s) {
    const int rows = (FH + s - 1) / s, waves = (groups(px) * s + 63) / 64;
    const int hrow = px == 4 ? 28 : 18, vrow = px == 4 ? 26 : 13;
    return groups(px) * s > 256 || rows * (s - 1) >= FH ? (1 << 30) : 1000 * ((rows + 4) * hrow + rows * vrow) + waves;
  }
  static constexpr int best_seg(int px) {
    int b = 1;
    for (int s = 2; s <= 16; s++)
      if (seg_cost(px, s) < seg_cost(px, b)) b = s;
    return b;
  }
#ifdef VH_DET_PX  // experiments: force the filter-pass layout (EXTRA="-DVH_DET_PX=4 -DVH_DET_SEG=4")
  static constexpr int PX = VH_DET_PX;
  static constexpr int SEG = VH_DET_SEG;
#else
  static constexpr int PX = seg_cost(2, best_seg(2)) < seg_cost(4, best_seg(4)) ? 2 : 4;
  static constexpr int SEG = best_seg(PX);
#endif
  static constexpr int G = groups(PX);
  static constexpr int FP = PX * G;
  static constexpr int ROWS = (FH + SEG - 1) / SEG;
  static constexpr int BIAS = 8192;  // stored responses are f + BIAS (positive int16; the NMS only compares)
  static_assert((G - 1) * PX / 4 + 1 <= DW - 1, "a group reads the staged dword it starts in and the next one");
  static_assert(G * SEG <= 256, "one lane per (group, row segment)");
  static constexpr int LAST = FH - (SEG - 1) * ROWS;  // rows of the last segment
  static_assert(LAST >= 1, "every segment has rows");
};

template <int N>
__global__ void __launch_bounds__(256)
detect_nms_fast_kernel(VhImages im, VhGeom g, uint64_t *__restrict__ rec, int32_t *__restrict__ chunk_count) {
  using T = DetTile<N>;
  constexpr int N1 = T::N1, WN = 2 * N + 1;
  VH_DET_SETPRIO();
  constexpr int X_BYTES = (T::IH * T::IP > 6144) ? T::IH * T::IP : 6144;  // image tile, later queues (4 KB) + codes (2 KB)
  __shared__ __attribute__((aligned(16))) int16_t sF1[T::FH * T::FP];
  __shared__ __attribute__((aligned(16))) int16_t sF2[T::FH * T::FP];
  // scratch: first the staged image tile, later the NMS candidate queues and pass flags
  __shared__ __attribute__((aligned(16))) uint8_t sX[X_BYTES];
  uint8_t *sI = sX;

  const int32_t id = blockIdx.z, tid = threadIdx.x;
  const uint8_t *__restrict__ I = vh_image_ptr(im, id);
  const int32_t bx0 = blockIdx.x * T::TBX, by0 = blockIdx.y * T::TBY;
  const int32_t fx0 = VH_MARGIN + bx0 * N1, fy0 = VH_MARGIN + by0 * N1;
  const int32_t ix0 = fx0 - 2, iy0 = fy0 - 2;
  const int32_t ax0 = ix0 - T::OFF;  // a multiple of 4 (DetTile::OFF)
  VH_DTICK_INIT;

  // 1. stage the image tile with aligned dword loads (zero outside the image).  All loads of a lane
  //    are issued before the first is stored: as a rolled load-store loop this stage was 4 dependent
  //    HBM round trips and 40 % of a workgroup's life.  Out-of-image dwords are loaded from a clamped
  //    (valid) address and zeroed afterwards, so that no load sits behind a branch.
  {
    constexpr int NLD = (T::IH * T::DW + 255) / 256;
    uint32_t stg[NLD];
    bool inside[NLD];
#pragma unroll
    for (int32_t i = 0; i < NLD; i++) {
      const int32_t k = min(tid + 256 * i, T::IH * T::DW - 1);
      const int32_t r = k / T::DW, c = k - r * T::DW;
      const int32_t gy = iy0 + r, gx = ax0 + 4 * c;
      stg[i] = *(const uint32_t *)(I + (int64_t)min(gy, g.Hm - 1) * g.bplm + min(gx, g.bplm - 4));
      inside[i] = gy < g.Hm && gx < g.bplm;
    }
    __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler pairs each load with its store again)
#pragma unroll
    for (int32_t i = 0; i < NLD; i++) {
      const int32_t k = tid + 256 * i;
      if (k < T::IH * T::DW) ((uint32_t *)sI)[k] = inside[i] ? stg[i] : 0u;
    }
  }
  __syncthreads();

  VH_DTICK(0);
  // 2. blob / checkerboard responses, PX columns per lane and two columns per register: the row
  //    sums live in the 16-bit halves of a dword and are added with plain 32-bit adds (full-rate
  //    instructions; every field stays in [0, 65535] through every intermediate, so no carry or
  //    borrow crosses the halves).  f2 = (1,1,0,-1,-1)^T (x) (1,1,0,-1,-1) (filter.cpp:339-347,
  //    :365-367), f1 = -S5x5 + 2*S3x3 + 7*centre (filter.cpp:461-463); both are stored + BIAS.
  //    Per row and lane: one 8-byte LDS read, byte permutes for the pairs (b_k, b_k+1) of the
  //    staged bytes and adds for the row sums, then the column pass and one store per plane --
  //    about a third of one-column-per-lane with byte loads.
  if (tid < T::G * T::SEG) {
    constexpr int PX = T::PX, NR = PX / 2;
    const int32_t seg = tid / T::G, gq = tid - seg * T::G;
    const int32_t r0 = seg * T::ROWS;
    {
      constexpr uint32_t HB = 512u * 0x00010001u, KB = (uint32_t)T::BIAS * 0x00010001u;
      const uint32_t *p = (const uint32_t *)sI + r0 * T::DW + (gq * PX) / 4;
      // PX 2: a group starts at byte 0 or 2 of its staged dword
      const uint32_t sel0 = 0x0c010c00u + ((PX == 2 && (gq & 1)) ? 0x00020002u : 0u);
      uint32_t *o1 = (uint32_t *)sF1 + (r0 * T::FP + PX * gq) / 2, *o2 = (uint32_t *)sF2 + (r0 * T::FP + PX * gq) / 2;
      // rolling state per register (register 0: columns 0,1 of the group; 1: columns 2,3)
      uint32_t h5[NR][5], h3[NR][5], pc[NR][4], cc[NR][5], hcp[NR], S5[NR];
#pragma unroll
      for (int32_t h = 0; h < NR; h++) S5[h] = 0u;
#pragma unroll
      for (int32_t t = 0; t < T::ROWS + 4; t++) {
        if (t - 4 < T::LAST || r0 + t - 4 < T::FH) {  // only the last segment can be short
          const uint32_t A = p[t * T::DW], B = p[t * T::DW + 1];
          // pk = bytes (k, k+1) of the group's staged bytes, zero-extended to the halves
          uint32_t pk[PX + 3];
#pragma unroll
          for (int32_t k = 0; k < PX + 3; k++) pk[k] = __builtin_amdgcn_perm(B, A, sel0 + (uint32_t)k * 0x00010001u);
          uint32_t n5[NR], n3[NR], nc[NR], nz[NR];
          if (PX == 4) {
            const uint32_t a01 = pk[0] + pk[1], a23 = pk[2] + pk[3], a34 = pk[3] + pk[4], a56 = pk[5] + pk[6];
            n3[0] = a23 + pk[1];             n3[NR - 1] = a34 + pk[5];                   // b+c+d
            n5[0] = n3[0] + pk[0] + pk[4];   n5[NR - 1] = n3[NR - 1] + pk[2] + pk[6];    // a+b+c+d+e
            nc[0] = (a01 + HB) - a34;        nc[NR - 1] = (a23 + HB) - a56;              // a+b-d-e + 512
            nz[0] = pk[2];                   nz[NR - 1] = pk[4];                         // c
          } else {
            n3[0] = pk[1] + pk[2] + pk[3];
            n5[0] = n3[0] + pk[0] + pk[4];
            nc[0] = ((pk[0] + pk[1]) + HB) - (pk[3] + pk[4]);
            nz[0] = pk[2];
          }
          uint32_t f1[NR], f2[NR];
#pragma unroll
          for (int32_t h = 0; h < NR; h++) {
            // slots 0..4 = staged rows t-4..t, the 5-row window of response row t-4
            const uint32_t old5 = h5[h][0];  // row t-5
#pragma unroll
            for (int32_t k = 0; k < 4; k++) { h5[h][k] = h5[h][k + 1]; h3[h][k] = h3[h][k + 1]; cc[h][k] = cc[h][k + 1]; }
            h5[h][4] = n5[h]; h3[h][4] = n3[h]; cc[h][4] = nz[h];
            S5[h] += n5[h];
            if (t >= 5) S5[h] -= old5;
            // pc slots 0..3 = hc(r) + hc(r+1) (+1024) for r = t-4..t-1
#pragma unroll
            for (int32_t k = 0; k < 3; k++) pc[h][k] = pc[h][k + 1];
            if (t > 0) pc[h][3] = hcp[h] + nc[h];
            hcp[h] = nc[h];
            if (t >= 4) {
              const uint32_t S3 = h3[h][1] + h3[h][2] + h3[h][3];
              f2[h] = (pc[h][0] + KB) - pc[h][3];
              f1[h] = ((S3 + S3) + (__umul24(cc[h][2], 7u) + KB)) - S5[h];
            }
          }
          if (t >= 4) {
            if (PX == 4) {
              *(uint2 *)(o1 + (t - 4) * (T::FP / 2)) = make_uint2(f1[0], f1[NR - 1]);
              *(uint2 *)(o2 + (t - 4) * (T::FP / 2)) = make_uint2(f2[0], f2[NR - 1]);
            } else {
              o1[(t - 4) * (T::FP / 2)] = f1[0];
              o2[(t - 4) * (T::FP / 2)] = f2[0];
            }
          }
        }
      }
    }
  }
  __syncthreads();  // responses complete; the image tile is dead from here on
  VH_DTICK(1);

  // 3. NMS (Neubeck/Van Gool alg. 4, matcher.cpp:381-466).  The dominance test
  //    "no strictly smaller value in the (2n+1)^2 window outside the block"
  //    (matcher.cpp:420-426) is, because the candidate is already the minimum of
  //    its own block, the same as "the candidate equals the minimum of the whole
  //    window".  The window is clipped at W-1-margin / H-1-margin on the high
  //    side only, which row/column replication reproduces.  (a) One lane per
  //    block finds the four block extrema (first in scan order,
  //    matcher.cpp:393-417) and applies the threshold (matcher.cpp:427,439,451,
  //    463); only ~13 % of them pass it on typical images, and those are queued
  //    in LDS.  (b) The queued candidates are checked one per lane by reading
  //    their whole window: far fewer instructions than window extrema for every
  //    pixel, and the queue keeps all lanes of the checking waves busy.
  const int32_t xlim = min(max((g.Wm - 1 - VH_MARGIN) - fx0, 0), T::FW - 1);
  const int32_t ylim = min(max((g.Hm - 1 - VH_MARGIN) - fy0, 0), T::FH - 1);
  const int32_t lby = tid / T::TBX, lbx = tid % T::TBX;
  const int32_t bx = bx0 + lbx, by = by0 + lby;
  const bool have_block = bx < g.nbx && by < g.nby;
  const int32_t fx = N + lbx * N1, fy = N + lby * N1;
  // queues: [0] minima, [1] maxima; entry = owner | type << 8 | (PACKED ? extremum : row << 3 | column of it in the block) << 16
  // With odd N a block row is N1/2 aligned dwords of the response rows, so (a) takes the block
  // extrema with packed 16-bit min/max over whole dwords -- the VALUE only; where in the block it
  // first occurs is looked up in (b), for the ~13 % that are candidates at all.
  constexpr bool PACKED = (N & 1) != 0;
  uint32_t *sQueue = (uint32_t *)sX;             // [2][512]
  uint16_t *sCode = (uint16_t *)(sQueue + 1024); // [256][4]: position code of each (block, type) that survives
  __shared__ int32_t sQueueN[2];
  if (tid < 2) sQueueN[tid] = 0;
  ((uint2 *)sCode)[tid] = make_uint2(VH_NO_CODE * 0x00010001u, VH_NO_CODE * 0x00010001u);
  __syncthreads();
  {
    typedef int16_t i16x2 __attribute__((ext_vector_type(2)));
    const int32_t lane = tid & 63;
#pragma unroll
    for (int32_t plane = 0; plane < 2; plane++) {
      const int16_t *b = (plane ? sF2 : sF1) + fy * T::FP + fx + T::OFF;
      int32_t vn, vx, pn = 0, px = 0;
      if (PACKED) {
        const i16x2 *bw = (const i16x2 *)b;  // fx + OFF and FP are even
        i16x2 mn = bw[0], mx = mn;
#pragma unroll
        for (int32_t j = 0; j < N1; j++) {
#pragma unroll
          for (int32_t d = 0; d < N1 / 2; d++) {
            if (d == 0 && j == 0) continue;
            const i16x2 w = bw[j * (T::FP / 2) + d];
            mn = __builtin_elementwise_min(mn, w);
            mx = __builtin_elementwise_max(mx, w);
          }
        }
        vn = min((int32_t)mn.x, (int32_t)mn.y);
        vx = max((int32_t)mx.x, (int32_t)mx.y);
      } else {
        // value and scan position in one key: min over (value << 6 | kk) is the smallest value at its FIRST
        // position in scan order, max over (value << 6 | 63 - kk) the largest value at its first position --
        // the reference's strict `<` / `else if >` updates (matcher.cpp:397-405; the `else` never matters:
        // a value below the running minimum cannot exceed the running maximum).  Stored responses are
        // f + 8192 in [0, 16383], kk = row << 3 | column <= 36.  One v_lshl_add per key and one
        // v_min3 / v_max3 per two pixels instead of two compares and four selects per pixel.
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
        for (int32_t j = 0; j < N1; j++) {
#pragma unroll
          for (int32_t i = 0; i < N1; i++) {
            const uint32_t cur = (uint32_t)(uint16_t)b[j * T::FP + i], kk = (uint32_t)((j << 3) | i);
            kmin = min(kmin, (cur << 6) + kk);
            kmax = max(kmax, (cur << 6) + (63u - kk));
          }
        }
        vn = (int32_t)(kmin >> 6); pn = (int32_t)(kmin & 63u);
        vx = (int32_t)(kmax >> 6); px = 63 - (int32_t)(kmax & 63u);
      }
#pragma unroll
      for (int32_t mm = 0; mm < 2; mm++) {  // 0: the block minimum, 1: the block maximum
        const bool cand = have_block && (mm ? vx >= T::BIAS + g.tau : vn <= T::BIAS - g.tau);
        const uint32_t payload = PACKED ? (uint32_t)(mm ? vx : vn) : (uint32_t)(mm ? px : pn);
        const uint64_t bal = __ballot(cand);
        if (bal) {  // wave-uniform
          int32_t base = 0;
          if (lane == 0) base = atomicAdd(&sQueueN[mm], (int32_t)__popcll(bal));
          base = __builtin_amdgcn_readfirstlane(base);
          const int32_t rank = (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
          if (cand) sQueue[mm * 512 + base + rank] = (uint32_t)tid | ((uint32_t)(2 * plane + mm) << 8) | (payload << 16);
        }
      }
    }
  }
  __syncthreads();
  VH_DTICK(2);
  {
    // waves 0-1 check the minima, waves 2-3 the maxima.  Tiles that do not touch
    // the high-side clipping limits (almost all) address the window relative to
    // its top-left corner, so that every read offset is an immediate.
    const int32_t mm = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int32_t nq = sQueueN[mm];
    const bool unclipped = xlim == T::FW - 1 && ylim == T::FH - 1;
    auto check = [&](auto is_max, auto no_clip) {
      for (int32_t e = tid & 127; e < nq; e += 128) {
        const uint32_t q = sQueue[mm * 512 + e];
        const int32_t owner = q & 255, type = (q >> 8) & 3;
        const int32_t ofy = N + (owner / T::TBX) * N1, ofx = N + (owner % T::TBX) * N1;  // the owner's block
        const int16_t *F = ((type & 2) ? sF2 : sF1) + T::OFF;
        int32_t bj = 0, bi = 0;  // the extremum's row and column in the block
        if (PACKED) {
          // first occurrence of the extremum in scan order (matcher.cpp:393-417: strict comparisons, the first one stays)
          const int32_t val = (int32_t)(q >> 16);
          const int16_t *bb = F + ofy * T::FP + ofx;
#pragma unroll
          for (int32_t idx = N1 * N1 - 1; idx >= 0; idx--) {
            const bool eq = bb[(idx / N1) * T::FP + idx % N1] == val;
            bj = eq ? idx / N1 : bj; bi = eq ? idx % N1 : bi;
          }
        } else {
          bj = (q >> 19) & 7; bi = (q >> 16) & 7;
        }
        const int32_t cy = ofy + bj, cx = ofx + bi;
        int32_t v[WN * WN];
        if (decltype(no_clip)::value) {
          const int16_t *w = F + (cy - N) * T::FP + (cx - N);
#pragma unroll
          for (int32_t j = 0; j < WN; j++)
#pragma unroll
            for (int32_t k = 0; k < WN; k++) v[j * WN + k] = w[j * T::FP + k];
        } else {
          int32_t col[WN];
#pragma unroll
          for (int32_t k = 0; k < WN; k++) col[k] = min(cx - N + k, xlim);
#pragma unroll
          for (int32_t j = 0; j < WN; j++) {
            const int16_t *row = F + min(cy - N + j, ylim) * T::FP;
#pragma unroll
            for (int32_t k = 0; k < WN; k++) v[j * WN + k] = row[col[k]];
          }
        }
        const int32_t centre = v[N * WN + N];
        int32_t ext = centre;
#pragma unroll
        for (int32_t k = 0; k < WN * WN; k++) ext = decltype(is_max)::value ? max(ext, v[k]) : min(ext, v[k]);
        if (ext == centre) sCode[owner * 4 + type] = (uint16_t)((bj << 6) | bi);
      }
    };
    if (mm) { if (unclipped) check(std::true_type{}, std::true_type{}); else check(std::true_type{}, std::false_type{}); }
    else { if (unclipped) check(std::false_type{}, std::true_type{}); else check(std::false_type{}, std::false_type{}); }
  }
  __syncthreads();
  VH_DTICK(3);
  uint32_t codes[4];
  {
    const uint2 cw = ((const uint2 *)sCode)[tid];
    codes[0] = cw.x & 0xFFFFu; codes[1] = cw.x >> 16; codes[2] = cw.y & 0xFFFFu; codes[3] = cw.y >> 16;
  }

  // 4. 8 bytes per block + the per-chunk survivor counts.  The counts are first
  //    accumulated in LDS and flushed with one global atomic per (workgroup,
  //    chunk): one atomic per block put ~8 k same-address atomics per image on 48
  //    counters and cost half of this kernel's time.
  __shared__ int32_t sChunk[16];  // a tile's 8 block rows span < 16 chunks of 1024 blocks (nbx <= 2048 checked by the launcher)
  const int32_t chunk_lo = (by0 * g.nbx + bx0) / VH_CHUNK;
  if (tid < 16) sChunk[tid] = 0;
  __syncthreads();
  if (have_block) {
    int32_t cnt = 0;
#pragma unroll
    for (int32_t c = 0; c < 4; c++) cnt += codes[c] != VH_NO_CODE ? 1 : 0;
    const int32_t blk = by * g.nbx + bx;
    rec[(int64_t)id * g.nblocks + blk] = (uint64_t)(codes[0] | (codes[1] << 16)) | ((uint64_t)(codes[2] | (codes[3] << 16)) << 32);
    if (cnt) atomicAdd(&sChunk[blk / VH_CHUNK - chunk_lo], cnt);
  }
  __syncthreads();
  if (tid < 16 && sChunk[tid]) atomicAdd(&chunk_count[(int64_t)id * g.nchunks + chunk_lo + tid], sChunk[tid]);
  VH_DTICK(4);
}

// ------------------------------------------------------------- emit_features
// One workgroup per VH_CHUNK consecutive NMS blocks (row-major == reference
// output order).  Phase A: ordered compaction of the chunk's survivors into an
// LDS list; phase B: 16 lanes per survivor compute its descriptor from the
// image and the 12-word record is stored with 48-byte-contiguous writes.
__device__ __forceinline__ int32_t wave_incl_scan(int32_t v) {
#pragma unroll
  for (int32_t d = 1; d < 64; d <<= 1) {
    const int32_t t = __shfl_up(v, d);
    if ((int32_t)(threadIdx.x & 63) >= d) v += t;
  }
  return v;
}

// sample-point offsets of the 16 (du,dv) pairs, byte order of matcher.cpp:482-513
__constant__ int8_t c_desc_dx[16] = {-3, -3, -1, -1, +3, +3, +1, +1, -1, -1, +1, +1, -5, -5, +5, +5};
__constant__ int8_t c_desc_dy[16] = {-1, +1, -1, +1, -1, +1, -1, +1, -5, +5, -5, +5, -3, +3, -3, +3};


#ifndef VH_EMIT_NF
#define VH_EMIT_NF 2
#endif
__global__ void __launch_bounds__(256)
emit_features_kernel(VhImages im, VhGeom g, const uint64_t *__restrict__ rec,
                     const int32_t *__restrict__ chunk_count, VhSets s) {
  int32_t *__restrict__ feat = s.feat;
  int32_t *__restrict__ count = s.count;
  const int32_t cap = s.cap;
  __shared__ uint32_t sList[4 * VH_CHUNK + 1];  // u | v<<14 | c<<28 (matching-resolution coords); last word = sink for empty slots
  __shared__ int32_t sWave[4], sWaveP[4];

  const int32_t id = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const uint8_t *__restrict__ I = vh_image_ptr(im, id);
  const int32_t set = vh_image_set(im, id);
  const int32_t n1 = g.n + 1;
  VH_DET_SETPRIO();
  VH_ETICK_INIT;

  // phase A: each lane owns VH_CHUNK/256 consecutive blocks; the four 16-bit codes of a
  // block are handled as two dwords so every extract is one 32-bit op.  The records and the
  // counts of the earlier chunks are requested together (one memory round trip, not two), from
  // clamped addresses so that no load sits behind a branch.
  constexpr int BPL = VH_CHUNK / 256;  // consecutive blocks per lane
  uint32_t clo[BPL], chi[BPL];
  {
    const uint2 *rp = reinterpret_cast<const uint2 *>(rec) + (int64_t)id * g.nblocks;
#pragma unroll
    for (int32_t k = 0; k < BPL; k++) {
      const uint2 c = rp[min(chunk * VH_CHUNK + tid * BPL + k, g.nblocks - 1)];
      clo[k] = c.x; chi[k] = c.y;
    }
  }
  // features emitted by earlier chunks of this image
  int32_t part = 0;
  for (int32_t k = tid; k < chunk; k += 256) part += chunk_count[(int64_t)id * g.nchunks + k];
  int32_t mine = 0;
#pragma unroll
  for (int32_t k = 0; k < BPL; k++) {
    if (chunk * VH_CHUNK + tid * BPL + k >= g.nblocks) clo[k] = chi[k] = ~0u;
    mine += ((clo[k] & 0xFFFFu) != VH_NO_CODE) + ((clo[k] >> 16) != VH_NO_CODE) + ((chi[k] & 0xFFFFu) != VH_NO_CODE) + ((chi[k] >> 16) != VH_NO_CODE);
  }
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
  const int32_t incl = wave_incl_scan(mine);
  if ((tid & 63) == 63) { sWave[tid >> 6] = incl; sWaveP[tid >> 6] = part; }
  __syncthreads();
  const int32_t base = sWaveP[0] + sWaveP[1] + sWaveP[2] + sWaveP[3];
  VH_ETICK(0);
  int32_t woff = 0;
  for (int32_t w = 0; w < (tid >> 6); w++) woff += sWave[w];
  const int32_t total = sWave[0] + sWave[1] + sWave[2] + sWave[3];
  int32_t pos = woff + incl - mine;
  {
    // position code = (row in block) << 6 | (column in block); block coordinates
    // advance incrementally (one division per lane, not one per block and code).
    // Appends are branch-free: an empty slot is written to the sink word.
    const int32_t blk0 = chunk * VH_CHUNK + tid * BPL;
    int32_t by = blk0 / g.nbx, bx = blk0 - by * g.nbx;
    const uint32_t org = (uint32_t)(g.n + VH_MARGIN) * ((1u << 14) + 1u);
#pragma unroll
    for (int32_t k = 0; k < BPL; k++) {
      const uint32_t pxy = org + (uint32_t)(bx * n1) + ((uint32_t)(by * n1) << 14);  // u | v<<14 of the block corner
#pragma unroll
      for (int32_t q = 0; q < 4; q++) {
        const uint32_t w = (q < 2) ? clo[k] : chi[k];
        const uint32_t pc = (q & 1) ? (w >> 16) : (w & 0xFFFFu);
        const bool ok = pc != VH_NO_CODE;
        // (pc & 63) lands in the u field, (pc >> 6) in the v field
        const uint32_t e = pxy + (pc & 63u) + ((pc >> 6) << 14) + ((uint32_t)q << 28);
        sList[ok ? pos : 4 * VH_CHUNK] = e;
        pos += ok ? 1 : 0;
      }
      if (++bx == g.nbx) { bx = 0; by++; }
    }
  }
  __syncthreads();
  if (chunk == g.nchunks - 1 && tid == 0) count[set] = base + total;
  VH_ETICK(1);
  // (phase B's lane roles, here because its first patch loads are requested now, ahead of phase A2)
  const int32_t grp = tid >> 4, k = tid & 15;
  const int32_t dx = c_desc_dx[k], dy = c_desc_dy[k];
  constexpr int NF = VH_EMIT_NF;  // features per lane group and trip
  const int32_t prk = min(k, 14) - 7;                    // patch row this lane fetches (lane 15: row 14 again, no branch)
  // The patch rows of trip i+1 are in flight while trip i is handed round, computed and stored
  // (one 16-byte load per lane and feature: 8 registers for the double buffer).
  typedef uint32_t u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
  struct Coords { int32_t fs[NF], us[NF], vs[NF], cs[NF]; bool lives[NF]; };
  auto fetch = [&](int32_t f0, Coords &q, u32x4a1 (&prow)[NF]) {
#pragma unroll
    for (int32_t h = 0; h < NF; h++) {
      q.fs[h] = f0 + 16 * h + grp;
      q.lives[h] = q.fs[h] < total;
      const uint32_t e = sList[q.lives[h] ? q.fs[h] : 0];  // dead lanes recompute feature 0 and drop the result
      q.us[h] = e & 0x3FFF; q.vs[h] = (e >> 14) & 0x3FFF; q.cs[h] = e >> 28;
      // 32-bit byte offsets from the (wave-uniform) image base: images are <= 2^28 bytes, rows < 2^14, strides
      // < 2^24.  Byte-granular: no alignment of the image or its stride is assumed.  u+8 <= W-1 for every feature.
#ifdef VH_EXP_NOPATCH  // timing-only build: the descriptor's instructions without its patch gathers (descriptors are wrong)
      prow[h] = u32x4a1{(uint32_t)q.us[h], (uint32_t)q.vs[h], 0u, 0u};
#else
      prow[h] = *(const u32x4a1 *)(I + (__umul24((uint32_t)(q.vs[h] + prk), (uint32_t)g.bplm) + (uint32_t)(q.us[h] - 7)));
#endif
    }
  };
  Coords qa, qb;
  u32x4a1 pa[NF], pb[NF];
  if (total > 0) fetch(0, qa, pa);

  // phase A2: bin histogram + per-bin staging + row histogram, one lane per
  // feature.  Kept out of the descriptor loop below: a returning atomic inside
  // it serialised one L2 round trip per loop trip.  createIndexVector's bin
  // (matcher.cpp:208-212) and the (class, v) row; the arbitrary arrival order of
  // the slots is undone by bin_sort.  The (class, v) row counts of the chunk --
  // its blocks span only a few pixel rows -- are first added up in LDS and flushed
  // with one global atomic per row that occurs (this phase is bound by the rate of
  // L2 atomics: ~3 per feature in round 1 -- bin slot, row count, row cursor in bin_sort -- ~1.15 now).
  constexpr int ROWS_LDS = 64;
  __shared__ int32_t sRow[4 * ROWS_LDS];
  __shared__ __attribute__((aligned(16))) uint16_t sLoc[4 * VH_CHUNK];  // rank of each feature of the chunk among those of its (class, v) row
  const int32_t blk_first = chunk * VH_CHUNK, blk_last = min(blk_first + VH_CHUNK, g.nblocks) - 1;
  const int32_t v_first = ((blk_first / g.nbx) * n1 + g.n + VH_MARGIN) * g.scale;                      // smallest v a feature of this chunk can have
  const int32_t v_span = ((blk_last / g.nbx) * n1 + g.n + VH_MARGIN + g.n) * g.scale + g.scale - v_first;  // ... and one past the largest, relative
  const bool rows_in_lds = v_span <= ROWS_LDS;
  int32_t *__restrict__ rowh = s.row_hist + (int64_t)set * 4 * s.H;
  // createIndexVector's bin of a feature (matcher.cpp:208-212)
  auto bin_of_entry = [&](uint32_t e) {
    const int32_t uu = (int32_t)(e & 0x3FFF) * g.scale, vv = (int32_t)((e >> 14) & 0x3FFF) * g.scale, c = e >> 28;
    const int32_t ubin = s.binsize == 1 ? uu : (int32_t)__umulhi((uint32_t)uu, s.inv_binsize);
    const int32_t vbin = s.binsize == 1 ? vv : (int32_t)__umulhi((uint32_t)vv, s.inv_binsize);
    return (c * s.ubn + min(ubin, s.ubn - 1)) * s.vbn + min(vbin, s.vbn - 1);
  };
  // The bin slot of the lane's first feature is requested here, ahead of the row ranks: the two
  // returning atomics of a feature are then one round trip deep, not two.
  const bool have0 = tid < total && base + tid < cap;
  const int32_t b0 = have0 ? bin_of_entry(sList[tid]) : 0;
  int32_t slot0 = 0;
  if (have0) slot0 = atomicAdd(&s.hist[(int64_t)set * s.nbins + b0], 1);
  if (rows_in_lds) {
    // (a) count the chunk's features per row in LDS, remembering each one's rank in its row;
    // (b) one returning global atomic per row that occurs reserves the chunk's range of the
    //     row (row_hist ends up as the row's total, bin_scan turns it into row_start);
    // (c) rank in row = reserved offset + rank in chunk: bin_sort needs no atomic for the row index
    for (int32_t k = tid; k < 4 * ROWS_LDS; k += 256) sRow[k] = 0;
    __syncthreads();
    for (int32_t f = tid; f < total && base + f < cap; f += 256) {  // (features beyond the capacity are dropped everywhere)
      const uint32_t e = sList[f];
      const int32_t vv = (int32_t)((e >> 14) & 0x3FFF) * g.scale, c = e >> 28;
      sLoc[f] = (uint16_t)atomicAdd(&sRow[c * ROWS_LDS + (vv - v_first)], 1);
    }
    __syncthreads();
    for (int32_t k = tid; k < 4 * ROWS_LDS; k += 256) {
      const int32_t cnt = sRow[k], c = k / ROWS_LDS, vv = v_first + k % ROWS_LDS;
      if (cnt) sRow[k] = atomicAdd(&rowh[c * s.H + vv], cnt);
    }
    __syncthreads();
  }
  VH_ETICK(2);
  for (int32_t f = tid; f < total; f += 256) {
    const int32_t fi = base + f;
    if (fi >= cap) break;
    const uint32_t e = sList[f];
    const int32_t vv = (int32_t)((e >> 14) & 0x3FFF) * g.scale, c = e >> 28;
    const int32_t b = f == tid ? b0 : bin_of_entry(e);
    const int32_t rowrel = rows_in_lds ? sRow[c * ROWS_LDS + (vv - v_first)] + (int32_t)sLoc[f] : atomicAdd(&rowh[c * s.H + vv], 1);
    const int32_t slot = f == tid ? slot0 : atomicAdd(&s.hist[(int64_t)set * s.nbins + b], 1);
#ifdef VH_CHECK
    { int32_t sl = slot; VH_CHECK_RANGE(s, 5, sl, 0, s.stage_cap); }
#endif
    if (slot < s.stage_cap) s.stage[((int64_t)set * s.nbins + b) * s.stage_cap + slot] = make_int2(fi, rowrel);
  }

  VH_ETICK(3);
  // phase B: 16 lanes per feature, lane k = sample point k; NF features per lane and loop trip.
  // The 16 windows of a feature (5x5 bytes each) all lie in the 15x15 patch around it.  Fetched
  // window by window (5 loads per lane) a wave-wide load touches ~8 cache lines per feature, 40
  // per feature in all, and the kernel is bound by exactly that: the rate at which the vector
  // memory pipe walks the distinct lines of a load, whatever their bytes.  So the patch is
  // fetched ONCE, row k by lane k as one 16-byte load at the (unaligned) byte u-7 -- 15 lines per
  // feature, plus the rows that straddle a line -- and handed round through LDS, from where every
  // lane takes its five row segments at offsets that depend on the lane alone.  A lane group
  // lives inside one wave and a wave's LDS instructions execute in order: no barrier is needed.
  const int32_t sh = g.scale - 1;  // scale is 1 or 2
  int32_t *__restrict__ out = feat + (int64_t)set * cap * 12;
  uint32_t *__restrict__ fuv = s.f_uv + (int64_t)set * cap;
  // lane L<4 writes the header word, lane 4+j the descriptor dword j = pair[2j] | pair[2j+1]<<16
  const int32_t gbase = (tid & 63) & ~15, j = (k >= 4) ? (k - 4) : 0;
  // (branch-free word select: the lane's role is loop-invariant; word 2, val, is zeroed on packing, matcher.cpp:667)
  const uint32_t m_u = k == 0 ? ~0u : 0u, m_v = k == 1 ? ~0u : 0u, m_c = k == 3 ? ~0u : 0u, m_d = k >= 4 ? ~0u : 0u;
  static_assert(16 * NF * 16 * 16 <= (int)sizeof(sLoc), "the patches reuse sLoc");
  __syncthreads();  // sLoc is dead from here on
  uint32_t *sPatch = (uint32_t *)sLoc + grp * (NF * 64);  // [feature of the trip][16 rows: 15 + lane 15's spare][4 dwords] of this lane group
  const int32_t rd0 = (dy + 5) * 4 + ((dx + 5) >> 2);         // first dword of the lane's window in a patch
  const uint32_t psel = 0x03020100u + (uint32_t)((dx + 5) & 3) * 0x01010101u;  // v_perm selector of its first four bytes there
  auto process = [&](const Coords &q, const u32x4a1 (&prow)[NF]) {
    const int32_t *fs = q.fs, *us = q.us, *vs = q.vs, *cs = q.cs;
    const bool *lives = q.lives;
    uint32_t lo[NF][5], hi[NF][5];
    uint32_t sel[NF];
#pragma unroll
    for (int32_t h = 0; h < NF; h++) sel[h] = psel;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int32_t h = 0; h < NF; h++)
      *(uint4 *)(sPatch + (h * 16 + k) * 4) = make_uint4(prow[h].x, prow[h].y, prow[h].z, prow[h].w);
#pragma unroll
    for (int32_t h = 0; h < NF; h++)
#pragma unroll
      for (int32_t r = 0; r < 5; r++) {
        lo[h][r] = sPatch[h * 64 + rd0 + 4 * r];
        hi[h][r] = sPatch[h * 64 + rd0 + 4 * r + 1];
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int32_t h = 0; h < NF; h++) {
      const int32_t u = us[h], v = vs[h];
      // 5x5 Sobel pair at (u+dx, v+dy): du = smooth_y (x) deriv_x, dv = deriv_y (x) smooth_x
      // (filter.cpp:288-318 column pass, :132-171 / :79-127 row passes)
      // With w = bytes x0..x0+3 and w' = bytes x0+1..x0+4 of a row, its sums are byte dot products
      // (v_dot4_u32_u8, accumulating): a+2b-2d-e = w.(1,2,0,0) - w'.(0,0,2,1) and
      // a+4b+6c+4d+e = w.(1,4,6,0) + w'.(0,0,4,1); the column weights (1,4,6,4,1) resp.
      // (1,2,0,-2,-1) are folded into the byte weights, positive and negative parts apart.
      uint32_t du_p = 0, du_n = 0, dv_p = 0, dv_n = 0;
#pragma unroll
      for (int32_t r = 0; r < 5; r++) {
        const uint32_t w = __builtin_amdgcn_perm(hi[h][r], lo[h][r], sel[h]);
        const uint32_t w1 = __builtin_amdgcn_perm(hi[h][r], lo[h][r], sel[h] + 0x01010101u);
        const uint32_t sw = (r == 0 || r == 4) ? 1u : (r == 2 ? 6u : 4u);
        du_p = __builtin_amdgcn_udot4(w, sw * 0x00000201u, du_p, false);
        du_n = __builtin_amdgcn_udot4(w1, sw * 0x01020000u, du_n, false);
        if (r != 2) {
          const uint32_t dw = (r == 0 || r == 4) ? 1u : 2u;
          uint32_t &acc = r < 2 ? dv_p : dv_n;
          acc = __builtin_amdgcn_udot4(w, dw * 0x00060401u, acc, false);
          acc = __builtin_amdgcn_udot4(w1, dw * 0x01040000u, acc, false);
        }
      }
      const int32_t a_du = (int32_t)(du_p - du_n), a_dv = (int32_t)(dv_p - dv_n);
      // arithmetic >>7, +128, unsigned saturation (filter.cpp:114-115,124 / :159-160,168)
      const uint32_t du = (uint32_t)min(255, max(0, (a_du >> 7) + 128));
      const uint32_t dv = (uint32_t)min(255, max(0, (a_dv >> 7) + 128));
      const uint32_t pair = du | (dv << 8);
      const uint32_t plo = __shfl(pair, gbase + 2 * j), phi = __shfl(pair, gbase + 2 * j + 1);
      const uint32_t word = (((uint32_t)u << sh) & m_u) | (((uint32_t)v << sh) & m_v) | ((uint32_t)cs[h] & m_c) | ((plo | (phi << 16)) & m_d);
      const int32_t fi = base + fs[h];
      const bool okf = lives[h] && fi < cap;
      if (okf && k < 12) out[(int64_t)fi * 12 + k] = (int32_t)word;
      if (okf && k == 12) fuv[fi] = (uint32_t)(u << sh) | ((uint32_t)(v << sh) << 16);
    }
  };
  // two register sets, taken in turns: a copy from "next" to "current" would have to wait for the loads
  if (total > 0) {
    for (int32_t f0 = 0;; f0 += 32 * NF) {  // all conditions are workgroup-uniform
      const bool more1 = f0 + 16 * NF < total;
      if (more1) fetch(f0 + 16 * NF, qb, pb);
      process(qa, pa);
      if (!more1) break;
      const bool more2 = f0 + 32 * NF < total;
      if (more2) fetch(f0 + 32 * NF, qa, pa);
      process(qb, pb);
      if (!more2) break;
    }
  }
  VH_ETICK(4);
}

// --------------------------------------------------------------------- planes
// Gradient / response planes on the valid interior, 0 elsewhere -- only for the
// optional I_du/I_dv outputs of computeFeatures and for vh_filters; the
// detection path never materialises them.
__global__ void planes_kernel(const uint8_t *__restrict__ I, int32_t bpl, int32_t H,
                              uint8_t *__restrict__ du, uint8_t *__restrict__ dv,
                              int16_t *__restrict__ f1, int16_t *__restrict__ f2) {
  const int32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= bpl) return;
  const int64_t o = (int64_t)y * bpl + x;
  int32_t r_du = 0, r_dv = 0, r_f1 = 0, r_f2 = 0;
  if (x >= 2 && x <= bpl - 3 && y >= 2 && y <= H - 3) {
    int32_t a_du = 0, a_dv = 0, a_f2 = 0, s5 = 0, s3 = 0, ctr = 0;
#pragma unroll
    for (int32_t r = 0; r < 5; r++) {
      const uint8_t *p = I + (int64_t)(y + r - 2) * bpl + (x - 2);
      const int32_t a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
      const int32_t sw = (r == 0 || r == 4) ? 1 : (r == 2 ? 6 : 4);
      const int32_t dw = (r == 0) ? 1 : (r == 1 ? 2 : (r == 2 ? 0 : (r == 3 ? -2 : -1)));
      const int32_t cw = (r < 2) ? 1 : (r == 2 ? 0 : -1);
      a_du += sw * (a + 2 * b - 2 * d - e);
      a_dv += dw * (a + 4 * b + 6 * c + 4 * d + e);
      a_f2 += cw * (a + b - d - e);
      s5 += a + b + c + d + e;
      if (r >= 1 && r <= 3) s3 += b + c + d;
      if (r == 2) ctr = c;
    }
    r_du = min(255, max(0, (a_du >> 7) + 128));
    r_dv = min(255, max(0, (a_dv >> 7) + 128));
    r_f2 = a_f2;
    if (x >= 3 && y >= 3) r_f1 = -s5 + 2 * s3 + 7 * ctr;
  }
  if (du) du[o] = (uint8_t)r_du;
  if (dv) dv[o] = (uint8_t)r_dv;
  if (f1) f1[o] = (int16_t)r_f1;
  if (f2) f2[o] = (int16_t)r_f2;
}

}  // namespace

// rows start on 4-byte boundaries: enables the dword-load fast paths
static bool images_dword_aligned(const VhImages &im, const VhGeom &g) {
  return (g.bplm % 4 == 0) && (im.stride % 4 == 0) && ((uintptr_t)im.base[0] % 4 == 0) &&
         (im.ncam < 2 || (uintptr_t)im.base[1] % 4 == 0);
}

void vh_launch_half_res(const VhImages &src, uint8_t *dst, const VhGeom &g, hipStream_t st) {
  dim3 grid((g.bplm + 255) / 256, g.Hm, src.S * src.ncam);
  hipLaunchKernelGGL(half_res_kernel, grid, dim3(256), 0, st, src, dst, g);
}

void vh_launch_detect_nms(const VhImages &im, const VhGeom &g, uint64_t *rec, int32_t *chunk_count,
                          hipStream_t st) {
  if (g.nblocks <= 0) return;
  if (images_dword_aligned(im, g) && g.n >= 1 && g.n <= 4 && g.nbx <= 2048) {
    dim3 grid((g.nbx + 31) / 32, (g.nby + 7) / 8, im.S * im.ncam);
    switch (g.n) {
      case 1: hipLaunchKernelGGL(detect_nms_fast_kernel<1>, grid, dim3(256), 0, st, im, g, rec, chunk_count); break;
      case 2: hipLaunchKernelGGL(detect_nms_fast_kernel<2>, grid, dim3(256), 0, st, im, g, rec, chunk_count); break;
      case 3: hipLaunchKernelGGL(detect_nms_fast_kernel<3>, grid, dim3(256), 0, st, im, g, rec, chunk_count); break;
      default: hipLaunchKernelGGL(detect_nms_fast_kernel<4>, grid, dim3(256), 0, st, im, g, rec, chunk_count); break;
    }
    return;
  }
  dim3 grid((g.nbx + g.tbx - 1) / g.tbx, (g.nby + g.tby - 1) / g.tby, im.S * im.ncam);
  const size_t lds = (size_t)g.IH * g.IWp + 2 * (size_t)g.FH * g.FWp * sizeof(int16_t) + 4 +
                     (size_t)g.tbx * g.tby * (16 + 8);
  hipLaunchKernelGGL(detect_nms_kernel, grid, dim3(256), lds, st, im, g, rec, chunk_count);
}

void vh_launch_emit_features(const VhImages &im, const VhGeom &g, const uint64_t *rec,
                             const int32_t *chunk_count, const VhSets &s, hipStream_t st) {
  if (g.nblocks <= 0) return;
  dim3 grid(g.nchunks, im.S * im.ncam);
  hipLaunchKernelGGL(emit_features_kernel, grid, dim3(256), 0, st, im, g, rec, chunk_count, s);
}

void vh_launch_planes(const uint8_t *img, int32_t bpl, int32_t H, uint8_t *du, uint8_t *dv,
                      int16_t *f1, int16_t *f2, hipStream_t st) {
  dim3 grid((bpl + 255) / 256, H);
  hipLaunchKernelGGL(planes_kernel, grid, dim3(256), 0, st, img, bpl, H, du, dv, f1, f2);
}
