// kernels_bin.hip -- position/class bin index of a feature set on gfx950.
//
// Replaces Matcher::createIndexVector (reference src/matcher.cpp:194-214).
// The reference keeps one std::vector<int32_t> per bin, filled in ascending
// feature order.  Here a feature set is *reordered* into bin order once
// (counting sort: histogram + per-bin staging [in emit_features for detected
// features, bin_hist/bin_fill for caller-supplied ones] -> scan -> per-bin
// order restore) and kept as structure-of-arrays (s_uv, s_idx, s_desc), so that
// the matcher streams candidates of a bin range as one contiguous, coalesced
// segment; a second copy ordered by (class, v) serves the 1-d stereo search.
//
// Bin numbering here is u-bin major, (c*ubn+ub)*vbn+vb, so that the order
// "u_bin outer, v_bin inner, list position innermost" in which
// Matcher::findMatch visits candidates (src/matcher.cpp:243-246) is simply
// ascending position -- the first-minimum tie-break becomes "lowest position".
#include "vh_dev.h"
#include <algorithm>

// Threads of the one-workgroup-per-set scan.  256, not 1024: a 1024-thread
// workgroup needs 16 free wave slots on ONE CU at once and starves beside the
// saturating flow search of the other stream.
#define VH_SCAN_T 256

namespace {

__device__ __forceinline__ int32_t bin_coord(int32_t x, int32_t binsize, int32_t nb) {
  // min((int)floor((float)x/(float)binsize), nb-1)   (matcher.cpp:208-209).
  // For 0 <= x < 2^24 the float quotient floors to the integer quotient.
  return min(x / binsize, nb - 1);
}

__device__ __forceinline__ int32_t feature_bin(const int32_t *__restrict__ f, const VhSets &s) {
  const int32_t u = f[0], v = f[1], c = f[3];
  return (c * s.ubn + bin_coord(u, s.binsize, s.ubn)) * s.vbn + bin_coord(v, s.binsize, s.vbn);
}

// Zero every per-frame counter of sets [set0, set0+nsets) -- feature counts,
// bin and row histograms and cursors -- plus an optional extra array, in one launch.
__global__ void zero_counters_kernel(VhSets s, int32_t set0, int32_t nsets, int32_t *__restrict__ extra, int64_t n_extra) {
  const int64_t nb = (int64_t)nsets * s.nbins, nr = (int64_t)nsets * 4 * s.H;
  const int64_t total = 2 * nb + 2 * nr + nsets + n_extra;
  int32_t *hist = s.hist + (int64_t)set0 * s.nbins, *cur = s.cursor + (int64_t)set0 * s.nbins;
  int32_t *rh = s.row_hist + (int64_t)set0 * 4 * s.H, *rc = s.row_cursor + (int64_t)set0 * 4 * s.H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t k = i;
    if (k < nb) { hist[k] = 0; continue; }
    k -= nb;
    if (k < nb) { cur[k] = 0; continue; }
    k -= nb;
    if (k < nr) { rh[k] = 0; continue; }
    k -= nr;
    if (k < nr) { rc[k] = 0; continue; }
    k -= nr;
    if (k < nsets) { s.count[set0 + k] = 0; continue; }
    k -= nsets;
    extra[k] = 0;
  }
}

__global__ void bin_hist_kernel(VhSets s, int32_t set0) {
  const int32_t set = set0 + blockIdx.y;
  const int32_t n = min(s.count[set], s.cap);
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t *f = s.feat + ((int64_t)set * s.cap + i) * 12;
    s.f_uv[(int64_t)set * s.cap + i] = (uint32_t)f[0] | ((uint32_t)f[1] << 16);  // caller-supplied features: emit_features did not write it
    atomicAdd(&s.hist[(int64_t)set * s.nbins + feature_bin(f, s)], 1);
    atomicAdd(&s.row_hist[(int64_t)set * 4 * s.H + f[3] * s.H + f[1]], 1);
  }
}

__device__ __forceinline__ int32_t scan_wave_inclusive(int32_t v) {
#pragma unroll
  for (int32_t d = 1; d < 64; d <<= 1) {
    const int32_t t = __shfl_up(v, d);
    if ((int32_t)(threadIdx.x & 63) >= d) v += t;
  }
  return v;
}

// Workgroup-wide exclusive scan of load(0..n-1): every element i is handed to
// emit(i, prefix, value); returns the total.  EPT consecutive elements per lane
// and trip, a wave scan of the lane sums and one LDS exchange of the
// VH_SCAN_T/64 wave totals -- two barriers per EPT*VH_SCAN_T elements.
template <int EPT, class Load, class Emit>
__device__ __forceinline__ int32_t scan_exclusive(int32_t n, int32_t *sWave, Load load, Emit emit) {
  const int32_t tid = threadIdx.x, w = tid >> 6;
  int32_t carry = 0;
  for (int32_t b0 = 0; b0 < n; b0 += EPT * VH_SCAN_T) {
    const int32_t b = b0 + EPT * tid;
    int32_t v[EPT];
    int32_t mine = 0;
#pragma unroll
    for (int32_t k = 0; k < EPT; k++) { v[k] = (b + k < n) ? load(b + k) : 0; mine += v[k]; }
    const int32_t incl = scan_wave_inclusive(mine);
    if ((tid & 63) == 63) sWave[w] = incl;
    __syncthreads();
    int32_t before = 0, total = 0;
#pragma unroll
    for (int32_t k = 0; k < VH_SCAN_T / 64; k++) {
      const int32_t x = sWave[k];
      before += (k < w) ? x : 0;
      total += x;
    }
    int32_t run = carry + before + incl - mine;
#pragma unroll
    for (int32_t k = 0; k < EPT; k++) {
      if (b + k < n) emit(b + k, run, v[k]);
      run += v[k];
    }
    carry += total;
    __syncthreads();  // sWave is rewritten by the next trip
  }
  return carry;
}

// Query tiles of the flow search: <= VH_TILE_Q consecutive features of one class in SNAKE order -- the (class, u-bin)
// columns one after the other, even columns in ascending bin order (ascending v), odd columns descending -- so that
// the lanes of every wave are filled (only the last tile of a class is partial) AND a tile that runs from the end of
// one column into the next stays compact: it holds the bottoms (or the tops) of two neighbouring columns.  Tiles cut
// at column ends instead leave ~10 % of the lanes idle; tiles over plain bin order make every crossing tile span the
// whole image height (tools/tile_model3.py: evaluated / in-window pairs 1.40 / 1.38 / 1.32 at KITTI size).
// A tile record is {first snake index, end, class, column of the first index}; a snake index k inside column
// [A, B) of the bin order is the position k (even column) or A + B - 1 - k (odd column): kernels_match.hip.
// One lane per (class, column) writes the tiles that start inside its column.
__device__ __forceinline__ void make_tiles(const VhSets &s, const int32_t *__restrict__ bs, int32_t set) {
  const int32_t span = s.ubn * s.vbn;
  int32_t tb[5];
  tb[0] = 0;
#pragma unroll
  for (int32_t c = 0; c < 4; c++) tb[c + 1] = tb[c] + (bs[(c + 1) * span] - bs[c * span] + VH_TILE_Q - 1) / VH_TILE_Q;
  int4 *__restrict__ tiles = s.tiles + (int64_t)set * s.max_tiles;
  for (int32_t i = threadIdx.x; i < 4 * s.ubn; i += VH_SCAN_T) {
    const int32_t c = i / s.ubn, col = i - c * s.ubn;
    const int32_t cls0 = bs[c * span], cls1 = bs[(c + 1) * span];
    const int32_t a = bs[(c * s.ubn + col) * s.vbn] - cls0, b = bs[(c * s.ubn + col + 1) * s.vbn] - cls0;
    for (int32_t t = (a + VH_TILE_Q - 1) / VH_TILE_Q; t * VH_TILE_Q < b; t++) {
      const int32_t k0 = cls0 + t * VH_TILE_Q, slot = tb[c] + t;
      if (slot < s.max_tiles) tiles[slot] = make_int4(k0, min(cls1, k0 + VH_TILE_Q), c, col);
    }
  }
  if (threadIdx.x == 0) s.tile_cnt[set] = min(tb[4], s.max_tiles);
}

// One workgroup per set: exclusive scan of the histogram into bin_start, and
// the list of query tiles the match kernel works through (make_tiles).
__global__ void __launch_bounds__(VH_SCAN_T) bin_scan_kernel(VhSets s, int32_t set0) {
  __shared__ int32_t sWave[VH_SCAN_T / 64];
  const int32_t set = set0 + blockIdx.x, tid = threadIdx.x;
  VH_DET_SETPRIO();
  const int32_t *__restrict__ hist = s.hist + (int64_t)set * s.nbins;
  int32_t *__restrict__ bs = s.bin_start + (int64_t)set * (s.nbins + 1);
  const int32_t nfeat = scan_exclusive<4>(s.nbins, sWave, [&](int32_t b) { return hist[b]; },
                                       [&](int32_t b, int32_t prefix, int32_t) { bs[b] = prefix; });
  if (tid == 0) bs[s.nbins] = nfeat;
  __syncthreads();  // bin_start is read back below by other lanes

  make_tiles(s, bs, set);

  // row index: exclusive scan of the (class, v) histogram
  const int32_t nrow = 4 * s.H;
  const int32_t *__restrict__ rh = s.row_hist + (int64_t)set * nrow;
  int32_t *__restrict__ rs = s.row_start + (int64_t)set * (nrow + 1);
  const int32_t nrowfeat = scan_exclusive<4>(nrow, sWave, [&](int32_t r) { return rh[r]; },
                                          [&](int32_t r, int32_t prefix, int32_t) { rs[r] = prefix; });
  if (tid == 0) rs[nrow] = nrowfeat;
}

__global__ void bin_fill_kernel(VhSets s, int32_t set0) {
  const int32_t set = set0 + blockIdx.y;
  const int32_t n = min(s.count[set], s.cap);
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t b = feature_bin(s.feat + ((int64_t)set * s.cap + i) * 12, s);
    const int32_t slot = s.bin_start[(int64_t)set * (s.nbins + 1) + b] + atomicAdd(&s.cursor[(int64_t)set * s.nbins + b], 1);
    s.tmp_idx[(int64_t)set * s.cap + slot] = i;
  }
}

// 16 lanes per bin (typical bins hold 5-20 features): the atomic fill left the
// bin's members in arbitrary order; rank-sort them back to ascending feature
// index (the reference's push_back order) and gather the bin-ordered
// structure-of-arrays and the row-ordered copy.
__global__ void __launch_bounds__(256) bin_sort_kernel(VhSets s, int32_t set0, int32_t staged) {
  const int32_t set = set0 + blockIdx.y;
  const int32_t bin = blockIdx.x * 16 + (threadIdx.x >> 4), gl = threadIdx.x & 15;
  if (bin >= s.nbins) return;
  VH_DET_SETPRIO();
  const int32_t *__restrict__ bs = s.bin_start + (int64_t)set * (s.nbins + 1);
  const int32_t p0 = bs[bin], p1 = bs[bin + 1], L = p1 - p0;
  if (L <= 0) return;
#ifdef VH_CHECK
  if (staged && gl == 0) { int32_t len = L; VH_CHECK_RANGE(s, 6, len, 0, s.stage_cap + 1); }
#endif
  // members of this bin, in arbitrary order: staged by emit_features (own features, with
  // their rank in their (class, v) row) or placed by bin_fill (caller-supplied features)
  const int2 *__restrict__ stg = s.stage + ((int64_t)set * s.nbins + bin) * s.stage_cap - p0;
  const int32_t *__restrict__ tmp = s.tmp_idx + (int64_t)set * s.cap;
  const int32_t *__restrict__ feat = s.feat + (int64_t)set * s.cap * 12;
  int32_t *__restrict__ sidx = s.s_idx + (int64_t)set * s.cap;
  uint32_t *__restrict__ suv = s.s_uv + (int64_t)set * s.cap;
  uint4 *__restrict__ sdesc = (uint4 *)(s.s_desc + (int64_t)set * s.cap * 8);
  const int32_t nrow = 4 * s.H;
  const int32_t *__restrict__ rs = s.row_start + (int64_t)set * (nrow + 1);
  int32_t *__restrict__ rcur = s.row_cursor + (int64_t)set * nrow;
  int32_t *__restrict__ rpos = s.r_pos + (int64_t)set * s.cap;
  for (int32_t e0 = 0; e0 < L; e0 += 16) {
    const int32_t e = e0 + gl;
    int32_t mine = 0x7FFFFFFF, rowrel = 0;
    if (e < L) {
      if (staged) { const int2 m2 = stg[p0 + e]; mine = m2.x; rowrel = m2.y; }
      else mine = tmp[p0 + e];
    }
    uint4 h = make_uint4(0, 0, 0, 0), d0 = h, d1 = h;
    if (e < L) {  // the record gather does not depend on the rank: issue it first
      const int32_t *f = feat + (int64_t)mine * 12;
      h = *(const uint4 *)f; d0 = *(const uint4 *)(f + 4); d1 = *(const uint4 *)(f + 8);
    }
    int32_t rank = 0;
    for (int32_t j0 = 0; j0 < L; j0 += 16) {
      const int32_t other = (j0 + gl < L) ? (staged ? stg[p0 + j0 + gl].x : tmp[p0 + j0 + gl]) : 0x7FFFFFFF;
      const int32_t m = min(16, L - j0);
      for (int32_t j = 0; j < m; j++) rank += (__shfl(other, j, 16) < mine) ? 1 : 0;
    }
    if (e < L) {
      const int32_t p = p0 + rank;
      sidx[p] = mine;
      suv[p] = (uint32_t)h.x | ((uint32_t)h.y << 16);
      sdesc[2 * (int64_t)p] = d0;
      sdesc[2 * (int64_t)p + 1] = d1;
      // row order for the stereo search: only the bin position is scattered (4 B), the search
      // gathers the records from the bin-ordered arrays.  Order inside a row is irrelevant (the
      // matcher minimises a (cost, bin position) key): detected features bring their rank in the
      // row along (emit_features), caller-supplied ones take a ticket here.
      const int32_t row = (int32_t)h.w * s.H + (int32_t)h.y;
      // (ranks within a row are 0 .. count-1, each taken once: emit_features' LDS ranks + reserved offsets, or the cursor)
      int32_t rp = rs[row] + (staged ? rowrel : atomicAdd(&rcur[row], 1));
      VH_CHECK_RANGE(s, 4, rp, rs[row], rs[row + 1]);
      if (rp < rs[nrow]) rpos[rp] = p;  // (one compare per feature: a rank beyond the row index -- an invariant broken upstream -- must not become a stray write)
    }
  }
}

// CSR in the reference's own bin numbering (c*vbn+vb)*ubn+ub, for
// vh_create_index (parity check of Matcher::createIndexVector).
__global__ void ref_index_kernel(VhSets s, int32_t set, int32_t *__restrict__ bs_ref,
                                 int32_t *__restrict__ list_ref) {
  // single workgroup: sequential over reference bins, parallel inside a bin
  __shared__ int32_t sPos;
  const int32_t *__restrict__ bs = s.bin_start + (int64_t)set * (s.nbins + 1);
  const int32_t *__restrict__ sidx = s.s_idx + (int64_t)set * s.cap;
  if (threadIdx.x == 0) sPos = 0;
  __syncthreads();
  for (int32_t rb = 0; rb < s.nbins; rb++) {
    const int32_t ub = rb % s.ubn, vb = (rb / s.ubn) % s.vbn, c = rb / (s.ubn * s.vbn);
    const int32_t ib = (c * s.ubn + ub) * s.vbn + vb;
    const int32_t p0 = bs[ib], L = bs[ib + 1] - p0, pos = sPos;
    if (threadIdx.x == 0) bs_ref[rb] = pos;
    for (int32_t e = threadIdx.x; e < L; e += blockDim.x) list_ref[pos + e] = sidx[p0 + e];
    __syncthreads();
    if (threadIdx.x == 0) sPos = pos + L;
    __syncthreads();
  }
  if (threadIdx.x == 0) bs_ref[s.nbins] = sPos;
}

}  // namespace

// Per-feature kernels use a grid-stride loop over a fixed, modest number of
// workgroups per set: the feature count lives on the device and a cap-sized
// grid would be ~75 % empty workgroups at typical densities.
static int32_t feature_blocks(const VhSets &s) { return std::min(std::max(s.cap / 1024, 8), 256); }

void vh_launch_zero_counters(const VhSets &s, int32_t set0, int32_t nsets, int32_t *extra, int64_t n_extra, hipStream_t st) {
  const int64_t total = 2 * (int64_t)nsets * s.nbins + 8 * (int64_t)nsets * s.H + nsets + n_extra;
  const int32_t blocks = (int32_t)std::min<int64_t>((total + 1023) / 1024, 2048);
  hipLaunchKernelGGL(zero_counters_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, st, s, set0, nsets, extra, n_extra);
}

void vh_launch_bin_hist(const VhSets &s, int32_t set0, int32_t nsets, hipStream_t st) {
  dim3 grid(feature_blocks(s), nsets);
  hipLaunchKernelGGL(bin_hist_kernel, grid, dim3(256), 0, st, s, set0);
}
void vh_launch_bin_scan(const VhSets &s, int32_t set0, int32_t nsets, hipStream_t st) {
  hipLaunchKernelGGL(bin_scan_kernel, dim3(nsets), dim3(VH_SCAN_T), 0, st, s, set0);
}
void vh_launch_bin_fill(const VhSets &s, int32_t set0, int32_t nsets, hipStream_t st) {
  dim3 grid(feature_blocks(s), nsets);
  hipLaunchKernelGGL(bin_fill_kernel, grid, dim3(256), 0, st, s, set0);
}
void vh_launch_bin_sort(const VhSets &s, int32_t set0, int32_t nsets, int32_t staged, hipStream_t st) {
  dim3 grid((s.nbins + 15) / 16, nsets);
  hipLaunchKernelGGL(bin_sort_kernel, grid, dim3(256), 0, st, s, set0, staged);
}
void vh_launch_ref_index(const VhSets &s, int32_t set, int32_t *bin_start_ref, int32_t *list_ref,
                         hipStream_t st) {
  hipLaunchKernelGGL(ref_index_kernel, dim3(1), dim3(256), 0, st, s, set, bin_start_ref, list_ref);
}
