// kernels_ego.hip -- batched stereo egomotion (RANSAC + Gauss-Newton) on gfx950, SURVEY 8(f-4).
//
// Replaces, for every camera stream of a batch in one launch:
//   VisualOdometryStereo::estimateMotion              (reference src/viso_stereo.cpp:54-157)
//   ...::getInlier / updateParameters / computeObservations / computeResidualsAndJacobian
//                                                     (src/viso_stereo.cpp:159-330)
//   Matrix::solve on the 6x6 normal equations         (src/matrix.cpp:417-504)
//   VisualOdometry::getRandomSample(N,3)              (src/viso.cpp:86-106), from caller-supplied rand() values
//
// One 256-thread workgroup per stream, double precision throughout, built with
// -ffp-contract=off so that a*b+c rounds twice as on the reference's x86 build:
//   1. 3-d points of the previous frame, one match per thread.
//   2. The ransac_iters hypotheses in parallel, one per thread: Gauss-Newton on its three
//      sampled matches (<= 22 updates, each accumulating J^T J row by row in the reference's
//      summation order, then the reference's Gauss-Jordan elimination), then its inlier
//      count over all matches.  Identical operation order to the reference per hypothesis;
//      only sin/cos come from the device library.
//   3. arg max of the inlier count, first hypothesis on ties (the reference's strict `>`
//      in iteration order), ordered inlier list of the winner.
//   4. Refinement on the inliers (<= 102 updates): the threads accumulate partial normal
//      equations over their share of the inliers, a fixed-shape tree joins them, thread 0
//      solves.  The summation order differs from the reference's sequential loop, so the
//      refined parameters agree with it to rounding (~1e-12), not bit for bit.
#include "vh_dev.h"
#include "../../include/viso_hip.h"
#include <math.h>

namespace {

struct EgoRot {
  double r[9], drx[9], dry[9], drz[9];
};

__device__ __forceinline__ void ego_rot(const double tr[6], EgoRot &R) {
  const double sx = sin(tr[0]), cx = cos(tr[0]), sy = sin(tr[1]), cy = cos(tr[1]), sz = sin(tr[2]), cz = cos(tr[2]);
  R.r[0] = +cy * cz; R.r[1] = -cy * sz; R.r[2] = +sy;
  R.r[3] = +sx * sy * cz + cx * sz; R.r[4] = -sx * sy * sz + cx * cz; R.r[5] = -sx * cy;
  R.r[6] = -cx * sy * cz + sx * sz; R.r[7] = +cx * sy * sz + sx * cz; R.r[8] = +cx * cy;
  R.drx[0] = 0; R.drx[1] = 0; R.drx[2] = 0;
  R.drx[3] = +cx * sy * cz - sx * sz; R.drx[4] = -cx * sy * sz - sx * cz; R.drx[5] = -cx * cy;
  R.drx[6] = +sx * sy * cz + cx * sz; R.drx[7] = -sx * sy * sz + cx * cz; R.drx[8] = -sx * cy;
  R.dry[0] = -sy * cz; R.dry[1] = +sy * sz; R.dry[2] = +cy;
  R.dry[3] = +sx * cy * cz; R.dry[4] = -sx * cy * sz; R.dry[5] = +sx * sy;
  R.dry[6] = -cx * cy * cz; R.dry[7] = +cx * cy * sz; R.dry[8] = -cx * sy;
  R.drz[0] = -cy * sz; R.drz[1] = -cy * cz; R.drz[2] = 0;
  R.drz[3] = -sx * sy * sz + cx * cz; R.drz[4] = -sx * sy * cz - cx * sz; R.drz[5] = 0;
  R.drz[6] = +cx * sy * sz + sx * cz; R.drz[7] = +cx * sy * cz - sx * sz; R.drz[8] = 0;
}

struct EgoObs { double u1c, v1c, u2c, v2c, X, Y, Z; };

// prediction of one match under (R, t): p_predict of computeResidualsAndJacobian (src/viso_stereo.cpp:317-321)
__device__ __forceinline__ void ego_predict(const vh_ego_params &e, const EgoRot &R, const double tr[6], const EgoObs &o, double p[4],
                                            double &X1c, double &Y1c, double &Z1c) {
  X1c = R.r[0] * o.X + R.r[1] * o.Y + R.r[2] * o.Z + tr[3];
  Y1c = R.r[3] * o.X + R.r[4] * o.Y + R.r[5] * o.Z + tr[4];
  Z1c = R.r[6] * o.X + R.r[7] * o.Y + R.r[8] * o.Z + tr[5];
  const double X2c = X1c - e.base;
  p[0] = e.f * X1c / Z1c + e.cu;
  p[1] = e.f * Y1c / Z1c + e.cv;
  p[2] = e.f * X2c / Z1c + e.cu;
  p[3] = e.f * Y1c / Z1c + e.cv;
}

// Adds the four rows of one match to the normal equations: acc[0..20] = upper triangle of
// J^T J (row-major, m <= n), acc[21..26] = J^T r; rows in the reference's order (u1, v1, u2, v2).
__device__ __forceinline__ void ego_accumulate(const vh_ego_params &e, const EgoRot &R, const double tr[6], const EgoObs &o, double acc[27]) {
  double p[4], X1c, Y1c, Z1c;
  ego_predict(e, R, tr, o, p, X1c, Y1c, Z1c);
  double weight = 1.0;
  if (e.reweighting) weight = 1.0 / (fabs(o.u1c - e.cu) / fabs(e.cu) + 0.05);
  const double X2c = X1c - e.base;
  double Jr[4][6];
#pragma unroll
  for (int32_t j = 0; j < 6; j++) {
    double X1cd, Y1cd, Z1cd;
    if (j == 0) { X1cd = 0; Y1cd = R.drx[3] * o.X + R.drx[4] * o.Y + R.drx[5] * o.Z; Z1cd = R.drx[6] * o.X + R.drx[7] * o.Y + R.drx[8] * o.Z; }
    else if (j == 1) { X1cd = R.dry[0] * o.X + R.dry[1] * o.Y + R.dry[2] * o.Z; Y1cd = R.dry[3] * o.X + R.dry[4] * o.Y + R.dry[5] * o.Z; Z1cd = R.dry[6] * o.X + R.dry[7] * o.Y + R.dry[8] * o.Z; }
    else if (j == 2) { X1cd = R.drz[0] * o.X + R.drz[1] * o.Y; Y1cd = R.drz[3] * o.X + R.drz[4] * o.Y; Z1cd = R.drz[6] * o.X + R.drz[7] * o.Y; }
    else { X1cd = j == 3 ? 1 : 0; Y1cd = j == 4 ? 1 : 0; Z1cd = j == 5 ? 1 : 0; }
    Jr[0][j] = weight * e.f * (X1cd * Z1c - X1c * Z1cd) / (Z1c * Z1c);
    Jr[1][j] = weight * e.f * (Y1cd * Z1c - Y1c * Z1cd) / (Z1c * Z1c);
    Jr[2][j] = weight * e.f * (X1cd * Z1c - X2c * Z1cd) / (Z1c * Z1c);
    Jr[3][j] = weight * e.f * (Y1cd * Z1c - Y1c * Z1cd) / (Z1c * Z1c);
  }
  const double obs[4] = {o.u1c, o.v1c, o.u2c, o.v2c};
#pragma unroll
  for (int32_t row = 0; row < 4; row++) {
    const double res = weight * (obs[row] - p[row]);
    int32_t k = 0;
#pragma unroll
    for (int32_t m = 0; m < 6; m++)
#pragma unroll
      for (int32_t n = m; n < 6; n++) acc[k++] += Jr[row][m] * Jr[row][n];
#pragma unroll
    for (int32_t m = 0; m < 6; m++) acc[21 + m] += Jr[row][m] * res;
  }
}

// Matrix::solve for the 6x6 system (src/matrix.cpp:417-504): Gauss-Jordan with full pivoting,
// singular below 1e-20.  acc as produced by ego_accumulate; on success b = the solution.
// Every array index is a compile-time constant after unrolling -- the pivot's row and column (data-dependent in
// the original) select among the six rows / columns by predicates -- so the system lives in registers: with
// dynamic indices it sat in private memory, and the 22 dependent solves of a hypothesis were 85 % of the
// kernel (1.2 of 1.4 ms per batch of bucketed lists, tools/ego_phases.py).  The arithmetic applied to the
// elements, and its order, are the original's.
__device__ bool ego_solve(const double acc[27], double b[6]) {
  double A[6][6];
  {
    int32_t k = 0;
#pragma unroll
    for (int32_t m = 0; m < 6; m++)
#pragma unroll
      for (int32_t n = m; n < 6; n++) { A[m][n] = acc[k]; A[n][m] = acc[k]; k++; }
#pragma unroll
    for (int32_t m = 0; m < 6; m++) b[m] = acc[21 + m];
  }
  int32_t ipiv[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int32_t i = 0; i < 6; i++) {
    double big = 0.0;
    int32_t irow = 0, icol = 0;
#pragma unroll
    for (int32_t j = 0; j < 6; j++)
#pragma unroll
      for (int32_t k = 0; k < 6; k++) {
        const double v = fabs(A[j][k]);
        if (ipiv[j] != 1 && ipiv[k] == 0 && v >= big) { big = v; irow = j; icol = k; }
      }
#pragma unroll
    for (int32_t q = 0; q < 6; q++) ipiv[q] += q == icol ? 1 : 0;
    // rows irow and icol change places (nothing moves when they are the same row)
    double ri[6], rc[6], bi = 0.0, bc = 0.0;
#pragma unroll
    for (int32_t l = 0; l < 6; l++) { ri[l] = 0.0; rc[l] = 0.0; }
#pragma unroll
    for (int32_t r = 0; r < 6; r++) {
#pragma unroll
      for (int32_t l = 0; l < 6; l++) { ri[l] = r == irow ? A[r][l] : ri[l]; rc[l] = r == icol ? A[r][l] : rc[l]; }
      bi = r == irow ? b[r] : bi; bc = r == icol ? b[r] : bc;
    }
#pragma unroll
    for (int32_t r = 0; r < 6; r++) {
#pragma unroll
      for (int32_t l = 0; l < 6; l++) A[r][l] = r == icol ? ri[l] : (r == irow ? rc[l] : A[r][l]);
      b[r] = r == icol ? bi : (r == irow ? bc : b[r]);
    }
    // the pivot row (now row icol) is ri, its right-hand side bi
    double piv = 0.0;
#pragma unroll
    for (int32_t l = 0; l < 6; l++) piv = l == icol ? ri[l] : piv;
    if (fabs(piv) < 1e-20) return false;
    const double pivinv = 1.0 / piv;
#pragma unroll
    for (int32_t l = 0; l < 6; l++) ri[l] = (l == icol ? 1.0 : ri[l]) * pivinv;
    bi *= pivinv;
#pragma unroll
    for (int32_t ll = 0; ll < 6; ll++) {
      double dum = 0.0;
#pragma unroll
      for (int32_t l = 0; l < 6; l++) dum = l == icol ? A[ll][l] : dum;
      const bool prow = ll == icol;
#pragma unroll
      for (int32_t l = 0; l < 6; l++) {
        const double cur = l == icol ? 0.0 : A[ll][l];
        A[ll][l] = prow ? ri[l] : cur - ri[l] * dum;
      }
      b[ll] = prow ? bi : b[ll] - bi * dum;
    }
  }
  return true;
}

__device__ __forceinline__ EgoObs ego_load(const vh_p_match *pm, const double *X, const double *Y, const double *Z, int32_t i) {
  EgoObs o;
  o.u1c = pm[i].u1c; o.v1c = pm[i].v1c; o.u2c = pm[i].u2c; o.v2c = pm[i].v2c;
  o.X = X[i]; o.Y = Y[i]; o.Z = Z[i];
  return o;
}

// squared reprojection error test of getInlier (src/viso_stereo.cpp:171-174)
__device__ __forceinline__ bool ego_is_inlier(const vh_ego_params &e, const EgoRot &R, const double tr[6], const EgoObs &o) {
  double p[4], a, b, c;
  ego_predict(e, R, tr, o, p, a, b, c);
  const double d0 = o.u1c - p[0], d1 = o.v1c - p[1], d2 = o.u2c - p[2], d3 = o.v2c - p[3];
  return d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3 < e.inlier_threshold * e.inlier_threshold;
}

#define EGO_T 256

// -DVH_EGO_TIMING: workgroup 0 adds the 100 MHz clock ticks of its phases to g_ego_t (tools/ego_phases.py)
__device__ unsigned long long g_ego_t[8];
#ifdef VH_EGO_TIMING
#define VH_ETK_INIT unsigned long long et_prev_ = wall_clock64()
#define VH_ETK(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_ego_t[k], now_ - et_prev_); et_prev_ = now_; } } while (0)
#else
#define VH_ETK_INIT do { } while (0)
#define VH_ETK(k) do { } while (0)
#endif

__global__ void __launch_bounds__(EGO_T)
ego_kernel(vh_ego_params e, const vh_p_match *__restrict__ pm_base, int64_t pm_stride, const int32_t *__restrict__ offsets,
           const int32_t *__restrict__ counts, int32_t count_cap, const int32_t *__restrict__ rand3, double *__restrict__ xyz,
           int64_t xyz_stride, double *__restrict__ tr_out, int32_t *__restrict__ ok_out, int32_t *__restrict__ ninl_out,
           int32_t *__restrict__ inl_out, int64_t inl_stride) {
  __shared__ double sTr[6];
  __shared__ double sAcc[EGO_T / 64][27];
  __shared__ unsigned long long sBestKey;  // inlier count << 32 | (2^31 - 1 - hypothesis), 0: none yet
  __shared__ int32_t sWave[EGO_T / 64], sFlag, sBase;
  const int32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // matches of this stream: a slice of a concatenated list (offsets) or a fixed-stride slot with a device-side count
  const vh_p_match *pm = offsets ? pm_base + offsets[s] : pm_base + (int64_t)s * pm_stride;
  const int32_t n = offsets ? offsets[s + 1] - offsets[s] : min(counts[s], count_cap);
  double *X = xyz + (int64_t)s * xyz_stride * 4, *Y = X + xyz_stride, *Z = Y + xyz_stride, *F = Z + xyz_stride;  // F: inlier flags of the winner
  int32_t *inl = inl_out ? inl_out + (offsets ? (int64_t)offsets[s] : (int64_t)s * inl_stride) : nullptr;
  if (n < 6) {  // src/viso_stereo.cpp:68-69
    if (tid == 0) { ok_out[s] = 0; ninl_out[s] = 0; for (int32_t m = 0; m < 6; m++) tr_out[6 * s + m] = 0; }
    return;
  }
  VH_ETK_INIT;
  // 1. project the matches of the previous image into 3d (src/viso_stereo.cpp:80-85)
  for (int32_t i = tid; i < n; i += EGO_T) {
    const float df = pm[i].u1p - pm[i].u2p;
    const double d = (double)(df > 0.0001f ? df : 0.0001f);
    X[i] = (pm[i].u1p - e.cu) * e.base / d;
    Y[i] = (pm[i].v1p - e.cv) * e.base / d;
    Z[i] = e.f * e.base / d;
  }
  if (tid == 0) sBestKey = 0;
  __syncthreads();
  VH_ETK(0);  // 3-d points

  // 2. hypotheses: one per thread
  for (int32_t k0 = 0; k0 < e.ransac_iters; k0 += EGO_T) {
    const int32_t k = k0 + tid;
    unsigned long long key = 0;
    double t6[6] = {0, 0, 0, 0, 0, 0};
    if (k < e.ransac_iters) {
      // getRandomSample(N,3): three draws without replacement from the ordered index list (src/viso.cpp:96-102)
      const int32_t *r = rand3 + ((int64_t)s * e.ransac_iters + k) * 3;
      // (rand() returns 0 .. RAND_MAX; the sign bit of a caller-supplied value is dropped rather than turned into a negative index)
      int32_t a = (r[0] & 0x7FFFFFFF) % n, b = (r[1] & 0x7FFFFFFF) % (n - 1), c = (r[2] & 0x7FFFFFFF) % (n - 2);
      b += b >= a ? 1 : 0;
      const int32_t lo = min(a, b), hi = max(a, b);
      c += c >= lo ? 1 : 0;
      c += c >= hi ? 1 : 0;
      const int32_t act[3] = {a, b, c};
      EgoObs o3[3];
      for (int32_t q = 0; q < 3; q++) o3[q] = ego_load(pm, X, Y, Z, act[q]);
      // minimise the reprojection errors of the sample (src/viso_stereo.cpp:104-110, :179-227)
      int32_t iter = 0;
      bool failed = false;
      for (;;) {
        EgoRot R;
        ego_rot(t6, R);
        double acc[27];
        for (int32_t q = 0; q < 27; q++) acc[q] = 0;
        for (int32_t q = 0; q < 3; q++) ego_accumulate(e, R, t6, o3[q], acc);
        double bsol[6];
        if (!ego_solve(acc, bsol)) { failed = true; break; }
        bool converged = true;
        for (int32_t m = 0; m < 6; m++) { t6[m] += bsol[m]; if (fabs(bsol[m]) > 1e-6) converged = false; }
        if (iter++ > 20 || converged) break;
      }
#ifdef VH_EGO_TIMING
      if (blockIdx.x == 0 && tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_ego_t[1], now_ - et_prev_); et_prev_ = now_; }  // Gauss-Newton on the sample (wave 0's)
#endif
      if (!failed) {  // its inliers over all matches (src/viso_stereo.cpp:113-119)
        EgoRot R;
        ego_rot(t6, R);
        int32_t cnt = 0;
        for (int32_t i = 0; i < n; i++) cnt += ego_is_inlier(e, R, t6, ego_load(pm, X, Y, Z, i)) ? 1 : 0;
        // more inliers win, the earlier hypothesis on ties (strict `>` in iteration order); a
        // hypothesis without inliers never replaces the initial empty set
        if (cnt > 0) key = ((unsigned long long)cnt << 32) | (unsigned long long)(0x7FFFFFFF - k);
      }
    }
    // 3. the best hypothesis so far: wave arg max, then across waves
    unsigned long long best = key;
#pragma unroll
    for (int32_t d = 32; d >= 1; d >>= 1) {
      const unsigned long long other = ((unsigned long long)(uint32_t)__shfl_xor((int32_t)(best >> 32), d) << 32) | (uint32_t)__shfl_xor((int32_t)(uint32_t)best, d);
      best = other > best ? other : best;
    }
    if (lane == 0 && best) atomicMax(&sBestKey, best);
    __syncthreads();
    if (key && key == sBestKey) for (int32_t m = 0; m < 6; m++) sTr[m] = t6[m];  // unique: the hypothesis number is part of the key
    __syncthreads();
  }
  VH_ETK(2);  // inlier counts + arg max
  const int32_t nbest = (int32_t)(sBestKey >> 32);
  // ordered inlier list of the winner (VisualOdometry::inliers)
  double tr[6];
  for (int32_t m = 0; m < 6; m++) tr[m] = nbest ? sTr[m] : 0.0;
  __syncthreads();
  if (nbest) {
    EgoRot R;
    ego_rot(tr, R);
    if (tid == 0) sBase = 0;
    __syncthreads();
    for (int32_t i0 = 0; i0 < n; i0 += EGO_T) {
      const int32_t i = i0 + tid;
      const bool in_ = i < n && ego_is_inlier(e, R, tr, ego_load(pm, X, Y, Z, i));
      if (i < n) F[i] = in_ ? 1.0 : 0.0;
      const uint64_t bal = __ballot(in_);
      if (lane == 0) sWave[w] = __popcll(bal);
      __syncthreads();
      int32_t off = sBase;
      for (int32_t q = 0; q < w; q++) off += sWave[q];
      if (in_ && inl) inl[off + __popcll(bal & ((1ull << lane) - 1))] = i;
      __syncthreads();
      if (tid == 0) sBase += sWave[0] + sWave[1] + sWave[2] + sWave[3];
      __syncthreads();
    }
  }
  VH_ETK(3);  // inlier list
  // 4. final optimisation on the inliers (src/viso_stereo.cpp:123-139)
  bool success = nbest >= 6;
  if (success) {
    int32_t iter = 0;
    for (;;) {
      EgoRot R;
      ego_rot(tr, R);
      double acc[27];
      for (int32_t q = 0; q < 27; q++) acc[q] = 0;
      for (int32_t i = tid; i < n; i += EGO_T) {
        const EgoObs o = ego_load(pm, X, Y, Z, i);
        if (F[i] != 0.0) ego_accumulate(e, R, tr, o, acc);  // the inlier set of the RANSAC winner, fixed
      }
      for (int32_t q = 0; q < 27; q++) {
        double v = acc[q];
#pragma unroll
        for (int32_t d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0) sAcc[w][q] = v;
      }
      __syncthreads();
      if (tid == 0) {
        double tot[27], bsol[6];
        for (int32_t q = 0; q < 27; q++) tot[q] = (sAcc[0][q] + sAcc[1][q]) + (sAcc[2][q] + sAcc[3][q]);
        int32_t flag = 2;  // failed
        if (ego_solve(tot, bsol)) {
          flag = 1;        // converged
          for (int32_t m = 0; m < 6; m++) { if (fabs(bsol[m]) > 1e-8) flag = 0; }
          for (int32_t m = 0; m < 6; m++) sAcc[0][m] = bsol[m];
        }
        sFlag = flag;
      }
      __syncthreads();
      const int32_t flag = sFlag;
      if (flag != 2) for (int32_t m = 0; m < 6; m++) tr[m] += sAcc[0][m];
      __syncthreads();
      if (flag == 2) { success = false; break; }                 // FAILED
      if (flag == 1) break;                                      // CONVERGED
      if (iter++ > 100) { success = false; break; }              // still UPDATED after 102 updates
    }
  }
  VH_ETK(4);  // refit
  if (tid == 0) {
    ok_out[s] = success ? 1 : 0;
    ninl_out[s] = nbest;
    for (int32_t m = 0; m < 6; m++) tr_out[6 * s + m] = success ? tr[m] : 0.0;  // the reference returns an empty vector on failure
  }
}

}  // namespace

extern "C" int32_t vh_debug_ego_timing(unsigned long long *out, int32_t reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ego_t), sizeof(g_ego_t)) != hipSuccess) return -3;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_ego_t), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}

void vh_launch_ego(const vh_ego_params &e, int32_t n_sets, const vh_p_match *pm, int64_t pm_stride, const int32_t *offsets,
                   const int32_t *counts, int32_t count_cap, const int32_t *rand3, double *xyz, int64_t xyz_stride, double *tr,
                   int32_t *ok, int32_t *ninl, int32_t *inl, int64_t inl_stride, hipStream_t st) {
  hipLaunchKernelGGL(ego_kernel, dim3(n_sets), dim3(EGO_T), 0, st, e, pm, pm_stride, offsets, counts, count_cap, rand3, xyz,
                     xyz_stride, tr, ok, ninl, inl, inl_stride);
}
