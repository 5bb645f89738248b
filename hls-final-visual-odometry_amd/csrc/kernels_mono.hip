// kernels_mono.hip -- batched monocular egomotion (8-point RANSAC + SVD) on gfx950, SURVEY 8(f-4).
//
// Replaces, for every camera stream of a batch:
//   VisualOdometryMono::estimateMotion          (reference src/viso_mono.cpp:41-160)
//   ...::normalizeFeaturePoints / fundamentalMatrix / getInlier / EtoRt / triangulateChieral /
//        smallerThanMedian                      (src/viso_mono.cpp:162-399)
//   Matrix::svd, operator*, lu/det              (src/matrix.cpp:579-802, :263-277, :400-415, :514-572)
//   VisualOdometry::getRandomSample(N,8)        (src/viso.cpp:86-106), from caller-supplied rand() values
//
// Double precision (single where the reference's p_match fields are float), built with
// -ffp-contract=off, every sum in the reference's order, so that each hypothesis' F -- and with it
// every inlier set -- is bit-identical to the reference's; only exp/sin/cos/asin come from the device
// library (the ground-plane vote and the final angles agree to rounding).
//
// Six launches per batch (S lists; DESIGN.md section 6 has the measurements behind each choice):
//   mono_norm     one workgroup per list: centroids and scales -- sequential sums (the order of a floating-point
//                 sum is part of the result) over terms all lanes fetched into LDS; normalised points as float4
//                 and, widened once, as doubles.
//   mono_hyp<0>   one THREAD per RANSAC hypothesis, grid (iters/128, S): 8-point system, Matrix::svd of the 8x9
//                 matrix in registers WITHOUT U (its null vector up to sign: svd_static.h), rank-2 F by a 3x3 SVD,
//                 Sampson inlier count over all matches (quotient-free unless the answer is in doubt); arg max per
//                 list with one 64-bit atomicMax (count << 32 | 2^31-1-k: more inliers win, the earlier
//                 hypothesis on ties, as the reference's strict `>`); every hypothesis leaves its F in scratch.
//                 A hypothesis whose +-F could differ in more than sign (flag of the 3x3 decomposition) has both
//                 candidates compared by its wave; only if they disagree on a match is it queued for
//   mono_hyp<1>   the same with U and the reference's sign, one lane per queued hypothesis (normally none).
//   mono_final_a  one workgroup per list: ordered inlier list of the winner (its stored F), F from all inliers
//                 (a workgroup-cooperative Matrix::svd of the N x 9 system: everything elementwise spread over the
//                 lanes, every sum a sequential chain over terms formed beforehand), E, the four (R,t) candidates.
//   mono_tri      one lane per (match, candidate): linear triangulation (4x4 SVD without U), chirality counts.
//   mono_final_c  one workgroup per list: the winning candidate, points in front, median distance, ground-plane
//                 vote, scale, angles.
#include "vh_dev.h"
#include "../../include/viso_hip.h"
#include <math.h>
#define SVD_HD __device__ __forceinline__
#include "svd_static.h"

namespace {

#define MONO_T 256
#ifndef MONO_LDS_ROWS
#define MONO_LDS_ROWS 640  // inlier sets of bucketed lists (a few hundred) fit; longer ones take the global-memory path.  mono_final_a holds
                           // the refit system (46 KB) AND the cooperative SVD's term scratch (41 KB) in LDS: 88 KB per workgroup -- gfx950's 160 KB only
#endif

__device__ __forceinline__ double sign_of(double a, double b) { return b >= 0.0 ? fabs(a) : -fabs(a); }

__device__ double pythag(double a, double b) {  // src/matrix.cpp:846-854
  const double absa = fabs(a), absb = fabs(b);
  if (absa > absb) { const double q = absb / absa; return absa * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q)); }
  if (absb == 0.0) return 0.0;
  const double q = absa / absb;
  return absb * sqrt(1.0 + (q == 0.0 ? 0.0 : q * q));
}

// The diagonalisation of the bidiagonal form (src/matrix.cpp:669-761) as a sequence of plane
// rotations: the scalars (w, rv1) are advanced here, each rotation of two columns of U or V is handed
// to `rot_u(col_a, col_b, c, s)` / `rot_v(...)`, a negated column of V to `neg_v(col)`.  The scalars
// never depend on U or V, so callers may apply the rotations to any partition of the rows.
template <class RotU, class RotV, class NegV>
__device__ __forceinline__ void svd_qr_phase(int n, double *w, double *rv1, double anorm, RotU rot_u, RotV rot_v, NegV neg_v) {
  for (int k = n - 1; k >= 0; k--) {
    for (int its = 0; its < 30; its++) {
      int flag = 1, l, nm = 0;
      for (l = k; l >= 0; l--) {
        nm = l - 1;
        if ((double)(fabs(rv1[l]) + anorm) == anorm) { flag = 0; break; }
        if ((double)(fabs(w[nm]) + anorm) == anorm) break;
      }
      double c, s, f, g, h, x, y, z;
      if (flag) {
        c = 0.0; s = 1.0;
        for (int i = l; i <= k; i++) {
          f = s * rv1[i];
          rv1[i] = c * rv1[i];
          if ((double)(fabs(f) + anorm) == anorm) break;
          g = w[i];
          h = pythag(f, g);
          w[i] = h;
          h = 1.0 / h;
          c = g * h;
          s = -f * h;
          rot_u(nm, i, c, s);
        }
      }
      z = w[k];
      if (l == k) {
        if (z < 0.0) { w[k] = -z; neg_v(k); }
        break;
      }
      x = w[l];
      nm = k - 1;
      y = w[nm];
      g = rv1[nm];
      h = rv1[k];
      f = ((y - z) * (y + z) + (g - h) * (g + h)) / (2.0 * h * y);
      g = pythag(f, 1.0);
      f = ((x - z) * (x + z) + h * ((y / (f + sign_of(g, f))) - h)) / x;
      c = s = 1.0;
      for (int j = l; j <= nm; j++) {
        const int i = j + 1;
        g = rv1[i];
        y = w[i];
        h = s * g;
        g = c * g;
        z = pythag(f, h);
        rv1[j] = z;
        c = f / z;
        s = h / z;
        f = x * c + g * s;
        g = g * c - x * s;
        h = y * s;
        y *= c;
        rot_v(j, i, c, s);
        z = pythag(f, h);
        w[j] = z;
        if (z) { z = 1.0 / z; c = f * z; s = h * z; }
        f = c * g + s * y;
        x = c * y - s * g;
        rot_u(j, i, c, s);
      }
      rv1[l] = 0.0;
      rv1[k] = f;
      w[k] = x;
    }
  }
}

// The shell sort of the singular values (src/matrix.cpp:770-790) as a sequence of column moves:
// save(col) -> temporary, move(dst, src), restore(dst) <- temporary.
template <class Save, class Move, class Restore>
__device__ __forceinline__ void svd_sort_phase(int n, double *w, Save save, Move move, Restore restore) {
  int inc = 1;
  do { inc *= 3; inc++; } while (inc <= n);
  do {
    inc /= 3;
    for (int i = inc; i < n; i++) {
      const double sw = w[i];
      save(i);
      int j = i;
      while (w[j - inc] < sw) {
        w[j] = w[j - inc];
        move(j, j - inc);
        j -= inc;
        if (j < inc) break;
      }
      w[j] = sw;
      restore(j);
    }
  } while (inc > 1);
}

// C = A (ma x na) * B (na x nb): Matrix::operator* (src/matrix.cpp:263-277), sums from 0, k ascending
__device__ void matmul(const double *A, int ma, int na, const double *B, int nb, double *C) {
  for (int i = 0; i < ma; i++)
    for (int j = 0; j < nb; j++) {
      double c = 0.0;
      for (int k = 0; k < na; k++) c += A[i * na + k] * B[k * nb + j];
      C[i * nb + j] = c;
    }
}
__device__ void transpose3(const double *A, double *T) {
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[j * 3 + i] = A[i * 3 + j];
}
// Matrix::svd of a 3x3 followed by U * diag(W with W[2] = 0) * ~V (src/viso_mono.cpp:262-265, :91-94).
// The factors live in registers (svd_static.h: every index static after unrolling).
// (returns svd_static's zero-pivot flag, see fundamental8)
__device__ __forceinline__ bool rank2_3x3(const double *M, double *out) {
  double U[3][3], w[3], V[3][3], a[9], D[9], UD[9], Vt[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) U[i][j] = M[i * 3 + j];
  const bool zero_pivot = svd_static<3, 3>(U, w, V);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { a[i * 3 + j] = U[i][j]; Vt[j * 3 + i] = V[i][j]; D[i * 3 + j] = 0.0; }
  D[0] = w[0]; D[4] = w[1]; D[8] = 0.0;
  matmul(a, 3, 3, D, 3, UD);
  matmul(UD, 3, 3, Vt, 3, out);
  return zero_pivot;
}

// Matrix::det of a 3x3 (src/matrix.cpp:400-415) over Matrix::lu (:514-572)
__device__ double det3(const double *M) {
  double a[3][3], vv[3], big, dum, sum, temp, d = 1.0;
  int imax = 0;
  bool ok = true;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a[i][j] = M[i * 3 + j];
  for (int i = 0; i < 3 && ok; i++) {
    big = 0.0;
    for (int j = 0; j < 3; j++) if ((temp = fabs(a[i][j])) > big) big = temp;
    if (big == 0.0) { ok = false; break; }
    vv[i] = 1.0 / big;
  }
  if (ok)
    for (int j = 0; j < 3; j++) {
      for (int i = 0; i < j; i++) {
        sum = a[i][j];
        for (int k = 0; k < i; k++) sum -= a[i][k] * a[k][j];
        a[i][j] = sum;
      }
      big = 0.0;
      for (int i = j; i < 3; i++) {
        sum = a[i][j];
        for (int k = 0; k < j; k++) sum -= a[i][k] * a[k][j];
        a[i][j] = sum;
        if ((dum = vv[i] * fabs(sum)) >= big) { big = dum; imax = i; }
      }
      if (j != imax) {
        for (int k = 0; k < 3; k++) { dum = a[imax][k]; a[imax][k] = a[j][k]; a[j][k] = dum; }
        d = -d;
        vv[imax] = vv[j];
      }
      if (j != 2) {
        dum = 1.0 / a[j][j];
        for (int i = j + 1; i < 3; i++) a[i][j] *= dum;
      }
    }
  for (int i = 0; i < 3; i++) d *= a[i][i];
  return d;
}

// -DVH_MONO_TIMING: workgroup 0 of mono_final adds the 100 MHz clock ticks of its phases to g_mono_t (tools/mono_phases.py)
__device__ unsigned long long g_mono_t[16];
#ifdef VH_MONO_TIMING
#define VH_MTICK_INIT unsigned long long mt_prev_ = wall_clock64()
#define VH_MTICK(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_mono_t[k], now_ - mt_prev_); mt_prev_ = now_; } } while (0)
#else
#define VH_MTICK_INIT do { } while (0)
#define VH_MTICK(k) do { } while (0)
#endif

struct __attribute__((aligned(16))) MonoPoint { double u1, v1, u2, v2; };  // normalised (u1p, v1p, u1c, v1c) as doubles

// hdr slots written by mono_final_a for mono_tri / mono_final_c
#define MONO_H_GO 32      /* 1: the later kernels have work (0: mono_final_a already reported the failure) */
#define MONO_H_NBEST 33
#define MONO_H_RT 34      /* Ra[9], Rb[9], t0[3] */
#define MONO_H_P2 56      /* P2 of the four (R, t) candidates, [4][12] */
#define MONO_H_P1 104     /* [K | 0], [12] */
#define MONO_H_CNT 116    /* int32[4]: points in front of both cameras per candidate */

struct MonoList {
  const vh_p_match *pm;  // the list's matches
  int32_t n;
  float4 *pn;            // normalised (u1p, v1p, u1c, v1c)
  double *A;             // [cap][9] refit system / its U factor
  double *X;             // [4 solutions][4][cap] triangulated points
  double *d, *dist;      // [cap] ground-plane coordinate / L1 distance of the points in front
  int32_t *idx;          // [cap] inlier indices of the winner
  double *hdr;           // [128] per list: Tp[9], Tc[9], [18] status; from [32] what mono_final's three kernels hand on (MONO_H_*)
  unsigned long long *key;
};

// bytes of scratch per list: pn 16*cap | A 72*cap | X 128*cap | d 8*cap | dist 8*cap | idx 4*cap | hdr 1024 | key 8, rounded to 16
__host__ __device__ inline int64_t mono_per_list(int64_t cap) { return (236 * cap + 1024 + 8 + 15) / 16 * 16; }

__device__ __forceinline__ MonoList mono_list(int32_t s, const vh_p_match *pm_base, int64_t pm_stride, const int32_t *offsets,
                                              const int32_t *counts, int32_t count_cap, uint8_t *scratch, int64_t cap) {
  MonoList L;
  L.pm = offsets ? pm_base + offsets[s] : pm_base + (int64_t)s * pm_stride;
  L.n = offsets ? offsets[s + 1] - offsets[s] : min(counts[s], count_cap);
  uint8_t *b = scratch + (int64_t)s * mono_per_list(cap);
  L.pn = (float4 *)b; b += 16 * cap;
  L.A = (double *)b; b += 72 * cap;
  L.X = (double *)b; b += 128 * cap;
  L.d = (double *)b; b += 8 * cap;
  L.dist = (double *)b; b += 8 * cap;
  L.hdr = (double *)b; b += 1024;
  L.key = (unsigned long long *)b; b += 8;
  L.idx = (int32_t *)b;
  return L;
}

// Behind the lists: the queue of hypotheses whose inliers the signed kernel has to count (mono_hyp): count, then entries s * iters + k.
__device__ __forceinline__ uint32_t *mono_queue(uint8_t *scratch, int32_t n_sets, int64_t cap) {
  return (uint32_t *)(scratch + (int64_t)n_sets * mono_per_list(cap));
}
__host__ __device__ inline int64_t mono_queue_bytes(int32_t n_sets, int32_t iters) { return 16 + ((int64_t)n_sets * iters * 4 + 15) / 16 * 16; }
// ... and behind the queue every hypothesis' F (+-F from the fast kernel), [n_sets * iters][9]: mono_final lists the
// winner's inliers with it instead of factorizing the winner's 8-point system again on one lane (125 us of latency).
__device__ __forceinline__ double *mono_hyp_F(uint8_t *scratch, int32_t n_sets, int64_t cap, int32_t iters) {
  return (double *)(scratch + (int64_t)n_sets * mono_per_list(cap) + mono_queue_bytes(n_sets, iters));
}

// s + term(0) + term(1) + .. in this order, terms at T[k * 4 + c] (the loads of a trip are independent of the chain: chain_sum)
__device__ __forceinline__ double chain_sum4(const double *T, int c, int m, double s) {
  int k = 0;
  for (; k + 8 <= m; k += 8) {
    double t[8];
#pragma unroll
    for (int q = 0; q < 8; q++) t[q] = T[(k + q) * 4 + c];
#pragma unroll
    for (int q = 0; q < 8; q++) s += t[q];
  }
  for (; k < m; k++) s += T[k * 4 + c];
  return s;
}

// ------------------------------------------------------------------ mono_norm
// normalizeFeaturePoints (src/viso_mono.cpp:187-233).  hdr[18] = 1: usable, 0: the reference returns
// an empty vector (N < 10 or a degenerate scale).
__global__ void __launch_bounds__(MONO_T)
mono_norm_kernel(const vh_p_match *__restrict__ pm_base, int64_t pm_stride, const int32_t *__restrict__ offsets,
                 const int32_t *__restrict__ counts, int32_t count_cap, uint8_t *__restrict__ scratch, int64_t cap) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *mono_queue(scratch, gridDim.x, cap) = 0u;  // (hypotheses for the signed recount, mono_hyp)
  __shared__ double sC[4], sS[2];
  // The sums of normalizeFeaturePoints are sequential chains over the matches (their order is part of the result):
  // the terms of a block of matches are fetched by all lanes into LDS, then one lane per sum walks them (chain_sum4).
  constexpr int32_t NB = 1024;
  __shared__ double sTerm[NB * 4];
  const int32_t s = blockIdx.x, tid = threadIdx.x;
  const MonoList L = mono_list(s, pm_base, pm_stride, offsets, counts, count_cap, scratch, cap);
  const int32_t N = L.n;
  if (tid == 0) { *L.key = 0ull; L.hdr[18] = 0.0; }
  if (N < 10) return;  // src/viso_mono.cpp:45-46
  {                    // centroids: one sequential sum per lane
    double c = 0;
    for (int32_t b0 = 0; b0 < N; b0 += NB) {
      const int32_t nb = min(NB, N - b0);
      for (int32_t i = tid; i < nb; i += MONO_T) {
        const vh_p_match &m = L.pm[b0 + i];
        sTerm[i * 4 + 0] = (double)m.u1p; sTerm[i * 4 + 1] = (double)m.v1p; sTerm[i * 4 + 2] = (double)m.u1c; sTerm[i * 4 + 3] = (double)m.v1c;
      }
      __syncthreads();
      if (tid < 4) c = chain_sum4(sTerm, tid, nb, c);
      __syncthreads();
    }
    if (tid < 4) sC[tid] = c / (double)N;
  }
  __syncthreads();
  const double cpu = sC[0], cpv = sC[1], ccu = sC[2], ccv = sC[3];
  float2 *fl = (float2 *)L.d;  // (scratch: the distances from the centroids)
  for (int32_t i = tid; i < N; i += MONO_T) {
    const vh_p_match &m = L.pm[i];
    float4 q;
    q.x = (float)((double)m.u1p - cpu); q.y = (float)((double)m.v1p - cpv);
    q.z = (float)((double)m.u1c - ccu); q.w = (float)((double)m.v1c - ccv);
    L.pn[i] = q;
    fl[i] = make_float2(sqrtf(q.x * q.x + q.y * q.y), sqrtf(q.z * q.z + q.w * q.w));  // float expressions in the reference
  }
  __syncthreads();
  {
    double acc = 0;
    for (int32_t b0 = 0; b0 < N; b0 += NB) {
      const int32_t nb = min(NB, N - b0);
      for (int32_t i = tid; i < nb; i += MONO_T) { const float2 v = fl[b0 + i]; sTerm[i * 4 + 0] = (double)v.x; sTerm[i * 4 + 1] = (double)v.y; }
      __syncthreads();
      if (tid < 2) acc = chain_sum4(sTerm, tid, nb, acc);
      __syncthreads();
    }
    if (tid < 2) sS[tid] = acc;
  }
  __syncthreads();
  double sp = sS[0], sc = sS[1];
  if (fabs(sp) < 1e-10 || fabs(sc) < 1e-10) return;
  sp = sqrt(2.0) * (double)N / sp;
  sc = sqrt(2.0) * (double)N / sc;
  for (int32_t i = tid; i < N; i += MONO_T) {
    float4 q = L.pn[i];
    q.x = (float)((double)q.x * sp); q.y = (float)((double)q.y * sp);
    q.z = (float)((double)q.z * sc); q.w = (float)((double)q.w * sc);
    L.pn[i] = q;
    // the same four floats widened once (exact), for mono_hyp's loop over the matches; the refit system's place is free until mono_final
    MonoPoint d4; d4.u1 = q.x; d4.v1 = q.y; d4.u2 = q.z; d4.v2 = q.w;
    ((MonoPoint *)L.A)[i] = d4;
  }
  if (tid == 0) {
    double *Tp = L.hdr, *Tc = L.hdr + 9;
    Tp[0] = sp; Tp[1] = 0; Tp[2] = -sp * cpu; Tp[3] = 0; Tp[4] = sp; Tp[5] = -sp * cpv; Tp[6] = 0; Tp[7] = 0; Tp[8] = 1;
    Tc[0] = sc; Tc[1] = 0; Tc[2] = -sc * ccu; Tc[3] = 0; Tc[4] = sc; Tc[5] = -sc * ccv; Tc[6] = 0; Tc[7] = 0; Tc[8] = 1;
    L.hdr[18] = 1.0;
  }
}

// fundamentalMatrix on the 8 sampled matches (src/viso_mono.cpp:235-266), one lane.
// SIGNED = true: F as the reference computes it, bit for bit.
// SIGNED = false: +F or -F -- the null vector of the 8x9 system is taken without Matrix::svd's sign normalisation,
//   which is the only consumer of that decomposition's U (72 doubles, its accumulation and half of the plane
//   rotations: svd_static.h).  Both callers only COUNT / LIST inliers, and the Sampson test is the same for F and
//   -F as long as the rank-2 projection of -F0 is the exact mirror image of that of F0; the one place where it need
//   not be (a Householder pivot that is exactly zero) is reported through the return value, and the caller then
//   takes the signed form.
template <bool SIGNED>
__device__ __forceinline__ bool fundamental8(const float4 *pn, const int32_t *act, double *F, double (&F0)[9]) {
  double Ur[8][9], wr[9], Vr[9][9];
#pragma unroll
  for (int32_t i = 0; i < 8; i++) {
    const float4 q = pn[act[i]];  // (u1p, v1p, u1c, v1c)
    Ur[i][0] = (double)(q.z * q.x); Ur[i][1] = (double)(q.z * q.y); Ur[i][2] = (double)q.z;  // float products
    Ur[i][3] = (double)(q.w * q.x); Ur[i][4] = (double)(q.w * q.y); Ur[i][5] = (double)q.w;
    Ur[i][6] = (double)q.x; Ur[i][7] = (double)q.y; Ur[i][8] = 1.0;
  }
  if (SIGNED) svd_static_last_v<8, 9>(Ur, wr, Vr, F0);
  else svd_static_last_v_unsigned<8, 9>(Ur, wr, Vr, F0);
  const bool zero_pivot = rank2_3x3(F0, F);
  return !SIGNED && zero_pivot;
}

// getRandomSample(N, 8) from eight rand() values (src/viso.cpp:96-102)
__device__ void sample8(const int32_t *r, int32_t N, int32_t *act) {
  int32_t taken[8], nt = 0;
  for (int32_t k = 0; k < 8; k++) {
    int32_t j = (int32_t)((uint32_t)(r[k] & 0x7FFFFFFF) % (uint32_t)(N - k));
    for (int32_t q = 0; q < nt; q++) if (j >= taken[q]) j++;
    act[k] = j;
    int32_t q = nt;
    for (; q > 0 && taken[q - 1] > j; q--) taken[q] = taken[q - 1];
    taken[q] = j; nt++;
  }
}

// Sampson distance test of getInlier (src/viso_mono.cpp:283-309)
__device__ __forceinline__ bool sampson_inlier(const double *F, const float4 q, double thr) {
  const double u1 = q.x, v1 = q.y, u2 = q.z, v2 = q.w;
  const double Fx1u = F[0] * u1 + F[1] * v1 + F[2], Fx1v = F[3] * u1 + F[4] * v1 + F[5], Fx1w = F[6] * u1 + F[7] * v1 + F[8];
  const double Ftx2u = F[0] * u2 + F[3] * v2 + F[6], Ftx2v = F[1] * u2 + F[4] * v2 + F[7];
  const double x2tFx1 = u2 * Fx1u + v2 * Fx1v + Fx1w;
  const double d = x2tFx1 * x2tFx1 / (Fx1u * Fx1u + Fx1v * Fx1v + Ftx2u * Ftx2u + Ftx2v * Ftx2v);
  return fabs(d) < thr;
}

// The same test for mono_hyp's loop (one lane = one hypothesis, every lane the same match): numerator and denominator
// as above, operation for operation, but the quotient is formed only when the answer is not already certain.
// With p = fl(thr * den): n < fl(p * (1 - 2^-50)) implies n / den < thr * (1 - 2^-51), whose rounding is still below
// thr; n > fl(p * (1 + 2^-50)) implies the rounded quotient is above thr; in between (a band of 2^-49 relative
// width), for a den outside the range where these error bounds hold, and for NaNs the division decides.
// Returns 1 / 0, or the answer + 2 when the division has to decide (sampson_settle).
__device__ __forceinline__ int32_t sampson_inlier_nodiv(const double *F, const MonoPoint q, double thr, double &n, double &den) {
  const double u1 = q.u1, v1 = q.v1, u2 = q.u2, v2 = q.v2;
  const double Fx1u = F[0] * u1 + F[1] * v1 + F[2], Fx1v = F[3] * u1 + F[4] * v1 + F[5], Fx1w = F[6] * u1 + F[7] * v1 + F[8];
  const double Ftx2u = F[0] * u2 + F[3] * v2 + F[6], Ftx2v = F[1] * u2 + F[4] * v2 + F[7];
  const double x2tFx1 = u2 * Fx1u + v2 * Fx1v + Fx1w;
  n = x2tFx1 * x2tFx1;
  den = Fx1u * Fx1u + Fx1v * Fx1v + Ftx2u * Ftx2u + Ftx2v * Ftx2v;
  const double p = thr * den;
  // (bitwise, not short-circuit: no control flow in the common case)
  const int32_t lo = n < p * (1.0 - 0x1p-50) ? 1 : 0, hi = n > p * (1.0 + 0x1p-50) ? 1 : 0;
  const int32_t certain = (lo | hi) & (den > 1e-200 ? 1 : 0) & (den < 1e200 ? 1 : 0);
  return lo | ((certain ^ 1) << 1);
}
__device__ __forceinline__ int32_t sampson_settle(int32_t r, double n, double den, double thr) {
  return (r & 2) ? (fabs(n / den) < thr ? 1 : 0) : r;
}

// ------------------------------------------------------------------- mono_hyp
// One lane per hypothesis.  The SIGNED = false launch does the work; a hypothesis that meets the mirror-image hazard
// of fundamental8 (a numerically singular 3x3: noise-free scenes have them, image data hardly ever) AND whose two
// candidate matrices disagree on a match does not vote there but is queued, and the SIGNED = true launch behind it
// counts the queued ones the reference's way into the same key.
// force_signed (VH_MONO_SIGNED=1, the tests): every hypothesis takes the second launch.
template <bool SIGNED>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(SIGNED ? 1 : 2, SIGNED ? 1 : 2)))  // (the fast form fits 256 registers: two waves per SIMD)
mono_hyp_kernel(vh_mono_params e, const vh_p_match *__restrict__ pm_base, int64_t pm_stride, const int32_t *__restrict__ offsets,
                const int32_t *__restrict__ counts, int32_t count_cap, const int32_t *__restrict__ rand8,
                uint8_t *__restrict__ scratch, int64_t cap, int32_t n_sets, int32_t force_signed) {
  uint32_t *queue = mono_queue(scratch, n_sets, cap);
  int32_t s, k;
  if (SIGNED) {
    const uint32_t q = blockIdx.x * 128u + threadIdx.x;
    if (q >= queue[0]) return;
    const uint32_t ent = queue[4 + q];
    s = (int32_t)(ent / (uint32_t)e.ransac_iters); k = (int32_t)(ent % (uint32_t)e.ransac_iters);
  } else {
    s = blockIdx.y; k = blockIdx.x * 128 + threadIdx.x;
  }
  const MonoList L = mono_list(s, pm_base, pm_stride, offsets, counts, count_cap, scratch, cap);
  // (no lane leaves before the cooperative check below: a wave's lanes are hypotheses of one list)
  const bool valid = L.hdr[18] != 0.0 && k < e.ransac_iters;
  if (SIGNED && !valid) return;
  int32_t act[8];
  double F[9];
  const MonoPoint *__restrict__ pd = (const MonoPoint *)L.A;
  const double thr = e.inlier_threshold;
  bool hazard = !SIGNED && force_signed != 0;
  if (SIGNED) {
    double F0[9];
    sample8(rand8 + ((int64_t)s * e.ransac_iters + k) * 8, L.n, act);
    fundamental8<true>(L.pn, act, F, F0);
  } else {
    // A hypothesis that raised fundamental8's flag: the reference's F is the rank-2 projection of +F0 or of -F0 -- which
    // one, only U knows.  Both are at hand: when they agree on every match (they differ, if at all, in the last bits)
    // the sign does not matter and the hypothesis is counted here after all; otherwise it goes to the signed kernel.
    // The comparison is made by the whole wave, lanes over matches, for one flagged lane after the other: on its own
    // lane it would double that lane's loop over the matches and with it the duration of its wave.
    double Fm[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool flagged = false;
    if (valid && !hazard) {
      double F0[9];
      sample8(rand8 + ((int64_t)s * e.ransac_iters + k) * 8, L.n, act);
      flagged = fundamental8<false>(L.pn, act, F, F0);
      if (flagged) {
#pragma unroll
        for (int32_t q = 0; q < 9; q++) F0[q] = -F0[q];
        rank2_3x3(F0, Fm);
      }
    }
    const int32_t lane = threadIdx.x & 63;
    for (uint64_t todo = __ballot(flagged); todo; todo &= todo - 1) {
      const int32_t j = __ffsll((unsigned long long)todo) - 1;
      double Fa[9], Fb[9];
#pragma unroll
      for (int32_t q = 0; q < 9; q++) { Fa[q] = __shfl(F[q], j); Fb[q] = __shfl(Fm[q], j); }
      int32_t differ = 0;
      for (int32_t i = lane; i < L.n; i += 64) {
        double n1, d1, n2, d2;
        differ |= sampson_settle(sampson_inlier_nodiv(Fa, pd[i], thr, n1, d1), n1, d1, thr) ^ sampson_settle(sampson_inlier_nodiv(Fb, pd[i], thr, n2, d2), n2, d2, thr);
      }
      const bool any = __ballot(differ != 0) != 0;
      if (lane == j) hazard = any;
    }
    if (!valid) return;
  }
  if (hazard) { queue[4 + atomicAdd(queue, 1u)] = (uint32_t)s * (uint32_t)e.ransac_iters + (uint32_t)k; return; }
  {
    double *Fk = mono_hyp_F(scratch, n_sets, cap, e.ransac_iters) + ((int64_t)s * e.ransac_iters + k) * 9;
#pragma unroll
    for (int32_t q = 0; q < 9; q++) Fk[q] = F[q];
  }
  int32_t cnt = 0;
  int32_t i = 0;
  for (; i + 4 <= L.n; i += 4) {  // four independent chains per trip
    double n4[4], d4[4];
    int32_t r4[4];
#pragma unroll
    for (int32_t q = 0; q < 4; q++) r4[q] = sampson_inlier_nodiv(F, pd[i + q], thr, n4[q], d4[q]);
    if (__builtin_expect(((r4[0] | r4[1] | r4[2] | r4[3]) & 2) != 0, 0)) {
#pragma unroll
      for (int32_t q = 0; q < 4; q++) r4[q] = sampson_settle(r4[q], n4[q], d4[q], thr);
    }
    cnt += r4[0] + r4[1] + r4[2] + r4[3];
  }
  for (; i < L.n; i++) {
    double n1, d1;
    cnt += sampson_settle(sampson_inlier_nodiv(F, pd[i], thr, n1, d1), n1, d1, thr);
  }
  // more inliers win, the earlier hypothesis on ties (strict `>` in iteration order, src/viso_mono.cpp:74);
  // a hypothesis without inliers never replaces the initial empty set
  if (cnt > 0) atomicMax(L.key, ((unsigned long long)cnt << 32) | (unsigned long long)(0x7FFFFFFF - k));
}

// ----------------------------------------------------------------- mono_final
// Workgroup-cooperative Matrix::svd of the tall system A[m][9] (m >= 9) -- every lane of the
// workgroup calls it.  The sums over the m rows of a column are sequential chains (their order is
// part of the result), so the parallelism is across what the algorithm leaves independent: the
// columns j > i of a left Householder step, the rows of a right one, the rows of every plane
// rotation and column move.  The scalars of the QR phase are advanced redundantly by every lane
// (same arithmetic, same values), so that phase needs no exchange at all.
// On return: A = U (sorted, signs flipped), sV[81] = V, and w[9] in every lane.
// Sequential sum (k ascending, from 0.0) of chain c of the term array T[row][8], rows [k0, m): what a
// `for (k = k0; k < m; k++) s += term(k)` of the original computes, with the terms formed beforehand by all lanes.
// One lane per chain; the loads of a trip do not depend on the chain, so they are in flight together and the trip
// costs eight dependent additions, not eight memory round trips.
__device__ __forceinline__ double chain_sum(const double *T, int c, int k0, int m) {
  double s = 0.0;
  int k = k0;
  for (; k + 8 <= m; k += 8) {
    double t[8];
#pragma unroll
    for (int q = 0; q < 8; q++) t[q] = T[(int64_t)(k + q) * 8 + c];
#pragma unroll
    for (int q = 0; q < 8; q++) s += t[q];
  }
  for (; k < m; k++) s += T[(int64_t)k * 8 + c];
  return s;
}

// T: [m][8] doubles of term scratch (LDS when A is).
__device__ void svd_tall9(double *A, int m, double *sV, double *sX, int *sNeg, double *w, double *col_tmp, double *T) {
  const int n = 9, tid = threadIdx.x;
  double rv1[9];
  int i, j, k, l = 0;
  double anorm, g, h, s, scale;
#define AA(r, q) A[(int64_t)(r) * 9 + (q)]
#define VV(r, q) sV[(r) * 9 + (q)]
#define TT(r, q) T[(int64_t)(r) * 8 + (q)]
  g = scale = anorm = 0.0;
  for (i = 0; i < n; i++) {
    l = i + 1;
    rv1[i] = scale * g;
    g = s = scale = 0.0;
    // left Householder step on column i (i < m always: m >= 9).  Every sum over the rows is a chain in the original's
    // order (chain_sum); everything elementwise -- terms, the division by the scale, the update of the columns -- is
    // spread over the workgroup.
    for (k = i + tid; k < m; k += MONO_T) TT(k, 0) = fabs(AA(k, i));
    __syncthreads();
    if (tid == 0) sX[0] = chain_sum(T, 0, i, m);
    __syncthreads();
    scale = sX[0];
    if (scale) {
      for (k = i + tid; k < m; k += MONO_T) { const double v = AA(k, i) / scale; AA(k, i) = v; TT(k, 0) = v * v; }
      __syncthreads();
      if (tid == 0) {
        const double ss = chain_sum(T, 0, i, m);
        const double ff = AA(i, i), gg = -sign_of(sqrt(ss), ff);
        sX[1] = gg; sX[2] = ff * gg - ss;
        AA(i, i) = ff - gg;
      }
      __syncthreads();
      g = sX[1]; h = sX[2];
      for (k = i + tid; k < m; k += MONO_T) {
        const double ai = AA(k, i);
        for (j = l; j < n; j++) TT(k, j - l) = ai * AA(k, j);
      }
      __syncthreads();
      if (tid < n - l) sX[4 + tid] = chain_sum(T, tid, i, m) / h;
      __syncthreads();
      for (k = i + tid; k < m; k += MONO_T) {
        const double ai = AA(k, i);
        for (j = l; j < n; j++) AA(k, j) += sX[4 + j - l] * ai;
        AA(k, i) = ai * scale;
      }
    }
    __syncthreads();
    w[i] = scale * g;
    g = s = scale = 0.0;
    if (i != n - 1) {  // right Householder step on row i
      if (tid == 0) {
        double sc = 0.0;
        for (k = l; k < n; k++) sc += fabs(AA(i, k));
        sX[0] = sc;
        if (sc) {
          double ss = 0.0;
          for (k = l; k < n; k++) { AA(i, k) /= sc; ss += AA(i, k) * AA(i, k); }
          const double ff = AA(i, l), gg = -sign_of(sqrt(ss), ff), hh = ff * gg - ss;
          AA(i, l) = ff - gg;
          sX[1] = gg;
          for (k = l; k < n; k++) sX[4 + k] = AA(i, k) / hh;
        }
      }
      __syncthreads();
      scale = sX[0];
      if (scale) {
        g = sX[1];
        for (k = l; k < n; k++) rv1[k] = sX[4 + k];
        for (j = l + tid; j < m; j += MONO_T) {
          for (s = 0.0, k = l; k < n; k++) s += AA(j, k) * AA(i, k);
          for (k = l; k < n; k++) AA(j, k) += s * rv1[k];
        }
        __syncthreads();
        if (tid == 0) for (k = l; k < n; k++) AA(i, k) *= scale;
      }
      __syncthreads();
    }
    const double t = fabs(w[i]) + fabs(rv1[i]);
    anorm = anorm > t ? anorm : t;
  }
  // accumulation of right-hand transformations: 9 x 9, lane j = column j
  {
    double gg = g;
    int ll = l;
    for (i = tid; i < 81; i += MONO_T) sV[i] = 0.0;
    __syncthreads();
    for (i = n - 1; i >= 0; i--) {
      if (i < n - 1) {
        if (gg) {
          if (tid >= ll && tid < n) VV(tid, i) = (AA(i, tid) / AA(i, ll)) / gg;
          __syncthreads();
          if (tid >= ll && tid < n) {
            j = tid;
            double ss = 0.0;
            for (k = ll; k < n; k++) ss += AA(i, k) * VV(k, j);
            for (k = ll; k < n; k++) VV(k, j) += ss * VV(k, i);
          }
          __syncthreads();
        }
        if (tid >= ll && tid < n) VV(i, tid) = VV(tid, i) = 0.0;
      }
      if (tid == 0) VV(i, i) = 1.0;
      __syncthreads();
      gg = rv1[i];
      ll = i;
    }
  }
  // accumulation of left-hand transformations
  for (i = n - 1; i >= 0; i--) {
    l = i + 1;
    g = w[i];
    if (tid >= l && tid < n) AA(i, tid) = 0.0;
    __syncthreads();
    if (g) {
      g = 1.0 / g;
      for (k = l + tid; k < m; k += MONO_T) {
        const double ai = AA(k, i);
        for (j = l; j < n; j++) TT(k, j - l) = ai * AA(k, j);
      }
      __syncthreads();
      if (tid < n - l) sX[4 + tid] = (chain_sum(T, tid, l, m) / AA(i, i)) * g;
      __syncthreads();
      for (k = i + tid; k < m; k += MONO_T) {
        const double ai = AA(k, i);
        for (j = l; j < n; j++) AA(k, j) += sX[4 + j - l] * ai;
        AA(k, i) = ai * g;
      }
    } else {
      for (j = i + tid; j < m; j += MONO_T) AA(j, i) = 0.0;
    }
    __syncthreads();
    if (tid == 0) ++AA(i, i);
    __syncthreads();
  }
  // QR phase and sort: every lane advances the scalars; lane r rotates / moves rows r, r + T, .. of U, lanes 0..8 row r of V
  svd_qr_phase(n, w, rv1, anorm,
    [&](int ca, int cb, double c, double s_) {
      for (int r = tid; r < m; r += MONO_T) { const double y = AA(r, ca), z = AA(r, cb); AA(r, ca) = y * c + z * s_; AA(r, cb) = z * c - y * s_; }
    },
    [&](int ca, int cb, double c, double s_) {
      if (tid < n) { const double x = VV(tid, ca), z = VV(tid, cb); VV(tid, ca) = x * c + z * s_; VV(tid, cb) = z * c - x * s_; }
    },
    [&](int col) { if (tid < n) VV(tid, col) = -VV(tid, col); });
  {
    // (the sort's temporary column: col_tmp[m] in global memory for U -- a lane only touches its own rows -- a register for V)
    double sv = 0.0;
    svd_sort_phase(n, w,
      [&](int col) { for (int r = tid; r < m; r += MONO_T) col_tmp[r] = AA(r, col); if (tid < n) sv = VV(tid, col); },
      [&](int dst, int src) { for (int r = tid; r < m; r += MONO_T) AA(r, dst) = AA(r, src); if (tid < n) VV(tid, dst) = VV(tid, src); },
      [&](int col) { for (int r = tid; r < m; r += MONO_T) AA(r, col) = col_tmp[r]; if (tid < n) VV(tid, col) = sv; });
  }
  // flip signs: count the negative elements of each (U column, V column)
  if (tid < n) sNeg[tid] = 0;
  __syncthreads();
  for (k = 0; k < n; k++) {
    int cnt = 0;
    for (int r = tid; r < m; r += MONO_T) cnt += AA(r, k) < 0.0 ? 1 : 0;
    if (tid < n) cnt += VV(tid, k) < 0.0 ? 1 : 0;
    if (cnt) atomicAdd(&sNeg[k], cnt);
  }
  __syncthreads();
  for (k = 0; k < n; k++)
    if (sNeg[k] > (m + n) / 2) {
      for (int r = tid; r < m; r += MONO_T) AA(r, k) = -AA(r, k);
      if (tid < n) VV(tid, k) = -VV(tid, k);
    }
  __syncthreads();
#undef AA
#undef VV
#undef TT
}

__global__ void __launch_bounds__(MONO_T)
mono_final_a_kernel(vh_mono_params e, const vh_p_match *__restrict__ pm_base, int64_t pm_stride, const int32_t *__restrict__ offsets,
                  const int32_t *__restrict__ counts, int32_t count_cap, const int32_t *__restrict__ rand8, uint8_t *__restrict__ scratch,
                  int64_t cap, double *__restrict__ tr_out, int32_t *__restrict__ ok_out, int32_t *__restrict__ ninl_out,
                  int32_t *__restrict__ inl_out, int64_t inl_stride) {
  __shared__ double sF[9], sV[81], sX[16];
  // The refit system lives in LDS when it fits (MONO_LDS_ROWS x 9 doubles): the cooperative SVD's sequential sums
  // are chains of dependent accumulations over a column, one load per step -- from global memory each step waited
  // out an L2 round trip (7.8 of the 15 ms a 256 x 400 x 2000 batch took), from LDS a few dozen cycles.
  __shared__ double sA[MONO_LDS_ROWS * 9];
  __shared__ double sT[MONO_LDS_ROWS * 8];        // the cooperative SVD's term scratch
  static_assert(sizeof(double) * MONO_LDS_ROWS * 17 <= 150 * 1024, "mono_final_a: refit system + term scratch must fit gfx950's 160 KB of LDS (lower MONO_LDS_ROWS for a smaller part)");
  __shared__ int32_t sNeg[9], sWave[MONO_T / 64], sBase;
  const int32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const MonoList L = mono_list(s, pm_base, pm_stride, offsets, counts, count_cap, scratch, cap);
  const int32_t N = L.n;
  int32_t *inl = inl_out ? inl_out + (offsets ? (int64_t)offsets[s] : (int64_t)s * inl_stride) : nullptr;
  auto fail = [&](int32_t ninl) {
    if (tid == 0) { ok_out[s] = 0; ninl_out[s] = ninl; for (int32_t q = 0; q < 6; q++) tr_out[6 * s + q] = 0.0; }
  };
  if (tid == 0) L.hdr[MONO_H_GO] = 0.0;
  if (L.hdr[18] == 0.0) { fail(0); return; }
  const unsigned long long key = *L.key;
  const int32_t nbest = (int32_t)(key >> 32);
  if (nbest == 0) { fail(0); return; }
  // the winner's F as mono_hyp left it (+-F: the test below is the same for both)
  VH_MTICK_INIT;
#ifdef VH_MONO_TIMING
  if (blockIdx.x == 0 && tid == 0) atomicAdd(&g_mono_t[15], (unsigned long long)*mono_queue(scratch, gridDim.x, cap));  // queued hypotheses of the call
#endif
  if (tid < 9) sF[tid] = mono_hyp_F(scratch, gridDim.x, cap, e.ransac_iters)[((int64_t)s * e.ransac_iters + (0x7FFFFFFF - (int32_t)(uint32_t)key)) * 9 + tid];
  if (tid == 0) sBase = 0;
  __syncthreads();
  VH_MTICK(0);  // the winner's F
  // its ordered inlier list (VisualOdometry::inliers)
  {
    double F[9];
    for (int32_t q = 0; q < 9; q++) F[q] = sF[q];
    for (int32_t i0 = 0; i0 < N; i0 += MONO_T) {
      const int32_t i = i0 + tid;
      const bool in_ = i < N && sampson_inlier(F, L.pn[min(i, N - 1)], e.inlier_threshold);
      const uint64_t bal = __ballot(in_);
      if (lane == 0) sWave[wv] = __popcll(bal);
      __syncthreads();
      int32_t off = sBase;
      for (int32_t q = 0; q < wv; q++) off += sWave[q];
      if (in_) {
        const int32_t p = off + __popcll(bal & ((1ull << lane) - 1));
        L.idx[p] = i;
        if (inl) inl[p] = i;
      }
      __syncthreads();
      if (tid == 0) sBase += sWave[0] + sWave[1] + sWave[2] + sWave[3];
      __syncthreads();
    }
  }
  if (nbest < 10) { fail(nbest); return; }  // src/viso_mono.cpp:80-81
  VH_MTICK(1);  // inlier list
  // F from all inliers: the nbest x 9 system (src/viso_mono.cpp:84, :242-256)
  double *Asys = nbest <= MONO_LDS_ROWS ? sA : L.A;
  for (int32_t i = tid; i < nbest; i += MONO_T) {
    const float4 q = L.pn[L.idx[i]];
    double *r = Asys + (int64_t)i * 9;
    r[0] = (double)(q.z * q.x); r[1] = (double)(q.z * q.y); r[2] = (double)q.z;
    r[3] = (double)(q.w * q.x); r[4] = (double)(q.w * q.y); r[5] = (double)q.w;
    r[6] = (double)q.x; r[7] = (double)q.y; r[8] = 1.0;
  }
  __syncthreads();
  double w9[9];
  VH_MTICK(2);  // refit system
  svd_tall9(Asys, nbest, sV, sX, sNeg, w9, L.d, nbest <= MONO_LDS_ROWS ? sT : L.X);
  VH_MTICK(3);  // its SVD
  if (tid == 0) {
    double F0[9], F[9], T1[9], T2[9], E[9], Kt[9];
    const double K[9] = {e.f, 0, e.cu, 0, e.f, e.cv, 0, 0, 1};
    for (int32_t q = 0; q < 9; q++) F0[q] = sV[q * 9 + 8];
    rank2_3x3(F0, F);
    // denormalise, essential matrix, rank 2 again (src/viso_mono.cpp:86-94)
    transpose3(L.hdr + 9, T1); matmul(T1, 3, 3, F, 3, T2); matmul(T2, 3, 3, L.hdr, 3, F);
    transpose3(K, Kt); matmul(Kt, 3, 3, F, 3, T2); matmul(T2, 3, 3, K, 3, E);
    rank2_3x3(E, T1);
    for (int32_t q = 0; q < 9; q++) E[q] = T1[q];
    // EtoRt (src/viso_mono.cpp:317-346)
    const double Wm[9] = {0, -1, 0, +1, 0, 0, 0, 0, 1}, Zm[9] = {0, +1, 0, -1, 0, 0, 0, 0, 0};
    double U[9], V[9], Ut[9], Vt[9], Wt[9], T[9];
    {
      double Ur[3][3], Sr[3], Vr[3][3];
#pragma unroll
      for (int32_t i = 0; i < 3; i++)
#pragma unroll
        for (int32_t q = 0; q < 3; q++) Ur[i][q] = E[i * 3 + q];
      svd_static<3, 3>(Ur, Sr, Vr);
#pragma unroll
      for (int32_t i = 0; i < 3; i++)
#pragma unroll
        for (int32_t q = 0; q < 3; q++) { U[i * 3 + q] = Ur[i][q]; V[i * 3 + q] = Vr[i][q]; }
    }
    transpose3(U, Ut); transpose3(V, Vt); transpose3(Wm, Wt);
    matmul(U, 3, 3, Zm, 3, T1); matmul(T1, 3, 3, Ut, 3, T);
    double *Ra = L.hdr + MONO_H_RT, *Rb = Ra + 9, *t0 = Ra + 18, *sK = L.hdr + MONO_H_P1;
    matmul(U, 3, 3, Wm, 3, T1); matmul(T1, 3, 3, Vt, 3, Ra);
    matmul(U, 3, 3, Wt, 3, T1); matmul(T1, 3, 3, Vt, 3, Rb);
    t0[0] = T[2 * 3 + 1]; t0[1] = T[0 * 3 + 2]; t0[2] = T[1 * 3 + 0];
    if (det3(Ra) < 0) for (int32_t q = 0; q < 9; q++) Ra[q] = -Ra[q];
    if (det3(Rb) < 0) for (int32_t q = 0; q < 9; q++) Rb[q] = -Rb[q];
    // projection matrices of triangulateChieral (src/viso_mono.cpp:370-375)
    for (int32_t i = 0; i < 3; i++) for (int32_t j = 0; j < 4; j++) sK[i * 4 + j] = j < 3 ? K[i * 3 + j] : 0.0;
    for (int32_t c = 0; c < 4; c++) {
      const double *R = c < 2 ? Ra : Rb;
      double Rt[12];
      for (int32_t i = 0; i < 3; i++) { for (int32_t j = 0; j < 3; j++) Rt[i * 4 + j] = R[i * 3 + j]; Rt[i * 4 + 3] = (c & 1) ? -t0[i] : t0[i]; }
      matmul(K, 3, 3, Rt, 4, L.hdr + MONO_H_P2 + 12 * c);
    }
    for (int32_t c = 0; c < 4; c++) ((int32_t *)(L.hdr + MONO_H_CNT))[c] = 0;
    L.hdr[MONO_H_NBEST] = (double)nbest;
    L.hdr[MONO_H_GO] = 1.0;
  }
  VH_MTICK(4);  // F, E, R|t candidates (one lane)
}

// ------------------------------------------------------------------- mono_tri
// triangulateChieral's linear triangulation (src/viso_mono.cpp:378-387): one lane per (match, candidate) -- one 4x4
// Matrix::svd each, in registers.  A kernel of its own so that the hundreds to thousands of independent
// factorizations of a list fill the chip instead of queueing on the 256 lanes of the list's mono_final workgroup.
__global__ void __launch_bounds__(256)
mono_tri_kernel(const vh_p_match *__restrict__ pm_base, int64_t pm_stride, const int32_t *__restrict__ offsets,
                const int32_t *__restrict__ counts, int32_t count_cap, uint8_t *__restrict__ scratch, int64_t cap) {
  const int32_t s = blockIdx.y, tid = threadIdx.x, lane = tid & 63, c = tid & 3;
  const MonoList L = mono_list(s, pm_base, pm_stride, offsets, counts, count_cap, scratch, cap);
  if (L.hdr[MONO_H_GO] == 0.0 || blockIdx.x * 64 >= L.n) return;
  const int32_t i = blockIdx.x * 64 + (tid >> 2);
  const double *sK = L.hdr + MONO_H_P1, *P2 = L.hdr + MONO_H_P2 + 12 * c;
  bool good = false;
  if (i < L.n) {
    const vh_p_match &m = L.pm[i];
    double J[4][4], w4[4], V4[4][4], x[4];
#pragma unroll
    for (int32_t j = 0; j < 4; j++) {
      J[0][j] = sK[2 * 4 + j] * m.u1p - sK[0 * 4 + j];
      J[1][j] = sK[2 * 4 + j] * m.v1p - sK[1 * 4 + j];
      J[2][j] = P2[2 * 4 + j] * m.u1c - P2[0 * 4 + j];
      J[3][j] = P2[2 * 4 + j] * m.v1c - P2[1 * 4 + j];
    }
    // the direction of the smallest singular value, up to sign (no U: svd_static.h) -- every use of it, here and in
    // mono_final_c, is a quotient or a product of two components, the same for x and -x
    svd_static_last_v_unsigned<4, 4>(J, w4, V4, x);
    double *Xc = L.X + (int64_t)c * 4 * cap;
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int32_t j = 0; j < 4; j++) Xc[(int64_t)j * cap + i] = x[j];
    for (int32_t j = 0; j < 4; j++) a += sK[2 * 4 + j] * x[j];
    for (int32_t j = 0; j < 4; j++) b += P2[2 * 4 + j] * x[j];
    good = a * x[3] > 0 && b * x[3] > 0;
  }
  const uint64_t bal = __ballot(good);
  if (lane < 4) {  // lane c of each wave adds up the wave's lanes of candidate c
    const int32_t n = __popcll(bal & (0x1111111111111111ull << lane));
    if (n) atomicAdd((int32_t *)(L.hdr + MONO_H_CNT) + lane, n);
  }
}

// --------------------------------------------------------------- mono_final_c
// The rest of estimateMotion (src/viso_mono.cpp:96-160) on the winning (R, t): points in front, median distance,
// ground-plane vote, scale, the six motion parameters.
__global__ void __launch_bounds__(MONO_T)
mono_final_c_kernel(vh_mono_params e, const vh_p_match *__restrict__ pm_base, int64_t pm_stride, const int32_t *__restrict__ offsets,
                    const int32_t *__restrict__ counts, int32_t count_cap, uint8_t *__restrict__ scratch, int64_t cap,
                    double *__restrict__ tr_out, int32_t *__restrict__ ok_out, int32_t *__restrict__ ninl_out) {
  __shared__ double sX[16], sM[21];
  __shared__ double sT[MONO_LDS_ROWS * 8];
  __shared__ int32_t sWave[MONO_T / 64], sBase, sFlag;
  const int32_t s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const MonoList L = mono_list(s, pm_base, pm_stride, offsets, counts, count_cap, scratch, cap);
  if (L.hdr[MONO_H_GO] == 0.0) return;  // (mono_final_a reported the failure)
  const int32_t N = L.n, nbest = (int32_t)L.hdr[MONO_H_NBEST];
  const int32_t *sCnt = (const int32_t *)(L.hdr + MONO_H_CNT);
  auto fail = [&](int32_t ninl) {
    if (tid == 0) { ok_out[s] = 0; ninl_out[s] = ninl; for (int32_t q = 0; q < 6; q++) tr_out[6 * s + q] = 0.0; }
  };
  if (tid < 21) sM[tid] = L.hdr[MONO_H_RT + tid];
  VH_MTICK_INIT;
  __syncthreads();
  int32_t cbest = -1, max_in = 0;
  for (int32_t c = 0; c < 4; c++) if (sCnt[c] > max_in) { max_in = sCnt[c]; cbest = c; }  // strict `>`, src/viso_mono.cpp:353
  if (cbest < 0) { fail(nbest); return; }  // (the reference dereferences an empty matrix here)
  // X / X(3,:), the points in front in match order, their L1 distance and ground-plane coordinate (src/viso_mono.cpp:100-128)
  const double *Xb = L.X + (int64_t)cbest * 4 * cap;
  const double n0 = cos(-e.pitch), n1 = sin(-e.pitch);
  if (tid == 0) sBase = 0;
  __syncthreads();
  for (int32_t i0 = 0; i0 < N; i0 += MONO_T) {
    const int32_t i = i0 + tid;
    double x = 0, y = 0, z = 0;
    if (i < N) {
      const double wq = Xb[(int64_t)3 * cap + i];
      if (wq != 0) { x = Xb[i] / wq; y = Xb[(int64_t)cap + i] / wq; z = Xb[(int64_t)2 * cap + i] / wq; }
    }
    const bool front = i < N && z > 0;
    const uint64_t bal = __ballot(front);
    if (lane == 0) sWave[wv] = __popcll(bal);
    __syncthreads();
    int32_t off = sBase;
    for (int32_t q = 0; q < wv; q++) off += sWave[q];
    if (front) {
      const int32_t p = off + __popcll(bal & ((1ull << lane) - 1));
      L.dist[p] = fabs(x) + fabs(y) + fabs(z);
      double dd = 0.0;
      dd += n0 * y; dd += n1 * z;
      L.d[p] = dd;
    }
    __syncthreads();
    if (tid == 0) sBase += sWave[0] + sWave[1] + sWave[2] + sWave[3];
    __syncthreads();
  }
  const int32_t np = sBase;
  if (np < 10) { fail(nbest); return; }
  VH_MTICK(6);  // points in front, compaction
  // median = element number np/2 of the sorted distances (smallerThanMedian): its rank by counting.  Both this and
  // the vote below read the whole list once per element: from LDS (the SVD's term scratch is free) when it fits.
  const bool staged = np <= MONO_LDS_ROWS * 8;
  const double *distv = staged ? sT : L.dist;
  if (staged) {
    for (int32_t i = tid; i < np; i += MONO_T) sT[i] = L.dist[i];
    __syncthreads();
  }
  const int32_t half = np / 2;
  for (int32_t i = tid; i < np; i += MONO_T) {
    const double v = distv[i];
    int32_t rank = 0, j = 0;
    for (; j + 8 <= np; j += 8) {  // (eight loads in flight per trip; the count does not care about the order)
      double o[8];
#pragma unroll
      for (int32_t q = 0; q < 8; q++) o[q] = distv[j + q];
#pragma unroll
      for (int32_t q = 0; q < 8; q++) rank += (o[q] < v || (o[q] == v && j + q < i)) ? 1 : 0;
    }
    for (; j < np; j++) { const double o = distv[j]; rank += (o < v || (o == v && j < i)) ? 1 : 0; }
    if (rank == half) sX[0] = v;
  }
  __syncthreads();
  const double median = sX[0];
  if (median > e.motion_threshold) { fail(nbest); return; }
  VH_MTICK(7);  // median
  const double sigma = median / 50.0, weight = 1.0 / (2.0 * sigma * sigma);
  // ground-plane vote (src/viso_mono.cpp:130-148): each lane sums the kernel over all j in order for its
  // candidates i (the sums replace the distances, which are dead); the first i with the largest sum wins
  const double *dv = staged ? sT : L.d;
  if (staged) {
    for (int32_t i = tid; i < np; i += MONO_T) sT[i] = L.d[i];
    __syncthreads();
  }
  for (int32_t i = tid; i < np; i += MONO_T) {
    double sum = 0;
    const double di = dv[i];
    if (di > median / e.motion_threshold) {
      int32_t j = 0;
      for (; j + 4 <= np; j += 4) {  // four independent exponentials per trip, added in the original's order
        double t[4];
#pragma unroll
        for (int32_t q = 0; q < 4; q++) { const double dq = dv[j + q] - di; t[q] = exp(-dq * dq * weight); }
#pragma unroll
        for (int32_t q = 0; q < 4; q++) sum += t[q];
      }
      for (; j < np; j++) { const double q = dv[j] - di; sum += exp(-q * q * weight); }
    }
    L.dist[i] = sum;
  }
  __syncthreads();
  VH_MTICK(8);  // ground-plane vote
  if (tid == 0) {
    double best_sum = 0;
    int32_t best_idx = 0;
    for (int32_t i = 0; i < np; i++) if (L.dist[i] > best_sum) { best_sum = L.dist[i]; best_idx = i; }
    const double dref = L.d[best_idx];
    int32_t okf = 1;
    if (fabs(dref) < 1e-20) okf = 0;  // (Matrix::operator/ exits the reference here)
    if (okf) {
      const double *R = cbest < 2 ? sM : sM + 9, *t0 = sM + 18;
      double t[3];
      for (int32_t q = 0; q < 3; q++) { const double tq = (cbest & 1) ? -t0[q] : t0[q]; t[q] = (tq * e.height) / dref; }
      const double ry = asin(R[0 * 3 + 2]);
      tr_out[6 * s + 0] = asin(-R[1 * 3 + 2] / cos(ry));
      tr_out[6 * s + 1] = ry;
      tr_out[6 * s + 2] = asin(-R[0 * 3 + 1] / cos(ry));
      tr_out[6 * s + 3] = t[0]; tr_out[6 * s + 4] = t[1]; tr_out[6 * s + 5] = t[2];
      ok_out[s] = 1; ninl_out[s] = nbest;
    }
    sFlag = okf;
  }
  __syncthreads();
  if (!sFlag) fail(nbest);
}

}  // namespace

extern "C" int32_t vh_debug_mono_timing(unsigned long long *out, int32_t reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mono_t), sizeof(g_mono_t)) != hipSuccess) return -3;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_mono_t), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}

int64_t vh_mono_scratch_bytes(int32_t n_sets, int64_t cap, int32_t ransac_iters) {
  return (int64_t)n_sets * mono_per_list(cap) + mono_queue_bytes(n_sets, ransac_iters) + (int64_t)n_sets * ransac_iters * 72;  // lists | queue | F of every hypothesis
}

void vh_launch_mono(const vh_mono_params &e, int32_t n_sets, const vh_p_match *pm, int64_t pm_stride, const int32_t *offsets,
                    const int32_t *counts, int32_t count_cap, const int32_t *rand8, uint8_t *scratch, int64_t cap, double *tr,
                    int32_t *ok, int32_t *ninl, int32_t *inl, int64_t inl_stride, hipStream_t st) {
  hipLaunchKernelGGL(mono_norm_kernel, dim3(n_sets), dim3(MONO_T), 0, st, pm, pm_stride, offsets, counts, count_cap, scratch, cap);
  static const int32_t force_signed = [] { const char *v = getenv("VH_MONO_SIGNED"); return v && atoi(v) ? 1 : 0; }();
  const dim3 hgrid((e.ransac_iters + 127) / 128, n_sets);
  hipLaunchKernelGGL(mono_hyp_kernel<false>, hgrid, dim3(128), 0, st, e, pm, pm_stride, offsets, counts, count_cap, rand8, scratch, cap, n_sets, force_signed);
  // (sized for a full queue; the workgroups beyond its length return at once)
  hipLaunchKernelGGL(mono_hyp_kernel<true>, dim3((uint32_t)(((int64_t)n_sets * e.ransac_iters + 127) / 128)), dim3(128), 0, st, e, pm, pm_stride, offsets,
                     counts, count_cap, rand8, scratch, cap, n_sets, force_signed);
  hipLaunchKernelGGL(mono_final_a_kernel, dim3(n_sets), dim3(MONO_T), 0, st, e, pm, pm_stride, offsets, counts, count_cap, rand8, scratch,
                     cap, tr, ok, ninl, inl, inl_stride);
  // (sized for lists as long as the capacity; the workgroups beyond a list's length return at once)
  hipLaunchKernelGGL(mono_tri_kernel, dim3((uint32_t)((cap + 63) / 64), n_sets), dim3(256), 0, st, pm, pm_stride, offsets, counts, count_cap, scratch, cap);
  hipLaunchKernelGGL(mono_final_c_kernel, dim3(n_sets), dim3(MONO_T), 0, st, e, pm, pm_stride, offsets, counts, count_cap, scratch, cap, tr, ok, ninl);
}
