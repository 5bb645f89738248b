// kernels_prior.hip -- quad matching with the motion prior of stock libviso2 on gfx950.
//
// The reference tree accepts Tr_delta in Matcher::matchFeatures and ignores it (src/matcher.cpp:93-111); what it does
// pin is the cost term findMatch adds when a predicted position (u_, v_) is given (src/matcher.cpp:257-262):
// cost = SAD + 4 * ||(u2, v2) - (u_, v_)|| in double, first strict minimum in visiting order.  Stock libviso2 uses it on
// ONE hop of the quad circle, previous right -> current right [upstream-recollection]: the 3-d point of the (1p, 2p)
// pair is moved by Tr_delta and projected into the current right image.  The prediction belongs to the DRIVING feature
// i1p, not to the query i2p, so this hop cannot be a table over the features of set 2p like the other three: it is
// evaluated per driving feature, after the 1p -> 2p table exists, one lane per circle, walking the query's bins in the
// reference's order (u-bin, v-bin, list position = ascending bin-order position) with the literal accept test and a
// double-precision compare.  Built with -ffp-contract=off: du * du + dv * dv rounds twice, as on the reference's x86 build.
#include "vh_dev.h"
#include <math.h>

namespace {

// tr: [S][16] row-major 4x4 per stream
__global__ void __launch_bounds__(128) quad_prior_kernel(VhSets s, VhMatchArgs a, const double *__restrict__ tr, double f, double cu, double cv,
                                                         double base, int32_t *__restrict__ best) {
  const int32_t stream = blockIdx.y;
  const int32_t set1p = vh_role_set(a.S, a.pair_cur, stream, 0), set2p = vh_role_set(a.S, a.pair_cur, stream, 1);
  const int32_t set2c = vh_role_set(a.S, a.pair_cur, stream, 3);
  const int64_t cap = s.cap;
  const int32_t n1p = s.bin_start[(int64_t)set1p * (s.nbins + 1) + s.nbins], n2p = s.bin_start[(int64_t)set2p * (s.nbins + 1) + s.nbins];
  const int32_t n2c = s.bin_start[(int64_t)set2c * (s.nbins + 1) + s.nbins];
  int32_t *__restrict__ T = best + (int64_t)stream * 4 * cap;
  const double *__restrict__ t = tr + 16 * (int64_t)stream;
  const int32_t *__restrict__ cbs = s.bin_start + (int64_t)set2c * (s.nbins + 1);
  const uint32_t *__restrict__ cuv = s.s_uv + (int64_t)set2c * cap;
  const uint4 *__restrict__ cdesc = (const uint4 *)(s.s_desc + (int64_t)set2c * cap * 8);
  const int32_t *__restrict__ cidx = s.s_idx + (int64_t)set2c * cap;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n1p; i += gridDim.x * blockDim.x) {
    if (n2p <= 0 || n2c <= 0) { T[1 * cap + i] = 0; continue; }
    const int32_t i2p = T[0 * cap + i];
    const uint32_t uv1 = s.f_uv[(int64_t)set1p * cap + i];
    const int32_t *q = s.feat + ((int64_t)set2p * cap + i2p) * 12;  // the query of this hop: previous right feature i2p
    const int32_t u2p = q[0], v2p = q[1], c = q[3];
    const uint4 a0 = *(const uint4 *)(q + 4), a1 = *(const uint4 *)(q + 8);
    const int32_t u1p = (int32_t)(uv1 & 0xFFFFu), v1p = (int32_t)(uv1 >> 16);
    // the prediction [upstream-recollection]
    double d = (double)u1p - (double)u2p;
    if (d < 1.0) d = 1.0;
    const double x1p = ((double)u1p - cu) * base / d, y1p = ((double)v1p - cv) * base / d, z1p = f * base / d;
    const double x2c = t[0] * x1p + t[1] * y1p + t[2] * z1p + t[3] - base;
    const double y2c = t[4] * x1p + t[5] * y1p + t[6] * z1p + t[7];
    const double z2c = t[8] * x1p + t[9] * y1p + t[10] * z1p + t[11];
    const double u_ = f * x2c / z2c + cu, v_ = f * y2c / z2c + cv;
    // findMatch (src/matcher.cpp:216-272), flow search window
    const int32_t u_lo = u2p - a.radius, u_hi = u2p + a.radius, v_lo = v2p - a.radius, v_hi = v2p + a.radius;
    const int32_t ub0 = min(max(u_lo, 0) / s.binsize, s.ubn - 1), ub1 = min(max(u_hi, 0) / s.binsize, s.ubn - 1);
    const int32_t vb0 = min(max(v_lo, 0) / s.binsize, s.vbn - 1), vb1 = min(max(v_hi, 0) / s.binsize, s.vbn - 1);
    double min_cost = 10000000;  // matcher.cpp:222
    int32_t min_pos = -1;
    for (int32_t ub = ub0; ub <= ub1; ub++) {
      const int32_t row = (c * s.ubn + ub) * s.vbn;
      for (int32_t p = cbs[row + vb0]; p < cbs[row + vb1 + 1]; p++) {
        const uint32_t uv2 = cuv[p];
        const int32_t u2 = uv2 & 0xFFFF, v2 = uv2 >> 16;
        if (u2 < u_lo || u2 > u_hi || v2 < v_lo || v2 > v_hi) continue;
        const uint4 b0 = cdesc[2 * (int64_t)p], b1 = cdesc[2 * (int64_t)p + 1];
        uint32_t sad = __builtin_amdgcn_sad_u8(a0.x, b0.x, 0);
        sad = __builtin_amdgcn_sad_u8(a0.y, b0.y, sad); sad = __builtin_amdgcn_sad_u8(a0.z, b0.z, sad); sad = __builtin_amdgcn_sad_u8(a0.w, b0.w, sad);
        sad = __builtin_amdgcn_sad_u8(a1.x, b1.x, sad); sad = __builtin_amdgcn_sad_u8(a1.y, b1.y, sad);
        sad = __builtin_amdgcn_sad_u8(a1.z, b1.z, sad); sad = __builtin_amdgcn_sad_u8(a1.w, b1.w, sad);
        double cost = (double)sad;
        if (u_ >= 0 && v_ >= 0) {
          const double du = (double)u2 - u_, dv = (double)v2 - v_;
          cost += 4 * sqrt(du * du + dv * dv);
        }
        if (cost < min_cost) { min_cost = cost; min_pos = p; }
      }
    }
    T[1 * cap + i] = min_pos >= 0 ? cidx[min_pos] : 0;  // (table slot 1 is indexed by the DRIVING feature here: chain_kernel, a.prior)
  }
}

}  // namespace

void vh_launch_quad_prior(const VhSets &s, const VhMatchArgs &a, const double *tr, double f, double cu, double cv, double base, int32_t *best,
                          hipStream_t st) {
  dim3 grid((uint32_t)((s.cap + 127) / 128 < 256 ? (s.cap + 127) / 128 : 256), a.S);
  hipLaunchKernelGGL(quad_prior_kernel, grid, dim3(128), 0, st, s, a, tr, f, cu, cv, base, best);
}
