// kernels_match.hip -- binned 32-byte SAD matching on gfx950.
//
// Replaces Matcher::findMatch (reference src/matcher.cpp:216-272) and the
// circle-match compositions of Matcher::matching (src/matcher.cpp:274-344).
//
// Restructuring relative to the reference (results identical):
//  * The reference calls findMatch on demand along each match chain.  findMatch
//    is a pure function of (query feature, candidate set), so here every pass
//    of a chain is evaluated for ALL features of its query set at once
//    (`match`), and the chains are then followed by table look-ups (`chain`).
//  * Mapping: one LANE per QUERY, one wavefront per tile of <= 64 consecutive
//    bin-ordered queries (of one class; of one (class, u-bin) column when the v
//    window does not span the image).  The candidate stream is then wave-uniform:
//    all 64 lanes walk the same bin range, candidates are staged 64 at a time in
//    a wave-private LDS chunk and broadcast-read, and the SAD is 8 v_sad_u8 /
//    v_sad_hi_u8 per lane and query with no cross-lane reduction at all.
//    Positions in bin order ARE the reference's visiting order, so its
//    first-minimum tie-break (strict `<`, src/matcher.cpp:264) is the minimum of
//    the key (SAD << 19 | position) -- or (SAD << 16 | position - class base),
//    which the v_sad_hi_u8 chain produces by itself -- whatever order
//    candidates arrive in.
//  * Each lane applies the reference's accept test on its own window
//    (src/matcher.cpp:249); the wave only walks the union of its lanes' bin
//    ranges (src/matcher.cpp:237-240), which never changes a lane's result
//    because a candidate inside a lane's window always lies inside that lane's
//    bin range.
#include "vh_dev.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#ifndef VH_FLOW_LDS_PAD_DEFAULT
#define VH_FLOW_LDS_PAD_DEFAULT 18000
#endif

namespace {

__device__ __forceinline__ int32_t wave_min(int32_t v) {
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ int32_t wave_max(int32_t v) {
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d));
  return v;
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(uint32_t x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ uint32_t as_u32(us2 x) { return __builtin_bit_cast(uint32_t, x); }

__device__ __forceinline__ uint32_t sad4(uint32_t a, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_sad_u8(a, b, acc);  // v_sad_u8: 4 byte-wise |a-b| summed into acc
}
__device__ __forceinline__ uint32_t sad4hi(uint32_t a, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_sad_hi_u8(a, b, acc);  // v_sad_hi_u8: the same sum added at bit 16 of acc
}

// One tile of the flow search: Q queries per lane (Q*64 consecutive bin-ordered
// queries per wave); every candidate record is read from LDS once per wave and
// used for Q queries.  At Q = 1 the broadcast reads (2.25 ds_read_b128 per
// candidate and wave, 4 LDS cycles each, four SIMDs sharing one LDS array) keep
// the LDS ~85 % busy beside the 8 v_sad_u8 per candidate; Q = 2 halves that --
// and was measured 15 % SLOWER on MI355X (flow search alone, S = 128: 1180 vs
// 1028 us): the kernel is bound by VALU issue, not by the LDS, and a 128-query
// tile spans more bin columns, so that its lanes evaluate ~10 % more candidates
// outside their own windows and fewer columns take the cheap accept tests.
// VH_FLOW_Q (vh_dev.h) therefore stays 1.
template <int Q, bool HI>
__device__ __forceinline__ void flow_tile(const VhSets &s, const VhMatchArgs &a, int32_t pass, int32_t stream,
                                          int32_t qset, int32_t cset, int32_t q0, int32_t q1, int32_t c,
                                          int32_t pbase, uint4 *wD, uint32_t *wU, int32_t *__restrict__ best) {
  const int32_t lane = threadIdx.x & 63;
  const uint32_t *__restrict__ quv = s.s_uv + (int64_t)qset * s.cap;
  const uint4 *__restrict__ qdesc = (const uint4 *)(s.s_desc + (int64_t)qset * s.cap * 8);
  const int32_t *__restrict__ qidx = s.s_idx + (int64_t)qset * s.cap;
  const uint32_t *__restrict__ cuv = s.s_uv + (int64_t)cset * s.cap;
  const uint4 *__restrict__ cdesc = (const uint4 *)(s.s_desc + (int64_t)cset * s.cap * 8);
  const int32_t *__restrict__ cidx = s.s_idx + (int64_t)cset * s.cap;
  const int32_t *__restrict__ cbs = s.bin_start + (int64_t)cset * (s.nbins + 1);
  const int32_t rv = a.pass[pass].flow ? a.radius : a.disp_tol;

  bool valid[Q];
  uint4 a0[Q], a1[Q];
  int32_t v_lo[Q];
  us2 lo2[Q];
  uint32_t best_key[Q];
  int32_t ub_lo = 0x7FFFFFFF, vb_lo = 0x7FFFFFFF, ub_hi = -1, vb_hi = -1;
  int32_t ulo_max = -0x40000000, uhi_min = 0x40000000, vlo_max = -0x40000000, vhi_min = 0x40000000;
#pragma unroll
  for (int32_t qi = 0; qi < Q; qi++) {
    const int32_t q = q0 + 64 * qi + lane;
    valid[qi] = q < q1;
    const int32_t ql = valid[qi] ? q : q0;
    const uint32_t uv1 = quv[ql];
    a0[qi] = qdesc[2 * (int64_t)ql]; a1[qi] = qdesc[2 * (int64_t)ql + 1];
    const int32_t u1 = uv1 & 0xFFFF, v1 = uv1 >> 16;
    // search window (matcher.cpp:231-234; stereo: v narrowed to +-disp_tolerance)
    const int32_t u_lo = u1 - a.radius, u_hi = u1 + a.radius, v_hi = v1 + rv;
    v_lo[qi] = v1 - rv;
    // Accept test of matcher.cpp:249 in packed 16-bit arithmetic: with
    // t = (u2,v2) - (u_lo,v_lo) (mod 2^16 per half), the candidate is inside the
    // window iff t.u <= 2*radius and t.v <= 2*rv, i.e. iff min(t, span) == t.
    // Exact because coordinates are < 2^14 and radii <= 2^14 (|u2-u1|+r < 2^15).
    lo2[qi] = us2{(unsigned short)u_lo, (unsigned short)v_lo[qi]};
    best_key[qi] = 0xFFFFFFFFu;
    if (valid[qi]) {
      // bins of interest (matcher.cpp:237-240); for x<0 the clamp to 0 makes the
      // truncating division equivalent to the reference's floor
      ub_lo = min(ub_lo, min(max(u_lo, 0) / s.binsize, s.ubn - 1));
      ub_hi = max(ub_hi, min(max(u_hi, 0) / s.binsize, s.ubn - 1));
      vb_lo = min(vb_lo, min(max(v_lo[qi], 0) / s.binsize, s.vbn - 1));
      vb_hi = max(vb_hi, min(max(v_hi, 0) / s.binsize, s.vbn - 1));
      ulo_max = max(ulo_max, u_lo); uhi_min = min(uhi_min, u_hi);
      vlo_max = max(vlo_max, v_lo[qi]); vhi_min = min(vhi_min, v_hi);
    }
  }
  const int32_t UB0 = __builtin_amdgcn_readfirstlane(wave_min(ub_lo));
  const int32_t UB1 = __builtin_amdgcn_readfirstlane(wave_max(ub_hi));
  const int32_t VB0 = __builtin_amdgcn_readfirstlane(wave_min(vb_lo));
  const int32_t VB1 = __builtin_amdgcn_readfirstlane(wave_max(vb_hi));
  const us2 span2 = {(unsigned short)(2 * a.radius), (unsigned short)(2 * rv)};
  // columns whose pixel range [ub*bs, ub*bs+bs-1] is inside EVERY query's u window
  const int32_t ULO_MAX = __builtin_amdgcn_readfirstlane(wave_max(ulo_max));
  const int32_t UHI_MIN = __builtin_amdgcn_readfirstlane(wave_min(uhi_min));
  // v-bins [VA0, VA1] whose pixel rows lie inside EVERY query's v window: in an
  // interior column their candidates need no accept test at all
  const int32_t VLO_MAX = __builtin_amdgcn_readfirstlane(wave_max(vlo_max));
  const int32_t VHI_MIN = __builtin_amdgcn_readfirstlane(wave_min(vhi_min));
  const int32_t VA0 = max(VB0, (max(VLO_MAX, 0) + s.binsize - 1) / s.binsize);
  const int32_t VA1 = min(VB1, (VHI_MIN + 1) / s.binsize - 1);

  // best = min over accepted candidates of (SAD << 19 | position): positions in
  // bin order are the reference's visiting order, so this key reproduces its
  // strict-< first-minimum rule (matcher.cpp:264).  SAD <= 8160 < 2^13.
  // HI: when the positions this tile can visit span less than 2^16, the key is
  // (SAD << 16 | position - pbase) instead and costs nothing to build: the SAD
  // chain runs on v_sad_hi_u8, which accumulates at bit 16, seeded with the
  // relative position in the low half.
  // TEST: 2 = full (u,v) window test, 1 = v only, 0 = none (see the column loop)
  auto make_key = [&](auto test, int32_t qi, uint32_t uv2, const uint4 &b0, const uint4 &b1, int32_t p) -> uint32_t {
    constexpr int TEST = decltype(test)::value;
    bool out = false;
    if (TEST == 2) {
      const us2 t = as_us2(uv2) - lo2[qi];
      const us2 m = __builtin_elementwise_min(t, span2);
      out = as_u32(t) != as_u32(m);
    } else if (TEST == 1) {
      out = (uint32_t)((int32_t)(uv2 >> 16) - v_lo[qi]) > (uint32_t)(2 * rv);
    }
    uint32_t key;
    if (HI) {
      key = sad4hi(a0[qi].x, b0.x, (uint32_t)(p - pbase));
      key = sad4hi(a0[qi].y, b0.y, key);
      key = sad4hi(a0[qi].z, b0.z, key);
      key = sad4hi(a0[qi].w, b0.w, key);
      key = sad4hi(a1[qi].x, b1.x, key);
      key = sad4hi(a1[qi].y, b1.y, key);
      key = sad4hi(a1[qi].z, b1.z, key);
      key = sad4hi(a1[qi].w, b1.w, key);
    } else {
      uint32_t sad = sad4(a0[qi].x, b0.x, 0);
      sad = sad4(a0[qi].y, b0.y, sad);
      sad = sad4(a0[qi].z, b0.z, sad);
      sad = sad4(a0[qi].w, b0.w, sad);
      sad = sad4(a1[qi].x, b1.x, sad);
      sad = sad4(a1[qi].y, b1.y, sad);
      sad = sad4(a1[qi].z, b1.z, sad);
      sad = sad4(a1[qi].w, b1.w, sad);
      key = (sad << 19) | (uint32_t)p;
    }
    return (TEST != 0 && out) ? 0xFFFFFFFFu : key;
  };
  // Candidate stream through a wave-private LDS chunk: lane j of the wave fetches
  // candidate p+j (coalesced 36 B per lane), the chunk is then consumed with
  // broadcast reads so that v_sad_u8 runs on VGPR operands (its SGPR-operand form
  // issues ~13 % slower, tools/ubench_valu.hip) and no scalar-load round trips
  // sit in the loop.  The next chunk's global loads are in flight while the
  // current one is consumed.
  for (int32_t ub = UB0; ub <= UB1; ub++) {
    const int32_t row = (c * s.ubn + ub) * s.vbn;
    const int32_t p0 = __builtin_amdgcn_readfirstlane(cbs[row + VB0]);
    const int32_t p1 = __builtin_amdgcn_readfirstlane(cbs[row + VB1 + 1]);
    const bool interior = ub * s.binsize >= ULO_MAX && ub * s.binsize + s.binsize - 1 <= UHI_MIN;
    // positions of the untested v-bins of this column (empty when VA0 > VA1)
    const int32_t pa0 = (interior && VA0 <= VA1) ? __builtin_amdgcn_readfirstlane(cbs[row + VA0]) : p1;
    const int32_t pa1 = (interior && VA0 <= VA1) ? __builtin_amdgcn_readfirstlane(cbs[row + VA1 + 1]) : p1;
    for (int32_t pc = p0; pc < p1; pc += 64) {
      const int32_t mcnt = min(64, p1 - pc);
      const int32_t pl = min(pc + lane, p1 - 1);
      const uint32_t gu = cuv[pl];
      const uint4 g0 = cdesc[2 * (int64_t)pl], g1 = cdesc[2 * (int64_t)pl + 1];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous chunk fully consumed
      wU[lane] = gu; wD[2 * lane] = g0; wD[2 * lane + 1] = g1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // chunk visible to every lane of the wave
      int32_t j = 0;
      // CPI candidates per trip, two per update: min(best, kA, kB) is one v_min3_u32
      constexpr int CPI = Q == 1 ? 4 : 2;
      auto run = [&](auto test, int32_t jend) {
        for (; j + CPI <= jend; j += CPI) {
#pragma unroll
          for (int32_t k = 0; k < CPI; k += 2) {
            const uint32_t uA = wU[j + k], uB = wU[j + k + 1];
            const uint4 dA0 = wD[2 * (j + k)], dA1 = wD[2 * (j + k) + 1], dB0 = wD[2 * (j + k) + 2], dB1 = wD[2 * (j + k) + 3];
#pragma unroll
            for (int32_t qi = 0; qi < Q; qi++) {
              const uint32_t kA = make_key(test, qi, uA, dA0, dA1, pc + j + k), kB = make_key(test, qi, uB, dB0, dB1, pc + j + k + 1);
              best_key[qi] = min(min(kA, kB), best_key[qi]);
            }
          }
        }
        for (; j < jend; j++) {
          const uint32_t uA = wU[j];
          const uint4 dA0 = wD[2 * j], dA1 = wD[2 * j + 1];
#pragma unroll
          for (int32_t qi = 0; qi < Q; qi++) best_key[qi] = min(best_key[qi], make_key(test, qi, uA, dA0, dA1, pc + j));
        }
      };
      if (interior) {
        // [0, ja): v test, [ja, jb): no test, [jb, mcnt): v test
        // (the untested range is shrunk inward to multiples of the unroll factor:
        // testing a few candidates that would not need it is always valid, and only
        // the chunk's tail is then left to the one-candidate loop)
        const int32_t ja = min((min(max(pa0 - pc, 0), mcnt) + CPI - 1) & ~(CPI - 1), mcnt);
        const int32_t jb = max(min(max(pa1 - pc, 0), mcnt) & ~(CPI - 1), ja);
        run(std::integral_constant<int, 1>{}, ja);
        run(std::integral_constant<int, 0>{}, jb);
        run(std::integral_constant<int, 1>{}, mcnt);
      } else {
        run(std::integral_constant<int, 2>{}, mcnt);
      }
    }
  }
#pragma unroll
  for (int32_t qi = 0; qi < Q; qi++) {
    if (valid[qi]) {
      // min_ind defaults to 0 when no candidate was accepted (matcher.cpp:221)
      const int32_t bp = HI ? pbase + (int32_t)(best_key[qi] & 0xFFFFu) : (int32_t)(best_key[qi] & 0x7FFFFu);
      const int32_t r = (best_key[qi] == 0xFFFFFFFFu) ? 0 : cidx[bp];
      best[((int64_t)stream * 4 + a.pass[pass].slot) * s.cap + qidx[q0 + 64 * qi + lane]] = r;
    }
  }
}

__global__ void __launch_bounds__(256)
match_kernel(VhSets s, VhMatchArgs a, int32_t *__restrict__ best) {
  __shared__ uint4 sDesc[4 * 128];
  __shared__ uint32_t sUv[4 * 64];
  const int32_t pass = blockIdx.y, stream = blockIdx.z;
  const int32_t qset = vh_role_set(a.S, a.pair_cur, stream, a.pass[pass].qset);
  const int32_t cset = vh_role_set(a.S, a.pair_cur, stream, a.pass[pass].cset);
  const int32_t ntile = s.tile_cnt[qset];
  uint4 *wD = sDesc + (threadIdx.x >> 6) * 128;   // [64][2] uint4
  uint32_t *wU = sUv + (threadIdx.x >> 6) * 64;   // [64]
  for (int32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); tile < ntile;
       tile += gridDim.x * 4) {
    const int4 t = s.tiles[(int64_t)qset * s.max_tiles + tile];
    const int32_t q0 = __builtin_amdgcn_readfirstlane(t.x), q1 = __builtin_amdgcn_readfirstlane(t.y);
    const int32_t c = __builtin_amdgcn_readfirstlane(t.z);
    // candidates of class c occupy the contiguous positions [pbase, pend) of the bin order
    const int32_t *cbs = s.bin_start + (int64_t)cset * (s.nbins + 1);
    const int32_t pbase = __builtin_amdgcn_readfirstlane(cbs[c * s.ubn * s.vbn]);
    const int32_t pend = __builtin_amdgcn_readfirstlane(cbs[(c + 1) * s.ubn * s.vbn]);
    // tiles hold up to VH_TILE_Q = 64 * VH_FLOW_Q queries
    if (pend - pbase <= 0xFFFF && !a.wide_keys) {
      if (VH_FLOW_Q > 1 && q1 - q0 > 64) flow_tile<VH_FLOW_Q, true>(s, a, pass, stream, qset, cset, q0, q1, c, pbase, wD, wU, best);
      else flow_tile<1, true>(s, a, pass, stream, qset, cset, q0, q1, c, pbase, wD, wU, best);
    } else {
      if (VH_FLOW_Q > 1 && q1 - q0 > 64) flow_tile<VH_FLOW_Q, false>(s, a, pass, stream, qset, cset, q0, q1, c, 0, wD, wU, best);
      else flow_tile<1, false>(s, a, pass, stream, qset, cset, q0, q1, c, 0, wD, wU, best);
    }
  }
}

// -------------------------------------------------------------- match (stereo)
// The 1-d stereo search (v window of +-match_disp_tolerance, stock libviso2;
// SURVEY App. A.7) accepts only a few dozen candidates per query, all within a
// handful of image rows, so walking whole 50x50 bins would waste >95 % of the
// work.  Same lane-per-query / wave-uniform-stream scheme as the flow search,
// but over the (class, v) ROW index on both sides: a tile is 64 consecutive
// row-ordered queries of one class (they span only a few rows), and the
// candidate stream is the single contiguous row range [vmin-tol, vmax+tol] of
// the candidate set.  The key still carries the candidate's BIN-order position,
// so the minimum is findMatch's first minimum in (u_bin, v_bin, list) order no
// matter in which order the rows are walked.
__global__ void __launch_bounds__(256)
match_rows_kernel(VhSets s, VhMatchArgs a, int32_t *__restrict__ best) {
  __shared__ uint4 sDesc[4 * 128];
  __shared__ uint2 sMeta[4 * 64];
  const int32_t pass = blockIdx.y, stream = blockIdx.z;
  const int32_t lane = threadIdx.x & 63;
  const int32_t qset = vh_role_set(a.S, a.pair_cur, stream, a.pass[pass].qset);
  const int32_t cset = vh_role_set(a.S, a.pair_cur, stream, a.pass[pass].cset);
  const int32_t nrow = 4 * s.H;
  const int32_t *__restrict__ qrs = s.row_start + (int64_t)qset * (nrow + 1);
  const int32_t *__restrict__ crs = s.row_start + (int64_t)cset * (nrow + 1);
  const int32_t *__restrict__ qpos = s.r_pos + (int64_t)qset * s.cap;
  const uint32_t *__restrict__ quv = s.s_uv + (int64_t)qset * s.cap;
  const uint4 *__restrict__ qdesc = (const uint4 *)(s.s_desc + (int64_t)qset * s.cap * 8);
  const int32_t *__restrict__ cpos = s.r_pos + (int64_t)cset * s.cap;
  const uint32_t *__restrict__ cuv = s.s_uv + (int64_t)cset * s.cap;
  const uint4 *__restrict__ cdesc = (const uint4 *)(s.s_desc + (int64_t)cset * s.cap * 8);
  const int32_t *__restrict__ qidx = s.s_idx + (int64_t)qset * s.cap;
  const int32_t *__restrict__ cidx = s.s_idx + (int64_t)cset * s.cap;
  // tile -> (class, query range): classes are contiguous in row order
  int32_t cls_q0[5];
#pragma unroll
  for (int32_t c = 0; c <= 4; c++) cls_q0[c] = __builtin_amdgcn_readfirstlane(qrs[c * s.H]);
  uint4 *wD = sDesc + (threadIdx.x >> 6) * 128;
  uint2 *wM = sMeta + (threadIdx.x >> 6) * 64;
  for (int32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));; tile += gridDim.x * 4) {
    int32_t c = -1, q0 = 0, q1 = 0, t = tile;
#pragma unroll
    for (int32_t k = 0; k < 4; k++) {
      const int32_t nt = (cls_q0[k + 1] - cls_q0[k] + 63) >> 6;
      if (c < 0 && t < nt) { c = k; q0 = cls_q0[k] + 64 * t; q1 = min(cls_q0[k + 1], q0 + 64); }
      if (c < 0) t -= nt;
    }
    if (c < 0) break;  // past the last tile (uniform)
    const int32_t q = q0 + lane;
    const bool valid = q < q1;
    const int32_t ql = valid ? q : q0;
    // row order holds bin positions; the records are gathered from the bin-ordered arrays
    const int32_t qp = qpos[ql];
    const uint2 qm = make_uint2(quv[qp], (uint32_t)qp);
    const uint4 a0 = qdesc[2 * (int64_t)qp], a1 = qdesc[2 * (int64_t)qp + 1];
    const int32_t u1 = qm.x & 0xFFFF, v1 = qm.x >> 16;
    // window: u1 +- radius, v1 +- disp_tolerance; packed accept test as in the flow search
    const us2 lo2 = {(unsigned short)(u1 - a.radius), (unsigned short)(v1 - a.disp_tol)};
    const us2 span2 = {(unsigned short)(2 * a.radius), (unsigned short)(2 * a.disp_tol)};
    const int32_t VLO = max(__builtin_amdgcn_readfirstlane(wave_min(valid ? v1 : 0x7FFFFFFF)) - a.disp_tol, 0);
    const int32_t VHI = min(__builtin_amdgcn_readfirstlane(wave_max(valid ? v1 : -1)) + a.disp_tol, s.H - 1);
    const int32_t r0 = __builtin_amdgcn_readfirstlane(crs[c * s.H + VLO]);
    const int32_t r1 = __builtin_amdgcn_readfirstlane(crs[c * s.H + VHI + 1]);
    // key = SAD << 16 | (bin-order position - class base) straight out of a
    // v_sad_hi_u8 chain when the class holds < 2^16 candidates (see flow_tile),
    // else SAD << 19 | position
    const int32_t *cbs = s.bin_start + (int64_t)cset * (s.nbins + 1);
    const int32_t pbase = __builtin_amdgcn_readfirstlane(cbs[c * s.ubn * s.vbn]);
    const bool hi = __builtin_amdgcn_readfirstlane(cbs[(c + 1) * s.ubn * s.vbn]) - pbase <= 0xFFFF && !a.wide_keys;
    uint32_t best_key = 0xFFFFFFFFu;
    for (int32_t rc = r0; rc < r1; rc += 64) {
      const int32_t mcnt = min(64, r1 - rc);
      const int32_t rl = min(rc + lane, r1 - 1);
      const int32_t cp = cpos[rl];
      const uint2 gm = make_uint2(cuv[cp], (uint32_t)(hi ? cp - pbase : cp));
      const uint4 g0 = cdesc[2 * (int64_t)cp], g1 = cdesc[2 * (int64_t)cp + 1];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous chunk fully consumed
      wM[lane] = gm; wD[2 * lane] = g0; wD[2 * lane + 1] = g1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // chunk visible to every lane of the wave
      // four candidates per trip (their LDS reads are issued together), then the rest
      auto key_of = [&](auto hi_, const uint2 cm, const uint4 &b0, const uint4 &b1) -> uint32_t {
        const us2 tt = as_us2(cm.x) - lo2;
        const us2 mm = __builtin_elementwise_min(tt, span2);
        const bool out = as_u32(tt) != as_u32(mm);
        uint32_t key;
        if (decltype(hi_)::value) {
          key = sad4hi(a0.x, b0.x, cm.y);
          key = sad4hi(a0.y, b0.y, key);
          key = sad4hi(a0.z, b0.z, key);
          key = sad4hi(a0.w, b0.w, key);
          key = sad4hi(a1.x, b1.x, key);
          key = sad4hi(a1.y, b1.y, key);
          key = sad4hi(a1.z, b1.z, key);
          key = sad4hi(a1.w, b1.w, key);
        } else {
          uint32_t sad = sad4(a0.x, b0.x, 0);
          sad = sad4(a0.y, b0.y, sad);
          sad = sad4(a0.z, b0.z, sad);
          sad = sad4(a0.w, b0.w, sad);
          sad = sad4(a1.x, b1.x, sad);
          sad = sad4(a1.y, b1.y, sad);
          sad = sad4(a1.z, b1.z, sad);
          sad = sad4(a1.w, b1.w, sad);
          key = (sad << 19) | cm.y;
        }
        return out ? 0xFFFFFFFFu : key;
      };
      auto consume = [&](auto hi_) {
        int32_t j = 0;
        for (; j + 4 <= mcnt; j += 4) {
          uint2 cm[4];
          uint4 b0[4], b1[4];
#pragma unroll
          for (int32_t k = 0; k < 4; k++) { cm[k] = wM[j + k]; b0[k] = wD[2 * (j + k)]; b1[k] = wD[2 * (j + k) + 1]; }
          const uint32_t k0 = key_of(hi_, cm[0], b0[0], b1[0]), k1 = key_of(hi_, cm[1], b0[1], b1[1]);
          const uint32_t k2 = key_of(hi_, cm[2], b0[2], b1[2]), k3 = key_of(hi_, cm[3], b0[3], b1[3]);
          best_key = min(min(min(k0, k1), min(k2, k3)), best_key);
        }
        for (; j < mcnt; j++) best_key = min(best_key, key_of(hi_, wM[j], wD[2 * j], wD[2 * j + 1]));
      };
      if (hi) consume(std::true_type{});
      else consume(std::false_type{});
    }
    if (valid) {
      // min_ind defaults to 0 when no candidate was accepted (matcher.cpp:221)
      const int32_t bp = hi ? pbase + (int32_t)(best_key & 0xFFFFu) : (int32_t)(best_key & 0x7FFFFu);
      const int32_t res = (best_key == 0xFFFFFFFFu) ? 0 : cidx[bp];
      best[((int64_t)stream * 4 + a.pass[pass].slot) * s.cap + qidx[qm.y]] = res;
    }
  }
}

// ------------------------------------------------------------ match (prior term)
// findMatch with the u_,v_ distance term (matcher.cpp:257-262).  The cost is no
// longer an integer, so the key trick does not apply: one lane per query walks
// its own bin range in the reference's order (u_bin, v_bin, list position) and
// keeps the first strict minimum, all in double exactly as the reference
// (sqrt is the correctly rounded IEEE one; du*du+dv*dv is exact in double).
__global__ void match_prior_kernel(VhSets s, VhMatchArgs a, double u_, double v_, int32_t *__restrict__ best) {
  const int32_t qset = vh_role_set(a.S, a.pair_cur, 0, a.pass[0].qset);
  const int32_t cset = vh_role_set(a.S, a.pair_cur, 0, a.pass[0].cset);
  const int32_t nq = min(s.count[qset], s.cap);
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  const int32_t *q = s.feat + ((int64_t)qset * s.cap + i) * 12;
  const int32_t u1 = q[0], v1 = q[1], c = q[3];
  const uint4 a0 = *(const uint4 *)(q + 4), a1 = *(const uint4 *)(q + 8);
  const int32_t rv = a.pass[0].flow ? a.radius : a.disp_tol;
  const int32_t u_lo = u1 - a.radius, u_hi = u1 + a.radius, v_lo = v1 - rv, v_hi = v1 + rv;
  const int32_t ub0 = min(max(u_lo, 0) / s.binsize, s.ubn - 1), ub1 = min(max(u_hi, 0) / s.binsize, s.ubn - 1);
  const int32_t vb0 = min(max(v_lo, 0) / s.binsize, s.vbn - 1), vb1 = min(max(v_hi, 0) / s.binsize, s.vbn - 1);
  const int32_t *__restrict__ cbs = s.bin_start + (int64_t)cset * (s.nbins + 1);
  const uint32_t *__restrict__ cuv = s.s_uv + (int64_t)cset * s.cap;
  const uint4 *__restrict__ cdesc = (const uint4 *)(s.s_desc + (int64_t)cset * s.cap * 8);
  double min_cost = 10000000;  // matcher.cpp:222
  int32_t min_pos = -1;
  for (int32_t ub = ub0; ub <= ub1; ub++) {
    const int32_t row = (c * s.ubn + ub) * s.vbn;
    for (int32_t p = cbs[row + vb0]; p < cbs[row + vb1 + 1]; p++) {
      const uint32_t uv2 = cuv[p];
      const int32_t u2 = uv2 & 0xFFFF, v2 = uv2 >> 16;
      if (u2 < u_lo || u2 > u_hi || v2 < v_lo || v2 > v_hi) continue;
      const uint4 b0 = cdesc[2 * (int64_t)p], b1 = cdesc[2 * (int64_t)p + 1];
      uint32_t sad = sad4(a0.x, b0.x, 0);
      sad = sad4(a0.y, b0.y, sad); sad = sad4(a0.z, b0.z, sad); sad = sad4(a0.w, b0.w, sad);
      sad = sad4(a1.x, b1.x, sad); sad = sad4(a1.y, b1.y, sad); sad = sad4(a1.z, b1.z, sad); sad = sad4(a1.w, b1.w, sad);
      double cost = (double)sad;
      if (u_ >= 0 && v_ >= 0) {
        const double du = (double)u2 - u_, dv = (double)v2 - v_;
        cost += 4 * sqrt(du * du + dv * dv);
      }
      if (cost < min_cost) { min_cost = cost; min_pos = p; }
    }
  }
  best[i] = min_pos >= 0 ? s.s_idx[(int64_t)cset * s.cap + min_pos] : 0;
}

// Survivors per emission chunk: one atomic per wave (the 64 lanes of a wave hold
// consecutive features of one 256-feature chunk), not one per lane -- 64
// same-address atomics per wave made the chain kernel 8x slower.
__device__ __forceinline__ void count_chunk(bool keep, int32_t *counter) {
  const uint64_t bal = __ballot(keep);
  if (bal && (threadIdx.x & 63) == (uint32_t)__builtin_ctzll(bal)) atomicAdd(counter, (int32_t)__popcll(bal));
}

// ---------------------------------------------------------------------- chain
// Follows the circle of each driving feature through the per-pass tables and
// records the index tuple (i1p,i2p,i1c,i2c), or z=-2 when the circle does not
// close / the disparity test fails.
//   flow   (matcher.cpp:308-336): 1c ->1p ->1c
//   stereo (stock libviso2): 1c ->2c ->1c, u1c >= u2c
//   quad   (stock libviso2): 1p ->2p ->2c ->1c ->1p,
//                                             u1p >= u2p and u1c >= u2c
// For flow the reference additionally keeps only the FIRST match per pixel of
// the current image (mask M, matcher.cpp:331-334): every closing feature bids
// for its pixel with atomicMax(epoch<<20 | (0xFFFFF - i1c)); the lowest i1c of
// this epoch wins, and no clearing between frames is needed.
__global__ void chain_kernel(VhSets s, VhMatchArgs a, int32_t method, const int32_t *__restrict__ best,
                             int4 *__restrict__ chain, uint32_t *__restrict__ mask, uint32_t epoch,
                             int32_t *__restrict__ mchunk, int32_t nchm) {
  const int32_t stream = blockIdx.y;
  const int32_t set1p = vh_role_set(a.S, a.pair_cur, stream, 0), set2p = vh_role_set(a.S, a.pair_cur, stream, 1);
  const int32_t set1c = vh_role_set(a.S, a.pair_cur, stream, 2), set2c = vh_role_set(a.S, a.pair_cur, stream, 3);
  const int32_t n1p = min(s.count[set1p], s.cap), n2p = min(s.count[set2p], s.cap);
  const int32_t n1c = min(s.count[set1c], s.cap), n2c = min(s.count[set2c], s.cap);
  const int32_t *__restrict__ T = best + (int64_t)stream * 4 * s.cap;
  const int64_t cap = s.cap;
  int4 *__restrict__ out = chain + (int64_t)stream * s.cap;
  const int32_t ndrive = (method == 2) ? n1p : n1c;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ndrive; i += gridDim.x * blockDim.x) {
  if (method == 0) {
    int4 r = make_int4(-1, -1, -2, -1);
    if (n1p > 0) {
      const int32_t i1p = T[0 * cap + i];
      const int32_t i1c2 = T[1 * cap + i1p];
      if (i1c2 == i) {
        r = make_int4(i1p, -1, i, -1);
        const int32_t *f = s.feat + ((int64_t)set1c * s.cap + i) * 12;
        atomicMax(&mask[(int64_t)stream * s.W * s.H + (int64_t)f[1] * s.W + f[0]],
                  (epoch << 20) | (0xFFFFFu - (uint32_t)i));
      }
    }
    out[i] = r;
  } else if (method == 1) {
    int4 r = make_int4(-1, -1, -2, -1);
    if (n2c > 0) {
      const int32_t i2c = T[0 * cap + i];
      const int32_t i1c2 = T[1 * cap + i2c];
      const int32_t u1c = s.feat[((int64_t)set1c * s.cap + i) * 12], u2c = s.feat[((int64_t)set2c * s.cap + i2c) * 12];
      if (i1c2 == i && u1c >= u2c) r = make_int4(-1, -1, i, i2c);
    }
    out[i] = r;
    count_chunk(r.z >= 0, mchunk + stream * nchm + (i >> 8));
  } else {
    int4 r = make_int4(-1, -1, -2, -1);
    if (n2p > 0 && n1c > 0 && n2c > 0) {
      const int32_t i2p = T[0 * cap + i];
      const int32_t i2c = T[1 * cap + i2p];
      const int32_t i1c = T[2 * cap + i2c];
      const int32_t i1p2 = T[3 * cap + i1c];
      const int32_t u1p = s.feat[((int64_t)set1p * s.cap + i) * 12], u2p = s.feat[((int64_t)set2p * s.cap + i2p) * 12];
      const int32_t u1c = s.feat[((int64_t)set1c * s.cap + i1c) * 12], u2c = s.feat[((int64_t)set2c * s.cap + i2c) * 12];
      if (i1p2 == i && u1p >= u2p && u1c >= u2c) r = make_int4(i, i2p, i1c, i2c);
    }
    out[i] = r;
    count_chunk(r.z >= 0, mchunk + stream * nchm + (i >> 8));
  }
  }  // grid-stride loop
}

// ------------------------------------------------------------------ flow_keep
// Flow only: after every closing feature has bid for its pixel, keep the winner
// (the reference's first writer, matcher.cpp:331-334), drop the others, and
// count the survivors per emission chunk.
__global__ void flow_keep_kernel(VhSets s, VhMatchArgs a, int4 *__restrict__ chain,
                                 const uint32_t *__restrict__ mask, uint32_t epoch,
                                 int32_t *__restrict__ mchunk, int32_t nchm) {
  const int32_t stream = blockIdx.y;
  const int32_t set1c = vh_role_set(a.S, a.pair_cur, stream, 2);
  const int32_t n1c = min(s.count[set1c], s.cap);
  int4 *__restrict__ ch = chain + (int64_t)stream * s.cap;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n1c; i += gridDim.x * blockDim.x) {
    const int4 r = ch[i];
    const int32_t *f = s.feat + ((int64_t)set1c * s.cap + i) * 12;
    const bool win = r.z >= 0 && mask[(int64_t)stream * s.W * s.H + (int64_t)f[1] * s.W + f[0]] == ((epoch << 20) | (0xFFFFFu - (uint32_t)i));
    if (r.z >= 0 && !win) ch[i].z = -2;
    count_chunk(win, mchunk + stream * nchm + (i >> 8));
  }
}

// --------------------------------------------------------------- emit_matches
// One 256-thread workgroup per 256 driving features: ordered compaction of the closed
// circles into p_match records (48 B, src/matcher.h:89-104), in ascending order
// of the driving feature index as the reference's loops emit them.  The offset of
// a chunk is the sum of the survivor counts of the chunks before it.
__global__ void __launch_bounds__(256)
emit_matches_kernel(VhSets s, VhMatchArgs a, int32_t method, const int4 *__restrict__ chain,
                    float *__restrict__ matches, int32_t mcap, int32_t *__restrict__ match_count,
                    int32_t *__restrict__ overflow, const int32_t *__restrict__ mchunk, int32_t nchm) {
  __shared__ int32_t sWave[4];
  __shared__ int32_t sBase;
  const int32_t chunk = blockIdx.x, stream = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int32_t sets[4];
#pragma unroll
  for (int32_t r = 0; r < 4; r++) sets[r] = vh_role_set(a.S, a.pair_cur, stream, r);
  const int32_t drive = (method == 2) ? sets[0] : sets[2];
  const int32_t n = min(s.count[drive], s.cap);
  if (chunk * 256 >= n && chunk != nchm - 1) return;
  const int4 *__restrict__ ch = chain + (int64_t)stream * s.cap;
  float *__restrict__ out = matches + (int64_t)stream * mcap * 12;
  // matches emitted by earlier chunks
  int32_t part = 0;
  for (int32_t k = tid; k < chunk; k += 256) part += mchunk[stream * nchm + k];
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) sWave[w] = part;
  __syncthreads();
  if (tid == 0) { int32_t t = 0; for (int32_t k = 0; k < 4; k++) t += sWave[k]; sBase = t; }
  __syncthreads();
  const int32_t base = sBase;
  __syncthreads();

  const int32_t i = chunk * 256 + tid;
  int4 r = make_int4(-1, -1, -2, -1);
  if (i < n) r = ch[i];
  const bool keep = r.z >= 0;
  uint32_t rec[12];
#pragma unroll
  for (int32_t k = 0; k < 12; k++) rec[k] = (k % 3 == 2) ? 0xFFFFFFFFu : __float_as_uint(-1.0f);
  if (keep) {
    const int32_t idx[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int32_t k = 0; k < 4; k++) {
      if (idx[k] >= 0) {
        const int32_t *f = s.feat + ((int64_t)sets[k] * s.cap + idx[k]) * 12;
        rec[3 * k + 0] = __float_as_uint((float)f[0]);
        rec[3 * k + 1] = __float_as_uint((float)f[1]);
      }
      rec[3 * k + 2] = (uint32_t)idx[k];
    }
  }
  const uint64_t bal = __ballot(keep);
  const int32_t before = __popcll(bal & ((1ull << lane) - 1));
  if (lane == 0) sWave[w] = __popcll(bal);
  __syncthreads();
  int32_t woff = 0, tot = 0;
#pragma unroll
  for (int32_t k = 0; k < 4; k++) { const int32_t c = sWave[k]; if (k < w) woff += c; tot += c; }
  const int32_t pos = base + woff + before;
  if (keep && pos < mcap) {
    uint4 *o = (uint4 *)(out + (int64_t)pos * 12);
    o[0] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
    o[1] = make_uint4(rec[4], rec[5], rec[6], rec[7]);
    o[2] = make_uint4(rec[8], rec[9], rec[10], rec[11]);
  }
  if (chunk == nchm - 1 && tid == 0) {
    match_count[stream] = base + tot;
    // a set this method read held more features than the capacity: the matching ran on
    // its first `cap` records only, which the host reports as VH_ERR_CAPACITY
    int32_t ov = 0;
#pragma unroll
    for (int32_t r = 0; r < 4; r++) {
      const bool used = method == 2 || r == 2 || (method == 0 && r == 0) || (method == 1 && r == 3);
      if (used && s.count[sets[r]] > s.cap) ov = 1;
    }
    overflow[stream] = ov;
  }
}

}  // namespace

// The passes are split by search type: 2-d flow search (wave-uniform candidate
// stream over bins) and 1-d stereo search (per-lane rows).
static VhMatchArgs filter_passes(const VhMatchArgs &a, int32_t flow) {
  VhMatchArgs r = a;
  r.npass = 0;
  for (int32_t k = 0; k < a.npass; k++)
    if ((a.pass[k].flow != 0) == (flow != 0)) r.pass[r.npass++] = a.pass[k];
  return r;
}
void vh_launch_match_stereo(const VhSets &s, const VhMatchArgs &a, int32_t *best, hipStream_t st) {
  VhMatchArgs sr = filter_passes(a, 0);
  static const int wide = [] { const char *e = getenv("VH_FLOW_WIDE_KEYS"); return e ? atoi(e) : 0; }();
  sr.wide_keys = wide;
  if (!sr.npass) return;
  dim3 grid(std::min(std::max(s.cap / 1024, 8), 1024), sr.npass, a.S);  // 4 tiles of 64 queries per workgroup per trip
  hipLaunchKernelGGL(match_rows_kernel, grid, dim3(256), 0, st, s, sr, best);
}
void vh_launch_match_prior(const VhSets &s, const VhMatchArgs &a, double u_, double v_, int32_t *best,
                           hipStream_t st) {
  hipLaunchKernelGGL(match_prior_kernel, dim3((s.cap + 127) / 128), dim3(128), 0, st, s, a, u_, v_, best);
}
void vh_launch_match_flow(const VhSets &s, const VhMatchArgs &a, int32_t *best, hipStream_t st) {
  VhMatchArgs fl = filter_passes(a, 1);
  static const int wide = [] { const char *e = getenv("VH_FLOW_WIDE_KEYS"); return e ? atoi(e) : 0; }();
  fl.wide_keys = wide;
  if (!fl.npass) return;
  // One workgroup per 4 tiles of the capacity-sized tile list (the kernel loops,
  // so any grid is correct).  At typical densities ~75 % of these workgroups
  // find no tile and exit at once; a tight grid (VH_FLOW_WGS=38) makes the flow
  // search itself 20 % shorter but starves the detect stream beside it: 61.6 k
  // vs 62.2 k pairs/s, and 58.4 k at 40 -- measured, so the full grid stays.
  static const int wgs = [] { const char *e = getenv("VH_FLOW_WGS"); return e ? atoi(e) : 0; }();
  const int32_t gx = wgs > 0 ? wgs : (s.max_tiles + 3) / 4;
  dim3 grid(gx, fl.npass, a.S);
  // Dynamic-LDS padding (unused by the kernel) caps the flow search at 5 workgroups
  // = 20 waves per CU.  Alone it runs as fast as at 8 waves/SIMD (it is VALU-issue
  // bound); beside the detect stream it leaves that stream enough wave slots to
  // finish frame t+1's detection inside the flow search of frame t instead of
  // being starved until it ends: +5 % throughput measured (sweep 0..60000 B,
  // VH_FLOW_LDS_PAD overrides).
  static const int pad = [] { const char *e = getenv("VH_FLOW_LDS_PAD"); return e ? atoi(e) : VH_FLOW_LDS_PAD_DEFAULT; }();
  hipLaunchKernelGGL(match_kernel, grid, dim3(256), (size_t)pad, st, s, fl, best);
}
void vh_launch_chain(const VhSets &s, const VhMatchArgs &a, int32_t method, const int32_t *best,
                     int4 *chain, uint32_t *mask, uint32_t epoch, int32_t *mchunk, hipStream_t st) {
  const int32_t nchm = (s.cap + 255) / 256;
  dim3 grid(std::min(std::max(s.cap / 1024, 8), 256), a.S);
  hipLaunchKernelGGL(chain_kernel, grid, dim3(256), 0, st, s, a, method, best, chain, mask, epoch, mchunk, nchm);
  if (method == 0)
    hipLaunchKernelGGL(flow_keep_kernel, grid, dim3(256), 0, st, s, a, chain, (const uint32_t *)mask, epoch, mchunk, nchm);
}
void vh_launch_emit_matches(const VhSets &s, const VhMatchArgs &a, int32_t method, const int4 *chain,
                            void *matches, int32_t mcap, int32_t *match_count, int32_t *overflow,
                            const int32_t *mchunk, hipStream_t st) {
  const int32_t nchm = (s.cap + 255) / 256;
  hipLaunchKernelGGL(emit_matches_kernel, dim3(nchm, a.S), dim3(256), 0, st, s, a, method, chain,
                     (float *)matches, mcap, match_count, overflow, mchunk, nchm);
}
