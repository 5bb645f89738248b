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
//  * Mapping (flow_tile / rows_tile below): one wavefront per tile of <= 32 consecutive queries
//    of one class -- in SNAKE order over the (class, u-bin) columns for the flow passes
//    (kernels_bin.hip: make_tiles; the queries' order is free, only the candidates' positions carry
//    the tie-break), in (class, v) row order for the 1-d stereo passes.  The 64 lanes are
//    4 phases of 16 lanes, lane (phase, l) holds queries l and l+16, so every phase holds the
//    whole tile; the candidate region of the tile -- the union of its queries' bin ranges -- is
//    streamed through two wave-private LDS buffers of 64 candidates filled by LDS-DMA
//    (walk_region_lds), and the four phases take candidates j, j+1, j+2, j+3 of a buffer in the
//    same step: one candidate stream shared by all phases, every record read from LDS serves two
//    queries per lane, 8 v_sad_u8 / v_sad_hi_u8 per pair, the partial minima joined by two
//    shuffles per tile.
//    Positions in bin order ARE the reference's visiting order, so its first-minimum tie-break
//    (strict `<`, src/matcher.cpp:264) is the minimum of the key (SAD << 19 | position) -- or
//    (SAD << 16 | position - class base), which the v_sad_hi_u8 chain produces by itself --
//    whatever order candidates arrive in.
//  * The accept test of src/matcher.cpp:249 is applied per pair only in the "tested" form of the
//    loops.  The default, speculative form applies none: every lane takes the minimum over the
//    whole region the wave walks, a superset of its own window, tests ONE candidate at the end --
//    its winner -- and the few queries whose winner lies outside their own window are searched
//    again exactly, in groups sharing one walk (finish_tile / RedoGroup).  Either way the result
//    is findMatch's; engine.hip picks the form per launch from the observed share of such queries.
#include "vh_dev.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#ifdef VH_FLOW_STATS
// debug build only (EXTRA=-DVH_FLOW_STATS): what the flow search executed --
// [0] tiles [1] chunks [2..4] trips without test / v test / full test [5] queries [6] columns [7] queries searched again
__device__ unsigned long long g_flow_stats[12];  // [8] stereo tiles [9] stereo trips [10] stereo queries searched again [11] stereo queries
extern "C" int32_t vh_debug_flow_stats(unsigned long long *out, int32_t reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flow_stats), sizeof(g_flow_stats)) != hipSuccess) return -3;
  if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_flow_stats), z, sizeof(z)) != hipSuccess) return -3; }
  return 0;
}
#define VH_STAT(k, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_flow_stats[k], (unsigned long long)(n)); } while (0)
#else
#define VH_STAT(k, n) do { } while (0)
#endif

// register budget of match_kernel: at least this many waves per SIMD must fit (the loop hides its LDS and L2
// round trips behind other waves; 99 registers / 4 waves measured 7 % slower than 68 / 7)
#ifndef VH_MATCH_WAVES
#define VH_MATCH_WAVES 7
#endif
#ifndef VH_SNAKE
#define VH_SNAKE 1  // 0 (experiments): query tiles over the plain bin order
#endif

namespace {

// features of a set that are in its bin order (the set's true count, s.count, can be larger: capacity)
__device__ __forceinline__ int32_t indexed_count(const VhSets &s, int32_t set) { return s.bin_start[(int64_t)set * (s.nbins + 1) + s.nbins]; }

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

// All-reduce over each 16-lane row of the wave with DPP row rotations (no LDS, no
// bpermute): afterwards every lane holds the extremum of its row.
template <bool MAX>
__device__ __forceinline__ int32_t row16_allreduce(int32_t v) {
#define VH_ROW_STEP(CTRL)                                                                   \
  {                                                                                         \
    const int32_t t = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);             \
    v = MAX ? max(v, t) : min(v, t);                                                        \
  }
  VH_ROW_STEP(0x128)  // row_ror:8
  VH_ROW_STEP(0x124)  // row_ror:4
  VH_ROW_STEP(0x122)  // row_ror:2
  VH_ROW_STEP(0x121)  // row_ror:1
#undef VH_ROW_STEP
  return v;
}

// The candidate stream of a tile of the flow search: every candidate of class c in
// the bins [UB0, UB1] x [VB0, VB1] (u-bin major, as Matcher::findMatch visits them,
// matcher.cpp:243-246), handed to `consume` in chunks of <= 64 consecutive positions,
// one record per lane (lane j: candidate min(pc+j, p1-1), so the lanes past the end
// of a column repeat its last candidate).  Software-pipelined: the column table
// (first/last position of up to 64 columns, one column per lane) costs one round
// trip per batch of columns, and the records of chunk k+1 are in flight while
// `consume` works on chunk k.  With 16..32 queries per tile a chunk is only a few
// hundred cycles of work, less than one L2 round trip; unpipelined, those round
// trips (two per column for the table, one per chunk for the records) bounded
// small tiles.
//   consume(pc, p1, pa0, pa1, pl, gu, g0, g1): chunk [pc, min(pc+64, p1)) of a column
//   ending at p1; [pa0, pa1) = its positions inside EVERY query's window (TESTED
//   only; pa0 > pa1 marks a column that is not inside every query's u window);
//   pl/gu/g0/g1 = this lane's candidate position, u|v<<16 (NEED_UV only) and descriptor.
template <bool TESTED, bool NEED_UV, class Consume>
__device__ __forceinline__ void walk_region(const VhSets &s, const int32_t *__restrict__ cbs, const uint32_t *__restrict__ cuv,
                                            const uint4 *__restrict__ cdesc, int32_t c, int32_t UB0, int32_t UB1, int32_t VB0,
                                            int32_t VB1, int32_t ULO_MAX, int32_t UHI_MIN, int32_t VA0, int32_t VA1,
                                            Consume consume) {
  const int32_t lane = threadIdx.x & 63;
  // When the v range covers every v-bin (2*radius >= H, as at KITTI size), the columns
  // [UB0, UB1] are one contiguous run of positions: walk it as a single column -- fewer,
  // fuller chunks and one rounding of the tail instead of one per column.
  const bool merged = !TESTED && VB0 == 0 && VB1 == s.vbn - 1;
  const int32_t UB1w = merged ? UB0 : UB1;
  for (int32_t cb = UB0; cb <= UB1w; cb += 64) {
    const int32_t ncb = min(64, UB1w - cb + 1);
    int32_t t_p0 = 0, t_p1 = 0, t_a0 = 0, t_a1 = 0;
    if (lane < ncb) {
      const int32_t row = (c * s.ubn + cb + lane) * s.vbn;
      t_p0 = cbs[row + VB0]; t_p1 = merged ? cbs[(c * s.ubn + UB1 + 1) * s.vbn] : cbs[row + VB1 + 1];
      if (TESTED) {
        const int32_t ubx = cb + lane;
        const bool in_ = ubx * s.binsize >= ULO_MAX && ubx * s.binsize + s.binsize - 1 <= UHI_MIN;
        // an interior column without untested bins has t_a0 == t_a1 == t_p1
        t_a0 = in_ ? ((VA0 <= VA1) ? cbs[row + VA0] : t_p1) : 1;
        t_a1 = in_ ? ((VA0 <= VA1) ? cbs[row + VA1 + 1] : t_p1) : 0;
      }
    }
    // chunk cursor: column ci of the batch, positions [pc, p1)
    int32_t ci = -1, pc = -64, p1 = 0;
    const auto advance = [&]() -> bool {  // to the next non-empty chunk; false past the last one (wave-uniform)
      pc += 64;
      while (pc >= p1) {
        if (++ci >= ncb) return false;
        pc = __builtin_amdgcn_readlane(t_p0, ci); p1 = __builtin_amdgcn_readlane(t_p1, ci);
      }
      return true;
    };
    if (!advance()) continue;
    int32_t pl = min(pc + lane, p1 - 1);
    uint32_t gu = 0;
    if (NEED_UV) gu = cuv[pl];
    uint4 g0 = cdesc[2 * (int64_t)pl], g1 = cdesc[2 * (int64_t)pl + 1];
    for (;;) {
      const int32_t c_pc = pc, c_p1 = p1, c_ci = ci, c_pl = pl;
      const uint32_t c_gu = gu;
      const uint4 c_g0 = g0, c_g1 = g1;
      const bool more = advance();
      if (more) {  // next chunk's records: in flight while this one is consumed
        pl = min(pc + lane, p1 - 1);
        if (NEED_UV) gu = cuv[pl];
        g0 = cdesc[2 * (int64_t)pl]; g1 = cdesc[2 * (int64_t)pl + 1];
      }
      int32_t pa0 = 1, pa1 = 0;
      if (TESTED) { pa0 = __builtin_amdgcn_readlane(t_a0, c_ci); pa1 = __builtin_amdgcn_readlane(t_a1, c_ci); }
      consume(c_pc, c_p1, pa0, pa1, c_pl, c_gu, c_g0, c_g1);
      if (!more) break;
    }
  }
}

// LDS-DMA (gfx950 global_load_lds_*): a wave-instruction copies 64 x 16 (or 4) bytes from per-lane global
// addresses straight to LDS at M0 + lane * size -- no VGPR destination, no ds_write.  hipcc does not count an asm
// memory operation, so completion is waited for by hand (vmcnt) before the chunk is read; the builtin form makes
// hipcc drain every DMA (vmcnt(0)) before ANY ds_read, which would serialise the prefetch of the next chunk.
__device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p);
}

// walk_region for the flow tiles (the stream is consumed from LDS): the same chunk cursor, but the records of a
// chunk travel global -> LDS by LDS-DMA into one of two wave-private buffers (wD: 2 x {64 first halves | 64 second
// halves}, wU: 2 x 64 u|v<<16 words, TESTED only), chunk k+1 in flight while `consume` reads chunk k.  No staging
// registers, no register copies of a software pipeline, no ds_write: ~10 instead of ~35 VALU instructions per chunk.
//   consume(pc, p1, pa0, pa1, cD, cU): chunk [pc, min(pc+64, p1)) staged at cD / cU (slot j: candidate min(pc+j, p1-1))
template <bool TESTED, class Consume>
__device__ __forceinline__ void walk_region_lds(const VhSets &s, const int32_t *__restrict__ cbs, const uint32_t *__restrict__ cuv,
                                                const uint4 *__restrict__ cdesc, int32_t c, int32_t UB0, int32_t UB1, int32_t VB0,
                                                int32_t VB1, int32_t ULO_MAX, int32_t UHI_MIN, int32_t VA0, int32_t VA1,
                                                uint4 *wD, uint32_t *wU, Consume consume) {
  const int32_t lane = threadIdx.x & 63;
  const uint32_t ldsD = lds_addr(wD), ldsU = lds_addr(wU);
  const bool merged = !TESTED && VB0 == 0 && VB1 == s.vbn - 1;
  const int32_t UB1w = merged ? UB0 : UB1;
  for (int32_t cb = UB0; cb <= UB1w; cb += 64) {
    const int32_t ncb = min(64, UB1w - cb + 1);
    int32_t t_p0 = 0, t_p1 = 0, t_a0 = 0, t_a1 = 0;
    if (lane < ncb) {
      const int32_t row = (c * s.ubn + cb + lane) * s.vbn;
      t_p0 = cbs[row + VB0]; t_p1 = merged ? cbs[(c * s.ubn + UB1 + 1) * s.vbn] : cbs[row + VB1 + 1];
      if (TESTED) {
        const int32_t ubx = cb + lane;
        const bool in_ = ubx * s.binsize >= ULO_MAX && ubx * s.binsize + s.binsize - 1 <= UHI_MIN;
        t_a0 = in_ ? ((VA0 <= VA1) ? cbs[row + VA0] : t_p1) : 1;
        t_a1 = in_ ? ((VA0 <= VA1) ? cbs[row + VA1 + 1] : t_p1) : 0;
      }
    }
    int32_t ci = -1, pc = -64, p1 = 0;
    const auto advance = [&]() -> bool {
      pc += 64;
      while (pc >= p1) {
        if (++ci >= ncb) return false;
        pc = __builtin_amdgcn_readlane(t_p0, ci); p1 = __builtin_amdgcn_readlane(t_p1, ci);
      }
      return true;
    };
    const auto issue = [&](int32_t b) {  // chunk [pc, p1) -> buffer b
      const int32_t pl = min(pc + lane, p1 - 1);
      const uint4 *g = cdesc + 2 * (int64_t)pl;
      glds16(g, ldsD + (uint32_t)b * 2048u); glds16(g + 1, ldsD + (uint32_t)b * 2048u + 1024u);
      if (TESTED) glds4(cuv + pl, ldsU + (uint32_t)b * 256u);
    };
    if (!advance()) continue;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of the buffers by an earlier walk has returned
    int32_t buf = 0;
    issue(0);
    for (;;) {
      const int32_t c_pc = pc, c_p1 = p1, c_ci = ci, c_buf = buf;
      const bool more = advance();
      if (more) {
        buf ^= 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // chunk k-1, the last reader of this buffer, is consumed
        issue(buf);
        if (TESTED) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // chunk k has landed (k+1 may be in flight)
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      int32_t pa0 = 1, pa1 = 0;
      if (TESTED) { pa0 = __builtin_amdgcn_readlane(t_a0, c_ci); pa1 = __builtin_amdgcn_readlane(t_a1, c_ci); }
      consume(c_pc, c_p1, pa0, pa1, (const uint4 *)(wD + c_buf * 128), (const uint32_t *)(wU + c_buf * 64));
      if (!more) break;
    }
  }
}

// One candidate (this lane's) against ONE wave-uniform query with the literal accept
// test of matcher.cpp:249: the key (SAD << 32 | position - pbase), or ~0.
// The slow, exact path behind the speculative searches below (lanes over candidates).
__device__ __forceinline__ uint64_t tested_key_uniform_query(const uint32_t (&qd)[8], us2 lo2, us2 span2, uint32_t uv2,
                                                             const uint4 &b0, const uint4 &b1, uint32_t relpos) {
  const us2 t = as_us2(uv2) - lo2;
  const us2 m = __builtin_elementwise_min(t, span2);
  uint32_t sad = sad4(qd[0], b0.x, 0);
  sad = sad4(qd[1], b0.y, sad); sad = sad4(qd[2], b0.z, sad); sad = sad4(qd[3], b0.w, sad);
  sad = sad4(qd[4], b1.x, sad); sad = sad4(qd[5], b1.y, sad); sad = sad4(qd[6], b1.z, sad); sad = sad4(qd[7], b1.w, sad);
  return as_u32(t) != as_u32(m) ? ~0ull : (((uint64_t)sad << 32) | relpos);
}
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t k, int32_t d) {
  const uint32_t lo = (uint32_t)__shfl_xor((int32_t)(uint32_t)k, d), hi = (uint32_t)__shfl_xor((int32_t)(uint32_t)(k >> 32), d);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t k) {
#pragma unroll
  for (int32_t d = 32; d >= 1; d >>= 1) k = min(k, shfl_xor_u64(k, d));
  return k;
}

// Match keys.  The minimum of (SAD, bin-order position relative to the first position of
// the class) in lexicographic order is findMatch's first strict minimum (matcher.cpp:264),
// because positions in bin order are the reference's visiting order.  Three encodings:
//   KEY_HI16  SAD << 16 | position: straight out of a v_sad_hi_u8 chain (it accumulates at
//             bit 16) seeded with the position; classes of < 2^16 - 64 features
//   KEY_W19   SAD << 19 | position from a v_sad_u8 chain and one v_lshl_or_b32
//             (SAD <= 8160 < 2^13); classes of <= VH_CLASS_POS_MAX features
//   KEY_64    SAD << 32 | position in 64 bits (compare + two selects per candidate):
//             any class size -- only images of more than 524 224 NMS blocks can need it
enum { KEY_HI16 = 0, KEY_W19 = 1, KEY_64 = 2 };
template <int KM> struct KeyT { typedef uint32_t type; };
template <> struct KeyT<KEY_64> { typedef uint64_t type; };
template <int KM>
__device__ __forceinline__ typename KeyT<KM>::type sad_key(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1, uint32_t seed) {
  if (KM == KEY_HI16) {
    uint32_t key = sad4hi(a0.x, b0.x, seed);
    key = sad4hi(a0.y, b0.y, key); key = sad4hi(a0.z, b0.z, key); key = sad4hi(a0.w, b0.w, key);
    key = sad4hi(a1.x, b1.x, key); key = sad4hi(a1.y, b1.y, key); key = sad4hi(a1.z, b1.z, key); key = sad4hi(a1.w, b1.w, key);
    return (typename KeyT<KM>::type)key;
  }
  uint32_t sad = sad4(a0.x, b0.x, 0);
  sad = sad4(a0.y, b0.y, sad); sad = sad4(a0.z, b0.z, sad); sad = sad4(a0.w, b0.w, sad);
  sad = sad4(a1.x, b1.x, sad); sad = sad4(a1.y, b1.y, sad); sad = sad4(a1.z, b1.z, sad); sad = sad4(a1.w, b1.w, sad);
  if (KM == KEY_W19) return (typename KeyT<KM>::type)((sad << 19) | seed);
  return (typename KeyT<KM>::type)(((uint64_t)sad << 32) | seed);
}
// any encoding -> SAD << 32 | position (or ~0 for "none")
template <int KM>
__device__ __forceinline__ uint64_t key_to_64(typename KeyT<KM>::type k) {
  if (KM == KEY_64) return (uint64_t)k;
  if ((uint32_t)k == 0xFFFFFFFFu) return ~0ull;
  const uint32_t kk = (uint32_t)k;
  return KM == KEY_HI16 ? (((uint64_t)(kk >> 16) << 32) | (kk & 0xFFFFu)) : (((uint64_t)(kk >> 19) << 32) | (kk & 0x7FFFFu));
}
__device__ __forceinline__ int32_t key_mode_of(int32_t class_count, int32_t wide_keys) {
  // (+64: the copies past the end of a run carry positions up to 63 past it)
  if (class_count + 64 <= 0x10000 && !wide_keys) return KEY_HI16;
  return class_count <= VH_CLASS_POS_MAX && wide_keys < 2 ? KEY_W19 : KEY_64;
}

// Second half of the speculative searches.  A query whose winner over the walked region lies
// outside its own window is searched again with the literal accept test of matcher.cpp:249, lanes
// over candidates.  Up to VH_REDO_G such queries of a tile share ONE walk over the union of their
// own regions: the candidate records are loaded once per group (round 2 walked once per query, and
// that stream of 40 KB per query, not the arithmetic, is what made the speculative form lose on
// images full of features without a partner), every lane tests its candidate against each query of
// the group (descriptors and windows wave-uniform, in scalar registers).  Measured on MI355X (KITTI, S = 256,
// speculative loops forced, k pairs/s at noise 0 / +-1 / +-2 grey levels; the tested loops: 101.9 / 67.2 / 48.5):
// G = 1: 101.3 / 55.3 / 38.0, G = 2: 101.1 / 60.3 / 42.1, G = 4: 100.9 / 62.7 / 43.9, G = 8: 93.1* / 58.2 / 42.0
// (* with the position tables of branch exp/position-tables).  Round 2's one-walk-per-query form: 51.9 / 34.0.
#ifndef VH_REDO_G
#define VH_REDO_G 2
#endif
struct RedoGroup {
  uint32_t qd[VH_REDO_G][8];  // descriptors (wave-uniform)
  us2 lo2[VH_REDO_G];         // window origins
  int32_t lane[VH_REDO_G], qi[VH_REDO_G], n;
  int32_t umin, umax, vmin, vmax;
};
// Takes up to VH_REDO_G (lane, query slot) pairs off the ballots `todo[Q]`; false when none is left.
template <int Q>
__device__ __forceinline__ bool redo_take(uint64_t (&todo)[Q], const uint4 (&a0)[Q], const uint4 (&a1)[Q], const uint32_t (&uv1)[Q],
                                          int32_t radius, int32_t rv, RedoGroup &g) {
  g.n = 0; g.umin = g.vmin = 0x7FFFFFFF; g.umax = g.vmax = -1;
#pragma unroll
  for (int32_t k = 0; k < VH_REDO_G; k++) {
    int32_t fl = -1, fq = 0;
#pragma unroll
    for (int32_t qi = 0; qi < Q; qi++)
      if (fl < 0 && todo[qi]) { fl = (int32_t)__builtin_ctzll(todo[qi]); fq = qi; todo[qi] &= todo[qi] - 1; }
    if (fl < 0) break;  // wave-uniform
    uint4 x0 = a0[0], x1 = a1[0];
    uint32_t xu = uv1[0];
#pragma unroll
    for (int32_t qi = 1; qi < Q; qi++) if (fq == qi) { x0 = a0[qi]; x1 = a1[qi]; xu = uv1[qi]; }
    g.qd[k][0] = __builtin_amdgcn_readlane(x0.x, fl); g.qd[k][1] = __builtin_amdgcn_readlane(x0.y, fl);
    g.qd[k][2] = __builtin_amdgcn_readlane(x0.z, fl); g.qd[k][3] = __builtin_amdgcn_readlane(x0.w, fl);
    g.qd[k][4] = __builtin_amdgcn_readlane(x1.x, fl); g.qd[k][5] = __builtin_amdgcn_readlane(x1.y, fl);
    g.qd[k][6] = __builtin_amdgcn_readlane(x1.z, fl); g.qd[k][7] = __builtin_amdgcn_readlane(x1.w, fl);
    const uint32_t quv1 = __builtin_amdgcn_readlane(xu, fl);
    const int32_t u1 = (int32_t)(quv1 & 0xFFFF), v1 = (int32_t)(quv1 >> 16);
    g.lo2[k] = us2{(unsigned short)(u1 - radius), (unsigned short)(v1 - rv)};
    g.lane[k] = fl; g.qi[k] = fq; g.n = k + 1;
    g.umin = min(g.umin, u1); g.umax = max(g.umax, u1); g.vmin = min(g.vmin, v1); g.vmax = max(g.vmax, v1);
  }
  return g.n > 0;
}

// What a tile hands to finish_tile: per query slot the minimum key over the walked region, joined over
// the phases (SAD << 32 | class-relative position, ~0: none), and the query itself.
template <int Q> struct TileOut {
  uint64_t k[Q];
  uint4 a0[Q], a1[Q];
  uint32_t uv1[Q];
  int32_t qpos[Q];  // the query's bin-order position
  bool valid[Q];
};

// The end of a tile, shared by every key encoding and by both kinds of pass (one copy of this code per
// kernel: inside the KM-templated tile functions it would be there six times).  Speculative form: the
// winner over the walked region is tested against the query's own window, and the queries that fail are
// searched again in groups (RedoGroup above).  The results go to the table as the reference's findMatch
// returns them: best[query's feature index] = winner's feature index.  (Round 3 also built the tables in
// position space -- entries = bin-order positions, stored at the queries' own positions, whole rows per
// store, the chain followed in position space: branch exp/position-tables.  The table writes fell from
// 14x to 1x their payload, but the chain then needs eleven gathers and a scattered record per circle
// instead of six gathers: 126 vs 65 us, and the step 99.1 k vs 102.0 k pairs/s on the same box.  The
// 4-byte scatter below costs no time; it stays.)
template <int Q, int P, bool SPEC, bool FLOW>
__device__ __forceinline__ void finish_tile(const VhSets &s, const VhMatchArgs &a, int32_t pass, int32_t stream, int32_t qset, int32_t cset,
                                            int32_t c, int32_t pbase, int32_t pcnt, const TileOut<Q> &o,
                                            int32_t *__restrict__ best, int32_t *__restrict__ redo_count) {
  constexpr int L = 64 / P;
  const int32_t lane = threadIdx.x & 63, ph = lane / L;
  const uint32_t *__restrict__ cuv = s.s_uv + (int64_t)cset * s.cap;
  const uint4 *__restrict__ cdesc = (const uint4 *)(s.s_desc + (int64_t)cset * s.cap * 8);
  const int32_t *__restrict__ cbs = s.bin_start + (int64_t)cset * (s.nbins + 1);
  const int32_t rv = FLOW ? a.radius : a.disp_tol;
  const us2 span2 = {(unsigned short)(2 * a.radius), (unsigned short)(2 * rv)};
  const int32_t *__restrict__ cidx = s.s_idx + (int64_t)cset * s.cap;
  const int32_t *__restrict__ qidx = s.s_idx + (int64_t)qset * s.cap;
  int32_t *__restrict__ tbl = best + ((int64_t)stream * 4 + a.pass[pass].slot) * s.cap;
  int32_t res[Q];
  uint64_t todo[Q];
#pragma unroll
  for (int32_t qi = 0; qi < Q; qi++) {
    res[qi] = -1;  // no candidate accepted
    bool fail = false;
    if (o.valid[qi] && ph == 0 && o.k[qi] != ~0ull) {
      // (a winner is never one of the copies past the end of a run -- the original has the same SAD at a lower
      //  position -- so its position lies inside the class)
      int32_t wp = (int32_t)(uint32_t)o.k[qi];
      VH_CHECK_RANGE(s, 7, wp, 0, pcnt);
      res[qi] = pbase + wp;
      if (SPEC) {  // the winner over the walked region: inside this query's own window?
        const int32_t u1 = o.uv1[qi] & 0xFFFF, v1 = o.uv1[qi] >> 16;
        const us2 lo2 = {(unsigned short)(u1 - a.radius), (unsigned short)(v1 - rv)};
        const us2 t = as_us2(cuv[res[qi]]) - lo2;
        const us2 m = __builtin_elementwise_min(t, span2);
        fail = as_u32(t) != as_u32(m);
      }
    }
    todo[qi] = SPEC ? __ballot(fail) : 0ull;
  }
  if (SPEC) {
    int32_t nfail = 0;
#pragma unroll
    for (int32_t qi = 0; qi < Q; qi++) nfail += (int32_t)__popcll(todo[qi]);
    if (nfail) {  // wave-uniform
      VH_STAT(FLOW ? 7 : 10, nfail);
      if (lane == 0) atomicAdd(redo_count, nfail);
      RedoGroup g;
      while (redo_take<Q>(todo, o.a0, o.a1, o.uv1, a.radius, rv, g)) {
        uint64_t kk[VH_REDO_G];
#pragma unroll
        for (int32_t k = 0; k < VH_REDO_G; k++) kk[k] = ~0ull;
        if (FLOW) {
          const auto bin_of = [&](int32_t x, int32_t nb) -> int32_t {
            const uint32_t xx = (uint32_t)max(x, 0);
            return min((int32_t)(s.binsize == 1 ? xx : __umulhi(xx, s.inv_binsize)), nb - 1);
          };
          // the union of the group's own bin ranges (matcher.cpp:237-240)
          walk_region<false, true>(s, cbs, cuv, cdesc, c, bin_of(g.umin - a.radius, s.ubn), bin_of(g.umax + a.radius, s.ubn),
                                   bin_of(g.vmin - rv, s.vbn), bin_of(g.vmax + rv, s.vbn), 0, 0, 0, 0,
            [&](int32_t, int32_t, int32_t, int32_t, int32_t pl, uint32_t gu, const uint4 &g0, const uint4 &g1) {
#pragma unroll
              for (int32_t k = 0; k < VH_REDO_G; k++)
                if (k < g.n) kk[k] = min(kk[k], tested_key_uniform_query(g.qd[k], g.lo2[k], span2, gu, g0, g1, (uint32_t)(pl - pbase)));
            });
        } else {
          // the union of the group's own rows [v - tol, v + tol]
          const int32_t *__restrict__ crs = s.row_start + (int64_t)cset * (4 * s.H + 1);
          const int32_t *__restrict__ cpos = s.r_pos + (int64_t)cset * s.cap;
          const int32_t x0 = __builtin_amdgcn_readfirstlane(crs[c * s.H + max(g.vmin - rv, 0)]);
          const int32_t x1 = __builtin_amdgcn_readfirstlane(crs[c * s.H + min(g.vmax + rv, s.H - 1) + 1]);
          for (int32_t x = x0 + lane; x < x1; x += 64) {
            int32_t cp = cpos[x];
            VH_CHECK_RANGE(s, 3, cp, pbase, pbase + pcnt);
            const uint32_t gu = cuv[cp];
            const uint4 g0 = cdesc[2 * (int64_t)cp], g1 = cdesc[2 * (int64_t)cp + 1];
#pragma unroll
            for (int32_t k = 0; k < VH_REDO_G; k++)
              if (k < g.n) kk[k] = min(kk[k], tested_key_uniform_query(g.qd[k], g.lo2[k], span2, gu, g0, g1, (uint32_t)(cp - pbase)));
          }
        }
#pragma unroll
        for (int32_t k = 0; k < VH_REDO_G; k++)
          if (k < g.n) {
            const uint64_t kf = wave_min_u64(kk[k]);
            int32_t wp = (int32_t)(uint32_t)kf;
            if (kf != ~0ull) VH_CHECK_RANGE(s, 7, wp, 0, pcnt);
            const int32_t r = kf == ~0ull ? -1 : pbase + wp;
#pragma unroll
            for (int32_t qi = 0; qi < Q; qi++) if (lane == g.lane[k] && g.qi[k] == qi) res[qi] = r;
          }
      }
    }
  }
#pragma unroll
  for (int32_t qi = 0; qi < Q; qi++)
    if (o.valid[qi] && ph == 0)  // min_ind defaults to 0 when no candidate was accepted (matcher.cpp:221)
      tbl[qidx[o.qpos[qi]]] = res[qi] < 0 ? 0 : cidx[res[qi]];
}

// One tile of the flow search.
//
// A tile is T = 64*Q/P consecutive queries of one class in snake order.  The wave's
// 64 lanes are P *phases* of L = 64/P lanes; lane (phase, l) holds the Q queries
// l, l+L, .. of the tile, so every phase holds the whole tile, and the P phases
// take the candidates j, j+1, .., j+P-1 of the staged chunk in the same step:
// one candidate stream shared by all phases, each candidate evaluated by exactly
// one phase.  Compared with one query per lane and a wave-uniform candidate
// (round 1: T = 64, P = 1, Q = 1):
//  * a 16- or 32-query tile is compact (about one 50x50 bin), so the union of its
//    lanes' windows exceeds each lane's own window by less: fewer evaluated pairs
//    outside the window (tools/tile_model3.py, tools/flow_stats.py);
//  * with Q = 2 every candidate record read from LDS serves two queries per
//    lane.  The record reads (4 LDS cycles per ds_read_b128 and wave, four SIMDs
//    sharing one LDS array) otherwise saturate the LDS beside 8 v_sad_u8 per
//    step: profiles/r02_ubench_valu.txt, last block.
// The minimum over a query's candidates is split over the P phases and joined
// with log2(P) shuffles per tile.
//
// SPEC (the default): *speculative* search.  The per-pair accept test of
// matcher.cpp:249 is not applied in the loop at all: every lane takes the minimum
// key over the whole union region the wave walks -- a superset of its own window
// -- which costs 8 v_sad_hi_u8 + 1.25 other VALU instructions per pair instead of
// 8 + 3.5..5.25 (the tests were 29 % of the loop's instructions, more than the
// pairs evaluated outside the window).  At the end each lane tests ONE candidate,
// its winner: if it lies inside the lane's window it is also the minimum over the
// window (keys are unique), i.e. exactly findMatch's result.  Otherwise (some
// out-of-window candidate resembled the query more than every in-window one:
// 0.5 % of the queries on the benchmark frames, ~10 % with two grey levels of
// sensor noise added, where most features have no true partner) the query is
// searched again by the whole wave with the literal test (flow_query_by_wave).
// Results are identical either way; only the time depends on the data.
// !SPEC keeps the tested loop: three accept-test classes per bin as in round 1.
//
// Staged chunk: 64 candidates, lane j loads candidate min(pc+j, p1-1).  Slots
// past the end of the column therefore hold copies of its last candidate; they
// are evaluated like real ones with positions p1, p1+1, ..: same SAD as the
// original at a larger position, so their keys are larger than the original's
// key and never change a minimum.  That lets every range of the chunk be rounded
// outward to whole trips of 2*P candidates without any tail code.
template <int Q, int P, int KM, bool SPEC>
__device__ __forceinline__ void flow_tile(const VhSets &s, const VhMatchArgs &a, int32_t pass, int32_t stream,
                                          int32_t qset, int32_t cset, int32_t q0, int32_t q1, int32_t c, int32_t col0,
                                          int32_t pbase, int32_t pcnt, uint4 *wD, uint32_t *wU, TileOut<Q> &out) {
  constexpr int L = 64 / P;
  static_assert(L == 8 || L == 16, "every 16-lane row must hold the whole tile (row-wise window reduction)");
  constexpr int TRIP = 2 * P;  // candidates per trip: two steps, joined by one v_min3_u32 per query
  const int32_t lane = threadIdx.x & 63, ph = lane / L, l = lane % L;
  const uint32_t *__restrict__ quv = s.s_uv + (int64_t)qset * s.cap;
  const uint4 *__restrict__ qdesc = (const uint4 *)(s.s_desc + (int64_t)qset * s.cap * 8);
  const uint32_t *__restrict__ cuv = s.s_uv + (int64_t)cset * s.cap;
  const uint4 *__restrict__ cdesc = (const uint4 *)(s.s_desc + (int64_t)cset * s.cap * 8);
  const int32_t *__restrict__ cbs = s.bin_start + (int64_t)cset * (s.nbins + 1);
  const int32_t rv = a.pass[pass].flow ? a.radius : a.disp_tol;

  bool valid[Q];
  uint4 a0[Q], a1[Q];
  uint32_t uv1[Q];
  int32_t v_lo[Q];
  us2 lo2[Q];
  typedef typename KeyT<KM>::type key_t;
  const key_t KNONE = (key_t)~(key_t)0;
  key_t best_key[Q];
  int32_t umin = 0x7FFFFFFF, umax = -1, vmin = 0x7FFFFFFF, vmax = -1;
  // [q0, q1) are SNAKE indices (kernels_bin.hip: make_tiles): inside the column [A, B) of the query set's bin order
  // the index k is the position k (even column) or A + B - 1 - k (odd column).  Column starts from col0 on, one per lane.
  int32_t qp[Q];
#pragma unroll
  for (int32_t qi = 0; qi < Q; qi++) {
    const int32_t q = q0 + L * qi + l;
    valid[qi] = q < q1;
    qp[qi] = valid[qi] ? q : q0;
  }
  {
    const int32_t *__restrict__ qbs = s.bin_start + (int64_t)qset * (s.nbins + 1);
    int32_t colb = col0, j = 0;
    int32_t csv = qbs[(c * s.ubn + min(colb + lane, s.ubn)) * s.vbn];
    int32_t A = __builtin_amdgcn_readlane(csv, 0);
    int32_t k[Q];
#pragma unroll
    for (int32_t qi = 0; qi < Q; qi++) k[qi] = qp[qi];
    for (;;) {
      const int32_t B = __builtin_amdgcn_readlane(csv, j + 1);
      if (VH_SNAKE && ((colb + j) & 1)) {
#pragma unroll
        for (int32_t qi = 0; qi < Q; qi++) if (k[qi] >= A && k[qi] < B) qp[qi] = A + B - 1 - k[qi];
      }
      if (B >= q1 || colb + j + 1 >= s.ubn) break;  // (wave-uniform)
      A = B;
      if (++j == 63) { colb += 63; j = 0; csv = qbs[(c * s.ubn + min(colb + lane, s.ubn)) * s.vbn]; }
    }
  }
#pragma unroll
  for (int32_t qi = 0; qi < Q; qi++) {
    const int32_t ql = qp[qi];
    uv1[qi] = quv[ql];
    a0[qi] = qdesc[2 * (int64_t)ql]; a1[qi] = qdesc[2 * (int64_t)ql + 1];
    const int32_t u1 = uv1[qi] & 0xFFFF, v1 = uv1[qi] >> 16;
    // search window (matcher.cpp:231-234; stereo: v narrowed to +-disp_tolerance).
    // Accept test of matcher.cpp:249 in packed 16-bit arithmetic: with
    // t = (u2,v2) - (u_lo,v_lo) (mod 2^16 per half), the candidate is inside the
    // window iff t.u <= 2*radius and t.v <= 2*rv, i.e. iff min(t, span) == t.
    // Exact because coordinates are < 2^14 and radii <= 2^14 (|u2-u1|+r < 2^15).
    v_lo[qi] = v1 - rv;
    lo2[qi] = us2{(unsigned short)(u1 - a.radius), (unsigned short)v_lo[qi]};
    best_key[qi] = KNONE;
    // (lanes past q1 repeat query q0: it is valid, so the extrema are unchanged)
    umin = min(umin, u1); umax = max(umax, u1); vmin = min(vmin, v1); vmax = max(vmax, v1);
  }
  // every 16-lane row holds all queries of the tile: row-wise extrema are the tile's
  umin = __builtin_amdgcn_readfirstlane(row16_allreduce<false>(umin));
  umax = __builtin_amdgcn_readfirstlane(row16_allreduce<true>(umax));
  vmin = __builtin_amdgcn_readfirstlane(row16_allreduce<false>(vmin));
  vmax = __builtin_amdgcn_readfirstlane(row16_allreduce<true>(vmax));
  // bins of interest of the union window (matcher.cpp:237-240; for x<0 the clamp to 0
  // makes the truncating division equivalent to the reference's floor); the bin of a
  // coordinate is monotone, so the union's bins follow from the extreme queries
  const auto bin_of = [&](int32_t x, int32_t nb) -> int32_t {
    const uint32_t xx = (uint32_t)max(x, 0);
    return min((int32_t)(s.binsize == 1 ? xx : __umulhi(xx, s.inv_binsize)), nb - 1);
  };
  const int32_t UB0 = bin_of(umin - a.radius, s.ubn), UB1 = bin_of(umax + a.radius, s.ubn);
  const int32_t VB0 = bin_of(vmin - rv, s.vbn), VB1 = bin_of(vmax + rv, s.vbn);
  const us2 span2 = {(unsigned short)(2 * a.radius), (unsigned short)(2 * rv)};
  // (tested loop) columns whose pixel range [ub*bs, ub*bs+bs-1] is inside EVERY query's u
  // window, and v-bins [VA0, VA1] whose pixel rows lie inside EVERY query's v window: in
  // an interior column their candidates need no accept test at all
  const int32_t ULO_MAX = umax - a.radius, UHI_MIN = umin + a.radius;
  const int32_t VLO_MAX = vmax - rv, VHI_MIN = vmin + rv;
  const int32_t VA0 = SPEC ? 0 : max(VB0, (max(VLO_MAX, 0) + s.binsize - 1) / s.binsize);
  const int32_t VA1 = SPEC ? 0 : min(VB1, (VHI_MIN + 1) / s.binsize - 1);

  // best = min over candidates of the key (sad_key): SAD, then bin-order position.
  // TEST: 2 = full (u,v) window test, 1 = v only, 0 = none
  auto make_key = [&](auto test, int32_t qi, uint32_t uv2, const uint4 &b0, const uint4 &b1, uint32_t seed) -> key_t {
    constexpr int TEST = decltype(test)::value;
    bool out = false;
    if (TEST == 2) {
      const us2 t = as_us2(uv2) - lo2[qi];
      const us2 m = __builtin_elementwise_min(t, span2);
      out = as_u32(t) != as_u32(m);
    } else if (TEST == 1) {
      out = (uint32_t)((int32_t)(uv2 >> 16) - v_lo[qi]) > (uint32_t)(2 * rv);
    }
    const key_t key = sad_key<KM>(a0[qi], a1[qi], b0, b1, seed);
    return (TEST != 0 && out) ? KNONE : key;
  };
  VH_STAT(0, 1); VH_STAT(5, q1 - q0); VH_STAT(6, UB1 - UB0 + 1);
  walk_region_lds<!SPEC>(s, cbs, cuv, cdesc, c, UB0, UB1, VB0, VB1, ULO_MAX, UHI_MIN, VA0, VA1, wD, wU,
    [&](int32_t pc, int32_t p1, int32_t pa0, int32_t pa1, const uint4 *cD, const uint32_t *cU) {
      VH_STAT(1, 1);
      const int32_t mcnt = min(64, p1 - pc);
      const int32_t jend = (mcnt + TRIP - 1) & ~(TRIP - 1);  // <= 64: trailing slots hold copies of the last candidate
      const uint4 *rd = cD + ph;                              // this phase's slot of step 0 (second half: +64)
      const uint32_t *ru = cU + ph;
      uint32_t seedA = (uint32_t)(pc - pbase + ph), seedB = seedA + P;
      int32_t j = 0;
      // one trip: candidates (j + ph) and (j + P + ph) at slot offset `o` of this phase's read pointers
      auto trip = [&](auto test, int32_t o) {
        const uint4 dA0 = rd[o], dA1 = rd[o + 64], dB0 = rd[o + P], dB1 = rd[o + 64 + P];
        uint32_t uA = 0, uB = 0;
        if (decltype(test)::value != 0) { uA = ru[o]; uB = ru[o + P]; }
#pragma unroll
        for (int32_t qi = 0; qi < Q; qi++) {
          const key_t kA = make_key(test, qi, uA, dA0, dA1, seedA + (uint32_t)o), kB = make_key(test, qi, uB, dB0, dB1, seedB + (uint32_t)o);
          best_key[qi] = min(min(kA, kB), best_key[qi]);
        }
      };
      auto run = [&](auto test, int32_t jstop) {
        if (jstop > j) VH_STAT(2 + decltype(test)::value, (jstop - j) / TRIP);
        // two trips per loop iteration: the pointer and seed updates (and the loop control) once per 16 candidates
        for (; j + 2 * TRIP <= jstop; j += 2 * TRIP) {
          trip(test, 0); trip(test, TRIP);
          rd += 2 * TRIP; ru += 2 * TRIP; seedA += 2 * TRIP; seedB += 2 * TRIP;
        }
        if (j < jstop) {
          trip(test, 0);
          j += TRIP; rd += TRIP; ru += TRIP; seedA += TRIP; seedB += TRIP;
        }
      };
      if (SPEC) {
        run(std::integral_constant<int, 0>{}, jend);
      } else if (pa0 <= pa1) {
        // interior column.  [0, ja): v test, [ja, jb): no test, [jb, jend): v test -- the
        // untested range shrunk inward to whole trips (testing a candidate that would
        // not need it is always valid)
        const int32_t ja = min((min(max(pa0 - pc, 0), mcnt) + TRIP - 1) & ~(TRIP - 1), jend);
        const int32_t jb = max(min(max(pa1 - pc, 0), mcnt) & ~(TRIP - 1), ja);
        run(std::integral_constant<int, 1>{}, ja);
        run(std::integral_constant<int, 0>{}, jb);
        run(std::integral_constant<int, 1>{}, jend);
      } else {
        run(std::integral_constant<int, 2>{}, jend);
      }
    });
  // join the phases: lanes l, l+L, l+2L, .. hold partial minima of the same query
#pragma unroll
  for (int32_t qi = 0; qi < Q; qi++) {
    uint64_t k = key_to_64<KM>(best_key[qi]);
#pragma unroll
    for (int32_t d = L; d < 64; d <<= 1) k = min(k, shfl_xor_u64(k, d));
    out.k[qi] = k; out.a0[qi] = a0[qi]; out.a1[qi] = a1[qi]; out.uv1[qi] = uv1[qi]; out.valid[qi] = valid[qi];
    out.qpos[qi] = qp[qi];
  }
}

template <bool SPEC>
__device__ __forceinline__ void flow_pass(const VhSets &s, const VhMatchArgs &a, int32_t pass, int32_t stream, int32_t qset,
                                          int32_t cset, uint4 *wD, uint32_t *wU, int32_t *__restrict__ best, int32_t *__restrict__ redo_count) {
  const int32_t ntile = s.tile_cnt[qset];
  for (int32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); tile < ntile;
       tile += gridDim.x * 4) {
    const int4 t = s.tiles[(int64_t)qset * s.max_tiles + tile];
    const int32_t q0 = __builtin_amdgcn_readfirstlane(t.x), q1 = __builtin_amdgcn_readfirstlane(t.y);
    const int32_t c = __builtin_amdgcn_readfirstlane(t.z), col0 = __builtin_amdgcn_readfirstlane(t.w);
    // candidates of class c occupy the contiguous positions [pbase, pend) of the bin order
    const int32_t *cbs = s.bin_start + (int64_t)cset * (s.nbins + 1);
    const int32_t pbase = __builtin_amdgcn_readfirstlane(cbs[c * s.ubn * s.vbn]);
    const int32_t pend = __builtin_amdgcn_readfirstlane(cbs[(c + 1) * s.ubn * s.vbn]);
    const int32_t km = key_mode_of(pend - pbase, a.wide_keys);
    TileOut<VH_FLOW_Q> out;
    if (km == KEY_HI16) flow_tile<VH_FLOW_Q, VH_FLOW_P, KEY_HI16, SPEC>(s, a, pass, stream, qset, cset, q0, q1, c, col0, pbase, pend - pbase, wD, wU, out);
    else if (km == KEY_W19) flow_tile<VH_FLOW_Q, VH_FLOW_P, KEY_W19, SPEC>(s, a, pass, stream, qset, cset, q0, q1, c, col0, pbase, pend - pbase, wD, wU, out);
    else flow_tile<VH_FLOW_Q, VH_FLOW_P, KEY_64, SPEC>(s, a, pass, stream, qset, cset, q0, q1, c, col0, pbase, pend - pbase, wD, wU, out);
    finish_tile<VH_FLOW_Q, VH_FLOW_P, SPEC, true>(s, a, pass, stream, qset, cset, c, pbase, pend - pbase, out, best, redo_count);
  }
}

// -------------------------------------------------------------- match (stereo)
// The 1-d stereo search (v window of +-match_disp_tolerance, stock libviso2;
// SURVEY App. A.7) accepts only a few dozen candidates per query, all within a
// handful of image rows, so walking whole 50x50 bins would waste >95 % of the
// work.  It runs over the (class, v) ROW index on both sides instead: a tile is
// T = 64*Q/P consecutive row-ordered queries of one class (32 queries span three
// or four image rows), the candidate stream is the single contiguous row range
// [vmin-tol, vmax+tol] of the candidate set, shared by P phases of lanes exactly
// as in the flow search (flow_tile), Q queries per lane.  The search is
// speculative in the same way: no accept test in the loop (the stream holds every
// u of those rows, the window only +-match_radius of them), the winner is tested
// once, and a query whose winner lies outside its window is searched again by the
// whole wave over its own bins (flow_query_by_wave).  The key carries the
// candidate's BIN-order position (staged next to its descriptor), so the minimum
// is findMatch's first minimum in (u_bin, v_bin, list) order no matter in which
// order the rows are walked.  Round 1 used 64-query tiles with the test in the
// loop: 100 evaluated candidates and 13 instructions per query and candidate
// where ~15 candidates are inside the window; this form evaluates ~70 at 9.4.
template <int Q, int P, int KM, bool SPEC>
__device__ __forceinline__ void rows_tile(const VhSets &s, const VhMatchArgs &a, int32_t pass, int32_t stream, int32_t qset,
                                          int32_t cset, int32_t q0, int32_t q1, int32_t c, int32_t pbase, int32_t pcnt, uint4 *wD,
                                          uint32_t *wU, uint32_t *wV, TileOut<Q> &out) {
  constexpr int L = 64 / P;
  static_assert(L == 8 || L == 16, "every 16-lane row must hold the whole tile");
  constexpr int TRIP = 2 * P;
  const int32_t lane = threadIdx.x & 63, ph = lane / L, l = lane % L;
  const int32_t nrow = 4 * s.H;
  const int32_t *__restrict__ crs = s.row_start + (int64_t)cset * (nrow + 1);
  const int32_t *__restrict__ qpos = s.r_pos + (int64_t)qset * s.cap;
  const uint32_t *__restrict__ quv = s.s_uv + (int64_t)qset * s.cap;
  const uint4 *__restrict__ qdesc = (const uint4 *)(s.s_desc + (int64_t)qset * s.cap * 8);
  const int32_t *__restrict__ cpos = s.r_pos + (int64_t)cset * s.cap;
  const uint32_t *__restrict__ cuv = s.s_uv + (int64_t)cset * s.cap;
  const uint4 *__restrict__ cdesc = (const uint4 *)(s.s_desc + (int64_t)cset * s.cap * 8);

  bool valid[Q];
  uint4 a0[Q], a1[Q];
  uint32_t uv1[Q];
  int32_t qp[Q];
  typedef typename KeyT<KM>::type key_t;
  key_t best_key[Q];
  us2 lo2[Q];
  const us2 span2 = {(unsigned short)(2 * a.radius), (unsigned short)(2 * a.disp_tol)};
  int32_t vmin = 0x7FFFFFFF, vmax = -1;
#pragma unroll
  for (int32_t qi = 0; qi < Q; qi++) {
    const int32_t q = q0 + L * qi + l;
    valid[qi] = q < q1;
    qp[qi] = qpos[valid[qi] ? q : q0];  // row order holds bin positions; the records are gathered from the bin-ordered arrays
    VH_CHECK_RANGE(s, 1, qp[qi], 0, s.cap);
    uv1[qi] = quv[qp[qi]];
    a0[qi] = qdesc[2 * (int64_t)qp[qi]]; a1[qi] = qdesc[2 * (int64_t)qp[qi] + 1];
    const int32_t u1 = uv1[qi] & 0xFFFF, v1 = uv1[qi] >> 16;
    lo2[qi] = us2{(unsigned short)(u1 - a.radius), (unsigned short)(v1 - a.disp_tol)};  // accept test as in flow_tile
    best_key[qi] = (key_t)~(key_t)0;
    vmin = min(vmin, v1); vmax = max(vmax, v1);
  }
  vmin = __builtin_amdgcn_readfirstlane(row16_allreduce<false>(vmin));
  vmax = __builtin_amdgcn_readfirstlane(row16_allreduce<true>(vmax));
  VH_STAT(8, 1); VH_STAT(11, q1 - q0);
  const int32_t VLO = max(vmin - a.disp_tol, 0), VHI = min(vmax + a.disp_tol, s.H - 1);
  const int32_t r0 = __builtin_amdgcn_readfirstlane(crs[c * s.H + VLO]);
  const int32_t r1 = __builtin_amdgcn_readfirstlane(crs[c * s.H + VHI + 1]);
  for (int32_t rc = r0; rc < r1; rc += 64) {
    const int32_t mcnt = min(64, r1 - rc);
    // slots past the end repeat the last candidate: same key, harmless.  Every slot of the row index in
    // [row_start[0], row_start[4H]) holds a position of the row's own class (bin_sort writes each exactly once).
    int32_t cp = cpos[min(rc + lane, r1 - 1)];
    VH_CHECK_RANGE(s, 2, cp, pbase, pbase + pcnt);
    const uint4 g0 = cdesc[2 * (int64_t)cp], g1 = cdesc[2 * (int64_t)cp + 1];
    uint32_t gu = 0;
    if (!SPEC) gu = cuv[cp];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous chunk fully consumed
    wD[lane] = g0; wD[64 + lane] = g1; wU[lane] = (uint32_t)(cp - pbase);
    if (!SPEC) wV[lane] = gu;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // chunk visible to every lane of the wave
    const int32_t jend = (mcnt + TRIP - 1) & ~(TRIP - 1);
    VH_STAT(9, jend / TRIP);
    const uint4 *rd = wD + ph;
    const uint32_t *ru = wU + ph, *rv = wV + ph;
    for (int32_t j = 0; j < jend; j += TRIP) {
      const uint4 dA0 = rd[0], dA1 = rd[64], dB0 = rd[P], dB1 = rd[64 + P];
      const uint32_t sA = ru[0], sB = ru[P];
      uint32_t uA = 0, uB = 0;
      if (!SPEC) { uA = rv[0]; uB = rv[P]; }
#pragma unroll
      for (int32_t qi = 0; qi < Q; qi++) {
        key_t kA = sad_key<KM>(a0[qi], a1[qi], dA0, dA1, sA), kB = sad_key<KM>(a0[qi], a1[qi], dB0, dB1, sB);
        if (!SPEC) {  // the literal accept test (matcher.cpp:249) per pair
          const us2 tA = as_us2(uA) - lo2[qi], tB = as_us2(uB) - lo2[qi];
          kA = as_u32(tA) != as_u32(__builtin_elementwise_min(tA, span2)) ? (key_t)~(key_t)0 : kA;
          kB = as_u32(tB) != as_u32(__builtin_elementwise_min(tB, span2)) ? (key_t)~(key_t)0 : kB;
        }
        best_key[qi] = min(min(kA, kB), best_key[qi]);
      }
      rd += TRIP; ru += TRIP; rv += TRIP;
    }
  }
  // join the phases
#pragma unroll
  for (int32_t qi = 0; qi < Q; qi++) {
    uint64_t k = key_to_64<KM>(best_key[qi]);
#pragma unroll
    for (int32_t d = L; d < 64; d <<= 1) k = min(k, shfl_xor_u64(k, d));
    out.k[qi] = k; out.a0[qi] = a0[qi]; out.a1[qi] = a1[qi]; out.uv1[qi] = uv1[qi]; out.valid[qi] = valid[qi];
    out.qpos[qi] = qp[qi];
  }
}

template <bool SPEC>
__device__ __forceinline__ void rows_pass(const VhSets &s, const VhMatchArgs &a, int32_t pass, int32_t stream, int32_t qset,
                                          int32_t cset, uint4 *wD, uint32_t *wU, uint32_t *wV, int32_t *__restrict__ best, int32_t *__restrict__ redo_count) {
  const int32_t nrow = 4 * s.H;
  const int32_t *__restrict__ qrs = s.row_start + (int64_t)qset * (nrow + 1);
  // tile -> (class, query range): classes are contiguous in row order
  int32_t cls_q0[5];
#pragma unroll
  for (int32_t c = 0; c <= 4; c++) cls_q0[c] = __builtin_amdgcn_readfirstlane(qrs[c * s.H]);
  constexpr int T = VH_TILE_Q;
  for (int32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));; tile += gridDim.x * 4) {
    int32_t c = -1, q0 = 0, q1 = 0, t = tile;
#pragma unroll
    for (int32_t k = 0; k < 4; k++) {
      const int32_t nt = (cls_q0[k + 1] - cls_q0[k] + T - 1) / T;
      if (c < 0 && t < nt) { c = k; q0 = cls_q0[k] + T * t; q1 = min(cls_q0[k + 1], q0 + T); }
      if (c < 0) t -= nt;
    }
    if (c < 0) break;  // past the last tile (uniform)
    const int32_t *cbs = s.bin_start + (int64_t)cset * (s.nbins + 1);
    const int32_t pbase = __builtin_amdgcn_readfirstlane(cbs[c * s.ubn * s.vbn]);
    const int32_t pend = __builtin_amdgcn_readfirstlane(cbs[(c + 1) * s.ubn * s.vbn]);
    const int32_t km = key_mode_of(pend - pbase, a.wide_keys);
    TileOut<VH_FLOW_Q> out;
    if (km == KEY_HI16) rows_tile<VH_FLOW_Q, VH_FLOW_P, KEY_HI16, SPEC>(s, a, pass, stream, qset, cset, q0, q1, c, pbase, pend - pbase, wD, wU, wV, out);
    else if (km == KEY_W19) rows_tile<VH_FLOW_Q, VH_FLOW_P, KEY_W19, SPEC>(s, a, pass, stream, qset, cset, q0, q1, c, pbase, pend - pbase, wD, wU, wV, out);
    else rows_tile<VH_FLOW_Q, VH_FLOW_P, KEY_64, SPEC>(s, a, pass, stream, qset, cset, q0, q1, c, pbase, pend - pbase, wD, wU, wV, out);
    finish_tile<VH_FLOW_Q, VH_FLOW_P, SPEC, false>(s, a, pass, stream, qset, cset, c, pbase, pend - pbase, out, best, redo_count);
  }
}

// Every pass of a matching method in ONE launch (blockIdx.y = pass): the 1-d stereo
// searches are chains of dependent L2 round trips around very little arithmetic
// (~75 candidates per tile), the 2-d flow searches are bound by VALU issue; as
// separate, serialised kernels the former cost as much time as ~half of the latter
// for 3 % of its instructions.  Mixed on the same CUs the flow waves fill the issue
// slots the stereo waves leave idle.  All passes of a method are independent of each
// other (each is a pure function of its two feature sets).
template <bool SPEC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(VH_MATCH_WAVES, 8)))
match_kernel(VhSets s, VhMatchArgs a, int32_t *__restrict__ best, int32_t *__restrict__ redo) {
  __shared__ uint4 sDesc[4 * 256];   // per wave: 2 buffers of 64 staged candidates, first | second descriptor half
  __shared__ uint32_t sAux[4 * 128]; // per wave: 2 x their u | v << 16 (tested flow loop) or key seeds (stereo)
  __shared__ uint32_t sAux2[SPEC ? 1 : 4 * 64];  // per wave: u | v << 16 (tested stereo loop)
  const int32_t pass = blockIdx.y, stream = blockIdx.z;
  if (a.prior && pass == 1) return;  // (searched per driving feature, with its prediction: kernels_prior.hip)
#ifdef VH_EXP_SKIP  // timing-only builds (tools/ab_bench.sh): 1 = no stereo passes, 2 = no flow passes; results are wrong
  if (VH_EXP_SKIP & (a.pass[pass].flow ? 2 : 1)) return;
#endif
  const int32_t qset = vh_role_set(a.S, a.pair_cur, stream, a.pass[pass].qset);
  const int32_t cset = vh_role_set(a.S, a.pair_cur, stream, a.pass[pass].cset);
  uint4 *wD = sDesc + (threadIdx.x >> 6) * 256;
  uint32_t *wU = sAux + (threadIdx.x >> 6) * 128;
  uint32_t *wV = sAux2 + (SPEC ? 0 : (threadIdx.x >> 6) * 64);
  // queries searched again by this stream's speculative passes: read by the host (with a lag) to
  // choose between the speculative and the tested loops (engine.hip: match policy)
  int32_t *redo_count = redo + stream;
  if (a.pass[pass].flow) flow_pass<SPEC>(s, a, pass, stream, qset, cset, wD, wU, best, redo_count);
  else rows_pass<SPEC>(s, a, pass, stream, qset, cset, wD, wU, wV, best, redo_count);
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
  const int32_t nq = indexed_count(s, qset);
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
// for its pixel with atomicMax(epoch << 24 | (2^24 - 1 - i1c)); the lowest i1c of
// this epoch wins, and no clearing between frames is needed.
__global__ void chain_kernel(VhSets s, VhMatchArgs a, int32_t method, const int32_t *__restrict__ best,
                             int4 *__restrict__ chain, uint32_t *__restrict__ mask, uint32_t epoch,
                             int32_t *__restrict__ mchunk, int32_t nchm) {
  const int32_t stream = blockIdx.y;
  const int32_t set1p = vh_role_set(a.S, a.pair_cur, stream, 0), set2p = vh_role_set(a.S, a.pair_cur, stream, 1);
  const int32_t set1c = vh_role_set(a.S, a.pair_cur, stream, 2), set2c = vh_role_set(a.S, a.pair_cur, stream, 3);
  const int32_t n1p = indexed_count(s, set1p), n2p = indexed_count(s, set2p);
  const int32_t n1c = indexed_count(s, set1c), n2c = indexed_count(s, set2c);
  const int32_t *__restrict__ T = best + (int64_t)stream * 4 * s.cap;
  const int64_t cap = s.cap;
  // coordinates in reference order, 4 B per feature (the 48-byte records would cost a 64-byte sector per look-up)
  const uint32_t *__restrict__ uv1p = s.f_uv + (int64_t)set1p * cap, *__restrict__ uv2p = s.f_uv + (int64_t)set2p * cap;
  const uint32_t *__restrict__ uv1c = s.f_uv + (int64_t)set1c * cap, *__restrict__ uv2c = s.f_uv + (int64_t)set2c * cap;
  int4 *__restrict__ out = chain + 2 * (int64_t)stream * s.cap;
  const int32_t ndrive = (method == 2) ? n1p : n1c;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ndrive; i += gridDim.x * blockDim.x) {
    int4 r = make_int4(-1, -1, -2, -1), c = make_int4(0, 0, 0, 0);
    if (method == 0) {
      if (n1p > 0) {
        const int32_t i1p = T[0 * cap + i];
        const int32_t i1c2 = T[1 * cap + i1p];
        if (i1c2 == i) {
          r = make_int4(i1p, -1, i, -1);
          c.x = (int32_t)uv1p[i1p]; c.z = (int32_t)uv1c[i];
          atomicMax(&mask[(int64_t)stream * s.W * s.H + (int64_t)((uint32_t)c.z >> 16) * s.W + ((uint32_t)c.z & 0xFFFFu)],
                    (epoch << VH_MASK_IDX_BITS) | (((1u << VH_MASK_IDX_BITS) - 1u) - (uint32_t)i));
        }
      }
    } else if (method == 1) {
      if (n2c > 0) {
        const int32_t i2c = T[0 * cap + i];
        const int32_t i1c2 = T[1 * cap + i2c];
        c.z = (int32_t)uv1c[i]; c.w = (int32_t)uv2c[i2c];
        if (i1c2 == i && ((uint32_t)c.z & 0xFFFFu) >= ((uint32_t)c.w & 0xFFFFu)) r = make_int4(-1, -1, i, i2c);
      }
    } else {
      if (n2p > 0 && n1c > 0 && n2c > 0) {
        const int32_t i2p = T[0 * cap + i];
        const int32_t i2c = T[1 * cap + (a.prior ? i : i2p)];
        const int32_t i1c = T[2 * cap + i2c];
        const int32_t i1p2 = T[3 * cap + i1c];
        c = make_int4((int32_t)uv1p[i], (int32_t)uv2p[i2p], (int32_t)uv1c[i1c], (int32_t)uv2c[i2c]);
        const uint32_t u1p = (uint32_t)c.x & 0xFFFFu, u2p = (uint32_t)c.y & 0xFFFFu, u1c = (uint32_t)c.z & 0xFFFFu, u2c = (uint32_t)c.w & 0xFFFFu;
        if (i1p2 == i && u1p >= u2p && u1c >= u2c) r = make_int4(i, i2p, i1c, i2c);
      }
    }
    out[2 * (int64_t)i] = r; out[2 * (int64_t)i + 1] = c;
    if (method != 0) count_chunk(r.z >= 0, mchunk + stream * nchm + (i >> 8));
  }
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
  const int32_t n1c = indexed_count(s, set1c);
  int4 *__restrict__ ch = chain + 2 * (int64_t)stream * s.cap;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n1c; i += gridDim.x * blockDim.x) {
    const int4 r = ch[2 * (int64_t)i];
    const uint32_t uv = (uint32_t)ch[2 * (int64_t)i + 1].z;
    const bool win = r.z >= 0 && mask[(int64_t)stream * s.W * s.H + (int64_t)(uv >> 16) * s.W + (uv & 0xFFFFu)] ==
                                     ((epoch << VH_MASK_IDX_BITS) | (((1u << VH_MASK_IDX_BITS) - 1u) - (uint32_t)i));
    if (r.z >= 0 && !win) ch[2 * (int64_t)i].z = -2;
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
                    int32_t *__restrict__ overflow, const int32_t *__restrict__ mchunk, int32_t nchm,
                    int32_t *__restrict__ redo, int32_t *__restrict__ mchunk_next, int4 *__restrict__ host_out,
                    float *__restrict__ host_matches) {
  __shared__ int32_t sWave[4];
  __shared__ int32_t sBase;
  const int32_t chunk = blockIdx.x, stream = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // the chunk counters of the NEXT launch (the other buffer; its last reader, the emission before this one, is done)
  // are zeroed here instead of by a memset of their own: two fill kernels and a launch gap per step
  if (tid == 0) mchunk_next[stream * nchm + chunk] = 0;
  int32_t sets[4];
#pragma unroll
  for (int32_t r = 0; r < 4; r++) sets[r] = vh_role_set(a.S, a.pair_cur, stream, r);
  const int32_t drive = (method == 2) ? sets[0] : sets[2];
  const int32_t n = indexed_count(s, drive);
  if (chunk * 256 >= n && chunk != nchm - 1) return;
  const int4 *__restrict__ ch = chain + 2 * (int64_t)stream * s.cap;
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
  int4 r = make_int4(-1, -1, -2, -1), c = make_int4(0, 0, 0, 0);
  if (i < n) { r = ch[2 * (int64_t)i]; c = ch[2 * (int64_t)i + 1]; }
  const bool keep = r.z >= 0;
  uint32_t rec[12];
#pragma unroll
  for (int32_t k = 0; k < 12; k++) rec[k] = (k % 3 == 2) ? 0xFFFFFFFFu : __float_as_uint(-1.0f);
  if (keep) {
    const int32_t idx[4] = {r.x, r.y, r.z, r.w};
    const uint32_t uv[4] = {(uint32_t)c.x, (uint32_t)c.y, (uint32_t)c.z, (uint32_t)c.w};
#pragma unroll
    for (int32_t k = 0; k < 4; k++) {
      if (idx[k] >= 0) {
        rec[3 * k + 0] = __float_as_uint((float)(uv[k] & 0xFFFFu));
        rec[3 * k + 1] = __float_as_uint((float)(uv[k] >> 16));
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
    if (host_matches) {  // small groups: the records also go straight to host-mapped memory (no download before getMatches)
      uint4 *h = (uint4 *)(host_matches + ((int64_t)stream * mcap + pos) * 12);
      h[0] = o[0]; h[1] = o[1]; h[2] = o[2];
    }
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
    // statistics of this launch for the host's loop policy: queries searched again / queries searched
    int32_t nq = 0;
#pragma unroll
    for (int32_t k = 0; k < 4; k++) if (k < a.npass) nq += indexed_count(s, vh_role_set(a.S, a.pair_cur, stream, a.pass[k].qset));
    // count, overflow flag and the launch's statistics also go straight to host-mapped memory: the host reads
    // them after the launch's event instead of through small device->host copies (each a blit kernel + a round trip)
    host_out[stream] = make_int4(base + tot, ov, redo[stream], nq);
    redo[stream] = 0;
  }
}

}  // namespace

void vh_launch_match_prior(const VhSets &s, const VhMatchArgs &a, double u_, double v_, int32_t *best,
                           hipStream_t st) {
  hipLaunchKernelGGL(match_prior_kernel, dim3((s.cap + 127) / 128), dim3(128), 0, st, s, a, u_, v_, best);
}
void vh_launch_match(const VhSets &s, const VhMatchArgs &a, int32_t *best, int32_t *redo, int32_t speculative, int32_t grid_x,
                     int32_t lds_bytes, hipStream_t st) {
  if (!a.npass) return;
  VhMatchArgs m = a;
  static const int wide = [] { const char *e = getenv("VH_FLOW_WIDE_KEYS"); return e ? atoi(e) : 0; }();
  m.wide_keys = wide;
  // Grid and occupancy.  The kernel loops over the tile list, so any grid is correct; what the choice decides is how the
  // searches share the chip with the detection chain of the next frame (round 5, MI355X, KITTI, S = 256, k pairs/s):
  //  * DEFAULT (no hint): one workgroup per 4 tiles of the capacity-sized tile list, width made odd (a multiple of 8 pins
  //    tile k of every row to one XCD: 128 -> 85.2, 129 -> 95.6 in round 2).  At typical densities 3 of 4 of these
  //    workgroups find no tile and leave at once; their churn is what lets the detection chain's workgroups in: 107.1-107.7.
  //  * TIGHT (grid_x = the tiles the sets really hold / 4, from the statistics of an earlier launch) alone is worse: the
  //    searches then hold all 7 wave slots per SIMD for their whole life, finish in 1 685 instead of 2 305 us, and the
  //    detection chain runs on after them on an empty chip: 100.3-100.5.
  //  * TIGHT + lds_bytes = 26 880 (21 of a CU's 128 LDS allocation units of 1 280 bytes: at most SIX search workgroups
  //    per CU, one wave slot per SIMD and 2 units left for everybody else): **110.5-111.8**; 20 units (153 600 in six
  //    workgroups) 104.2-105.2, 22 units (five workgroups) 104.8-105.1 -- the window is one allocation unit wide, because
  //    it is the arithmetic of what fits beside what (detect_nms 14 units, emit_features 21), and it moves with those.
  //    Same rule: 1080p 18.5 -> 19.2, KITTI + noise (tested loops) 73.2 -> 74.9, mono flow 140.7 -> 144.7, S = 32 / 64 /
  //    128 streams 93.0 -> 97.8 / 102.7 -> 104.9 / 105.7 -> 105.8.  4K (detect_nms<3>: 22 units, 69 registers) wants FIVE
  //    workgroups instead: 21 / 22 / 23 / 24 / 25 / 26 units = 3.67 / 3.94 / 4.08 / 4.08 / 4.08 / 3.96 against 3.68 -- the
  //    caller picks the units per detector (engine.hip: match_queued).
  // VH_FLOW_WGS / VH_FLOW_LDS_PAD override both (experiments).
  static const int wgs = [] { const char *e = getenv("VH_FLOW_WGS"); return e ? atoi(e) : 0; }();
  static const int pad_env = [] { const char *e = getenv("VH_FLOW_LDS_PAD"); return e ? atoi(e) : -1; }();
  const int32_t gx = wgs > 0 ? wgs : (grid_x > 0 ? (grid_x | 1) : (((s.max_tiles + 3) / 4) | 1));
  dim3 grid(gx, m.npass, a.S);
  static const size_t static_lds[2] = {
      [] { hipFuncAttributes at{}; return hipFuncGetAttributes(&at, (const void *)match_kernel<false>) == hipSuccess ? at.sharedSizeBytes : (size_t)0; }(),
      [] { hipFuncAttributes at{}; return hipFuncGetAttributes(&at, (const void *)match_kernel<true>) == hipSuccess ? at.sharedSizeBytes : (size_t)0; }()};
  const size_t have = static_lds[speculative ? 1 : 0];
  const int32_t pad = pad_env >= 0 ? pad_env : (wgs > 0 || grid_x <= 0 || lds_bytes <= 0 || have == 0 || (size_t)lds_bytes <= have ? 0 : (int32_t)((size_t)lds_bytes - have));
  if (speculative) hipLaunchKernelGGL(match_kernel<true>, grid, dim3(256), pad, st, s, m, best, redo);
  else hipLaunchKernelGGL(match_kernel<false>, grid, dim3(256), pad, st, s, m, best, redo);
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
                            const int32_t *mchunk, int32_t *redo, int32_t *mchunk_next, void *host_out, void *host_matches,
                            hipStream_t st) {
  const int32_t nchm = (s.cap + 255) / 256;
  hipLaunchKernelGGL(emit_matches_kernel, dim3(nchm, a.S), dim3(256), 0, st, s, a, method, chain,
                     (float *)matches, mcap, match_count, overflow, mchunk, nchm, redo, mchunk_next, (int4 *)host_out, (float *)host_matches);
}
