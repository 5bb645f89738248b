// VALU issue-rate microbenchmark for gfx950 (round 2): settles whether a wave64
// vector instruction occupies a SIMD for 2 or for 4 shader cycles when several
// waves are resident, for the instruction forms of the match inner loop.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip ; run on the GPU box.
//
// Method: every wave runs ITERS trips of 32 instructions of ONE kind on 8
// INDEPENDENT accumulators (inline asm, so that the form measured is the form
// named and nothing fuses), brackets the loop with s_memtime (shader clock) and
// s_memrealtime (100 MHz), and stores both.  The grid is 256 CUs x W workgroups
// of 4 waves: W waves per SIMD when the dispatcher spreads them evenly (HW_ID is
// recorded to check that).  Reported per mode and W:
//   cyc/instr/wave  = median over waves of (dt_shader / instructions)
//   cyc/instr/SIMD  = that / W            (the SIMD's issue cost per wave-instruction)
//   GHz             = dt_shader / dt_real * 0.1
//   wall rate       = all wave-instructions / hipEvent time / 1024 SIMDs, in shader cycles
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define OPS_PER_TRIP 32

enum Mode { ADD_VOP2 = 0, ADD_VOP3, SAD_U8, SAD_HI_U8, SAD_U8_SGPR, MIN_U32, MIN3_U32, PK_SUB_U16, PK_MIN_U16,
            DOT4, FMA_F32, PK_FMA_F32, CNDMASK, LSHL_OR, AND_B32, PERM_B32, SAD_U16, MQSAD, LOOP_BODY, SAD_LDS, NMODES };
static const char *kNames[NMODES] = {"v_add_u32 (VOP2)", "v_add_u32_e64 (VOP3)", "v_sad_u8 vgpr", "v_sad_hi_u8 vgpr", "v_sad_u8 sgpr src1",
                                     "v_min_u32 (VOP2)", "v_min3_u32", "v_pk_sub_u16", "v_pk_min_u16", "v_dot4_u32_u8", "v_fma_f32",
                                     "v_pk_fma_f32", "v_cndmask_b32 vcc", "v_lshl_or_b32", "v_and_b32 (VOP2)", "v_perm_b32", "v_sad_u16",
                                     "v_qsad_pk_u16_u8", "match body 8sad+3acc+min3/2", "8 v_sad_u8 + 2.25 ds_read bcast"};

#define REP8(OP)                                                                                                     \
  OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

template <int MODE>
__global__ void __launch_bounds__(256) k(uint64_t *__restrict__ stamps, uint32_t *__restrict__ sink, const uint32_t *__restrict__ in, int iters) {
  __shared__ uint4 sD[4 * 128];
  __shared__ uint32_t sU[4 * 64];
  const uint32_t t = threadIdx.x + blockIdx.x * 256;
  uint32_t a = in[t & 1023], b = a * 2654435761u;
  uint32_t x0 = a ^ 1, x1 = a ^ 2, x2 = a ^ 3, x3 = a ^ 4, x4 = a ^ 5, x5 = a ^ 6, x6 = a ^ 7, x7 = a ^ 8;
  uint32_t sb = __builtin_amdgcn_readfirstlane(in[blockIdx.x & 1023]);
  float fa = 1.0001f, fb = 0.5f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f}, pa = {1.0001f, 0.9999f}, pb = {0.5f, 0.25f};
  uint64_t q0 = x0, q1 = x1, q2 = x2, q3 = x3, qa = ((uint64_t)a << 32) | b;
  for (int i = threadIdx.x; i < 4 * 128; i += 256) sD[i] = make_uint4(a + i, b + i, a ^ i, b ^ i);
  sU[threadIdx.x] = a;
  __syncthreads();
  const uint4 *wD = sD + (threadIdx.x >> 6) * 128;
  const uint32_t *wU = sU + (threadIdx.x >> 6) * 64;
  asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a), "v"(b) : "vcc");
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t c0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (MODE == ADD_VOP2) {
#define OP(X) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
      REP32(OP)
#undef OP
    } else if (MODE == ADD_VOP3) {
#define OP(X) asm volatile("v_add_u32_e64 %0, %1, %0" : "+v"(X) : "v"(a));
      REP32(OP)
#undef OP
    } else if (MODE == SAD_U8) {
#define OP(X) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
      REP32(OP)
#undef OP
    } else if (MODE == SAD_HI_U8) {
#define OP(X) asm volatile("v_sad_hi_u8 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
      REP32(OP)
#undef OP
    } else if (MODE == SAD_U8_SGPR) {
#define OP(X) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(X) : "v"(a), "s"(sb));
      REP32(OP)
#undef OP
    } else if (MODE == MIN_U32) {
#define OP(X) asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
      REP32(OP)
#undef OP
    } else if (MODE == MIN3_U32) {
#define OP(X) asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
      REP32(OP)
#undef OP
    } else if (MODE == PK_SUB_U16) {
#define OP(X) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(X) : "v"(a));
      REP32(OP)
#undef OP
    } else if (MODE == PK_MIN_U16) {
#define OP(X) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(X) : "v"(a));
      REP32(OP)
#undef OP
    } else if (MODE == DOT4) {
#define OP(X) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
      REP32(OP)
#undef OP
    } else if (MODE == FMA_F32) {
#define OP(X) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(X) : "v"(fa), "v"(fb));
      REP32(OP)
#undef OP
    } else if (MODE == PK_FMA_F32) {
#define OP(X) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(X) : "v"(pa), "v"(pb));
      OP(p0) OP(p1) OP(p2) OP(p3) OP(p0) OP(p1) OP(p2) OP(p3) OP(p0) OP(p1) OP(p2) OP(p3) OP(p0) OP(p1) OP(p2) OP(p3)
      OP(p0) OP(p1) OP(p2) OP(p3) OP(p0) OP(p1) OP(p2) OP(p3) OP(p0) OP(p1) OP(p2) OP(p3) OP(p0) OP(p1) OP(p2) OP(p3)
#undef OP
    } else if (MODE == CNDMASK) {
#define OP(X) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(X) : "v"(a) : "vcc");
      REP32(OP)
#undef OP
    } else if (MODE == LSHL_OR) {
#define OP(X) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(X) : "v"(a));
      REP32(OP)
#undef OP
    } else if (MODE == AND_B32) {
#define OP(X) asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
      REP32(OP)
#undef OP
    } else if (MODE == PERM_B32) {
#define OP(X) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(X) : "v"(a), "v"(b));
      REP32(OP)
#undef OP
    } else if (MODE == SAD_U16) {
#define OP(X) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
      REP32(OP)
#undef OP
    } else if (MODE == MQSAD) {
#define OP(X) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(X) : "v"(qa), "v"(b));
      OP(q0) OP(q1) OP(q2) OP(q3) OP(q0) OP(q1) OP(q2) OP(q3) OP(q0) OP(q1) OP(q2) OP(q3) OP(q0) OP(q1) OP(q2) OP(q3)
      OP(q0) OP(q1) OP(q2) OP(q3) OP(q0) OP(q1) OP(q2) OP(q3) OP(q0) OP(q1) OP(q2) OP(q3) OP(q0) OP(q1) OP(q2) OP(q3)
#undef OP
    } else if (MODE == LOOP_BODY) {
      // the shape of the shipped flow loop: per candidate pair (A,B): 16 v_sad_hi_u8, 2 x (pk_sub, pk_min, cmp, cndmask), 1 min3
      // = 25 VALU per 2 candidates; here 32 instructions = 2.56 candidates' worth, register operands only
#define SAD(X) asm volatile("v_sad_hi_u8 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
      SAD(x0) SAD(x1) SAD(x0) SAD(x1) SAD(x0) SAD(x1) SAD(x0) SAD(x1) SAD(x0) SAD(x1) SAD(x0) SAD(x1) SAD(x0) SAD(x1) SAD(x0) SAD(x1)
      asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(x2) : "v"(x4), "v"(a));
      asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(x3) : "v"(x2), "v"(b));
      asm volatile("v_cmp_ne_u32 vcc, %0, %1" ::"v"(x2), "v"(x3) : "vcc");
      asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x0) : "v"(a) : "vcc");
      asm volatile("v_pk_sub_u16 %0, %1, %2" : "=v"(x5) : "v"(x4), "v"(b));
      asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(x6) : "v"(x5), "v"(b));
      asm volatile("v_cmp_ne_u32 vcc, %0, %1" ::"v"(x5), "v"(x6) : "vcc");
      asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x1) : "v"(a) : "vcc");
      asm volatile("v_min3_u32 %0, %1, %2, %0" : "+v"(x7) : "v"(x0), "v"(x1));
      SAD(x0) SAD(x1) SAD(x0) SAD(x1) SAD(x0) SAD(x1) SAD(x0)
#undef SAD
    } else if (MODE == SAD_LDS) {
      // 4 candidates per trip as the shipped loop reads them: per candidate one broadcast ds_read_b32 + two ds_read_b128,
      // then 8 v_sad_u8 on the loaded registers (32 VALU per trip)
      const int j = (i & 15) * 4;
#pragma unroll
      for (int kq = 0; kq < 4; kq++) {
        const uint32_t u = wU[j + kq];
        const uint4 d0 = wD[2 * (j + kq)], d1 = wD[2 * (j + kq) + 1];
        asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(d0.x));
        asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(d0.y));
        asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(d0.z));
        asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x0) : "v"(a), "v"(d0.w));
        asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x0) : "v"(b), "v"(d1.x));
        asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x0) : "v"(b), "v"(d1.y));
        asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x0) : "v"(b), "v"(d1.z));
        asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x0) : "v"(u), "v"(d1.w));
      }
    }
  }
  const uint64_t c1 = __builtin_amdgcn_s_memtime();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[4 * w + 0] = c1 - c0; stamps[4 * w + 1] = r1 - r0; stamps[4 * w + 2] = hw; stamps[4 * w + 3] = r0;
  }
  sink[t] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (uint32_t)(p0.x + p1.x + p2.x + p3.x + p0.y) + (uint32_t)(q0 + q1 + q2 + q3) +
            (uint32_t)fa;
}

template <int MODE> void run(uint64_t *stamps, uint32_t *sink, uint32_t *in, int W, int iters) {
  const int blocks = 256 * W, waves = blocks * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, stamps, sink, in, iters / 8 + 1);  // warm-up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, stamps, sink, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(4 * (size_t)waves);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc(waves), ghz(waves);
  std::vector<int> per_simd(1 << 16, 0);
  uint64_t rmin = ~0ull, rmax = 0;
  for (int w = 0; w < waves; w++) {
    cyc[w] = (double)h[4 * w] / ((double)iters * OPS_PER_TRIP);
    ghz[w] = (double)h[4 * w] / (double)h[4 * w + 1] * 0.1;
    const uint32_t hw = (uint32_t)h[4 * w + 2];
    // HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13](+)  -- key on (se,sh,cu,simd) is enough to see the spread
    per_simd[(hw >> 4) & 0xFFFF & ~0xC]++;  // drop the pipe bits
    rmin = std::min(rmin, h[4 * w + 3]); rmax = std::max(rmax, h[4 * w + 3] + h[4 * w + 1]);
  }
  std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
  const double med = cyc[waves / 2], clk = ghz[waves / 2];
  const double total_instr = (double)waves * iters * OPS_PER_TRIP;
  const double wall_cyc_per_instr_simd = (ms * 1e-3 * clk * 1e9) / (total_instr / 1024.0);
  printf("%-30s W=%d  cyc/instr/wave med %6.2f (p10 %6.2f p90 %6.2f)  => /SIMD %5.2f   clock %.3f GHz   wall %.3f ms => %5.2f cyc/instr/SIMD\n",
         kNames[MODE], W, med, cyc[waves / 10], cyc[waves * 9 / 10], med / W, clk, ms, wall_cyc_per_instr_simd);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

template <int MODE> void sweep(uint64_t *stamps, uint32_t *sink, uint32_t *in) {
  for (int W : {1, 2, 4, 8}) run<MODE>(stamps, sink, in, W, 4096 / W);
}

int main() {
  uint64_t *stamps; uint32_t *sink, *in;
  hipMalloc(&stamps, 8 * 4 * 4 * 256 * 8); hipMalloc(&sink, 4 * 256 * 256 * 8); hipMalloc(&in, 4096);
  std::vector<uint32_t> h(1024); for (int i = 0; i < 1024; i++) h[i] = i * 2654435761u + 12345u;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  // heat the chip first so that the clock is the loaded one
  for (int r = 0; r < 50; r++) hipLaunchKernelGGL(k<SAD_U8>, dim3(2048), dim3(256), 0, 0, stamps, sink, in, 2048);
  hipDeviceSynchronize();
  sweep<ADD_VOP2>(stamps, sink, in); sweep<ADD_VOP3>(stamps, sink, in); sweep<SAD_U8>(stamps, sink, in); sweep<SAD_HI_U8>(stamps, sink, in);
  sweep<SAD_U8_SGPR>(stamps, sink, in); sweep<MIN_U32>(stamps, sink, in); sweep<MIN3_U32>(stamps, sink, in); sweep<PK_SUB_U16>(stamps, sink, in);
  sweep<PK_MIN_U16>(stamps, sink, in); sweep<DOT4>(stamps, sink, in); sweep<FMA_F32>(stamps, sink, in); sweep<PK_FMA_F32>(stamps, sink, in);
  sweep<CNDMASK>(stamps, sink, in); sweep<LSHL_OR>(stamps, sink, in); sweep<AND_B32>(stamps, sink, in); sweep<PERM_B32>(stamps, sink, in);
  sweep<SAD_U16>(stamps, sink, in); sweep<MQSAD>(stamps, sink, in); sweep<LOOP_BODY>(stamps, sink, in); sweep<SAD_LDS>(stamps, sink, in);
  return 0;
}
