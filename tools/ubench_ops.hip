// Issue-cost table of single VALU instructions on gfx950 (companion of ubench_valu.hip):
// for each opcode, 32 instructions per trip on 8 independent accumulators, W = 1/2/8 waves
// per SIMD, cycles per wave-instruction per SIMD from the wall clock and the measured
// shader clock (s_memtime / s_memrealtime).  2.x = full rate (32 lanes/clk), 4.x = half rate.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench_ops tools/ubench_ops.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(OP) OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define KERNEL(NAME, ASM)                                                                                         \
  __global__ void __launch_bounds__(256) k_##NAME(uint64_t *__restrict__ stamps, uint32_t *__restrict__ sink,     \
                                                  const uint32_t *__restrict__ in, int iters) {                   \
    const uint32_t t = threadIdx.x + blockIdx.x * 256;                                                            \
    uint32_t a = in[t & 1023], b = a * 2654435761u;                                                               \
    uint32_t x0 = a ^ 1, x1 = a ^ 2, x2 = a ^ 3, x3 = a ^ 4, x4 = a ^ 5, x5 = a ^ 6, x6 = a ^ 7, x7 = a ^ 8;       \
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a), "v"(b) : "vcc");                                            \
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();                                                         \
    const uint64_t c0 = __builtin_amdgcn_s_memtime();                                                             \
    for (int i = 0; i < iters; i++) {                                                                             \
      REP32(ASM)                                                                                                  \
    }                                                                                                             \
    const uint64_t c1 = __builtin_amdgcn_s_memtime();                                                             \
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();                                                         \
    if ((threadIdx.x & 63) == 0) {                                                                                \
      const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);                                               \
      stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = r1 - r0;                                                       \
    }                                                                                                             \
    sink[t] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                                              \
  }

#define A_ADD(X) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
#define A_SUB(X) asm volatile("v_sub_u32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
#define A_OR(X) asm volatile("v_or_b32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
#define A_XOR(X) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
#define A_MAXU(X) asm volatile("v_max_u32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
#define A_MINI(X) asm volatile("v_min_i32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
#define A_LSHL(X) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(X));
#define A_LSHR(X) asm volatile("v_lshrrev_b32_e32 %0, 1, %0" : "+v"(X));
#define A_ASHR(X) asm volatile("v_ashrrev_i32_e32 %0, 1, %0" : "+v"(X));
#define A_MOV(X) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(X) : "v"(a));
#define A_CNDMASK(X) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(X) : "v"(a));
#define A_CMP(X) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" ::"v"(X), "v"(a) : "vcc");
#define A_CMP64(X) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1" ::"v"(X), "v"(a) : "s20", "s21");
#define A_CMPX(X) asm volatile("v_cmp_ne_u32_e32 vcc, %0, %1" ::"v"(X), "v"(a) : "vcc");
#define A_ADD3(X) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(X) : "v"(a), "v"(b));
#define A_ANDOR(X) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(X) : "v"(a), "v"(b));
#define A_OR3(X) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(X) : "v"(a), "v"(b));
#define A_BFE(X) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(X));
#define A_BFI(X) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(X) : "v"(a), "v"(b));
#define A_MAD24(X) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(X) : "v"(a), "v"(b));
#define A_MUL24(X) asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(X) : "v"(a));
#define A_ADDCO(X) asm volatile("v_add_co_u32_e32 %0, vcc, %1, %0" : "+v"(X) : "v"(a) : "vcc");
#define A_ADDC(X) asm volatile("v_addc_co_u32_e32 %0, vcc, %1, %0, vcc" : "+v"(X) : "v"(a) : "vcc");
#define A_SUBCO(X) asm volatile("v_sub_co_u32_e32 %0, vcc, %1, %0" : "+v"(X) : "v"(a) : "vcc");
#define A_ADDF(X) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
#define A_MAXF(X) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(X) : "v"(a));
#define A_MAXF_ABS(X) asm volatile("v_max_f32_e64 %0, |%1|, |%0|" : "+v"(X) : "v"(a));
#define A_SUBF_ABS(X) asm volatile("v_sub_f32_e64 %0, |%1|, %0" : "+v"(X) : "v"(a));
#define A_CMPF(X) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" ::"v"(X), "v"(a) : "vcc");
#define A_MED3(X) asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(X) : "v"(a), "v"(b));
#define A_MAX3(X) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(X) : "v"(a), "v"(b));
#define A_PKADD(X) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(X) : "v"(a));
#define A_PKSUBCL(X) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(X) : "v"(a));
#define A_PKMAX(X) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(X) : "v"(a));
#define A_ALIGNBIT(X) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(X) : "v"(a));
#define A_LSHLADD(X) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(X) : "v"(a));
#define A_XAD(X) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(X) : "v"(a), "v"(b));
#define A_SADU32(X) asm volatile("v_sad_u32 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
#define A_MSAD(X) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
#define A_SADLIT(X) asm volatile("v_sad_u8 %0, %1, %2, 0" : "=v"(X) : "v"(a), "v"(b));
#define A_DOT4I(X) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
#define A_DOT8(X) asm volatile("v_dot8_u32_u4 %0, %1, %2, %0" : "+v"(X) : "v"(a), "v"(b));
#define A_READLANE(X) asm volatile("v_readlane_b32 s20, %0, 3" ::"v"(X) : "s20");
#define A_DPPMOV(X) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(X) : "v"(a));
#define A_DPPADD(X) asm volatile("v_add_u32_dpp %0, %1, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(X) : "v"(a));
#define A_DPPMIN(X) asm volatile("v_min_u32_dpp %0, %1, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(X) : "v"(a));
#define A_SDWA(X) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(X) : "v"(a));
#define A_SUBSDWA(X) asm volatile("v_sub_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0" : "+v"(X) : "v"(a));
#define A_CMPSDWA(X) asm volatile("v_cmp_lt_u32_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:WORD_0" ::"v"(X), "v"(a) : "vcc");

#define LIST(F)                                                                                                        \
  F(add_u32, A_ADD) F(sub_u32, A_SUB) F(or_b32, A_OR) F(xor_b32, A_XOR) F(max_u32, A_MAXU) F(min_i32, A_MINI)          \
  F(lshlrev_b32, A_LSHL) F(lshrrev_b32, A_LSHR) F(ashrrev_i32, A_ASHR) F(mov_b32, A_MOV) F(cndmask_vcc, A_CNDMASK)      \
  F(cmp_lt_u32_vcc, A_CMP) F(cmp_lt_u32_e64_sgpr, A_CMP64) F(cmp_ne_u32_vcc, A_CMPX) F(add3_u32, A_ADD3)               \
  F(and_or_b32, A_ANDOR) F(or3_b32, A_OR3) F(bfe_u32, A_BFE) F(bfi_b32, A_BFI) F(mad_u32_u24, A_MAD24)                  \
  F(mul_u32_u24, A_MUL24) F(add_co_u32, A_ADDCO) F(addc_co_u32, A_ADDC) F(sub_co_u32, A_SUBCO) F(add_f32, A_ADDF)       \
  F(max_f32, A_MAXF) F(max_f32_abs_e64, A_MAXF_ABS) F(sub_f32_abs_e64, A_SUBF_ABS) F(cmp_gt_f32_vcc, A_CMPF)            \
  F(med3_u32, A_MED3) F(max3_u32, A_MAX3) F(pk_add_u16, A_PKADD) F(pk_sub_u16_clamp, A_PKSUBCL) F(pk_max_u16, A_PKMAX)  \
  F(alignbit_b32, A_ALIGNBIT) F(lshl_add_u32, A_LSHLADD) F(xad_u32, A_XAD) F(sad_u32, A_SADU32) F(msad_u8, A_MSAD)      \
  F(sad_u8_acc0, A_SADLIT) F(dot4_i32_i8, A_DOT4I) F(dot8_u32_u4, A_DOT8) F(readlane_b32, A_READLANE)                   \
  F(mov_b32_dpp, A_DPPMOV) F(add_u32_dpp, A_DPPADD) F(min_u32_dpp, A_DPPMIN) F(add_u32_sdwa, A_SDWA)                    \
  F(sub_u32_sdwa, A_SUBSDWA) F(cmp_lt_u32_sdwa, A_CMPSDWA)

#define DEF(NAME, ASM) KERNEL(NAME, ASM)
LIST(DEF)

typedef void (*kfn)(uint64_t *, uint32_t *, const uint32_t *, int);
struct Entry { const char *name; kfn fn; };
#define ENT(NAME, ASM) {#NAME, k_##NAME},
static Entry kTable[] = {LIST(ENT)};

int main() {
  uint64_t *stamps; uint32_t *sink, *in;
  (void)hipMalloc(&stamps, 8 * 2 * 4 * 256 * 8); (void)hipMalloc(&sink, 4 * 256 * 256 * 8); (void)hipMalloc(&in, 4096);
  std::vector<uint32_t> h(1024); for (int i = 0; i < 1024; i++) h[i] = i * 2654435761u + 12345u;
  (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  for (int r = 0; r < 100; r++) hipLaunchKernelGGL(kTable[0].fn, dim3(2048), dim3(256), 0, 0, stamps, sink, in, 1024);
  (void)hipDeviceSynchronize();
  printf("%-22s  cycles per wave-instruction per SIMD (wall clock x measured shader clock), W waves per SIMD\n", "opcode");
  for (auto &e : kTable) {
    printf("%-22s", e.name);
    for (int W : {1, 2, 8}) {
      const int blocks = 256 * W, waves = blocks * 4, iters = 8192 / W;
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, stamps, sink, in, iters / 8 + 1);
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, stamps, sink, in, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<uint64_t> s(2 * (size_t)waves);
      (void)hipMemcpy(s.data(), stamps, s.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> ghz(waves), cyc(waves);
      for (int w = 0; w < waves; w++) { ghz[w] = (double)s[2 * w] / (double)s[2 * w + 1] * 0.1; cyc[w] = (double)s[2 * w] / (iters * 32.0); }
      std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
      const double clk = ghz[waves / 2];
      const double per = (ms * 1e-3 * clk * 1e9) / ((double)waves * iters * 32.0 / 1024.0);
      printf("   W=%d %5.2f (wave %5.2f, %.2f GHz)", W, per, cyc[waves / 2] / W, clk);
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    printf("\n");
  }
  return 0;
}
