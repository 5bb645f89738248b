// Instruction-rate microbenchmark for the ops of the match inner loop (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 4096

template <int MODE> __global__ void __launch_bounds__(256) k(uint32_t *out, const uint32_t *in) {
  const uint32_t t = threadIdx.x + blockIdx.x * 256;
  uint32_t a0 = in[t & 1023], a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  uint32_t best = 0xFFFF, bpos = 0, acc = 0, acc2 = 0;
  int32_t ulo = (int32_t)(a0 & 255), uhi = ulo + 400, vlo = (int32_t)(a1 & 255), vhi = vlo + 400;
  uint32_t s = __builtin_amdgcn_readfirstlane(in[blockIdx.x & 1023]);
  for (int i = 0; i < ITERS; i++) {
    s = s * 1664525u + 1013904223u;  // uniform "candidate"
    const uint32_t b0 = s, b1 = s ^ 0x11111111u, b2 = s ^ 0x22222222u, b3 = s ^ 0x33333333u, b4 = s ^ 0x44444444u,
                   b5 = s ^ 0x55555555u, b6 = s ^ 0x66666666u, b7 = s ^ 0x77777777u;
    if (MODE == 0) {  // 8 dependent v_sad_u8, SGPR operand
      uint32_t x = __builtin_amdgcn_sad_u8(a0, b0, 0);
      x = __builtin_amdgcn_sad_u8(a1, b1, x); x = __builtin_amdgcn_sad_u8(a2, b2, x); x = __builtin_amdgcn_sad_u8(a3, b3, x);
      x = __builtin_amdgcn_sad_u8(a4, b4, x); x = __builtin_amdgcn_sad_u8(a5, b5, x); x = __builtin_amdgcn_sad_u8(a6, b6, x);
      x = __builtin_amdgcn_sad_u8(a7, b7, x);
      acc += x;
    } else if (MODE == 1) {  // 8 v_sad in two independent chains of 4
      uint32_t x = __builtin_amdgcn_sad_u8(a0, b0, 0), y = __builtin_amdgcn_sad_u8(a4, b4, 0);
      x = __builtin_amdgcn_sad_u8(a1, b1, x); y = __builtin_amdgcn_sad_u8(a5, b5, y);
      x = __builtin_amdgcn_sad_u8(a2, b2, x); y = __builtin_amdgcn_sad_u8(a6, b6, y);
      x = __builtin_amdgcn_sad_u8(a3, b3, x); y = __builtin_amdgcn_sad_u8(a7, b7, y);
      acc += x; acc2 += y;
    } else if (MODE == 2) {  // 8 dependent v_add/xor (full-rate baseline)
      uint32_t x = a0 + b0; x = (x ^ a1) + b1; x = (x ^ a2) + b2; x = (x ^ a3) + b3;
      acc += x;
    } else if (MODE == 3) {  // full inner-loop body as in match_kernel
      const int32_t u2 = b0 & 0xFFFF, v2 = b0 >> 16;
      const bool in_ = (u2 >= ulo) & (u2 <= uhi) & (v2 >= vlo) & (v2 <= vhi);
      uint32_t x = __builtin_amdgcn_sad_u8(a0, b0, 0);
      x = __builtin_amdgcn_sad_u8(a1, b1, x); x = __builtin_amdgcn_sad_u8(a2, b2, x); x = __builtin_amdgcn_sad_u8(a3, b3, x);
      x = __builtin_amdgcn_sad_u8(a4, b4, x); x = __builtin_amdgcn_sad_u8(a5, b5, x); x = __builtin_amdgcn_sad_u8(a6, b6, x);
      x = __builtin_amdgcn_sad_u8(a7, b7, x);
      if (in_ && x < best) { best = x; bpos = i; }
    } else if (MODE == 4) {  // only the window test + update
      const int32_t u2 = b0 & 0xFFFF, v2 = b0 >> 16;
      const bool in_ = (u2 >= ulo) & (u2 <= uhi) & (v2 >= vlo) & (v2 <= vhi);
      const uint32_t x = b1 & 0x1FFF;
      if (in_ && x < best) { best = x; bpos = i; }
    } else if (MODE == 5) {  // 8 x v_sad_u16
      uint32_t x = __builtin_amdgcn_sad_u16(a0, b0, 0);
      x = __builtin_amdgcn_sad_u16(a1, b1, x); x = __builtin_amdgcn_sad_u16(a2, b2, x); x = __builtin_amdgcn_sad_u16(a3, b3, x);
      x = __builtin_amdgcn_sad_u16(a4, b4, x); x = __builtin_amdgcn_sad_u16(a5, b5, x); x = __builtin_amdgcn_sad_u16(a6, b6, x);
      x = __builtin_amdgcn_sad_u16(a7, b7, x);
      acc += x;
    } else if (MODE == 6) {  // 8 dependent v_sad_u8 with VGPR operands only
      const uint32_t c = b0 + t;
      uint32_t x = __builtin_amdgcn_sad_u8(a0, c, 0);
      x = __builtin_amdgcn_sad_u8(a1, c, x); x = __builtin_amdgcn_sad_u8(a2, c, x); x = __builtin_amdgcn_sad_u8(a3, c, x);
      x = __builtin_amdgcn_sad_u8(a4, c, x); x = __builtin_amdgcn_sad_u8(a5, c, x); x = __builtin_amdgcn_sad_u8(a6, c, x);
      x = __builtin_amdgcn_sad_u8(a7, c, x);
      acc += x;
    }
  }
  out[t] = acc + acc2 + best + bpos;
}

template <int MODE> void run(const char *name, int nvalu, uint32_t *out, uint32_t *in, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // waves per SIMD = blocks*4 / 1024
  const double iters_per_simd = (double)blocks * 4 / 1024 * ITERS;
  const double cyc = ms * 1e-3 * 2.4e9 / iters_per_simd;
  printf("%-28s blocks %5d  %8.3f ms  %6.1f cyc/iter/SIMD @2.4GHz  (%d VALU listed => %.2f cyc each)\n", name, blocks, ms, cyc, nvalu, cyc / nvalu);
}

int main() {
  uint32_t *out, *in;
  hipMalloc(&out, 4 * 256 * 8192); hipMalloc(&in, 4096);
  std::vector<uint32_t> h(1024); for (int i = 0; i < 1024; i++) h[i] = i * 2654435761u;
  hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
  for (int blocks : {256, 1024, 2048, 8192}) {
    run<0>("8 dep v_sad_u8 (sgpr)", 8, out, in, blocks);
    run<1>("2x4 v_sad_u8 (sgpr)", 8, out, in, blocks);
    run<6>("8 dep v_sad_u8 (vgpr)", 8, out, in, blocks);
    run<5>("8 dep v_sad_u16", 8, out, in, blocks);
    run<2>("8 dep add/xor", 8, out, in, blocks);
    run<3>("full body (16 VALU)", 16, out, in, blocks);
    run<4>("window+update (8 VALU)", 8, out, in, blocks);
  }
  return 0;
}
