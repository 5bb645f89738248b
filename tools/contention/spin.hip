// spin.hip -- synthetic competitors for tools/vote_contention.py: kernels that load ONE resource of the chip
// (full-rate VALU issue, v_sad_u8 issue, LDS traffic, dependent HBM gathers) for a chosen number of iterations.
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {
__global__ __launch_bounds__(256) void spin_fma(float *out, int iters) {
  float a = threadIdx.x, b = 1.0f, c = 2.0f, d = 3.0f;
  for (int i = 0; i < iters; i++) {
    a = a * 1.0001f + 0.5f; b = b * 0.9999f + a; c = c * 1.0002f + 0.25f; d = d * 0.9998f + c;
  }
  if (a + b + c + d == 12345.0f) out[0] = a;
}
__global__ __launch_bounds__(256) void spin_sad(uint32_t *out, int iters) {
  uint32_t a = threadIdx.x * 0x01010101u, b = 0x10203040u, s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int i = 0; i < iters; i++) {
    s0 = __builtin_amdgcn_sad_u8(a, b, s0); s1 = __builtin_amdgcn_sad_u8(a ^ s0, b, s1);
    s2 = __builtin_amdgcn_sad_u8(a, b ^ s1, s2); s3 = __builtin_amdgcn_sad_u8(a ^ s2, b, s3);
  }
  if (s0 + s1 + s2 + s3 == 0x12345u) out[0] = s0;
}
__global__ __launch_bounds__(256) void spin_lds(uint32_t *out, int iters) {
  __shared__ uint32_t sh[4096];
  for (int k = threadIdx.x; k < 4096; k += 256) sh[k] = k;
  __syncthreads();
  uint32_t s = 0, j = threadIdx.x;
  for (int i = 0; i < iters; i++) { s += sh[j & 4095]; j += 257; s += sh[(j + s) & 4095]; }
  if (s == 0x12345u) out[0] = s;
}
// every lane chases its own pointer chain through a large table (one 64-byte sector per step)
__global__ __launch_bounds__(64) void spin_chase(const uint32_t *table, uint32_t mask, uint32_t *out, int iters) {
  uint32_t j = (blockIdx.x * 64 + threadIdx.x) * 2654435761u;
  for (int i = 0; i < iters; i++) j = table[(size_t)(j & mask) * 16] + i;
  if (j == 0x12345u) out[0] = j;
}
}  // namespace

extern "C" int spin_launch(int kind, int blocks, int iters, void *buf, unsigned mask, void *stream) {
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) hipLaunchKernelGGL(spin_fma, dim3(blocks), dim3(256), 0, st, (float *)buf, iters);
  else if (kind == 1) hipLaunchKernelGGL(spin_sad, dim3(blocks), dim3(256), 0, st, (uint32_t *)buf, iters);
  else if (kind == 2) hipLaunchKernelGGL(spin_lds, dim3(blocks), dim3(256), 0, st, (uint32_t *)buf, iters);
  else hipLaunchKernelGGL(spin_chase, dim3(blocks), dim3(64), 0, st, (const uint32_t *)buf, mask, (uint32_t *)buf, iters);
  return (int)hipGetLastError();
}
