// Drives the drop-in Matcher::computeFeatures (include/viso_hip_matcher.hpp) exactly as the reference's own
// member is called (src/matcher.cpp:585-672; oracle/ref_harness.cpp calls the reference's the same way):
// references to null pointers in, _mm_malloc blocks out, released here with _mm_free.
//   shim_compute_features <image.bin> W H bpl nms_n nms_tau multi_stage half_resolution <out.bin>
// out.bin: num1, max1[12 num1], num2, max2[12 num2], dm[3], I_du, I_dv (dm[2] x dm[1] each), has_full, I_du_full, I_dv_full
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "viso_hip_matcher.hpp"

int main(int argc, char **argv) {
  if (argc < 10) return 2;
  const int32_t W = atoi(argv[2]), H = atoi(argv[3]), bpl = atoi(argv[4]);
  Matcher::parameters param;
  param.nms_n = atoi(argv[5]); param.nms_tau = atoi(argv[6]); param.multi_stage = atoi(argv[7]); param.half_resolution = atoi(argv[8]);
  std::vector<uint8_t> img((size_t)bpl * H);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(img.data(), 1, img.size(), f) != img.size()) return 3;
  fclose(f);
  Matcher M(param);
  if (!M.ok()) return 4;
  int32_t dims[3] = {W, H, bpl};
  int32_t *max1 = 0, *max2 = 0, num1 = 0, num2 = 0;
  uint8_t *I_du = 0, *I_dv = 0, *I_du_full = 0, *I_dv_full = 0;
  M.computeFeatures(img.data(), dims, max1, num1, max2, num2, I_du, I_dv, I_du_full, I_dv_full);
  if (!I_du || !I_dv) return 5;
  int32_t dm[3] = {W, H, bpl};
  if (param.half_resolution) { dm[0] = W / 2; dm[1] = H / 2; dm[2] = dm[0] + 15 - (dm[0] - 1) % 16; }
  FILE *o = fopen(argv[9], "wb");
  if (!o) return 6;
  fwrite(&num1, 4, 1, o); if (num1) fwrite(max1, 48, (size_t)num1, o);
  fwrite(&num2, 4, 1, o); if (num2) fwrite(max2, 48, (size_t)num2, o);
  fwrite(dm, 4, 3, o);
  fwrite(I_du, 1, (size_t)dm[2] * dm[1], o); fwrite(I_dv, 1, (size_t)dm[2] * dm[1], o);
  const int32_t has_full = (I_du_full && I_dv_full) ? 1 : 0;
  fwrite(&has_full, 4, 1, o);
  if (has_full) { fwrite(I_du_full, 1, (size_t)bpl * H, o); fwrite(I_dv_full, 1, (size_t)bpl * H, o); }
  fclose(o);
  if (((uintptr_t)max2 & 15) || ((uintptr_t)I_du & 15)) return 7;  // 16-byte aligned, as _mm_load_si128 on the records needs (src/matcher.cpp:226-227)
  if (max1) _mm_free(max1);
  if (max2) _mm_free(max2);
  _mm_free(I_du); _mm_free(I_dv);
  if (I_du_full) _mm_free(I_du_full);
  if (I_dv_full) _mm_free(I_dv_full);
  return 0;
}
