// Drives include/viso_hip_matcher.hpp the way VisualOdometryStereo::process does
// (reference src/viso_stereo.cpp:33-52): pushBack(I1,I2,dims,replace) ->
// matchFeatures(2) -> [bucketFeatures] -> getMatches(), frame after frame; with
// <mono> = 1 the way VisualOdometryMono::process does (src/viso_mono.cpp:33-39):
// pushBack(I,dims,replace) -> matchFeatures(0) -> bucketFeatures -> getMatches().
//
//   shim_stereo_loop <frames.bin> <W> <H> <bpl> <n_frames> <bucket:0|1> <out.bin> [mono:0|1]
//
// frames.bin: n_frames x {left, right} raw u8 images of H*bpl bytes
// (the layout the reference's demo reads from its .dat files, src/demo.cpp:107-110).
// out.bin: per frame {int32 count, count x p_match(48 B)}.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "viso_hip_matcher.hpp"

int main(int argc, char **argv) {
  if (argc != 8 && argc != 9) { std::fprintf(stderr, "usage\n"); return 2; }
  const int mono = argc == 9 ? std::atoi(argv[8]) : 0;
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), bpl = std::atoi(argv[4]);
  const int nf = std::atoi(argv[5]), bucket = std::atoi(argv[6]);
  std::FILE *fi = std::fopen(argv[1], "rb"), *fo = std::fopen(argv[7], "wb");
  if (!fi || !fo) return 3;
  Matcher::parameters param;  // the reference's defaults
  Matcher *matcher = new Matcher(param);  // src/viso.cpp:32
  if (!matcher->ok()) return 4;
  matcher->setIntrinsics(645.24, 635.96, 194.13, 0.5707);
  std::vector<uint8_t> I1((size_t)H * bpl), I2((size_t)H * bpl);
  int32_t dims[3] = {W, H, bpl};
  for (int t = 0; t < nf; t++) {
    if (std::fread(I1.data(), 1, I1.size(), fi) != I1.size()) return 5;
    if (std::fread(I2.data(), 1, I2.size(), fi) != I2.size()) return 5;
    if (mono) {
      matcher->pushBack(I1.data(), dims, false);
      matcher->matchFeatures(0);
    } else {
      matcher->pushBack(I1.data(), I2.data(), dims, false);
      matcher->matchFeatures(2);
    }
    if (bucket) matcher->bucketFeatures(2, 50, 50);  // VisualOdometry::bucketing defaults (src/viso.h:44-53)
    std::vector<Matcher::p_match> p_matched = matcher->getMatches();
    const int32_t n = (int32_t)p_matched.size();
    std::fwrite(&n, 4, 1, fo);
    if (n) std::fwrite(p_matched.data(), sizeof(Matcher::p_match), (size_t)n, fo);
  }
  delete matcher;  // src/viso.cpp:39
  std::fclose(fi); std::fclose(fo);
  return 0;
}
