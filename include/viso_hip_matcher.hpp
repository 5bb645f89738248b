// viso_hip_matcher.hpp -- drop-in C++ `Matcher` over the C ABI of libviso_hip.so.
//
// Same public surface as the reference's class (src/matcher.h:40-143 of
// Chang-Tun-Yu/HLS-final-Visual-Odometry): nested `parameters` and `p_match`
// with identical field order and layout, the constructor, setIntrinsics,
// both pushBack overloads, matchFeatures, bucketFeatures and getMatches, all
// with the reference's signatures and void returns.  src/viso.cpp,
// src/viso_stereo.cpp and src/viso_mono.cpp therefore compile against this
// header unchanged once it is on the include path AS "matcher.h"
// (see INTEGRATION.md); `Matrix` only has to be a declared type, because
// Tr_delta is accepted and ignored exactly as the reference does
// (src/matcher.cpp:93-111).
//
// Differences a maintainer should know about (all documented in DESIGN.md):
//  * detection and matching run on the GPU (hand-written HIP kernels); the
//    stock SSE Matcher::computeFeatures / matching semantics are reproduced
//    bit for bit, NOT the 1024x284-only HLS restatement the reference's
//    pushBack/matchFeatures currently call (finding 2);
//  * matchFeatures ends with removeOutliers like the reference's
//    (src/matcher.cpp:108), host side and bit-identical to it; set
//    `outlier_removal = false` for the bare Matcher::matching result.  Stereo
//    matches (method 1, absent from the reference) are never filtered: its
//    vote compares flows, which stereo records do not carry;
//  * errors that the reference ignores (capacity overrun, bad dims) are
//    reported on std::cerr and leave the match list empty instead of
//    corrupting memory.
#ifndef VISO_HIP_MATCHER_HPP
#define VISO_HIP_MATCHER_HPP

// the same system headers the reference's matcher.h pulls in (src/matcher.h:25-32):
// its callers (viso_stereo.cpp, viso_mono.cpp) rely on getting <math.h> and
// <algorithm> through it
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <cstring>
#include <iostream>
#include <vector>

#include "viso_hip.h"

class Matrix;  // src/matrix.h; only passed through

class Matcher {
 public:
  // parameter settings (src/matcher.h:45-72)
  struct parameters {
    int32_t nms_n;
    int32_t nms_tau;
    int32_t match_binsize;
    int32_t match_radius;
    int32_t match_disp_tolerance;
    int32_t outlier_disp_tolerance;
    int32_t outlier_flow_tolerance;
    int32_t multi_stage;
    int32_t half_resolution;
    int32_t refinement;
    double f, cu, cv, base;
    parameters() {
      nms_n = 2;
      nms_tau = 50;
      match_binsize = 50;
      match_radius = 200;
      match_disp_tolerance = 2;
      outlier_disp_tolerance = 5;
      outlier_flow_tolerance = 5;
      multi_stage = 0;
      half_resolution = 0;
      refinement = 0;
      f = cu = cv = base = 0;
    }
  };

  // structure for storing matches (src/matcher.h:89-104)
  struct p_match {
    float u1p, v1p; int32_t i1p;
    float u2p, v2p; int32_t i2p;
    float u1c, v1c; int32_t i1c;
    float u2c, v2c; int32_t i2c;
    p_match() {}
    p_match(float u1p, float v1p, int32_t i1p, float u2p, float v2p, int32_t i2p, float u1c, float v1c,
            int32_t i1c, float u2c, float v2c, int32_t i2c)
        : u1p(u1p), v1p(v1p), i1p(i1p), u2p(u2p), v2p(v2p), i2p(i2p),
          u1c(u1c), v1c(v1c), i1c(i1c), u2c(u2c), v2c(v2c), i2c(i2c) {}
  };

  // constructor (src/matcher.cpp:32-41); `device` selects the GPU of this stream
  explicit Matcher(parameters param, int32_t device = 0) : outlier_removal(true), param(param), handle(0) {
    static_assert(sizeof(parameters) == sizeof(vh_params), "parameters must mirror vh_params");
    static_assert(sizeof(p_match) == sizeof(vh_p_match) && sizeof(p_match) == 48, "p_match must be 48 bytes");
    vh_params p;
    std::memcpy(&p, &param, sizeof(p));
    const int32_t rc = vh_create(&p, device, &handle);
    if (rc != VH_OK) {
      std::cerr << "ERROR: viso_hip: " << vh_error_string(rc) << " " << vh_last_error() << std::endl;
      handle = 0;
    }
  }
  ~Matcher() { if (handle) vh_destroy(handle); }

  // intrinsics (src/matcher.h:81-86)
  void setIntrinsics(double f, double cu, double cv, double base) {
    param.f = f; param.cu = cu; param.cv = cv; param.base = base;
    if (handle) vh_set_intrinsics(handle, f, cu, cv, base);
  }

  // src/matcher.h:116, src/matcher.cpp:51-91
  void pushBack(uint8_t *I1, uint8_t *I2, int32_t *dims, const bool replace) {
    if (!handle) return;
    const int32_t rc = vh_push_back(handle, I1, I2, dims, replace ? 1 : 0);
    if (rc == VH_ERR_INVALID_ARG) std::cerr << "ERROR: Image dimension mismatch!" << std::endl;
    else if (rc != VH_OK) report("pushBack", rc);
  }
  // src/matcher.h:122
  void pushBack(uint8_t *I1, int32_t *dims, const bool replace) { pushBack(I1, 0, dims, replace); }

  // src/matcher.h:128, src/matcher.cpp:93-111 (0 = flow, 1 = stereo, 2 = quad)
  void matchFeatures(int32_t method, Matrix *Tr_delta = 0) {
    (void)Tr_delta;
    if (!handle) return;
    int32_t rc = vh_match_features(handle, method, 0);
    if (rc != VH_OK && rc != VH_ERR_STATE) { report("matchFeatures", rc); return; }
    if (rc == VH_OK && outlier_removal && (rc = vh_remove_outliers(handle)) != VH_OK) report("removeOutliers", rc);
  }

  // src/matcher.h:132, src/matcher.cpp:140-187
  void bucketFeatures(int32_t max_features, float bucket_width, float bucket_height) {
    if (!handle) return;
    const int32_t rc = vh_bucket_features(handle, max_features, bucket_width, bucket_height);
    if (rc != VH_OK) report("bucketFeatures", rc);
  }

  // src/matcher.h:138-143
  std::vector<Matcher::p_match> getMatches() {
    std::vector<Matcher::p_match> out;
    if (!handle) return out;
    int32_t n = 0;
    int32_t rc = vh_get_matches(handle, 0, 0, &n);
    if ((rc != VH_OK && rc != VH_ERR_CAPACITY) || n <= 0) return out;
    out.resize((size_t)n);
    rc = vh_get_matches(handle, reinterpret_cast<vh_p_match *>(out.data()), n, &n);
    if (rc != VH_OK) { report("getMatches", rc); out.clear(); }
    return out;
  }

  // The ring buffer's packed feature records {u,v,0,class,d1..d8}
  // (max2p/max2c of the reference, src/matcher.h:252); which = VH_SET_*.
  std::vector<int32_t> getFeatures(int32_t which) {
    std::vector<int32_t> out;
    if (!handle) return out;
    int32_t n = 0;
    int32_t rc = vh_get_features(handle, which, 0, 0, &n);
    if ((rc != VH_OK && rc != VH_ERR_CAPACITY) || n <= 0) return out;
    out.resize((size_t)n * VH_FEATURE_WORDS);
    rc = vh_get_features(handle, which, out.data(), n, &n);
    if (rc != VH_OK) { report("getFeatures", rc); out.clear(); }
    return out;
  }

  bool ok() const { return handle != 0; }

  // matchFeatures finishes with removeOutliers (src/matcher.cpp:108) unless cleared
  bool outlier_removal;

 private:
  Matcher(const Matcher &);             // one handle per camera stream
  Matcher &operator=(const Matcher &);
  void report(const char *where, int32_t rc) {
    std::cerr << "ERROR: viso_hip " << where << ": " << vh_error_string(rc) << " " << vh_last_error() << std::endl;
  }
  parameters param;
  vh_matcher *handle;
};

#endif  // VISO_HIP_MATCHER_HPP
