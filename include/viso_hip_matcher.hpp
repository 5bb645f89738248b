// viso_hip_matcher.hpp -- drop-in C++ `Matcher` over the C ABI of libviso_hip.so.
//
// Same public surface as the reference's class (src/matcher.h:40-143 of
// Chang-Tun-Yu/HLS-final-Visual-Odometry): nested `parameters` and `p_match`
// with identical field order and layout, the constructor, setIntrinsics,
// both pushBack overloads, matchFeatures, bucketFeatures and getMatches, all
// with the reference's signatures and void returns -- plus computeFeatures,
// the private member the reference's pushBack is built on (src/matcher.h:209),
// with its signature and its _mm_malloc ownership.  src/viso.cpp,
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
#include <mm_malloc.h>  // _mm_malloc / _mm_free: the ownership contract of computeFeatures (src/matcher.h:208)

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
  explicit Matcher(parameters param, int32_t device = 0) : outlier_removal(true), param(param), handle(0), device(device) {
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

  // src/matcher.h:209, src/matcher.cpp:585-672 -- the detector on one image, with the reference's signature and
  // ownership: max1 / max2 (sparse set, only with param.multi_stage / dense set; 0 when empty) and the Sobel planes
  // I_du, I_dv (at matching resolution: half size with param.half_resolution) are _mm_malloc blocks the CALLER
  // releases with _mm_free; I_du_full / I_dv_full (the full-resolution planes) are produced only with
  // param.half_resolution and left untouched otherwise, as the reference does (:606-613).  (A private member in
  // the reference; public here, where nothing else calls it.)  The records are the reference's bit for bit; the
  // planes agree with filter::sobel5x5 on the valid interior -- 2 pixels in from every edge -- outside of which
  // the reference's SSE row passes hold wrapped-around values nothing reads.  On an error the
  // outputs are 0 and a message goes to std::cerr.
  void computeFeatures(uint8_t *I, const int32_t *dims, int32_t *&max1, int32_t &num1, int32_t *&max2, int32_t &num2,
                       uint8_t *&I_du, uint8_t *&I_dv, uint8_t *&I_du_full, uint8_t *&I_dv_full) {
    max1 = 0; max2 = 0; num1 = 0; num2 = 0; I_du = 0; I_dv = 0;
    if (!I || !dims || dims[0] < 1 || dims[1] < 1 || dims[2] < dims[0]) { std::cerr << "ERROR: Image dimension mismatch!" << std::endl; return; }
    vh_params p;
    std::memcpy(&p, &param, sizeof(p));
    int32_t dm[3] = {dims[0], dims[1], dims[2]};
    if (param.half_resolution) {  // getHalfResolutionDimensions, src/matcher.cpp:566-570
      dm[0] = dims[0] / 2; dm[1] = dims[1] / 2;
      dm[2] = dm[0] > 0 ? dm[0] + 15 - (dm[0] - 1) % 16 : 16;
    }
    const size_t plane = (size_t)dm[2] * (size_t)dm[1], plane_full = (size_t)dims[2] * (size_t)dims[1];
    // at most one feature per class and NMS block of (nms_n + 1)^2 pixels (src/matcher.cpp:381-466)
    const int32_t blk = param.nms_n + 1;
    const int64_t bound = std::max<int64_t>(4 * ((int64_t)dm[0] / blk + 1) * ((int64_t)dm[1] / blk + 1), 64);
    std::vector<int32_t> t1, t2((size_t)bound * VH_FEATURE_WORDS);
    if (param.multi_stage) t1.resize((size_t)bound * VH_FEATURE_WORDS);
    uint8_t *du = (uint8_t *)_mm_malloc(std::max<size_t>(plane, 16), 16), *dv = (uint8_t *)_mm_malloc(std::max<size_t>(plane, 16), 16);
    int32_t n1 = 0, n2 = 0;
    int32_t rc = (du && dv) ? vh_compute_features(&p, device, I, dims, t1.empty() ? 0 : t1.data(), (int32_t)bound, &n1, t2.data(), (int32_t)bound, &n2, du, dv)
                            : VH_ERR_CAPACITY;
    uint8_t *duf = 0, *dvf = 0;
    if (rc == VH_OK && param.half_resolution) {  // filter::sobel5x5 on the full-resolution image (:610-613)
      duf = (uint8_t *)_mm_malloc(std::max<size_t>(plane_full, 16), 16); dvf = (uint8_t *)_mm_malloc(std::max<size_t>(plane_full, 16), 16);
      rc = (duf && dvf) ? vh_filters(device, I, dims[2], dims[1], duf, dvf, 0, 0) : VH_ERR_CAPACITY;
    }
    int32_t *m1 = 0, *m2 = 0;
    if (rc == VH_OK && n1 > 0) { m1 = (int32_t *)_mm_malloc(sizeof(int32_t) * VH_FEATURE_WORDS * (size_t)n1, 16); if (!m1) rc = VH_ERR_CAPACITY; }
    if (rc == VH_OK && n2 > 0) { m2 = (int32_t *)_mm_malloc(sizeof(int32_t) * VH_FEATURE_WORDS * (size_t)n2, 16); if (!m2) rc = VH_ERR_CAPACITY; }
    if (rc != VH_OK) {
      report("computeFeatures", rc);
      void *blocks[6] = {du, dv, duf, dvf, m1, m2};
      for (int k = 0; k < 6; k++) if (blocks[k]) _mm_free(blocks[k]);
      return;
    }
    if (m1) std::memcpy(m1, t1.data(), sizeof(int32_t) * VH_FEATURE_WORDS * (size_t)n1);
    if (m2) std::memcpy(m2, t2.data(), sizeof(int32_t) * VH_FEATURE_WORDS * (size_t)n2);
    max1 = m1; num1 = n1; max2 = m2; num2 = n2; I_du = du; I_dv = dv;
    if (param.half_resolution) { I_du_full = duf; I_dv_full = dvf; }
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
  int32_t device;
};

#endif  // VISO_HIP_MATCHER_HPP
