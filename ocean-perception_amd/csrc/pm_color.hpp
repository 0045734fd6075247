// pm_color.hpp -- per-pixel colour arithmetic of the stereo-ready enhancement (device inline functions only), shared by
// the imaging kernels (pm_enhance.hpp) and the prep kernel that folds the enhancement's per-pixel tail into its load
// (pm_kernels.hpp::k_prep_bgr).  Every float operation is a single IEEE operation in the order of
// oracle/pm_enhance_oracle.c; reference: src/vehicle/imaging/normalization.cpp:43-69,178-185.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

namespace pm {

// ---- cv::cvtColor BGR2HSV / HSV2BGR on floats -----------------------------------------------------------------------
__device__ __forceinline__ void bgr2hsv_d(float b, float g, float r, float& h, float& s, float& v) {
  v = b;
  float vmin = b;
  if (g > v) v = g;
  if (r > v) v = r;
  if (g < vmin) vmin = g;
  if (r < vmin) vmin = r;
  float diff = v - vmin;
  s = diff / (fabsf(v) + FLT_EPSILON);
  diff = 60.f / (diff + FLT_EPSILON);
  if (v == r) h = (g - b) * diff;
  else if (v == g) h = (b - r) * diff + 120.f;
  else h = (r - g) * diff + 240.f;
  if (h < 0.f) h += 360.f;
}

__device__ __forceinline__ void hsv2bgr_d(float h, float s, float v, float& b, float& g, float& r) {
  if (s == 0.f) {
    b = g = r = v;
    return;
  }
  h = h * (6.f / 360.f);
  if (h < 0.f) {
    do h += 6.f; while (h < 0.f);
  } else if (h >= 6.f) {
    do h -= 6.f; while (h >= 6.f);
  }
  int sector = (int)floorf(h);
  h -= (float)sector;
  if ((unsigned)sector >= 6u) {
    sector = 0;
    h = 0.f;
  }
  const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * h), t3 = v * (1.f - s * (1.f - h));
  // sector_data = {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}
  switch (sector) {
    case 0: b = t1; g = t3; r = t0; break;
    case 1: b = t1; g = t0; r = t2; break;
    case 2: b = t3; g = t0; r = t1; break;
    case 3: b = t0; g = t2; r = t1; break;
    case 4: b = t0; g = t1; r = t3; break;
    default: b = t2; g = t1; r = t0; break;
  }
}

// Normalize's per-pixel part (normalization.cpp:43-69): the HSV value channel stretched, V' = V * alpha + beta
__device__ __forceinline__ void normalize_px(float b, float g, float r, float alpha, float beta, float& ob, float& og,
                                             float& orr) {
  float h, s, v;
  bgr2hsv_d(b, g, r, h, s, v);
  v = v * alpha + beta;
  hsv2bgr_d(h, s, v, ob, og, orr);
}
// alpha = (float)(1 / (vmax - vmin)), beta = (float)(-vmin / (vmax - vmin)) from the min / max bits of the 1/8 image
__device__ __forceinline__ void stretch_coeffs(const unsigned* __restrict__ mm, float& alpha, float& beta) {
  const double vmin = (double)__uint_as_float(mm[0]), vmax = (double)__uint_as_float(mm[1]);
  alpha = (float)(1.0 / (vmax - vmin));
  beta = (float)(-vmin / (vmax - vmin));
}
// One channel of NormalizeColorIlluminant before its Normalize: CastImage3bTo3f(I) / (2 * blur), 0 where the divisor is 0
// (what k_blur_cols<DIVIDE, ORIG_U8> writes)
__device__ __forceinline__ float illuminant_div(uint8_t orig, float blur) {
  const float cast = (float)(1.0 / 255.0);
  const float num = (float)orig * cast;
  const float d = blur * 2.0f;
  return d != 0.f ? num / d : 0.f;
}
// BGR2GRAY + convertTo(CV_8U, 255) with saturate_cast<uchar>
__device__ __forceinline__ uint8_t gray_u8(float b, float g, float r) {
  float gr = b * 0.114f;
  gr = gr + g * 0.587f;
  gr = gr + r * 0.299f;
  return (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(gr * 255.f, 0, 0u);
}
// The whole per-pixel tail of pm_stereo_ready from the 8-bit pixel and its blurred illuminant: q = I / (2 blur),
// J1 = Normalize(q) (the one inside NormalizeColorIlluminant), J2 = Normalize(J1) (enhance_test.cpp:69), gray.
// mm = {min1, max1, min2, max2} bits.
__device__ __forceinline__ uint8_t stereo_ready_gray(const uint8_t* __restrict__ bgr8, const float* __restrict__ blur,
                                                     size_t px, const unsigned* __restrict__ mm) {
  float a1, b1, a2, b2;
  stretch_coeffs(mm, a1, b1);
  stretch_coeffs(mm + 2, a2, b2);
  const float q0 = illuminant_div(bgr8[px * 3], blur[px * 3]), q1 = illuminant_div(bgr8[px * 3 + 1], blur[px * 3 + 1]),
              q2 = illuminant_div(bgr8[px * 3 + 2], blur[px * 3 + 2]);
  float jb, jg, jr, kb, kg, kr;
  normalize_px(q0, q1, q2, a1, b1, jb, jg, jr);
  normalize_px(jb, jg, jr, a2, b2, kb, kg, kr);
  return gray_u8(kb, kg, kr);
}

// What pm_match_bgr_device hands to the prep kernel (pm_kernels.hpp::k_prep_bgr) in place of 8-bit gray images.
struct BgrSource {
  const uint8_t* left;   // [B][rows][cols][3]
  const uint8_t* right;
  const float* blur_l;   // [B][rows][cols][3] GaussianBlur(CastImage3bTo3f(I))
  const float* blur_r;
  const unsigned* mm;    // [B][2 images][4]: value min / max bits of the two Normalize stages
};
}  // namespace pm
