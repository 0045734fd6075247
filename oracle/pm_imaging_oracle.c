/* pm_imaging_oracle.c -- CPU restatement of the per-pixel range-dependent stages that follow stereo
 * (SURVEY.md 8f-3).  TEST INFRASTRUCTURE ONLY, like pm_oracle.c: the checker of include/pm/imaging.h's
 * device kernels, never linked into or called from the product.
 *
 * PARITY UNPINNED: the reference holds no expected outputs for these functions (the tests under test/imaging only
 * displays images), OpenCV is not available here, and cv::exp is OpenCV's own table-based exponential.
 * The restatement follows the reference's operation order in float with the C library's expf; the
 * device results are compared to a stated relative tolerance, not bit for bit.
 *
 *   pmo_disp_to_range        StereoCamera::DispToDepth, src/vehicle/vision_core/stereo_camera.cpp:49-53
 *   pmo_remove_backscatter   imaging::RemoveBackscatter, src/vehicle/imaging/backscatter.cpp:277-308
 *   pmo_correct_attenuation  imaging::CorrectAttenuation + SetMaxRangeWhereZero, attenuation.cpp:255-299
 *   pmo_compute_intensity    ComputeIntensity, src/vehicle/vision_core/image_util.cpp:97-102
 *   pmo_find_dark            imaging::FindDarkFast, backscatter.cpp:41-78
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

void pmo_disp_to_range(const float* disp, size_t n, double fx, double baseline, float* range) {
  const double fxb = fx * baseline; /* fx() * Baseline() / disp, left to right */
  for (size_t i = 0; i < n; ++i) range[i] = disp[i] > 0.f ? (float)(fxb / (double)disp[i]) : 0.f;
}

/* bgr: interleaved [n][3].  z = range + threshold_inv(range, 1e-3 -> 20); e = exp(z * -beta);
 * bs = e * -B + B (the cv::Mat expression B * (1 - e) folds into one scale-and-shift); out = max(I - bs, 0). */
void pmo_remove_backscatter(const float* bgr, const float* range, size_t n, const float B[3], const float beta_B[3],
                            float* out) {
  for (size_t i = 0; i < n; ++i) {
    const float r = range[i];
    const float z = r > 1e-3f ? r : r + 20.0f;
    for (int c = 0; c < 3; ++c) {
      const float e = expf(z * (-beta_B[c]));
      const float bs = e * (-B[c]) + B[c];
      const float o = bgr[i * 3 + c] - bs;
      out[i * 3 + c] = o > 0.f ? o : 0.f;
    }
  }
}

/* X = (a_b, a_g, a_r, b_b, ..., c_b, ..., d_b, ...).  z = range + (range > 0 ? 0 : max(range));
 * beta_cz = z * (a * exp(z * b) + c * exp(z * d)); out = I * exp(beta_cz). */
void pmo_correct_attenuation(const float* bgr, const float* range, size_t n, const float X[12], float* out) {
  float rmax = 0.f;
  for (size_t i = 0; i < n; ++i) rmax = range[i] > rmax ? range[i] : rmax;
  for (size_t i = 0; i < n; ++i) {
    const float r = range[i];
    const float z = r > 0.f ? r : r + rmax;
    for (int c = 0; c < 3; ++c) {
      const float e1 = expf(z * X[3 + c]), e2 = expf(z * X[9 + c]);
      const float w = e1 * X[c] + e2 * X[6 + c];
      const float bz = z * w;
      out[i * 3 + c] = bgr[i * 3 + c] * expf(bz);
    }
  }
}

void pmo_compute_intensity(const float* bgr, size_t n, float* gray) {
  for (size_t i = 0; i < n; ++i) {
    float s = bgr[i * 3] * 0.114f;
    s = s + bgr[i * 3 + 1] * 0.587f;
    s = s + bgr[i * 3 + 2] * 0.299f;
    gray[i] = s;
  }
}

static int dark_count(const float* intensity, const float* range, size_t n, float thr, uint8_t* mask) {
  int c = 0;
  for (size_t i = 0; i < n; ++i) {
    const int dark = (intensity[i] <= thr) && (range[i] > 0.1f);
    mask[i] = dark ? 255 : 0;
    c += dark;
  }
  return c;
}

float pmo_find_dark(const float* intensity, const float* range, int rows, int cols, float percentile, uint8_t* mask) {
  const size_t n = (size_t)rows * cols;
  const float N = (float)(rows * cols);
  const int n_desired = (int)(percentile * N);
  float low = 0.f, high = 0.5f;
  const float first = (float)(1.5 * percentile);
  int n_dark = dark_count(intensity, range, n, first, mask);
  if (n_dark < n_desired) low = first;
  else if (n_dark > n_desired) high = first;
  else return first;
  for (int iter = 0; iter < 8; ++iter) {
    const float thr = (high + low) / 2.0f;
    n_dark = dark_count(intensity, range, n, thr, mask);
    if (n_dark < n_desired) low = thr;
    else if (n_dark > n_desired) high = thr;
    else return thr;
  }
  return (high + low) / 2.0f;
}
