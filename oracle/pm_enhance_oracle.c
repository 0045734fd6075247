/* pm_enhance_oracle.c -- CPU restatement of the range-free "stereo-ready" enhancement that precedes stereo
 * (SURVEY.md 8f-2): J = Normalize(NormalizeColorIlluminant(I)), gray = BGR2GRAY(J), as in
 * test/imaging/enhance_test.cpp:69-73 and the commented intent of test/stereo_matching/sgbm_test.cpp:66-84.
 * TEST INFRASTRUCTURE ONLY (the checker of the device kernels in pm_enhance.hpp).
 *
 * PARITY UNPINNED: the reference holds no expected outputs for this chain and OpenCV 3.4.0 is not available
 * here; the OpenCV primitives below (getGaussianKernel, the separable filter's summation order, cv::divide,
 * BGR<->HSV on floats, INTER_LINEAR resize, minMaxLoc, the scale-and-shift folding of cv::Mat expressions,
 * convertTo rounding) are restated from the OpenCV 3.4 sources as remembered -- they are this build's own
 * definitions.  Every float operation is a single IEEE operation (-ffp-contract=off), so the device kernels
 * can match bit for bit.
 *
 *   CastImage3bTo3f             src/vehicle/vision_core/image_util.cpp:25-31 (scale 1.0/255.0 as float)
 *   EstimateIlluminantGaussian  src/vehicle/imaging/illuminant.cpp:10-21 (GaussianBlur, BORDER_REPLICATE, x2)
 *   NormalizeColorIlluminant    src/vehicle/imaging/normalization.cpp:178-185 (ksize = NextOddInt(cols/3), sigma = ksize/4)
 *   Normalize                   src/vehicle/imaging/normalization.cpp:43-69 (HSV value stretch, min/max of a 1/8 resize)
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* core/math_util.hpp:22-25 */
static int next_odd_int(int x) { return x + (1 - x % 2); }

/* cv::getGaussianKernel(n, sigma, CV_32F), sigma > 0 */
void pmo_gaussian_kernel(int n, double sigma, float* k) {
  const double scale2x = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    const double t = exp(scale2x * x * x);
    k[i] = (float)t;
    sum += k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cv::GaussianBlur(src, dst, Size(ksize, ksize), sigma, sigma, BORDER_REPLICATE) on an interleaved float image
 * with `ch` channels.  Row pass: the generic row filter, taps added left to right.  Column pass: the symmetric
 * column filter, centre tap first, then pairs (S[+j] + S[-j]) * k[c + j]. */
void pmo_gaussian_blur(const float* src, int rows, int cols, int ch, int ksize, double sigma, float* dst) {
  float* k = (float*)malloc(sizeof(float) * (size_t)ksize);
  float* tmp = (float*)malloc(sizeof(float) * (size_t)rows * cols * ch);
  pmo_gaussian_kernel(ksize, sigma, k);
  const int c = ksize / 2;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
      for (int q = 0; q < ch; ++q) {
        float s = k[0] * src[((size_t)y * cols + clampi(x - c, 0, cols - 1)) * ch + q];
        for (int t = 1; t < ksize; ++t) s = s + k[t] * src[((size_t)y * cols + clampi(x - c + t, 0, cols - 1)) * ch + q];
        tmp[((size_t)y * cols + x) * ch + q] = s;
      }
#pragma omp parallel for schedule(static)
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x)
      for (int q = 0; q < ch; ++q) {
        float s = k[c] * tmp[((size_t)y * cols + x) * ch + q];
        for (int j = 1; j <= c; ++j) {
          const float a = tmp[((size_t)clampi(y + j, 0, rows - 1) * cols + x) * ch + q];
          const float b = tmp[((size_t)clampi(y - j, 0, rows - 1) * cols + x) * ch + q];
          s = s + k[c + j] * (a + b);
        }
        dst[((size_t)y * cols + x) * ch + q] = s;
      }
  free(tmp);
  free(k);
}

void pmo_cast_3b_to_3f(const uint8_t* bgr, size_t n_values, float* out) {
  const float s = (float)(1.0 / 255.0);
  for (size_t i = 0; i < n_values; ++i) out[i] = (float)bgr[i] * s;
}

/* cv::cvtColor BGR2HSV on floats (RGB2HSV_f, hrange 360) */
static void bgr2hsv(float b, float g, float r, float* hh, float* ss, float* vv) {
  float v = b, vmin = b;
  if (g > v) v = g;
  if (r > v) v = r;
  if (g < vmin) vmin = g;
  if (r < vmin) vmin = r;
  float diff = v - vmin;
  const float s = diff / (fabsf(v) + FLT_EPSILON);
  diff = 60.f / (diff + FLT_EPSILON);
  float h;
  if (v == r) h = (g - b) * diff;
  else if (v == g) h = (b - r) * diff + 120.f;
  else h = (r - g) * diff + 240.f;
  if (h < 0.f) h += 360.f;
  *hh = h;
  *ss = s;
  *vv = v;
}

/* cv::cvtColor HSV2BGR on floats (HSV2RGB_native, hscale = 6/360) */
static void hsv2bgr(float h, float s, float v, float* bb, float* gg, float* rr) {
  static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
  if (s == 0.f) {
    *bb = *gg = *rr = v;
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
  float tab[4];
  tab[0] = v;
  tab[1] = v * (1.f - s);
  tab[2] = v * (1.f - s * h);
  tab[3] = v * (1.f - s * (1.f - h));
  *bb = tab[sector_data[sector][0]];
  *gg = tab[sector_data[sector][1]];
  *rr = tab[sector_data[sector][2]];
}

/* One axis of cv::resize INTER_LINEAR: source index and the two weights of destination index d. */
static void linear_coeff(int d, int ssize, int dsize, int* s0, float* w0, float* w1) {
  const double scale = (double)ssize / dsize;
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) {
    f = 0.f;
    s = 0;
  }
  if (s >= ssize - 1) {
    f = 0.f;
    s = ssize - 1;
  }
  *s0 = s;
  *w0 = 1.f - f;
  *w1 = f;
}

/* min and max of cv::resize(V, size / 8) (INTER_LINEAR): horizontal pass per source row, then vertical. */
void pmo_value_minmax_eighth(const float* V, int rows, int cols, double* vmin, double* vmax) {
  const int dr = rows / 8, dc = cols / 8;
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (int dy = 0; dy < dr; ++dy) {
    int sy;
    float b0, b1;
    linear_coeff(dy, rows, dr, &sy, &b0, &b1);
    const int sy1 = sy + 1 < rows ? sy + 1 : rows - 1;
    for (int dx = 0; dx < dc; ++dx) {
      int sx;
      float a0, a1;
      linear_coeff(dx, cols, dc, &sx, &a0, &a1);
      const int sx1 = sx + 1 < cols ? sx + 1 : cols - 1;
      const float r0 = V[(size_t)sy * cols + sx] * a0 + V[(size_t)sy * cols + sx1] * a1;
      const float r1 = V[(size_t)sy1 * cols + sx] * a0 + V[(size_t)sy1 * cols + sx1] * a1;
      const float val = r0 * b0 + r1 * b1;
      if (val < lo) lo = val;
      if (val > hi) hi = val;
    }
  }
  *vmin = lo;
  *vmax = hi;
}

/* imaging::Normalize: HSV, V' = V * (1/(vmax-vmin)) + (-vmin/(vmax-vmin)) (the folded Mat expression), back. */
void pmo_normalize(const float* bgr, int rows, int cols, float* out) {
  const size_t n = (size_t)rows * cols;
  float* V = (float*)malloc(sizeof(float) * n);
  for (size_t i = 0; i < n; ++i) {
    float h, s;
    bgr2hsv(bgr[i * 3], bgr[i * 3 + 1], bgr[i * 3 + 2], &h, &s, &V[i]);
  }
  double vmin, vmax;
  pmo_value_minmax_eighth(V, rows, cols, &vmin, &vmax);
  const float alpha = (float)(1.0 / (vmax - vmin)), beta = (float)(-vmin / (vmax - vmin));
  for (size_t i = 0; i < n; ++i) {
    float h, s, v;
    bgr2hsv(bgr[i * 3], bgr[i * 3 + 1], bgr[i * 3 + 2], &h, &s, &v);
    v = v * alpha + beta;
    hsv2bgr(h, s, v, &out[i * 3], &out[i * 3 + 1], &out[i * 3 + 2]);
  }
  free(V);
}

/* imaging::NormalizeColorIlluminant: Normalize(bgr / (2 * GaussianBlur(bgr))) */
void pmo_normalize_color_illuminant(const float* bgr, int rows, int cols, float* out) {
  const size_t n3 = (size_t)rows * cols * 3;
  const int ksize = next_odd_int(cols / 3);
  const double sigma = (float)ksize / 4.0f;
  float* il = (float*)malloc(sizeof(float) * n3);
  pmo_gaussian_blur(bgr, rows, cols, 3, ksize, sigma, il);
  for (size_t i = 0; i < n3; ++i) {
    const float d = il[i] * 2.0f;
    il[i] = d != 0.f ? bgr[i] / d : 0.f; /* cv::divide: x / 0 = 0 */
  }
  pmo_normalize(il, rows, cols, out);
  free(il);
}

/* The chain on an 8-bit BGR image: J (float BGR, optional) and the 8-bit gray image stereo consumes
 * (gray.convertTo(CV_8UC1, 255): saturate_cast<uchar>(rint(g * 255))). */
void pmo_stereo_ready(const uint8_t* bgr8, int rows, int cols, float* J_out, uint8_t* gray8) {
  const size_t n = (size_t)rows * cols;
  float* I = (float*)malloc(sizeof(float) * n * 3);
  float* J = (float*)malloc(sizeof(float) * n * 3);
  pmo_cast_3b_to_3f(bgr8, n * 3, I);
  /* enhance_test.cpp:69: Normalize(NormalizeColorIlluminant(I)) -- NormalizeColorIlluminant already ends with a
   * Normalize (normalization.cpp:184), so the value channel is stretched twice */
  pmo_normalize_color_illuminant(I, rows, cols, J);
  memcpy(I, J, sizeof(float) * n * 3);
  pmo_normalize(I, rows, cols, J);
  for (size_t i = 0; i < n; ++i) {
    float g = J[i * 3] * 0.114f;
    g = g + J[i * 3 + 1] * 0.587f;
    g = g + J[i * 3 + 2] * 0.299f;
    const float r = nearbyintf(g * 255.f);
    gray8[i] = (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
  }
  if (J_out) memcpy(J_out, J, sizeof(float) * n * 3);
  free(I);
  free(J);
}
