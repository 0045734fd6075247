/*
 * pm_seed_oracle.c -- CPU restatement of the sparse seeder (TEST INFRASTRUCTURE, parity unpinned; see
 * pm_oracle.h).  Reference: src/vehicle/feature_tracking/feature_detector.cpp:44-57,89-122 (GFTT through
 * cv::GFTTDetector), stereo_matcher.cpp:22-116 (cv::matchTemplate TM_SQDIFF_NORMED + minMaxLoc) and
 * src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:414-442 (scatter + dilate).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "pm_oracle.h"

void pmo_seed_params_default(pmo_seed_params* p) {
  p->max_features = 200;
  p->min_distance = 20;
  p->quality_level = 0.01;
  p->block_size = 5;
  p->templ_cols = 31;
  p->templ_rows = 11;
  p->max_disp = 128;
  p->max_matching_cost = 0.15;
  p->use_harris = 0;
  p->harris_k = 0.04;
}

static inline int refl101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

/* Sobel 3x3 derivatives (integers), REFLECT_101. */
static void sobel_xy(const uint8_t* im, int rows, int cols, int* dx, int* dy) {
  for (int y = 0; y < rows; ++y) {
    const uint8_t* r0 = im + (size_t)refl101(y - 1, rows) * cols;
    const uint8_t* r1 = im + (size_t)y * cols;
    const uint8_t* r2 = im + (size_t)refl101(y + 1, rows) * cols;
    for (int x = 0; x < cols; ++x) {
      const int xm = refl101(x - 1, cols), xp = refl101(x + 1, cols);
      dx[(size_t)y * cols + x] = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
      dy[(size_t)y * cols + x] = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
    }
  }
}

/* Corner response of cv::goodFeaturesToTrack: cornerMinEigenVal, or cornerHarris when use_harris is set
 * (feature_detector.cpp:44-57 hands gftt_use_harris_corner_detector / gftt_k to cv::GFTTDetector::create). */
void pmo_corner_response_map(const uint8_t* img, int rows, int cols, int block_size, int use_harris, double harris_k,
                             float* eig) {
  const size_t n = (size_t)rows * cols;
  int* dx = (int*)malloc(sizeof(int) * n);
  int* dy = (int*)malloc(sizeof(int) * n);
  sobel_xy(img, rows, cols, dx, dy);
  const int h = block_size / 2;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      long long sxx = 0, sxy = 0, syy = 0;
      for (int j = -h; j <= h; ++j) {
        const size_t ro = (size_t)refl101(y + j, rows) * cols;
        for (int i = -h; i <= h; ++i) {
          const size_t o = ro + refl101(x + i, cols);
          const long long gx = dx[o], gy = dy[o];
          sxx += gx * gx;
          sxy += gx * gy;
          syy += gy * gy;
        }
      }
      if (use_harris) {
        /* calcHarris (OpenCV 3.4 imgproc/corner.cpp): float a = cov0, b = cov1, c = cov2;
         * dst = (float)(a*c - b*b - k*(a + c)*(a + c)) -- a*c, b*b and their difference are binary32 operations,
         * k is a double, so the trace term and the final difference are binary64 */
        const float a = (float)sxx, b = (float)sxy, c = (float)syy;
        const float ac = a * c, bb = b * b;
        const float det = ac - bb;
        const float tr = a + c;
        const double kt = harris_k * (double)tr;
        const double ktt = kt * (double)tr;
        eig[(size_t)y * cols + x] = (float)((double)det - ktt);
      } else {
        /* calcMinEigenVal: a = cov0*0.5, b = cov1, c = cov2*0.5; (a+c) - sqrt((a-c)^2 + b^2) */
        const float a = (float)sxx * 0.5f, b = (float)sxy, c = (float)syy * 0.5f;
        const float t = a - c;
        const float tt = t * t, bb = b * b;
        const float s = a + c;
        eig[(size_t)y * cols + x] = s - sqrtf(tt + bb);
      }
    }
  free(dx);
  free(dy);
}

void pmo_min_eig_map(const uint8_t* img, int rows, int cols, int block_size, float* eig) {
  pmo_corner_response_map(img, rows, cols, block_size, 0, 0.0, eig);
}

typedef struct { float v; int idx; } cand_t;
/* greaterThanPtr of cv::goodFeaturesToTrack: larger value first, ties: larger address (raster index) first */
static int cand_cmp(const void* pa, const void* pb) {
  const cand_t* a = (const cand_t*)pa;
  const cand_t* b = (const cand_t*)pb;
  if (a->v > b->v) return -1;
  if (a->v < b->v) return 1;
  return a->idx > b->idx ? -1 : (a->idx < b->idx ? 1 : 0);
}

int pmo_gftt_detect(const uint8_t* img, int rows, int cols, const pmo_seed_params* p, int* xs, int* ys, int cap) {
  const size_t n = (size_t)rows * cols;
  float* eig = (float*)malloc(sizeof(float) * n);
  pmo_corner_response_map(img, rows, cols, p->block_size, p->use_harris, p->harris_k, eig);
  float maxv = 0.f; /* the response is >= 0 up to rounding; a non-positive maximum means "no corners" */
  for (size_t i = 0; i < n; ++i)
    if (eig[i] > maxv) maxv = eig[i];
  /* threshold(eig, maxVal*qualityLevel, 0, THRESH_TOZERO) on a 32F image */
  const float thr = (float)((double)maxv * p->quality_level);
  for (size_t i = 0; i < n; ++i)
    if (!(eig[i] > thr)) eig[i] = 0.f;
  cand_t* cand = (cand_t*)malloc(sizeof(cand_t) * n);
  size_t nc = 0;
  for (int y = 1; y < rows - 1; ++y)
    for (int x = 1; x < cols - 1; ++x) {
      const float v = eig[(size_t)y * cols + x];
      if (v == 0.f) continue;
      float m = v; /* 3x3 dilate */
      for (int j = -1; j <= 1; ++j)
        for (int i = -1; i <= 1; ++i) {
          const float u = eig[(size_t)(y + j) * cols + (x + i)];
          if (u > m) m = u;
        }
      if (v == m) {
        cand[nc].v = v;
        cand[nc].idx = y * cols + x;
        ++nc;
      }
    }
  qsort(cand, nc, sizeof(cand_t), cand_cmp);
  int count = 0;
  const long long md2 = (long long)p->min_distance * p->min_distance;
  const int limit = p->max_features < cap ? p->max_features : cap;
  for (size_t k = 0; k < nc && count < limit; ++k) {
    const int y = cand[k].idx / cols, x = cand[k].idx - y * cols;
    int good = 1;
    if (p->min_distance >= 1)
      for (int j = 0; j < count; ++j) {
        const long long ddx = x - xs[j], ddy = y - ys[j];
        if (ddx * ddx + ddy * ddy < md2) {
          good = 0;
          break;
        }
      }
    if (good) {
      xs[count] = x;
      ys[count] = y;
      ++count;
    }
  }
  free(cand);
  free(eig);
  return count;
}

double pmo_match_rectified(const uint8_t* left, const uint8_t* right, int rows, int cols, float kx, float ky,
                           const pmo_seed_params* p) {
  const int tc = p->templ_cols, tr = p->templ_rows, md = p->max_disp;
  const int stripe_rows = tr + 2;
  const int rx = (int)roundf(kx), ry = (int)roundf(ky);
  int ty = ry - (tr - 1) / 2;
  if (ty < 0 || ty + tr >= rows) return -1.0;
  int offset_x = 0;
  int tx = rx - (tc - 1) / 2;
  if (tx < 0) {
    offset_x = tx;
    tx = 0;
  }
  if (tx + tc >= cols) {
    if (offset_x != 0) return -1.0; /* LOG(FATAL) in the reference */
    offset_x = (tx + tc) - (cols - 1);
    tx -= offset_x;
  }
  const int sy = ry - (stripe_rows - 1) / 2;
  if (sy < 0 || sy + stripe_rows >= rows) return -1.0;
  int sx = rx + (tc - 1) / 2 - md;
  if (sx + md > cols - 1) sx -= (sx + md) - (cols - 1);
  if (sx < 0) sx = 0;
  if (sx + md > cols || tx < 0) return -1.0; /* the cv::Mat ROI of the reference would throw */
  const int rw = md - tc + 1, rh = stripe_rows - tr + 1;
  long long t2 = 0;
  for (int j = 0; j < tr; ++j)
    for (int i = 0; i < tc; ++i) {
      const long long t = left[(size_t)(ty + j) * cols + tx + i];
      t2 += t * t;
    }
  float best = 0.f;
  int bx = 0, by = 0, have = 0;
  for (int v = 0; v < rh; ++v)
    for (int u = 0; u < rw; ++u) {
      long long num = 0, i2 = 0;
      for (int j = 0; j < tr; ++j) {
        const uint8_t* T = left + (size_t)(ty + j) * cols + tx;
        const uint8_t* I = right + (size_t)(sy + v + j) * cols + sx + u;
        for (int i = 0; i < tc; ++i) {
          const long long d = (long long)T[i] - (long long)I[i];
          num += d * d;
          i2 += (long long)I[i] * I[i];
        }
      }
      /* TM_SQDIFF_NORMED = sum (T-I)^2 / sqrt(sum T^2 * sum I^2); 1 when the denominator vanishes */
      const double den = sqrt((double)t2 * (double)i2);
      const float r = den > 0.0 ? (float)((double)num / den) : 1.f;
      if (!have || r < best) { /* minMaxLoc: first minimum in row-major order */
        best = r;
        bx = u;
        by = v;
        have = 1;
      }
    }
  (void)by;
  const int mx = bx + sx + (tc - 1) / 2 + offset_x;
  if ((double)best < p->max_matching_cost && kx >= (float)mx) return (double)(float)(kx - (float)mx);
  return -1.0;
}

void pmo_sparse_init(const uint8_t* left, const uint8_t* right, int rows, int cols, int dilate_factor,
                     const pmo_seed_params* p, float* seed) {
  const size_t n = (size_t)rows * cols;
  int* xs = (int*)malloc(sizeof(int) * (size_t)(p->max_features > 0 ? p->max_features : 1));
  int* ys = (int*)malloc(sizeof(int) * (size_t)(p->max_features > 0 ? p->max_features : 1));
  const int cnt = pmo_gftt_detect(left, rows, cols, p, xs, ys, p->max_features);
  float* sparse = (float*)calloc(n, sizeof(float));
  for (int i = 0; i < cnt; ++i) {
    const float d = (float)pmo_match_rectified(left, right, rows, cols, (float)xs[i], (float)ys[i], p);
    if (d >= 0) sparse[(size_t)ys[i] * cols + xs[i]] = d;
  }
  const int k = (int)pow(2.0, (double)dilate_factor) + 1; /* patchmatch_gpu.cu:436 */
  pmo_dilate_rect(sparse, seed, rows, cols, k);
  free(sparse);
  free(xs);
  free(ys);
}

/* Patchmatch::Initialize (src/vehicle/stereo_matching/patchmatch.cpp:52-87): the scatter of SparseInit, dilation with
 * dilate_size = (int)pow(2, f - 1) + 1 (:75), cv::resize(INTER_NEAREST) to size / f (:79; OpenCV 3.4
 * resizeNN_: sx = min(cvFloor(x * ifx), cols - 1) with ifx = 1 / ((double)dst_cols / src_cols)) and
 * disps /= pow(2, f) (:81 -- 2^f although the map shrinks by f: quirk Q1, reproduced).  out: (rows/f) x (cols/f). */
void pmo_cpu_initialize(const uint8_t* left, const uint8_t* right, int rows, int cols, int downsample_factor,
                        const pmo_seed_params* p, float* out) {
  const size_t n = (size_t)rows * cols;
  const int f = downsample_factor;
  int* xs = (int*)malloc(sizeof(int) * (size_t)(p->max_features > 0 ? p->max_features : 1));
  int* ys = (int*)malloc(sizeof(int) * (size_t)(p->max_features > 0 ? p->max_features : 1));
  const int cnt = pmo_gftt_detect(left, rows, cols, p, xs, ys, p->max_features);
  float* sparse = (float*)calloc(n, sizeof(float));
  float* dil = (float*)malloc(sizeof(float) * n);
  for (int i = 0; i < cnt; ++i) {
    const float d = (float)pmo_match_rectified(left, right, rows, cols, (float)xs[i], (float)ys[i], p);
    if (d >= 0) sparse[(size_t)ys[i] * cols + xs[i]] = d;
  }
  const int k = (int)pow(2.0, (double)(f - 1)) + 1;
  pmo_dilate_rect(sparse, dil, rows, cols, k);
  const int orows = rows / f, ocols = cols / f;
  const double ifx = 1.0 / ((double)ocols / (double)cols), ify = 1.0 / ((double)orows / (double)rows);
  const double div = pow(2.0, (double)f);
  for (int y = 0; y < orows; ++y) {
    int sy = (int)floor((double)y * ify);
    if (sy > rows - 1) sy = rows - 1;
    for (int x = 0; x < ocols; ++x) {
      int sx = (int)floor((double)x * ifx);
      if (sx > cols - 1) sx = cols - 1;
      out[(size_t)y * ocols + x] = (float)((double)dil[(size_t)sy * cols + sx] / div);
    }
  }
  free(dil);
  free(sparse);
  free(xs);
  free(ys);
}
