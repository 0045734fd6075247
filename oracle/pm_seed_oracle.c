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
  p->subpixel_corners = 0;
  p->subpix_winsize = 10;
  p->subpix_zerozone = -1;
  p->subpix_maxiters = 10;
  p->subpix_epsilon = 0.01f;
  p->subpixel_refinement = 0;
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

/* ---- cv::cornerSubPix (OpenCV 3.4 modules/imgproc/src/cornersubpix.cpp) on an 8-bit image ----------------------------
 * FeatureDetector::Detect refines the detected corners with it when subpixel_corners is set
 * (feature_detector.cpp:110-120), StereoMatcher::MatchRectified the match when subpixel_refinement is
 * (stereo_matcher.cpp:94-103).  Restated from the published algorithm like the other OpenCV primitives (parity
 * unpinned): Gaussian-like window mask in float, the (2 win + 3)^2 neighbourhood sampled by getRectSubPix 8u -> 32f,
 * central differences in float, the 2 x 2 normal equations accumulated in double in raster order, at most max_iters
 * steps, stop when the step is below eps; a corner that moved further than the window keeps its start. */
static inline int sp_floor(float v) {
  int i = (int)v;
  return i - (v < (float)i);
}
/* getRectSubPix 8u -> 32f: getRectSubPix_8u32f of samplers.cpp (window inside the image: the running form with
 * a = max(a, 0.0001) and prev = t * ((1 - a) / a)), else getRectSubPix_Cn_<uchar, float, float> with adjustRect. */
static void sp_get_rect_8u32f(const uint8_t* src, int rows, int cols, int ww, int wh, float cx, float cy, float* dst) {
  cx -= (float)(ww - 1) * 0.5f;
  cy -= (float)(wh - 1) * 0.5f;
  const int ipx = sp_floor(cx), ipy = sp_floor(cy);
  if (0 <= ipx && ipx + ww < cols && 0 <= ipy && ipy + wh < rows && ww > 0 && wh > 0) {
    float a = cx - (float)ipx;
    const float b = cy - (float)ipy;
    a = a > 0.0001f ? a : 0.0001f;
    const float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
    const double s = (1. - (double)a) / (double)a;
    const uint8_t* r = src + (size_t)ipy * cols + ipx;
    for (int i = 0; i < wh; ++i, r += cols, dst += ww) {
      const float t0 = b1 * (float)r[0], t1 = b2 * (float)r[cols];
      float prev = (1.f - a) * (t0 + t1);
      for (int j = 0; j < ww; ++j) {
        const float u0 = a12 * (float)r[j + 1], u1 = a22 * (float)r[j + 1 + cols];
        const float t = u0 + u1;
        dst[j] = prev + t;
        prev = (float)((double)t * s);
      }
    }
    return;
  }
  const float a = cx - (float)ipx, b = cy - (float)ipy;
  const float ia = 1.f - a, ib = 1.f - b;
  const float a11 = ia * ib, a12 = a * ib, a21 = ia * b, a22 = a * b, b1 = ib, b2 = b;
  /* adjustRect (samplers.cpp): clip the window, replicate the border */
  int rx, ry, rw, rh;
  ptrdiff_t off = 0;
  if (ipx >= 0) { off += ipx; rx = 0; } else { rx = -ipx; if (rx > ww) rx = ww; }
  if (ipx < cols - ww) rw = ww; else { rw = cols - ipx - 1; if (rw < 0) { off += rw; rw = 0; } }
  if (ipy >= 0) { off += (ptrdiff_t)ipy * cols; ry = 0; } else ry = -ipy;
  if (ipy < rows - wh) rh = wh; else { rh = rows - ipy - 1; if (rh < 0) { off += (ptrdiff_t)rh * cols; rh = 0; } }
  const uint8_t* r = src + (off - rx);
  for (int i = 0; i < wh; ++i, dst += ww) {
    const uint8_t* r2 = r + cols;
    if (i < ry || i >= rh) r2 -= cols;
    float s0 = (float)r[rx] * b1 + (float)r2[rx] * b2;
    for (int j = 0; j < rx; ++j) dst[j] = s0;
    s0 = (float)r[rw] * b1 + (float)r2[rw] * b2;
    for (int j = rw; j < ww; ++j) dst[j] = s0;
    for (int j = rx; j < rw; ++j) {
      float v = (float)r[j] * a11;
      v = v + (float)r[j + 1] * a12;
      v = v + (float)r2[j] * a21;
      v = v + (float)r2[j + 1] * a22;
      dst[j] = v;
    }
    if (i < rh) r = r2;
  }
}
void pmo_subpix_mask(int win, int zero_zone, float* mask) {
  const int ww = 2 * win + 1;
  for (int i = 0; i < ww; ++i) {
    const float y = (float)(i - win) / (float)win;
    const float vy = expf(-y * y);
    for (int j = 0; j < ww; ++j) {
      const float x = (float)(j - win) / (float)win;
      mask[i * ww + j] = (float)(vy * expf(-x * x));
    }
  }
  if (zero_zone >= 0 && zero_zone * 2 + 1 < ww)
    for (int i = win - zero_zone; i <= win + zero_zone; ++i)
      for (int j = win - zero_zone; j <= win + zero_zone; ++j) mask[i * ww + j] = 0.f;
}
void pmo_corner_subpix(const uint8_t* img, int rows, int cols, float* xs, float* ys, int n, int win, int zero_zone,
                       int max_iters, double eps) {
  const int ww = 2 * win + 1, bw = ww + 2;
  max_iters = max_iters < 1 ? 1 : (max_iters > 100 ? 100 : max_iters);
  eps = eps > 0. ? eps : 0.;
  eps *= eps;
  float* mask = (float*)malloc(sizeof(float) * (size_t)ww * ww);
  float* buf = (float*)malloc(sizeof(float) * (size_t)bw * bw);
  pmo_subpix_mask(win, zero_zone, mask);
  for (int p = 0; p < n; ++p) {
    const float tx = xs[p], ty = ys[p];
    float ix = tx, iy = ty;
    int iter = 0;
    double err = 0.;
    do {
      double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
      sp_get_rect_8u32f(img, rows, cols, bw, bw, ix, iy, buf);
      const float* sp = buf + bw + 1;
      for (int i = 0, k = 0; i < ww; ++i, sp += bw) {
        const double py = i - win;
        for (int j = 0; j < ww; ++j, ++k) {
          const double m = mask[k];
          const float fgx = sp[j + 1] - sp[j - 1], fgy = sp[j + bw] - sp[j - bw];
          const double tgx = fgx, tgy = fgy;
          const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
          const double px = j - win;
          a += gxx;
          b += gxy;
          c += gyy;
          bb1 += gxx * px + gxy * py;
          bb2 += gxy * px + gyy * py;
        }
      }
      const double det = a * c - b * b;
      if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
      const double scale = 1.0 / det;
      const float nx = (float)((double)ix + c * scale * bb1 - b * scale * bb2);
      const float ny = (float)((double)iy - b * scale * bb1 + a * scale * bb2);
      const float dxs = nx - ix, dys = ny - iy;
      const float e0 = dxs * dxs, e1 = dys * dys;
      err = (double)(e0 + e1);
      ix = nx;
      iy = ny;
      if (ix < 0 || ix >= (float)cols || iy < 0 || iy >= (float)rows) break;
    } while (++iter < max_iters && err > eps);
    if (fabsf(ix - tx) > (float)win || fabsf(iy - ty) > (float)win) {
      ix = tx;
      iy = ty;
    }
    xs[p] = ix;
    ys[p] = iy;
  }
  free(mask);
  free(buf);
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
  const int mx = bx + sx + (tc - 1) / 2 + offset_x;
  float mpx = (float)mx;
  if (p->subpixel_refinement) { /* stereo_matcher.cpp:94-103: cornerSubPix on the right image, 10 x 10, 40 steps, 0.001 */
    float mpy = (float)(by + sy + (tr - 1) / 2);
    pmo_corner_subpix(right, rows, cols, &mpx, &mpy, 1, 10, -1, 40, 0.001);
  }
  if ((double)best < p->max_matching_cost && kx >= mpx) return (double)(float)(kx - mpx);
  return -1.0;
}

/* FeatureDetector::Detect (feature_detector.cpp:89-122): GFTT corners, optionally refined by cornerSubPix. */
static int detect_corners(const uint8_t* img, int rows, int cols, const pmo_seed_params* p, float* fx, float* fy) {
  const int cap = p->max_features > 0 ? p->max_features : 1;
  int* xs = (int*)malloc(sizeof(int) * (size_t)cap);
  int* ys = (int*)malloc(sizeof(int) * (size_t)cap);
  const int cnt = pmo_gftt_detect(img, rows, cols, p, xs, ys, p->max_features);
  for (int i = 0; i < cnt; ++i) {
    fx[i] = (float)xs[i];
    fy[i] = (float)ys[i];
  }
  if (p->subpixel_corners)
    pmo_corner_subpix(img, rows, cols, fx, fy, cnt, p->subpix_winsize, p->subpix_zerozone, p->subpix_maxiters,
                      (double)p->subpix_epsilon);
  free(xs);
  free(ys);
  return cnt;
}

void pmo_sparse_init(const uint8_t* left, const uint8_t* right, int rows, int cols, int dilate_factor,
                     const pmo_seed_params* p, float* seed) {
  const size_t n = (size_t)rows * cols;
  float* xs = (float*)malloc(sizeof(float) * (size_t)(p->max_features > 0 ? p->max_features : 1));
  float* ys = (float*)malloc(sizeof(float) * (size_t)(p->max_features > 0 ? p->max_features : 1));
  const int cnt = detect_corners(left, rows, cols, p, xs, ys);
  float* sparse = (float*)calloc(n, sizeof(float));
  for (int i = 0; i < cnt; ++i) {
    const float d = (float)pmo_match_rectified(left, right, rows, cols, xs[i], ys[i], p);
    /* disps.at<float>(std::round(kp.y), std::round(kp.x)) = d (patchmatch_gpu.cu:431) */
    const int ry = (int)roundf(ys[i]), rx = (int)roundf(xs[i]);
    if (d >= 0 && ry >= 0 && ry < rows && rx >= 0 && rx < cols) sparse[(size_t)ry * cols + rx] = d;
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
  float* xs = (float*)malloc(sizeof(float) * (size_t)(p->max_features > 0 ? p->max_features : 1));
  float* ys = (float*)malloc(sizeof(float) * (size_t)(p->max_features > 0 ? p->max_features : 1));
  const int cnt = detect_corners(left, rows, cols, p, xs, ys);
  float* sparse = (float*)calloc(n, sizeof(float));
  float* dil = (float*)malloc(sizeof(float) * n);
  for (int i = 0; i < cnt; ++i) {
    const float d = (float)pmo_match_rectified(left, right, rows, cols, xs[i], ys[i], p);
    const int ry = (int)roundf(ys[i]), rx = (int)roundf(xs[i]); /* patchmatch.cpp:68-71 */
    if (d >= 0 && ry >= 0 && ry < rows && rx >= 0 && rx < cols) sparse[(size_t)ry * cols + rx] = d;
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
