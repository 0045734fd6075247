/*
 * pm_oracle.c -- CPU restatement of the reference PatchMatch stereo path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see pm_oracle.h for the full statement.
 *
 * Build: gcc -O3 -march=native -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 * -O3 -march=native are the reference's flags (CMakeLists.txt:17-25); -ffp-contract=off makes
 * every float operation a single IEEE rounding so the HIP kernels can match bit for bit.
 *
 * All `reference file:line` citations are relative to /root/reference.
 */
#include "pm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================================== */
/* OpenCV 3.4 primitives                                                                     */
/* ======================================================================================== */

/* cvRound(double/float): round half to even (SSE2 cvtsd2si / lrint under the default mode). */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_floor_f(float v) {
  int i = (int)v;
  return i - (v < (float)i);
}
/* saturate_cast<uchar>(float): cvRound then clamp to [0,255] (core/saturate.hpp). */
static inline uint8_t sat_u8_f(float v) {
  int iv = cv_round_f(v);
  return (uint8_t)((unsigned)iv <= 255u ? iv : (iv > 0 ? 255 : 0));
}

/* cv::RNG: multiply-with-carry, core/core.hpp RNG::next():
 *   state = (uint64)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state */
#define PMO_RNG_COEFF 4164903690u
static inline uint32_t rng_next(uint64_t* state) {
  *state = (uint64_t)(uint32_t)(*state) * PMO_RNG_COEFF + (uint32_t)(*state >> 32);
  return (uint32_t)(*state);
}

void pmo_rng_raw(uint32_t* dst, size_t n, uint64_t seed) {
  uint64_t state = seed ? seed : 0xffffffffu; /* RNG::RNG(uint64 _state) */
  for (size_t i = 0; i < n; ++i) dst[i] = rng_next(&state);
}

/* RNG::fill(CV_32F, UNIFORM, a, b) -> randf_32f (modules/core/src/rand.cpp):
 *   scale = (float)((b - a) * 2^-32), shift = (float)((a + b) * 0.5)
 *   out[i] = (float)(int)next() * scale + shift     (single precision, mul then add)
 * over the elements of a continuous matrix in row-major order. */
void pmo_rng_fill_uniform(float* dst, size_t n, double lo, double hi, uint64_t seed) {
  uint64_t state = seed ? seed : 0xffffffffu;
  const float scale = (float)((hi - lo) * 2.3283064365386962890625e-10);
  const float shift = (float)((hi + lo) * 0.5);
  for (size_t i = 0; i < n; ++i) {
    const float t = (float)(int32_t)rng_next(&state);
    const float m = t * scale;
    dst[i] = m + shift;
  }
}

/* cv::borderInterpolate(p, len, BORDER_REFLECT_101). */
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

/* ComputeGradient (test/stereo_matching/patchmatch_test.cpp:48-64):
 *   Dx = Sobel(im, CV_32F, 1, 0, 3), Dy = Sobel(im, CV_32F, 0, 1, 3)   kernels [-1 0 1]x[1 2 1]^T,
 *   unnormalised, BORDER_DEFAULT = REFLECT_101;  G = sqrt(Dx^2 + Dy^2).
 * All intermediate values are integers < 2^24, hence exact in binary32; the only rounding is sqrtf.
 * GradientMagnitude (src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:307-319) computes the same values
 * (cu::createSobelFilter ksize 3 + cu::magnitude on the u8->f32 converted image). */
void pmo_gradient_magnitude(const uint8_t* im, int rows, int cols, float* g) {
  for (int y = 0; y < rows; ++y) {
    const uint8_t* r0 = im + (size_t)reflect101(y - 1, rows) * cols;
    const uint8_t* r1 = im + (size_t)y * cols;
    const uint8_t* r2 = im + (size_t)reflect101(y + 1, rows) * cols;
    for (int x = 0; x < cols; ++x) {
      const int xm = reflect101(x - 1, cols), xp = reflect101(x + 1, cols);
      const int dx = (r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
      const int dy = (r2[xm] - r0[xm]) + 2 * (r2[x] - r0[x]) + (r2[xp] - r0[xp]);
      const float fx = (float)dx, fy = (float)dy;
      const float sx = fx * fx, sy = fy * fy;
      g[(size_t)y * cols + x] = sqrtf(sx + sy);
    }
  }
}

/* cv::dilate, rectangular (2k+1)^2 element anchored at its centre; samples outside the image are
 * ignored (morphologyDefaultBorderValue).  Used by Patchmatch::Initialize (patchmatch.cpp:75-78)
 * and PatchmatchGpu::SparseInit (patchmatch_gpu.cu:436-439). */
void pmo_dilate_rect(const float* src, float* dst, int rows, int cols, int k) {
  float* tmp = (float*)malloc(sizeof(float) * (size_t)rows * cols);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int x0 = x - k < 0 ? 0 : x - k, x1 = x + k >= cols ? cols - 1 : x + k;
      float m = src[(size_t)y * cols + x0];
      for (int xx = x0 + 1; xx <= x1; ++xx) {
        const float v = src[(size_t)y * cols + xx];
        if (v > m) m = v;
      }
      tmp[(size_t)y * cols + x] = m;
    }
  for (int y = 0; y < rows; ++y) {
    int y0 = y - k < 0 ? 0 : y - k, y1 = y + k >= rows ? rows - 1 : y + k;
    for (int x = 0; x < cols; ++x) {
      float m = tmp[(size_t)y0 * cols + x];
      for (int yy = y0 + 1; yy <= y1; ++yy) {
        const float v = tmp[(size_t)yy * cols + x];
        if (v > m) m = v;
      }
      dst[(size_t)y * cols + x] = m;
    }
  }
  free(tmp);
}

/* cv::resize INTER_LINEAR, CV_8UC1 (pm_oracle.h).  OpenCV 3.4 modules/imgproc/src/resize.cpp: resize() -> the
 * is_area_fast shortcut for an exact 2x2 shrink (ResizeAreaFast_Invoker / ResizeAreaFastVec for uchar), else
 * resizeGeneric_ with HResizeLinear<uchar, int, short, INTER_RESIZE_COEF_SCALE> and the uchar VResizeLinear. */
void pmo_resize_linear_u8(const uint8_t* src, int rows, int cols, uint8_t* dst, int drows, int dcols) {
  if (drows * 2 == rows && dcols * 2 == cols) { /* inv_scale == 2 exactly in both axes */
    for (int y = 0; y < drows; ++y) {
      const uint8_t* s0 = src + (size_t)(2 * y) * cols;
      const uint8_t* s1 = s0 + cols;
      for (int x = 0; x < dcols; ++x)
        dst[(size_t)y * dcols + x] = (uint8_t)((s0[2 * x] + s0[2 * x + 1] + s1[2 * x] + s1[2 * x + 1] + 2) >> 2);
    }
    return;
  }
  const double scale_x = (double)cols / dcols, scale_y = (double)rows / drows;
  int* xofs = (int*)malloc(sizeof(int) * (size_t)dcols);
  short* alpha = (short*)malloc(sizeof(short) * 2 * (size_t)dcols);
  int* hbuf[2];
  hbuf[0] = (int*)malloc(sizeof(int) * (size_t)dcols);
  hbuf[1] = (int*)malloc(sizeof(int) * (size_t)dcols);
  for (int dx = 0; dx < dcols; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floor(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= cols - 1) { fx = 0.f; sx = cols - 1; }
    xofs[dx] = sx;
    /* saturate_cast<short>((1 - fx) * 2048), saturate_cast<short>(fx * 2048): cvRound, ties to even */
    alpha[2 * dx] = (short)lrintf((1.f - fx) * 2048.f);
    alpha[2 * dx + 1] = (short)lrintf(fx * 2048.f);
  }
  for (int dy = 0; dy < drows; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floor(fy);
    fy -= (float)sy; /* the vertical pass keeps its weights at the border and clamps the row indices instead */
    const short b0 = (short)lrintf((1.f - fy) * 2048.f), b1 = (short)lrintf(fy * 2048.f);
    for (int k = 0; k < 2; ++k) {
      int r = sy + k;
      r = r < 0 ? 0 : (r > rows - 1 ? rows - 1 : r);
      const uint8_t* srow = src + (size_t)r * cols;
      for (int dx = 0; dx < dcols; ++dx) {
        const int sx = xofs[dx], sx1 = sx + 1 < cols ? sx + 1 : cols - 1;
        hbuf[k][dx] = srow[sx] * alpha[2 * dx] + srow[sx1] * alpha[2 * dx + 1];
      }
    }
    for (int dx = 0; dx < dcols; ++dx)
      dst[(size_t)dy * dcols + dx] =
          (uint8_t)((((b0 * (hbuf[0][dx] >> 4)) >> 16) + ((b1 * (hbuf[1][dx] >> 4)) >> 16) + 2) >> 2);
  }
  free(xofs);
  free(alpha);
  free(hbuf[0]);
  free(hbuf[1]);
}

/* ForegroundTextureMask (src/vehicle/stereo_matching/patchmatch.cpp:19-49).  cv::morphologyEx(MORPH_GRADIENT) with a
 * (2k+1)^2 rectangle = dilate - erode; the default border value never wins a min or a max, i.e. the window is clipped at
 * the image border.  `gradient > min_grad` compares the 8-bit gradient with a double and gives 255 / 0. */
static void morph_gradient_u8(const uint8_t* src, int rows, int cols, int k, uint8_t* grad) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      int mn = 255, mx = 0;
      for (int yy = y - k < 0 ? 0 : y - k; yy <= (y + k > rows - 1 ? rows - 1 : y + k); ++yy)
        for (int xx = x - k < 0 ? 0 : x - k; xx <= (x + k > cols - 1 ? cols - 1 : x + k); ++xx) {
          const int v = src[(size_t)yy * cols + xx];
          if (v < mn) mn = v;
          if (v > mx) mx = v;
        }
      grad[(size_t)y * cols + x] = (uint8_t)(mx - mn);
    }
}
int pmo_foreground_texture_mask(const uint8_t* gray, int rows, int cols, int ksize, double min_grad, int downsize,
                                uint8_t* mask) {
  if (downsize < 1 || downsize > 8) return -1; /* CHECK at patchmatch.cpp:25 */
  const int k = ksize / downsize;
  if (k <= 1) return -1;                       /* CHECK_GT(scaled_ksize, 1) at :27 */
  if (downsize > 1) {
    const int srows = rows / downsize, scols = cols / downsize; /* gray.size() / downsize: integer division */
    const size_t n = (size_t)srows * scols;
    uint8_t* small = (uint8_t*)malloc(n);
    uint8_t* grad = (uint8_t*)malloc(n);
    pmo_resize_linear_u8(gray, rows, cols, small, srows, scols);
    morph_gradient_u8(small, srows, scols, k, grad);
    for (size_t i = 0; i < n; ++i) grad[i] = (double)grad[i] > min_grad ? 255 : 0;
    pmo_resize_linear_u8(grad, srows, scols, mask, rows, cols);
    free(small);
    free(grad);
  } else {
    morph_gradient_u8(gray, rows, cols, k, mask);
    for (size_t i = 0; i < (size_t)rows * cols; ++i) mask[i] = (double)mask[i] > min_grad ? 255 : 0;
  }
  return 0;
}

void pmo_flip_h_u8(const uint8_t* src, uint8_t* dst, int rows, int cols) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) dst[(size_t)y * cols + x] = src[(size_t)y * cols + (cols - 1 - x)];
}
void pmo_flip_h_f32(const float* src, float* dst, int rows, int cols) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) dst[(size_t)y * cols + x] = src[(size_t)y * cols + (cols - 1 - x)];
}

/* adjustRect of modules/imgproc/src/samplers.cpp: clips the window against the image and returns
 * the (possibly shifted) source origin as an element offset; r = {x, y, width, height} in window
 * coordinates with OpenCV's meaning (x..width is the bilinear span, the rest replicates). */
typedef struct { int x, y, width, height; } rect_i;
static ptrdiff_t adjust_rect(int step, int src_w, int src_h, int win_w, int win_h, int ipx, int ipy,
                             rect_i* pr) {
  rect_i rect;
  ptrdiff_t off = 0;
  if (ipx >= 0) {
    off += ipx;
    rect.x = 0;
  } else {
    rect.x = -ipx;
    if (rect.x > win_w) rect.x = win_w;
  }
  if (ipx < src_w - win_w) {
    rect.width = win_w;
  } else {
    rect.width = src_w - ipx - 1;
    if (rect.width < 0) {
      off += rect.width;
      rect.width = 0;
    }
  }
  if (ipy >= 0) {
    off += (ptrdiff_t)ipy * step;
    rect.y = 0;
  } else {
    rect.y = -ipy;
  }
  if (ipy < src_h - win_h) {
    rect.height = win_h;
  } else {
    rect.height = src_h - ipy - 1;
    if (rect.height < 0) {
      off += (ptrdiff_t)rect.height * step;
      rect.height = 0;
    }
  }
  *pr = rect;
  return off - rect.x;
}

/* cv::getRectSubPix 8u->8u = getRectSubPix_Cn_<uchar,uchar,int,scale_fixpt,cast_8u>
 * (modules/imgproc/src/samplers.cpp):  weights cvRound(w * 2^16), result (s + 2^15) >> 16;
 * window top-left = centre - (size-1)/2; replicate border through adjustRect.
 * Called by GetPatchSubpix (src/vehicle/stereo_matching/patchmatch.cpp:98-103). */
void pmo_get_rect_subpix_u8(const uint8_t* src, int rows, int cols, int pw, int ph, float cx,
                            float cy, uint8_t* dst) {
  cx -= (float)(pw - 1) * 0.5f;
  cy -= (float)(ph - 1) * 0.5f;
  const int ipx = cv_floor_f(cx), ipy = cv_floor_f(cy);
  const float a = cx - (float)ipx, b = cy - (float)ipy;
  const float ia = 1.f - a, ib = 1.f - b;
#define FIX(v) cv_round_f((v) * 65536.f)
  const int a11 = FIX(ia * ib), a12 = FIX(a * ib), a21 = FIX(ia * b), a22 = FIX(a * b);
  const int b1 = FIX(ib), b2 = FIX(b);
#undef FIX
#define C8(s) ((uint8_t)(((s) + (1 << 15)) >> 16))
  const int step = cols;
  if (0 <= ipx && ipx < cols - pw && 0 <= ipy && ipy < rows - ph) {
    const uint8_t* s = src + (size_t)ipy * step + ipx;
    for (int i = 0; i < ph; ++i, s += step, dst += pw)
      for (int j = 0; j < pw; ++j) {
        const int s0 = s[j] * a11 + s[j + 1] * a12 + s[j + step] * a21 + s[j + step + 1] * a22;
        dst[j] = C8(s0);
      }
  } else {
    rect_i r;
    const uint8_t* s = src + adjust_rect(step, cols, rows, pw, ph, ipx, ipy, &r);
    for (int i = 0; i < ph; ++i, dst += pw) {
      const uint8_t* s2 = s + step;
      if (i < r.y || i >= r.height) s2 -= step;
      int s0 = s[r.x] * b1 + s2[r.x] * b2;
      for (int j = 0; j < r.x; ++j) dst[j] = C8(s0);
      s0 = s[r.width] * b1 + s2[r.width] * b2;
      for (int j = r.width; j < pw; ++j) dst[j] = C8(s0);
      for (int j = r.x; j < r.width; ++j) {
        s0 = s[j] * a11 + s[j + 1] * a12 + s2[j] * a21 + s2[j + 1] * a22;
        dst[j] = C8(s0);
      }
      if (i < r.height) s = s2;
    }
  }
#undef C8
}

/* cv::getRectSubPix 32f->32f = getRectSubPix_Cn_<float,float,float,nop,nop>: float weights,
 * s = s00*a11 + s01*a12 + s10*a21 + s11*a22 evaluated left to right. */
void pmo_get_rect_subpix_f32(const float* src, int rows, int cols, int pw, int ph, float cx,
                             float cy, float* dst) {
  cx -= (float)(pw - 1) * 0.5f;
  cy -= (float)(ph - 1) * 0.5f;
  const int ipx = cv_floor_f(cx), ipy = cv_floor_f(cy);
  const float a = cx - (float)ipx, b = cy - (float)ipy;
  const float ia = 1.f - a, ib = 1.f - b;
  const float a11 = ia * ib, a12 = a * ib, a21 = ia * b, a22 = a * b;
  const float b1 = ib, b2 = b;
  const int step = cols;
  if (0 <= ipx && ipx < cols - pw && 0 <= ipy && ipy < rows - ph) {
    const float* s = src + (size_t)ipy * step + ipx;
    for (int i = 0; i < ph; ++i, s += step, dst += pw)
      for (int j = 0; j < pw; ++j) {
        float s0 = s[j] * a11;
        s0 = s0 + s[j + 1] * a12;
        s0 = s0 + s[j + step] * a21;
        s0 = s0 + s[j + step + 1] * a22;
        dst[j] = s0;
      }
  } else {
    rect_i r;
    const float* s = src + adjust_rect(step, cols, rows, pw, ph, ipx, ipy, &r);
    for (int i = 0; i < ph; ++i, dst += pw) {
      const float* s2 = s + step;
      if (i < r.y || i >= r.height) s2 -= step;
      float s0 = s[r.x] * b1 + s2[r.x] * b2;
      for (int j = 0; j < r.x; ++j) dst[j] = s0;
      s0 = s[r.width] * b1 + s2[r.width] * b2;
      for (int j = r.width; j < pw; ++j) dst[j] = s0;
      for (int j = r.x; j < r.width; ++j) {
        s0 = s[j] * a11;
        s0 = s0 + s[j + 1] * a12;
        s0 = s0 + s2[j] * a21;
        s0 = s0 + s2[j + 1] * a22;
        dst[j] = s0;
      }
      if (i < r.height) s = s2;
    }
  }
}

/* ======================================================================================== */
/* SEM_CPU                                                                                   */
/* ======================================================================================== */

/* cv::mean returns sum * (1./N) in double (modules/core/src/stat.cpp); the reference casts that to
 * float (patchmatch_test.cpp:25). */
static inline float mean_from_sum(double sum, int n) { return (float)(sum * (1. / (double)n)); }

/* alpha * ec + (1 - alpha) * eg (patchmatch_test.cpp:44) as this build DEFINES it: two products and a sum, three roundings.
 * The reference is compiled with g++'s default -ffp-contract=fast (CMakeLists.txt:17-25), so in its binary the expression
 * may be ONE fused multiply-add on top of one product -- which of the two products is fused is the compiler's choice --
 * and nothing in the reference pins it.  -DPMO_FUNCTOR_FMA=1 / =2 build those two contracted forms, for the sensitivity
 * count of tools/fp_contract_sensitivity.py only (profiles/r06_fp_contract_sensitivity.txt); the shipped oracle and the
 * engine use the uncontracted form. */
static inline float functor_mix(float alpha, float error_color, float error_grad) {
  const float one_minus = 1.f - alpha;
#if defined(PMO_FUNCTOR_FMA) && PMO_FUNCTOR_FMA == 1
  return fmaf(alpha, error_color, one_minus * error_grad);
#elif defined(PMO_FUNCTOR_FMA) && PMO_FUNCTOR_FMA == 2
  return fmaf(one_minus, error_grad, alpha * error_color);
#else
  const float t0 = alpha * error_color;
  const float t1 = one_minus * error_grad;
  return t0 + t1;
#endif
}

/* L1GradientCostFunction (test/stereo_matching/patchmatch_test.cpp:30-45).  The functor is declared
 * with Image1b gradient parameters while Patchmatch passes Image1f patches (patchmatch.hpp:18), so
 * each gradient patch goes through Mat::convertTo(CV_8U) = saturate_cast<uchar>(cvRound(v)) before
 * the L1 distance.  alpha*ec + (1-alpha)*eg is float arithmetic. */
float pmo_cpu_functor(const uint8_t* pl, const uint8_t* pr, const float* gl, const float* gr, int n,
                      const pmo_functor* f) {
  long sc = 0, sg = 0;
  for (int i = 0; i < n; ++i) {
    sc += abs((int)pl[i] - (int)pr[i]);
    sg += abs((int)sat_u8_f(gl[i]) - (int)sat_u8_f(gr[i]));
  }
  const float error_color = fminf(mean_from_sum((double)sc, n), f->tau_color);
  const float error_grad = fminf(mean_from_sum((double)sg, n), f->tau_grad);
  return functor_mix(f->alpha, error_color, error_grad);
}

#define PMO_MAX_PATCH (31 * 31)

/* The four GetPatchSubpix + functor calls of PropagateNeighbors (patchmatch.cpp:171-181). */
float pmo_cpu_cost_literal(const pmo_images* im, int pw, int ph, float x, float y, float d,
                           const pmo_functor* f) {
  uint8_t ref[PMO_MAX_PATCH], p0[PMO_MAX_PATCH];
  float gref[PMO_MAX_PATCH], g0[PMO_MAX_PATCH];
  pmo_get_rect_subpix_u8(im->il, im->rows, im->cols, pw, ph, x, y, ref);
  pmo_get_rect_subpix_f32(im->gl, im->rows, im->cols, pw, ph, x, y, gref);
  const float xr = x - d;
  pmo_get_rect_subpix_u8(im->ir, im->rows, im->cols, pw, ph, xr, y, p0);
  pmo_get_rect_subpix_f32(im->gr, im->rows, im->cols, pw, ph, xr, y, g0);
  return pmo_cpu_functor(ref, p0, gref, g0, pw * ph, f);
}

/* Same value without materialising patches.  On the path the window centre has an integer y
 * (b = 0) and 0 <= x-d-(pw-1)/2 <= cols-pw, so every sample is the two-tap horizontal lerp of
 * getRectSubPix; its border branch is reached only with a = 0 (last interior column / row) where
 * it degenerates to a plain copy.  Valid for pw/2 <= x <= cols-pw/2-1, ph/2 <= y <= rows-ph/2-1,
 * 0 <= d <= x - pw/2. */
float pmo_cpu_cost_direct(const pmo_images* im, int pw, int ph, int x, int y, float d,
                          const pmo_functor* f) {
  const int cols = im->cols;
  float cx = (float)x - d;
  cx -= (float)(pw - 1) * 0.5f;
  const int ipx = cv_floor_f(cx);
  const float a = cx - (float)ipx;
  const float ia = 1.f - a;
  const int a11 = cv_round_f(ia * 65536.f), a12 = cv_round_f(a * 65536.f);
  const int x0 = x - pw / 2, y0 = y - ph / 2;
  long sc = 0, sg = 0;
  for (int i = 0; i < ph; ++i) {
    const size_t row = (size_t)(y0 + i) * cols;
    const uint8_t* l = im->il + row + x0;
    const float* g = im->gl + row + x0;
    const uint8_t* r = im->ir + row;
    const float* gr = im->gr + row;
    for (int j = 0; j < pw; ++j) {
      const int c0 = ipx + j;
      const int c1 = c0 + 1 < cols ? c0 + 1 : cols - 1;
      const int v = (r[c0] * a11 + r[c1] * a12 + (1 << 15)) >> 16;
      sc += abs((int)l[j] - v);
      float s0 = gr[c0] * ia;
      s0 = s0 + gr[c1] * a;
      sg += abs((int)sat_u8_f(g[j]) - (int)sat_u8_f(s0));
    }
  }
  const int n = pw * ph;
  const float error_color = fminf(mean_from_sum((double)sc, n), f->tau_color);
  const float error_grad = fminf(mean_from_sum((double)sg, n), f->tau_grad);
  return functor_mix(f->alpha, error_color, error_grad);
}

static inline float cpu_cost(const pmo_images* im, int pw, int ph, int x, int y, float d,
                             const pmo_functor* f, int literal) {
  return literal ? pmo_cpu_cost_literal(im, pw, ph, (float)x, (float)y, d, f)
                 : pmo_cpu_cost_direct(im, pw, ph, x, y, d, f);
}

/* Patchmatch::AddNoise (patchmatch.cpp:143-155): a FRESH cv::RNG(123) per call, uniform
 * [-amount, amount) over the whole image, added where mask != 0 (everywhere if the mask is empty),
 * then max(disp, 0). */
void pmo_cpu_add_noise(float* disp, int rows, int cols, float amount, const uint8_t* mask,
                       uint64_t seed) {
  const size_t n = (size_t)rows * cols;
  float* noise = (float*)malloc(sizeof(float) * n);
  pmo_rng_fill_uniform(noise, n, -(double)amount, (double)amount, seed);
  for (size_t i = 0; i < n; ++i) {
    float v = disp[i];
    if (!mask || mask[i]) v = v + noise[i];
    disp[i] = v > 0.f ? v : 0.f; /* cv::max(disp, 0) */
  }
  free(noise);
}

/* PropagateNeighbors with (x_offset, y_offset) (patchmatch.cpp:158-196). */
static inline void propagate_neighbors(const pmo_images* im, float* disp, int x, int y, int pw,
                                       int ph, int xo, int yo, const pmo_functor* f, int literal) {
  const int cols = im->cols;
  float d0 = disp[(size_t)y * cols + x];
  /* d0 = fmin(fmax(d0, 0), (float)x - patch_width / 2) */
  const float hi = (float)x - (float)(pw / 2);
  d0 = d0 > 0.f ? d0 : 0.f;
  d0 = d0 < hi ? d0 : hi;
  const float dl = disp[(size_t)(y + yo) * cols + (x + xo)];
  const float c0 = cpu_cost(im, pw, ph, x, y, d0, f, literal);
  float best = d0;
  if (((float)x - dl) >= (float)(pw / 2)) {
    const float c1 = cpu_cost(im, pw, ph, x, y, dl, f, literal);
    if (c1 < c0) best = dl; /* Argmin: first strict minimum (patchmatch.cpp:115-126) */
  }
  disp[(size_t)y * cols + x] = best;
}

static inline int cpu_skip(int x, int y, int w, int h, int pw, int ph) {
  return y < (ph / 2) || x < (pw / 2) || y > (h - ph / 2 - 1) || x > (w - pw / 2 - 1);
}

/* Patchmatch::Propagate (patchmatch.cpp:248-311): four full-image passes,
 *   A raster  y=1..h-1, x=1..w-1, neighbour (-1, 0)      B raster,  neighbour (0,-1)
 *   C reverse y=h-2..0, x=w-2..0, neighbour (+1, 0)      D reverse, neighbour (0,+1)
 * each skipping a patch/2 border.  In A and C a pixel depends only on its predecessor in the same
 * row, in B and D only on its predecessor in the same column, so rows (columns) are independent and
 * may be run in any order / in parallel with identical results. */
void pmo_cpu_propagate(const pmo_images* im, float* disp, int ph, int pw, const pmo_functor* f,
                       int pass_mask, int literal, int nthreads) {
  const int w = im->cols, h = im->rows;
  if (nthreads < 1) nthreads = 1;
  if (pass_mask & 1) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int y = 1; y < h; ++y)
      for (int x = 1; x < w; ++x)
        if (!cpu_skip(x, y, w, h, pw, ph)) propagate_neighbors(im, disp, x, y, pw, ph, -1, 0, f, literal);
  }
  if (pass_mask & 2) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int x = 1; x < w; ++x)
      for (int y = 1; y < h; ++y)
        if (!cpu_skip(x, y, w, h, pw, ph)) propagate_neighbors(im, disp, x, y, pw, ph, 0, -1, f, literal);
  }
  if (pass_mask & 4) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int y = h - 2; y >= 0; --y)
      for (int x = w - 2; x >= 0; --x)
        if (!cpu_skip(x, y, w, h, pw, ph)) propagate_neighbors(im, disp, x, y, pw, ph, 1, 0, f, literal);
  }
  if (pass_mask & 8) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
    for (int x = w - 2; x >= 0; --x)
      for (int y = h - 2; y >= 0; --y)
        if (!cpu_skip(x, y, w, h, pw, ph)) propagate_neighbors(im, disp, x, y, pw, ph, 0, 1, f, literal);
  }
}

/* Patchmatch::RemoveBackground (patchmatch.cpp:314-360). */
void pmo_cpu_remove_background(const pmo_images* im, float* disp, int ph, int pw,
                               const pmo_functor* f, float win_by_factor, int literal,
                               int nthreads) {
  const int w = im->cols, h = im->rows;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
  for (int y = 1; y < h; ++y)
    for (int x = 1; x < w; ++x) {
      if (cpu_skip(x, y, w, h, pw, ph)) continue;
      float d0 = disp[(size_t)y * w + x];
      const float hi = (float)x - (float)(pw / 2);
      d0 = d0 > 0.f ? d0 : 0.f;
      d0 = d0 < hi ? d0 : hi;
      const float c = cpu_cost(im, pw, ph, x, y, d0, f, literal);
      const float c_bg = cpu_cost(im, pw, ph, x, y, 0.f, f, literal);
      if (c > (c_bg / win_by_factor)) disp[(size_t)y * w + x] = 0.f;
    }
}

/* ======================================================================================== */
/* SEM_GPU                                                                                   */
/* ======================================================================================== */

/* GetSubpixel (patchmatch_gpu.cu:18-42): floor/ceil neighbours, lerp rows first then columns. */
float pmo_gpu_get_subpixel(const float* im, int rows, int cols, float row, float col) {
  (void)rows;
  const int row0 = (int)floorf(row), row1 = (int)ceilf(row);
  const int col0 = (int)floorf(col), col1 = (int)ceilf(col);
  const float c00 = im[(size_t)row0 * cols + col0], c01 = im[(size_t)row0 * cols + col1];
  const float c10 = im[(size_t)row1 * cols + col0], c11 = im[(size_t)row1 * cols + col1];
  const float trow = row - (float)row0, tcol = col - (float)col0;
  const float ir = 1.0f - trow, ic = 1.0f - tcol;
  float c0 = ir * c00;
  c0 = c0 + trow * c10;
  float c1 = ir * c01;
  c1 = c1 + trow * c11;
  float r = ic * c0;
  r = r + tcol * c1;
  return r;
}

static inline float u8_subpixel(const uint8_t* im, int cols, int row, float col) {
  /* integer row: trow = 0, so c0 = c00, c1 = c01 exactly */
  const int col0 = (int)floorf(col), col1 = (int)ceilf(col);
  const float c0 = (float)im[(size_t)row * cols + col0], c1 = (float)im[(size_t)row * cols + col1];
  const float tcol = col - (float)col0, ic = 1.0f - tcol;
  float r = ic * c0;
  r = r + tcol * c1;
  return r;
}
static inline float f32_subpixel(const float* im, int cols, int row, float col) {
  const int col0 = (int)floorf(col), col1 = (int)ceilf(col);
  const float c0 = im[(size_t)row * cols + col0], c1 = im[(size_t)row * cols + col1];
  const float tcol = col - (float)col0, ic = 1.0f - tcol;
  float r = ic * c0;
  r = r + tcol * c1;
  return r;
}

/* L1GradientCost3x3 (patchmatch_gpu.cu:72-114): five taps (corners + centre of the 3x3), in the
 * order (-1,-1) (-1,+1) (0,0) (+1,-1) (+1,+1); yr is always an integer on the path.  The images
 * are the u8 inputs converted to f32 (patchmatch_gpu.cu:346-349), i.e. exact. */
float pmo_gpu_cost5(const pmo_images* im, int yl, int xl, float yr, float xr, float alpha) {
  static const int dy[5] = {-1, -1, 0, 1, 1}, dx[5] = {-1, 1, 0, -1, 1};
  const int cols = im->cols;
  const int yri = (int)yr;
  const float one_minus = 1.f - alpha;
  float cost = 0.f;
  for (int t = 0; t < 5; ++t) {
    const float xs = xr + (float)dx[t];
    const float il = (float)im->il[(size_t)(yl + dy[t]) * cols + (xl + dx[t])];
    const float gl = im->gl[(size_t)(yl + dy[t]) * cols + (xl + dx[t])];
    const float e0 = fabsf(il - u8_subpixel(im->ir, cols, yri + dy[t], xs));
    const float e1 = fabsf(gl - f32_subpixel(im->gr, cols, yri + dy[t], xs));
    const float t0 = alpha * e0;
    const float t1 = one_minus * e1;
    const float s = t0 + t1;
    cost = cost + s;
  }
  return cost;
}

/* AddForegroundNoise (patchmatch_gpu.cu:298-304): mask = disp > 0; disp = noise*scale + disp;
 * disp *= mask; disp = max(disp, 0).  (scale is a power of two on the path, so the scaleAdd is
 * exact whether or not it is fused.) */
void pmo_gpu_add_foreground_noise(float* disp, const float* unit_noise, size_t n, float scale) {
  for (size_t i = 0; i < n; ++i) {
    const float d = disp[i];
    if (d > 0.f) {
      const float m = unit_noise[i] * scale;
      const float v = m + d;
      disp[i] = v > 0.f ? v : 0.f;
    } else {
      disp[i] = 0.f;
    }
  }
}

/* PropagateRow (patchmatch_gpu.cu:116-172) as ONE stripe (blockDim.x = 1): rows r..H-r-1,
 * direction +1: cols r..W-r-2 ascending, direction -1: cols W-r-1..r+1 descending (the loop end is
 * exclusive, :156).  The cost is always the 5-tap one (patch_size only sets the radius, Q4). */
void pmo_gpu_propagate_row(const pmo_images* im, float* disp, int direction, int patch_size,
                           float alpha, int nthreads) {
  const int r = patch_size / 2, W = im->cols, H = im->rows;
  const int min_col = r, max_col = W - r - 1;
  const int start = direction > 0 ? min_col : max_col, end = direction > 0 ? max_col : min_col;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
  for (int row = r; row <= H - r - 1; ++row) {
    const float y = (float)row;
    for (int col = start; direction > 0 ? col < end : col > end; col += direction) {
      const float x = (float)col;
      const float d0 = disp[(size_t)row * W + col];
      const float d1 = disp[(size_t)row * W + (col - direction)];
      const float cost0 = pmo_gpu_cost5(im, row, col, y, fmaxf(x - d0, (float)r), alpha);
      const float cost1 = pmo_gpu_cost5(im, row, col, y, fmaxf(x - d1, (float)r), alpha);
      if (cost1 < cost0) disp[(size_t)row * W + col] = fminf(d1, x - (float)r);
    }
  }
}

/* PropagateCol (patchmatch_gpu.cu:175-230), one stripe per column. */
void pmo_gpu_propagate_col(const pmo_images* im, float* disp, int direction, int patch_size,
                           float alpha, int nthreads) {
  const int r = patch_size / 2, W = im->cols, H = im->rows;
  const int min_row = r, max_row = H - r - 1;
  const int start = direction > 0 ? min_row : max_row, end = direction > 0 ? max_row : min_row;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
  for (int col = r; col <= W - r - 1; ++col) {
    const float x = (float)col;
    for (int row = start; direction > 0 ? row < end : row > end; row += direction) {
      const float y = (float)row;
      const float d0 = disp[(size_t)row * W + col];
      const float d1 = disp[(size_t)(row - direction) * W + col];
      const float cost0 = pmo_gpu_cost5(im, row, col, y, fmaxf(x - d0, (float)r), alpha);
      const float cost1 = pmo_gpu_cost5(im, row, col, y, fmaxf(x - d1, (float)r), alpha);
      if (cost1 < cost0) disp[(size_t)row * W + col] = fminf(d1, x - (float)r);
    }
  }
}

/* MaskBackground (patchmatch_gpu.cu:233-270). */
void pmo_gpu_mask_background(const pmo_images* im, float* disp, int patch_size, float alpha,
                             float improve_factor, int nthreads) {
  const int r = patch_size / 2, W = im->cols, H = im->rows;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
  for (int row = r; row <= H - r - 1; ++row)
    for (int col = r; col <= W - r - 1; ++col) {
      const float y = (float)row, x = (float)col;
      const float d1 = disp[(size_t)row * W + col];
      const float cost0 = pmo_gpu_cost5(im, row, col, y, x, alpha);
      const float cost1 = pmo_gpu_cost5(im, row, col, y, fmaxf(x - d1, (float)r), alpha);
      const float thr = improve_factor * cost0;
      if (!(cost1 < thr)) disp[(size_t)row * W + col] = 0.f;
    }
}

/* MaskOcclusions (patchmatch_gpu.cu:273-295): float image indices truncate to int; the thresholds
 * are double literals, so the comparison is carried out in double (Q13). */
void pmo_gpu_mask_occlusions(float* displ, const float* dispr, int rows, int cols) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const float dl = displ[(size_t)y * cols + x];
      const int xr = (int)fmaxf((float)x - dl, 0.f);
      const float dr = dispr[(size_t)y * cols + xr];
      if ((double)dr > 1.4 * (double)dl || (double)dr < 0.7 * (double)dl)
        displ[(size_t)y * cols + x] = 0.f;
    }
}

/* ======================================================================================== */
/* pipelines                                                                                 */
/* ======================================================================================== */

void pmo_params_default(pmo_params* p, int semantics) {
  memset(p, 0, sizeof(*p));
  p->semantics = semantics;
  p->n_iters = 3;          /* patchmatch_gpu.h:86 */
  p->cost_alpha = 0.9f;    /* patchmatch_gpu.h:85 */
  p->functor.alpha = 0.7f; /* patchmatch_test.cpp:35-37 */
  p->functor.tau_color = 50.0f;
  p->functor.tau_grad = 20.0f;
  for (int i = 0; i < PMO_MAX_ITERS; ++i) {
    p->noise_amp[i] = (float)(32.0 / pow(2.0, (double)(float)i)); /* patchmatch_gpu.cu:395 */
    p->patch_w[i] = 3;
    p->patch_h[i] = 3;
  }
  p->bg_patch_w = 3;
  p->bg_patch_h = 3;
  p->bg_factor = semantics == PMO_SEM_GPU ? 0.8f /* patchmatch_gpu.h:88 */
                                          : 1.5f /* patchmatch_test.cpp:183 */;
  p->noise_seed = 123;
  p->left_right_check = 1;
  p->literal = 0;
  p->nthreads = 1;
}

void pmo_match_view(const pmo_params* p, const pmo_images* im, float* disp) {
  const int rows = im->rows, cols = im->cols;
  const size_t n = (size_t)rows * cols;
  if (p->semantics == PMO_SEM_GPU) {
    /* unit noise: cv::RNG(123).fill(UNIFORM, -1, 1), created once (patchmatch_gpu.cu:339-344) */
    float* unit = (float*)malloc(sizeof(float) * n);
    pmo_rng_fill_uniform(unit, n, -1.0, 1.0, p->noise_seed);
    for (int it = 0; it < p->n_iters; ++it) {
      pmo_gpu_add_foreground_noise(disp, unit, n, p->noise_amp[it]);
      pmo_gpu_propagate_row(im, disp, 1, 3, p->cost_alpha, p->nthreads);
      pmo_gpu_propagate_col(im, disp, 1, 3, p->cost_alpha, p->nthreads);
      pmo_gpu_propagate_row(im, disp, -1, 3, p->cost_alpha, p->nthreads);
      pmo_gpu_propagate_col(im, disp, -1, 3, p->cost_alpha, p->nthreads);
    }
    pmo_gpu_mask_background(im, disp, 3, p->cost_alpha, p->bg_factor, p->nthreads);
    free(unit);
  } else {
    uint8_t* mask = (uint8_t*)malloc(n);
    for (int it = 0; it < p->n_iters; ++it) {
      for (size_t i = 0; i < n; ++i) mask[i] = disp[i] > 0.f ? 255 : 0; /* `disp > 0` */
      pmo_cpu_add_noise(disp, rows, cols, p->noise_amp[it], mask, p->noise_seed);
      pmo_cpu_propagate(im, disp, p->patch_h[it], p->patch_w[it], &p->functor, 15, p->literal,
                        p->nthreads);
    }
    pmo_cpu_remove_background(im, disp, p->bg_patch_h, p->bg_patch_w, &p->functor, p->bg_factor,
                              p->literal, p->nthreads);
    free(mask);
  }
}

void pmo_match(const pmo_params* p, const uint8_t* left, const uint8_t* right, int rows, int cols,
               const float* seed_l, const float* seed_r, float* disp_l, float* disp_r) {
  const size_t n = (size_t)rows * cols;
  float* gl = (float*)malloc(sizeof(float) * n);
  float* gr = (float*)malloc(sizeof(float) * n);
  pmo_gradient_magnitude(left, rows, cols, gl);
  pmo_gradient_magnitude(right, rows, cols, gr);

  pmo_images lv = {rows, cols, left, right, gl, gr};
  if (seed_l) memcpy(disp_l, seed_l, sizeof(float) * n);
  else memset(disp_l, 0, sizeof(float) * n);
  pmo_match_view(p, &lv, disp_l);

  if (p->left_right_check) {
    /* right view = same algorithm on the horizontally mirrored (R, L) pair (patchmatch_gpu.cu:357-368) */
    uint8_t* lf = (uint8_t*)malloc(n);
    uint8_t* rf = (uint8_t*)malloc(n);
    float* glf = (float*)malloc(sizeof(float) * n);
    float* grf = (float*)malloc(sizeof(float) * n);
    float* df = (float*)malloc(sizeof(float) * n);
    pmo_flip_h_u8(left, lf, rows, cols);
    pmo_flip_h_u8(right, rf, rows, cols);
    pmo_flip_h_f32(gl, glf, rows, cols);
    pmo_flip_h_f32(gr, grf, rows, cols);
    if (seed_r) pmo_flip_h_f32(seed_r, df, rows, cols);
    else memset(df, 0, sizeof(float) * n);
    pmo_images rv = {rows, cols, rf, lf, grf, glf};
    pmo_match_view(p, &rv, df);
    pmo_flip_h_f32(df, disp_r, rows, cols);
    pmo_gpu_mask_occlusions(disp_l, disp_r, rows, cols);
    free(lf);
    free(rf);
    free(glf);
    free(grf);
    free(df);
  }
  free(gl);
  free(gr);
}
