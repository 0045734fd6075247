// pm_texmask.hpp -- ForegroundTextureMask (src/vehicle/stereo_matching/patchmatch.cpp:19-49): a mask of the pixels whose
// neighbourhood has texture, from the morphological gradient (dilate - erode over a (2k + 1)^2 rectangle) of the gray
// image, optionally computed on an image shrunk by `downsize` and blown up again with cv::resize(INTER_LINEAR).
// Nothing in the reference calls it; it is here because it is the one function of stereo_matching/patchmatch.{hpp,cpp}
// the engine did not have.  Integer arithmetic throughout: bit-identical to oracle/pm_oracle.c::pmo_foreground_texture_mask.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pm {

// cv::resize(src, dst, Size(dcols, drows), 0, 0, INTER_LINEAR) on CV_8UC1 (OpenCV 3.4 resize.cpp; oracle:
// pmo_resize_linear_u8): the exact 2 x 2 shrink takes the area shortcut (a + b + c + d + 2) >> 2, everything else the
// 11-bit fixed-point bilinear form with its two roundings.
__global__ void __launch_bounds__(256) k_resize_linear_u8(const uint8_t* __restrict__ src, int rows, int cols,
                                                          uint8_t* __restrict__ dst, int drows, int dcols) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
  if (dx >= dcols) return;
  if (drows * 2 == rows && dcols * 2 == cols) {
    const uint8_t* s0 = src + (size_t)(2 * dy) * cols;
    const uint8_t* s1 = s0 + cols;
    dst[(size_t)dy * dcols + dx] = (uint8_t)((s0[2 * dx] + s0[2 * dx + 1] + s1[2 * dx] + s1[2 * dx + 1] + 2) >> 2);
    return;
  }
  const double scale_x = (double)cols / dcols, scale_y = (double)rows / drows;
  float fx = (float)((dx + 0.5) * scale_x - 0.5);
  int sx = (int)floor((double)fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= cols - 1) { fx = 0.f; sx = cols - 1; }
  const int a0 = (int)(short)__float2int_rn((1.f - fx) * 2048.f), a1 = (int)(short)__float2int_rn(fx * 2048.f);
  float fy = (float)((dy + 0.5) * scale_y - 0.5);
  const int sy = (int)floor((double)fy);
  fy -= (float)sy;
  const int b0 = (int)(short)__float2int_rn((1.f - fy) * 2048.f), b1 = (int)(short)__float2int_rn(fy * 2048.f);
  const int sx1 = sx + 1 < cols ? sx + 1 : cols - 1;
  int hb[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int r = sy + k;
    r = r < 0 ? 0 : (r > rows - 1 ? rows - 1 : r);
    const uint8_t* srow = src + (size_t)r * cols;
    hb[k] = srow[sx] * a0 + srow[sx1] * a1;
  }
  dst[(size_t)dy * dcols + dx] = (uint8_t)((((b0 * (hb[0] >> 4)) >> 16) + ((b1 * (hb[1] >> 4)) >> 16) + 2) >> 2);
}

// horizontal pass of the rectangle's min / max (the window is clipped at the image border: cv::morphologyEx's default
// border value never wins a min or a max)
__global__ void __launch_bounds__(256) k_morph_rows(const uint8_t* __restrict__ src, int rows, int cols, int k,
                                                    uint8_t* __restrict__ lo, uint8_t* __restrict__ hi) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= cols) return;
  const uint8_t* r = src + (size_t)y * cols;
  const int x0 = max(x - k, 0), x1 = min(x + k, cols - 1);
  int mn = 255, mx = 0;
  for (int xx = x0; xx <= x1; ++xx) {
    const int v = r[xx];
    mn = min(mn, v);
    mx = max(mx, v);
  }
  lo[(size_t)y * cols + x] = (uint8_t)mn;
  hi[(size_t)y * cols + x] = (uint8_t)mx;
}
// vertical pass, gradient = dilate - erode, mask = gradient > min_grad ? 255 : 0
__global__ void __launch_bounds__(256) k_morph_cols_threshold(const uint8_t* __restrict__ lo,
                                                              const uint8_t* __restrict__ hi, int rows, int cols, int k,
                                                              double min_grad, uint8_t* __restrict__ mask) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= cols) return;
  const int y0 = max(y - k, 0), y1 = min(y + k, rows - 1);
  int mn = 255, mx = 0;
  for (int yy = y0; yy <= y1; ++yy) {
    mn = min(mn, (int)lo[(size_t)yy * cols + x]);
    mx = max(mx, (int)hi[(size_t)yy * cols + x]);
  }
  mask[(size_t)y * cols + x] = (double)(mx - mn) > min_grad ? 255 : 0;
}

}  // namespace pm
