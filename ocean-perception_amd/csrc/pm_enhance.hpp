// pm_enhance.hpp -- the range-free "stereo-ready" enhancement in front of stereo (include/pm/imaging.h;
// SURVEY.md 8f-2): gray = BGR2GRAY(Normalize(NormalizeColorIlluminant(CastImage3bTo3f(bgr8)))), the chain of
// test/imaging/enhance_test.cpp:69-73 / test/stereo_matching/sgbm_test.cpp:66-84.
//   NormalizeColorIlluminant  normalization.cpp:178-185  bgr / (2 * GaussianBlur(bgr, NextOddInt(cols/3), ksize/4, REPLICATE))
//   Normalize                 normalization.cpp:43-69    HSV value stretch by the min / max of a 1/8 bilinear resize
// Same operation order as oracle/pm_enhance_oracle.c, every float op a single IEEE op: bit-identical results.
//
// Kernels: row pass of the separable Gaussian (taps added left to right, window staged in LDS), column pass
// (centre tap, then symmetric pairs; (T + 2c) x W tile in LDS) fused with the division by the illuminant,
// min / max of the 1/8-resized value channel, and the per-pixel HSV stretch + gray + 8-bit conversion.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "pm_color.hpp"

namespace pm {

__device__ __forceinline__ int clampi_d(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Several images of one size through one launch (blockIdx.z = image).
constexpr int kBlurBatch = 8;
struct BlurBatch {
  const void* src[kBlurBatch];  // 8-bit or float source images
  float* dst[kBlurBatch];
  float* tmp;                   // row-pass output: image z at tmp + z * tmp_stride
  size_t tmp_stride;
};

// ---- row pass: dst[y][x][q] = sum_t k[t] * src[y][clamp(x - c + t)][q], t = 0 .. ksize-1 in order -------------
// SRC_U8: the source is the 8-bit image, cast like CastImage3bTo3f (x * (float)(1/255.)).
// One thread per FOUR consecutive pixels: every window sample is read from LDS once and feeds four outputs with
// four different taps (register blocking: the first version, one pixel per thread, was LDS-bandwidth bound --
// 3 x 427 reads per pixel).  Each output still adds its taps in ascending order, so the result is unchanged.
// LDS is planar per channel with one pad word after every FOUR samples (index p + p / 4): a thread's four pixels start
// at 4 * lane, i.e. at word 5 * lane -- an odd lane stride, conflict-free -- and window sample u = 4 k + m of a thread
// lies at its base + 5 k + m: the tap loop advances one pointer per four taps and addresses the samples by immediate
// offsets.  (The first layout, a pad word every 32, cost an address computation per tap: 17 vector instructions per
// tap against the 12 packed multiplies and adds that do the work.)
// grid = (ceil(cols / 512), rows, images), block = 128, dynamic LDS = (CH * (pad(512 + ksize - 1) + 1) + ksize) floats.
constexpr int kBlurRowThreads = 128;
constexpr int kBlurRowPx = 4 * kBlurRowThreads;
__host__ __device__ inline int blur_skew(int p) { return p + (p >> 2); }
template <bool SRC_U8, int CH>
__global__ void __launch_bounds__(kBlurRowThreads) k_blur_rows(BlurBatch bb, int rows, int cols, int ksize,
                                                               const float* __restrict__ taps) {
  extern __shared__ float lds[];
  const void* __restrict__ src = bb.src[blockIdx.z];
  float* __restrict__ dst = bb.tmp + (size_t)blockIdx.z * bb.tmp_stride;
  constexpr int R = 4;
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int y = blockIdx.y, x0 = blockIdx.x * kBlurRowPx, c = ksize / 2;
  const int span = kBlurRowPx + ksize - 1;
  const int plane = blur_skew(span) + 1;
  float* s_k = lds + plane * CH;
  const float cast = (float)(1.0 / 255.0);
  for (int e = threadIdx.x; e < span * CH; e += kBlurRowThreads) {
    const int p = e / CH, q = e - p * CH;
    const int sx = clampi_d(x0 - c + p, 0, cols - 1);
    const size_t o = ((size_t)y * cols + sx) * CH + q;
    lds[q * plane + blur_skew(p)] = SRC_U8 ? (float)((const uint8_t*)src)[o] * cast : ((const float*)src)[o];
  }
  for (int e = threadIdx.x; e < ksize; e += kBlurRowThreads) s_k[e] = taps[e];
  __syncthreads();
  const int p0 = threadIdx.x * R;  // first output pixel of this thread, relative to x0
  if (x0 + p0 >= cols) return;
  const float* base = lds + 5 * threadIdx.x;  // = lds + blur_skew(p0); sample u at base[u + (u >> 2)]
  float s[R][CH];
  // window position u (sample x0 - c + p0 + u) contributes tap j = u - r to output r.
  // prologue u < R: the first taps of outputs 0 .. u (the sum starts WITH k[0] * w, it is not added to 0)
#pragma unroll
  for (int u = 0; u < R; ++u) {
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const float w = base[q * plane + u];
#pragma unroll
      for (int r = 0; r <= u; ++r) {
        const float prod = s_k[u - r] * w;
        s[r][q] = (u == r) ? prod : s[r][q] + prod;
      }
    }
  }
  // main part: every output has a tap at this position.  Outputs are paired (0,1) and (2,3): packed-f32
  // multiply and add, the same two IEEE operations per tap as the scalar form.  The taps are uniform: they come
  // through the scalar unit (s_load), not from LDS.
  f2 a[2][CH];
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    a[0][q] = (f2){s[0][q], s[1][q]};
    a[1][q] = (f2){s[2][q], s[3][q]};
  }
  auto tap = [&](const float* g, int m, int u) {  // sample u = 4 k + m at g[m], g = base + 5 k
    const f2 k01 = {taps[u], taps[u - 1]}, k23 = {taps[u - 2], taps[u - 3]};
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const float w = g[q * plane + m];
      const f2 ww = {w, w};
      a[0][q] = a[0][q] + k01 * ww;
      a[1][q] = a[1][q] + k23 * ww;
    }
  };
  int u = R;
  const float* g = base + 5;  // group k = 1
  for (; u + 3 < ksize; u += 4, g += 5) {
    tap(g, 0, u);
    tap(g, 1, u + 1);
    tap(g, 2, u + 2);
    tap(g, 3, u + 3);
  }
  if (u < ksize) tap(g, 0, u);
  if (u + 1 < ksize) tap(g, 1, u + 1);
  if (u + 2 < ksize) tap(g, 2, u + 2);
#pragma unroll
  for (int q = 0; q < CH; ++q) {
    s[0][q] = a[0][q].x;
    s[1][q] = a[0][q].y;
    s[2][q] = a[1][q].x;
    s[3][q] = a[1][q].y;
  }
  // epilogue: the last taps of outputs 1 .. R-1
#pragma unroll
  for (int v = 0; v < R - 1; ++v) {
    const int ue = ksize + v;
    const int sp = ue + (ue >> 2);
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const float w = base[q * plane + sp];
#pragma unroll
      for (int r = v + 1; r < R; ++r) s[r][q] = s[r][q] + s_k[ue - r] * w;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int x = x0 + p0 + r;
    if (x < cols)
#pragma unroll
      for (int q = 0; q < CH; ++q) dst[((size_t)y * cols + x) * CH + q] = s[r][q];
  }
}

// ---- column pass over the image seen as [rows][width] floats (width = cols * ch):
// s = k[c] * S[y]; s += k[c + j] * (S[y + j] + S[y - j]), j = 1 .. c.  DIVIDE: the result is the illuminant / 2;
// the kernel writes orig / (2 * s) (0 where the divisor is 0), orig being the 8-bit (cast) or float source image.
// A thread owns TWO adjacent columns (packed-f32 adds and multiplies: the same IEEE operations, two per
// instruction) and walks its share of the tile's T output rows.
// grid = (ceil(width / W), ceil(rows / T), images), block = 128 or 256, dynamic LDS = ((T + 2c + 3) * (W + 2) + c + 1)
// floats, W a power of two.  T = (threads / (W / 2)) * 4: every thread has exactly one group of four rows (a tile of 66 rows
// for 32 row groups left half the threads without work for the whole tap loop).
template <bool DIVIDE, bool ORIG_U8>
__global__ void __launch_bounds__(256) k_blur_cols(BlurBatch bb, int rows, int width, int ksize,
                                                   const float* __restrict__ taps, int W, int T) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  extern __shared__ float lds[];
  const float* __restrict__ tmp = bb.tmp + (size_t)blockIdx.z * bb.tmp_stride;
  const void* __restrict__ orig = bb.src[blockIdx.z];
  float* __restrict__ dst = bb.dst[blockIdx.z];
  const int c = ksize / 2;
  const int x0 = blockIdx.x * W, y0 = blockIdx.y * T;
  const int span = T + 2 * c;
  const int P = W + 2;  // row pitch: four rows apart (the row groups of a wavefront) are 4 P = 8 (mod 32) banks apart, not 0
  const int NT = blockDim.x;
  float* s_k = lds + (span + 3) * P;  // taps[c .. ksize-1], behind three spare rows (the four-row windows of the last
                                      // row group read up to three rows past the tile; those outputs are not stored)
  {
    // W is a power of two (8, 16 or 32): every thread keeps its column and walks down the rows -- no division per
    // element (with a run-time W the fill cost more vector instructions than the taps)
    const int fw = threadIdx.x & (W - 1), fr = threadIdx.x / W, fstep = NT / W;
    const int sx = min(x0 + fw, width - 1);
    for (int r = fr; r < span; r += fstep) {
      const int sy = clampi_d(y0 - c + r, 0, rows - 1);
      lds[r * P + fw] = tmp[(size_t)sy * width + sx];
    }
  }
  for (int e = threadIdx.x; e <= c; e += NT) s_k[e] = taps[c + e];
  __syncthreads();
  const int W2 = W / 2;
  const int w = (threadIdx.x % W2) * 2, g = threadIdx.x / W2, G = NT / W2;
  const int x = x0 + w;
  if (x >= width) return;
  const float cast = (float)(1.0 / 255.0);
  auto finish = [&](int yy, f2 s) {
    const size_t o = (size_t)(y0 + yy) * width + x;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (x + u >= width) break;
      const float sv = u == 0 ? s.x : s.y;
      if (DIVIDE) {
        const float num = ORIG_U8 ? (float)((const uint8_t*)orig)[o + u] * cast : ((const float*)orig)[o + u];
        const float d = sv * 2.0f;
        dst[o + u] = d != 0.f ? num / d : 0.f;
      } else {
        dst[o + u] = sv;
      }
    }
  };
  // FOUR consecutive output rows per thread: the samples S[y + r + j] and S[y + r - j] of the four rows r at tap j
  // are two sliding windows of four values each, so a tap costs TWO new LDS reads for four outputs instead of eight
  // (the one-row form spent 2 LDS reads per 3 vector instructions and was LDS-bound).  The windows live in
  // registers as circular buffers indexed by (row offset & 3); the tap loop is unrolled by four so that every index
  // is a compile-time constant.  Each output still adds its taps in the same order: bit-identical.
  constexpr int RB = 4;
  for (int yb = g * RB; yb < T && y0 + yb < rows; yb += G * RB) {
    const float* col = lds + (yb + c) * P + w;  // S[y0 + yb] of this column pair
    const float kc = s_k[0];
    f2 acc[RB], up[RB], dn[RB];  // up[k & 3] = S[yb + k], k = j .. j + 3;  dn[k & 3] = S[yb + k], k = -j .. 3 - j
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const f2 v = *(const f2*)(col + r * P);
      acc[r] = v * (f2){kc, kc};
      up[r] = v;
      dn[r] = v;
    }
    // before tap j: up holds S[yb + j - 1 .. yb + j + 2], dn holds S[yb - j + 1 .. yb - j + 4]
    auto tap = [&](int j, int jm) {  // jm = j & 3 as a compile-time constant at every call site
      // slide: up gains S[yb + j + 3] in place of S[yb + j - 1]; dn gains S[yb - j] in place of S[yb - j + 4]
      up[(jm + 3) & 3] = *(const f2*)(col + (j + 3) * P);
      dn[(4 - jm) & 3] = *(const f2*)(col - j * P);
      const float kj = taps[c + j];  // uniform: a scalar load, not an LDS read
      const f2 kk = {kj, kj};
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const f2 pair = up[(r + jm) & 3] + dn[(r + 4 - jm) & 3];
        acc[r] = acc[r] + pair * kk;
      }
    };
    int j = 1;
    for (; j + 3 <= c; j += 4) {  // j = 1 (mod 4) at the top
      tap(j, 1);
      tap(j + 1, 2);
      tap(j + 2, 3);
      tap(j + 3, 0);
    }
    if (j <= c) tap(j, 1);
    if (j + 1 <= c) tap(j + 1, 2);
    if (j + 2 <= c) tap(j + 2, 3);
#pragma unroll
    for (int r = 0; r < RB; ++r)
      if (yb + r < T && y0 + yb + r < rows) finish(yb + r, acc[r]);
  }
}

__device__ __forceinline__ float value_of(const float* __restrict__ q, size_t px) {
  const float b = q[px * 3], g = q[px * 3 + 1], r = q[px * 3 + 2];
  float v = b;
  if (g > v) v = g;
  if (r > v) v = r;
  return v;
}

// One axis of cv::resize INTER_LINEAR (the oracle's linear_coeff)
__device__ __forceinline__ void linear_coeff_d(int d, int ssize, int dsize, int& s0, float& w0, float& w1) {
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
  s0 = s;
  w0 = 1.f - f;
  w1 = f;
}

// ---- min / max of resize(V, size / 8); mm[0] = min bits (init 0x7f7fffff), mm[1] = max bits (init 0); V >= 0 ----
__global__ void __launch_bounds__(256) k_value_minmax(const float* __restrict__ q, int rows, int cols, unsigned* mm) {
  __shared__ float s_lo[4], s_hi[4];
  const int dr = rows / 8, dc = cols / 8;
  float lo = FLT_MAX, hi = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dr * dc; i += gridDim.x * blockDim.x) {
    const int dy = i / dc, dx = i - dy * dc;
    int sy, sx;
    float b0, b1, a0, a1;
    linear_coeff_d(dy, rows, dr, sy, b0, b1);
    linear_coeff_d(dx, cols, dc, sx, a0, a1);
    const int sy1 = min(sy + 1, rows - 1), sx1 = min(sx + 1, cols - 1);
    const float r0 = value_of(q, (size_t)sy * cols + sx) * a0 + value_of(q, (size_t)sy * cols + sx1) * a1;
    const float r1 = value_of(q, (size_t)sy1 * cols + sx) * a0 + value_of(q, (size_t)sy1 * cols + sx1) * a1;
    const float val = r0 * b0 + r1 * b1;
    lo = val < lo ? val : lo;
    hi = val > hi ? val : hi;
  }
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, ofs, 64));
    hi = fmaxf(hi, __shfl_xor(hi, ofs, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    hi = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    atomicMin(&mm[0], __float_as_uint(lo));
    atomicMax(&mm[1], __float_as_uint(hi));
  }
}

// ---- the same min / max with the value computed on the fly from the 8-bit image and its blurred illuminant (the fused
// path of pm_match_bgr_device: neither q = I / (2 blur) nor J1 = Normalize(q) is ever written to memory).
// STAGE 1: V of q -> mm[0..1];  STAGE 2: V of J1 = Normalize(q; mm[0..1]) -> mm[2..3].
template <int STAGE>
__device__ __forceinline__ float fused_value(const uint8_t* __restrict__ bgr8, const float* __restrict__ blur, size_t px,
                                             float a1, float b1) {
  float b = illuminant_div(bgr8[px * 3], blur[px * 3]), g = illuminant_div(bgr8[px * 3 + 1], blur[px * 3 + 1]),
        r = illuminant_div(bgr8[px * 3 + 2], blur[px * 3 + 2]);
  if (STAGE == 2) {
    float jb, jg, jr;
    normalize_px(b, g, r, a1, b1, jb, jg, jr);
    b = jb;
    g = jg;
    r = jr;
  }
  float v = b;
  if (g > v) v = g;
  if (r > v) v = r;
  return v;
}
// blockIdx.y = image: bb.src / bb.dst are the 8-bit images and their blurred illuminants, image y's words at mm + 4 y
template <int STAGE>
__global__ void __launch_bounds__(256) k_value_minmax_fused(BlurBatch bb, int rows, int cols, unsigned* mm_all) {
  __shared__ float s_lo[4], s_hi[4];
  const uint8_t* __restrict__ bgr8 = (const uint8_t*)bb.src[blockIdx.y];
  const float* __restrict__ blur = bb.dst[blockIdx.y];
  unsigned* mm = mm_all + 4 * blockIdx.y;
  const int dr = rows / 8, dc = cols / 8;
  float a1 = 0.f, b1 = 0.f;
  if (STAGE == 2) stretch_coeffs(mm, a1, b1);
  float lo = FLT_MAX, hi = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dr * dc; i += gridDim.x * blockDim.x) {
    const int dy = i / dc, dx = i - dy * dc;
    int sy, sx;
    float w0, w1, u0, u1;
    linear_coeff_d(dy, rows, dr, sy, w0, w1);
    linear_coeff_d(dx, cols, dc, sx, u0, u1);
    const int sy1 = min(sy + 1, rows - 1), sx1 = min(sx + 1, cols - 1);
    const float r0 = fused_value<STAGE>(bgr8, blur, (size_t)sy * cols + sx, a1, b1) * u0 +
                     fused_value<STAGE>(bgr8, blur, (size_t)sy * cols + sx1, a1, b1) * u1;
    const float r1 = fused_value<STAGE>(bgr8, blur, (size_t)sy1 * cols + sx, a1, b1) * u0 +
                     fused_value<STAGE>(bgr8, blur, (size_t)sy1 * cols + sx1, a1, b1) * u1;
    const float val = r0 * w0 + r1 * w1;
    lo = val < lo ? val : lo;
    hi = val > hi ? val : hi;
  }
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, ofs, 64));
    hi = fmaxf(hi, __shfl_xor(hi, ofs, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fminf(fminf(s_lo[0], s_lo[1]), fminf(s_lo[2], s_lo[3]));
    hi = fmaxf(fmaxf(s_hi[0], s_hi[1]), fmaxf(s_hi[2], s_hi[3]));
    unsigned* out = mm + (STAGE == 2 ? 2 : 0);
    atomicMin(&out[0], __float_as_uint(lo));
    atomicMax(&out[1], __float_as_uint(hi));
  }
}

// ---- Normalize's per-pixel part + BGR2GRAY + convertTo(CV_8U, 255) ------------------------------------------------
// V' = V * (float)(1 / (vmax - vmin)) + (float)(-vmin / (vmax - vmin)); J (optional) and gray8 (optional) out.
__global__ void __launch_bounds__(256) k_normalize_gray(const float* __restrict__ q, size_t n_px,
                                                        const unsigned* __restrict__ mm, float* __restrict__ J,
                                                        uint8_t* __restrict__ gray8) {
  float alpha, beta;
  stretch_coeffs(mm, alpha, beta);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += (size_t)gridDim.x * blockDim.x) {
    float b, g, r;
    normalize_px(q[i * 3], q[i * 3 + 1], q[i * 3 + 2], alpha, beta, b, g, r);
    if (J) {
      J[i * 3] = b;
      J[i * 3 + 1] = g;
      J[i * 3 + 2] = r;
    }
    if (gray8) gray8[i] = gray_u8(b, g, r);
  }
}

}  // namespace pm
