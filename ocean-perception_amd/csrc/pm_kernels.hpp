// pm_kernels.hpp -- gfx950 kernels of the PatchMatch stereo engine.
// `file:line` citations are relative to the reference tree (/root/reference).
#pragma once

#include "pm_color.hpp"
#include "pm_device.hpp"
#include "pm_sweep_defs.hpp"

namespace pm {

// ---------------------------------------------------------------------------------------------
// prep: u8 pair -> {L, R, mirrored L, mirrored R} as u8, Sobel magnitude as f32 and saturated u8.
// Replaces upload+convertTo (patchmatch_gpu.cu:346-349), GradientMagnitude x2 (:351-352,
// :307-319: Sobel x, Sobel y, magnitude = 3 passes per image) and the four cu::flip (:357-360)
// with one pass: 2 B/px read, 4*(1+4+1) B/px written.
// Sobel: kernels [-1 0 1]x[1 2 1]^T, unnormalised, BORDER_REFLECT_101 (OpenCV default); every
// intermediate is an integer < 2^24, the only rounding is the correctly rounded sqrt.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sobel_mag(const uint8_t* im, size_t stride, int rows, int cols, int x, int y) {
  const uint8_t* r0 = im + (size_t)reflect101(y - 1, rows) * stride;
  const uint8_t* r1 = im + (size_t)y * stride;
  const uint8_t* r2 = im + (size_t)reflect101(y + 1, rows) * stride;
  const int xm = reflect101(x - 1, cols), xp = reflect101(x + 1, cols);
  const int dx = ((int)r0[xp] - (int)r0[xm]) + 2 * ((int)r1[xp] - (int)r1[xm]) + ((int)r2[xp] - (int)r2[xm]);
  const int dy = ((int)r2[xm] - (int)r0[xm]) + 2 * ((int)r2[x] - (int)r0[x]) + ((int)r2[xp] - (int)r0[xp]);
  const float fx = (float)dx, fy = (float)dy;
  const float sx = fx * fx, sy = fy * fy;
  return sqrtf(sx + sy);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt); __fsqrt_rn is the 1-ulp native form
}

// in_left / in_right: [B][rows][in_stride] u8.  grid = (ceil(cols/256), rows, B).
// view_sel: -1 = all four planes; 0 = the direct copies only (what view 0 works on), 1 = the mirrored copies only (view 1):
// with the two views on their own streams each stream prepares its own planes and nothing waits for the other.
// seeds.on != 0: also what k_seed does for the view(s) of this launch (seed maps -> disparity planes), one launch less
// at the head of a Match.
struct PrepSeeds {
  const float* l;  // tightly packed [B][rows][cols] seed maps in left / right image coordinates, or null = all background
  const float* r;
  int on;
};
__global__ void __launch_bounds__(256) k_prep(PlaneSet ps, const uint8_t* __restrict__ in_left,
                                              const uint8_t* __restrict__ in_right, size_t in_stride, int view_sel,
                                              PrepSeeds seeds) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z;
  if (x >= ps.cols) return;
  if (seeds.on) {
    const size_t sp = (size_t)ps.rows * ps.cols, so = (size_t)b * sp + (size_t)y * ps.cols;
    const size_t o = state_at(x, y, ps.pitch);
    if (view_sel != 1) ps.disp[((size_t)b * 2 + 0) * ps.splane + o] = seeds.l ? seeds.l[so + x] : 0.f;
    if (ps.n_views > 1 && view_sel != 0)
      ps.disp[((size_t)b * 2 + 1) * ps.splane + o] = seeds.r ? seeds.r[so + (ps.cols - 1 - x)] : 0.f;
  }
  const size_t in_plane = (size_t)ps.rows * in_stride;
  const uint8_t* srcs[2] = {in_left + (size_t)b * in_plane, in_right + (size_t)b * in_plane};
  const int xm = ps.cols - 1 - x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint8_t p = srcs[i][(size_t)y * in_stride + x];
    const float g = sobel_mag(srcs[i], in_stride, ps.rows, ps.cols, x, y);
    const uint8_t g8 = (uint8_t)sat_u8(g);
    const size_t direct = ((size_t)b * 4 + i) * ps.plane + (size_t)y * ps.pitch + x;
    const size_t mirror = ((size_t)b * 4 + 2 + i) * ps.plane + (size_t)y * ps.pitch + xm;
    const uint16_t pk = (uint16_t)(p | ((unsigned)g8 << 8));
    if (view_sel != 1) {
      ps.img8[direct] = p;
      ps.g32[direct] = g;
      ps.g8[direct] = g8;
      ps.pk16[direct] = pk;
    }
    if (view_sel != 0) {
      ps.img8[mirror] = p;
      ps.g32[mirror] = g;
      ps.g8[mirror] = g8;
      ps.pk16[mirror] = pk;
    }
  }
}

// k_prep with the stereo-ready enhancement's per-pixel tail folded into its load (BASELINE configs[4], "enhancement fused
// into the cost kernel's load path"): the inputs are the 8-bit BGR images and their blurred illuminants (the two
// 427-tap Gaussian passes stay kernels of their own, pm_enhance.hpp); a block computes the enhanced 8-bit gray values
// of its 64x8 tile plus a 1-pixel ring ONCE into LDS (pm_color.hpp::stereo_ready_gray: I / (2 blur), two HSV value
// stretches, gray -- exactly the values pm_stereo_ready would have written) and then does what k_prep does.  Neither
// the quotient image, nor the stretched images, nor the gray image ever goes through HBM.
// src/vehicle/imaging/normalization.cpp:43-69,178-185.  grid = (ceil(cols/64), ceil(rows/8), B), block = 256.
constexpr int kPrepBgrTileH = 8;  // rows per block: (66 x 10) / (64 x 8) = 1.29 enhanced-gray evaluations per pixel (4 rows: 1.55)
__global__ void __launch_bounds__(256) k_prep_bgr(PlaneSet ps, BgrSource src) {
  constexpr int TW = 64, TH = kPrepBgrTileH, LW = TW + 2, LH = TH + 2;
  __shared__ uint8_t s_gray[2][LH][LW + 2];
  const int b = blockIdx.z;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const size_t ipx = (size_t)ps.rows * ps.cols;
  for (int e = threadIdx.x; e < 2 * LH * LW; e += blockDim.x) {
    const int i = e / (LH * LW), r = (e - i * LH * LW) / LW, c = e - i * LH * LW - r * LW;
    // the ring follows the Sobel border rule (BORDER_REFLECT_101); positions beyond the image are never used
    const int gy = reflect101(min(y0 - 1 + r, ps.rows), ps.rows), gx = reflect101(min(x0 - 1 + c, ps.cols), ps.cols);
    const uint8_t* img = (i == 0 ? src.left : src.right) + (size_t)b * ipx * 3;
    const float* blur = (i == 0 ? src.blur_l : src.blur_r) + (size_t)b * ipx * 3;
    s_gray[i][r][c] = stereo_ready_gray(img, blur, (size_t)gy * ps.cols + gx, src.mm + ((size_t)b * 2 + i) * 4);
  }
  __syncthreads();
  const int tx = threadIdx.x & (TW - 1);
  const int x = x0 + tx;
  if (x >= ps.cols) return;
  const int xm = ps.cols - 1 - x;
  for (int ty = threadIdx.x / TW; ty < TH; ty += 256 / TW) {
  const int y = y0 + ty;
  if (y >= ps.rows) break;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const uint8_t(*g)[LW + 2] = s_gray[i];
    const int r = ty + 1, c = tx + 1;
    const uint8_t p = g[r][c];
    const int dx = ((int)g[r - 1][c + 1] - (int)g[r - 1][c - 1]) + 2 * ((int)g[r][c + 1] - (int)g[r][c - 1]) +
                   ((int)g[r + 1][c + 1] - (int)g[r + 1][c - 1]);
    const int dy = ((int)g[r + 1][c - 1] - (int)g[r - 1][c - 1]) + 2 * ((int)g[r + 1][c] - (int)g[r - 1][c]) +
                   ((int)g[r + 1][c + 1] - (int)g[r - 1][c + 1]);
    const float fx = (float)dx, fy = (float)dy;
    const float sx = fx * fx, sy = fy * fy;
    const float gm = sqrtf(sx + sy);
    const uint8_t g8 = (uint8_t)sat_u8(gm);
    const size_t direct = ((size_t)b * 4 + i) * ps.plane + (size_t)y * ps.pitch + x;
    const size_t mirror = ((size_t)b * 4 + 2 + i) * ps.plane + (size_t)y * ps.pitch + xm;
    ps.img8[direct] = p;
    ps.img8[mirror] = p;
    ps.g32[direct] = gm;
    ps.g32[mirror] = gm;
    ps.g8[direct] = g8;
    ps.g8[mirror] = g8;
    const uint16_t pk = (uint16_t)(p | ((unsigned)g8 << 8));
    ps.pk16[direct] = pk;
    ps.pk16[mirror] = pk;
  }
  }
}

// One view from caller-supplied FLOAT images and gradients -- the inputs of
// PatchmatchGpu::Match(const cu::GpuMat& iml, imr, Gl, Gr, cu::GpuMat& disp) (patchmatch_gpu.h:104-108): iml / imr hold
// the 8-bit image values as floats, as convertTo(CV_32F) leaves them (patchmatch_gpu.cu:346-349), Gl / Gr their
// gradient magnitudes.  Fills planes 0 (reference) and 1 (target) of pair 0; `stride` in elements.
__global__ void __launch_bounds__(256) k_prep_view(PlaneSet ps, const float* __restrict__ iml,
                                                   const float* __restrict__ imr, const float* __restrict__ gl,
                                                   const float* __restrict__ gr, size_t stride) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= ps.cols) return;
  const float* im[2] = {iml, imr};
  const float* gm[2] = {gl, gr};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const size_t si = (size_t)y * stride + x;
    const uint8_t p = (uint8_t)sat_u8(im[i][si]);
    const float g = gm[i][si];
    const uint8_t g8 = (uint8_t)sat_u8(g);
    const size_t o = (size_t)i * ps.plane + (size_t)y * ps.pitch + x;
    ps.img8[o] = p;
    ps.g32[o] = g;
    ps.g8[o] = g8;
    ps.pk16[o] = (uint16_t)(p | ((unsigned)g8 << 8));
  }
}

// Transposes `planes` planes of rows x cols (pitch `sp`) into cols x rows (pitch `dp`) through a
// 64x64 LDS tile (+1 column of padding: conflict-free for 4-byte elements, 2-way for bytes) so that
// both the reads and the writes are coalesced.  Block (bx, by, bz) of a (ceil(cols/64), ceil(rows/64), planes) grid, 256 threads.
template <typename T>
__device__ __forceinline__ void transpose_block(const T* __restrict__ src, T* __restrict__ dst, int rows, int cols, int sp,
                                                int dp, size_t src_plane, size_t dst_plane, int bx, int by, int bz,
                                                void* lds) {
  T(*tile)[65] = (T(*)[65])lds;
  const T* s = src + (size_t)bz * src_plane;
  T* d = dst + (size_t)bz * dst_plane;
  const int x0 = bx * 64, y0 = by * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int y = y0 + r, x = x0 + tx;
    if (y < rows && x < cols) tile[r][tx] = s[(size_t)y * sp + x];
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int x = x0 + r, y = y0 + tx;
    if (x < cols && y < rows) {
      const T val = tile[tx][r];
      d[(size_t)x * dp + y] = val;
      if (x == cols - 1)
        for (int p = 1; p <= kTransPad; ++p) d[(size_t)(x + p) * dp + y] = val;
    }
  }
}

// Line-triple planes of the run engine (PlaneSet::rpg / cpg): built from the plain and the transposed planes once per
// Match.  Block (bx, lg, z) of a (ceil(len / 256), ceil(lines / 4), B * 2) grid with z = b * 2 + view (a thread builds
// the records of four consecutive lines from six loaded ones); `rows_mode` != 0: row triples (len = cols,
// line = image row, source planes img8 / g32 with pitch), else column triples on the transposed planes (len = rows,
// line = image column incl. the replicated pad columns, source timg8 / tg32 with pitch_t).
// Lines beyond the last one repeat it (they are only ever the unused twelfth line of a window).
constexpr int kLinesPerSetupThread = 4;  // a thread of the row-triple / quad sections builds this many consecutive lines
__device__ __forceinline__ void triples_block(const PlaneSet& ps, int rows_mode, int bx, int lg, int z) {
  const int e = bx * blockDim.x + threadIdx.x;
  const int view = z & 1, b = z >> 1;
  const int itgt = view == 0 ? 1 : 2;
  const int len = rows_mode ? ps.cols : ps.rows, nl = rows_mode ? ps.nrl : ps.ncl;
  const int l0 = lg * kLinesPerSetupThread;
  if (e >= len || l0 >= nl) return;
  const int lmax = rows_mode ? ps.rows - 1 : ps.cols + kTransPad - 1;
  const size_t sp = rows_mode ? (size_t)ps.pitch : (size_t)ps.pitch_t;
  const size_t tp = ((size_t)b * 4 + itgt) * (rows_mode ? ps.plane : ps.plane_t);
  const float* g = (rows_mode ? ps.g32 : ps.tg32) + tp;
  const uint8_t* c = (rows_mode ? ps.img8 : ps.timg8) + tp;
  // lines l0 .. l0 + 5 once (a record holds three consecutive lines: four records share six lines)
  float gv[kLinesPerSetupThread + 2];
  uint32_t cv[kLinesPerSetupThread + 2];
#pragma unroll
  for (int k = 0; k < kLinesPerSetupThread + 2; ++k) {
    const size_t so = (size_t)min(l0 + k, lmax) * sp + e;
    gv[k] = g[so];
    cv[k] = c[so];
  }
  float4* dst = (float4*)(rows_mode ? ps.rpg : ps.cpg);
#pragma unroll
  for (int k = 0; k < kLinesPerSetupThread; ++k) {
    if (l0 + k >= nl) break;
    // one 16-byte record per element: three gradients and three colour bytes -- ONE aligned global_load_dwordx4 in the sweeps
    float4 rec;
    rec.x = gv[k];
    rec.y = gv[k + 1];
    rec.z = gv[k + 2];
    rec.w = __builtin_bit_cast(float, cv[k] | (cv[k + 1] << 8) | (cv[k + 2] << 16));
    dst[((size_t)z * nl + (l0 + k)) * sp + e] = rec;
  }
}

// Reference quads of the row sweeps (PlaneSet::rqk).  Block (bx, lg, z) of a (ceil(cols / 256), ceil(nrl / 4), B * 2) grid,
// z = b * 2 + view: rows 4 lg .. 4 lg + 3.  Rows beyond the last one repeat it (only ever the unused twelfth row).
__device__ __forceinline__ void quads_block(const PlaneSet& ps, int bx, int lg, int z) {
  const int e = bx * blockDim.x + threadIdx.x;
  const int view = z & 1, b = z >> 1;
  const int l0 = lg * kLinesPerSetupThread;
  if (e >= ps.cols || l0 >= ps.nrl) return;
  const int iref = view == 0 ? 0 : 3;
  const size_t rp = ((size_t)b * 4 + iref) * ps.plane;
  uint32_t pk[kLinesPerSetupThread + 3];  // rows l0 .. l0 + 6 once (a quad holds four consecutive rows)
#pragma unroll
  for (int k = 0; k < kLinesPerSetupThread + 3; ++k)
    pk[k] = ps.pk16[rp + (size_t)min(l0 + k, ps.rows - 1) * ps.pitch + e];
#pragma unroll
  for (int k = 0; k < kLinesPerSetupThread; ++k) {
    if (l0 + k >= ps.nrl) break;
    uint32_t cw = 0u, gw = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cw |= (pk[k + j] & 0xffu) << (8 * j);
      gw |= (pk[k + j] >> 8) << (8 * j);
    }
    const size_t dst = (((size_t)z * ps.nrl + (l0 + k)) * ps.pitch + e) * 2;
    ps.rqk[dst] = cw;
    ps.rqk[dst + 1] = gw;
  }
}

// Column triples (PlaneSet::cpg) straight from the ROW-MAJOR target planes, transposed through LDS: the same records
// triples_block(rows_mode = 0) builds from the transposed planes (line x of the transposed plane = image column
// min(x, cols - 1): the pad lines replicate the last column), without waiting for those planes to exist -- so every
// derived plane of a Match comes out of ONE launch.  Block (bx, by, z): lines 32 bx .. 32 bx + 31, rows 64 by .. 64 by + 63;
// `lds` holds 64 x 35 floats + 64 x 36 bytes.
__device__ __forceinline__ void col_triples_tile(const PlaneSet& ps, int bx, int by, int z, void* lds) {
  constexpr int TL = 32, TE = 64, GW = TL + 3, CW = TL + 4;
  float(*tg)[GW] = (float(*)[GW])lds;
  uint8_t(*tc)[CW] = (uint8_t(*)[CW])((float*)lds + TE * GW);
  const int view = z & 1, b = z >> 1;
  const int itgt = view == 0 ? 1 : 2;
  const size_t tp = ((size_t)b * 4 + itgt) * ps.plane;
  const float* g = ps.g32 + tp;
  const uint8_t* c = ps.img8 + tp;
  const int x0 = bx * TL, y0 = by * TE;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int r = w; r < TE; r += 4) {
    const size_t row = (size_t)min(y0 + r, ps.rows - 1) * ps.pitch;
    if (lane < TL + 2) {
      const int x = min(x0 + lane, ps.cols - 1);
      tg[r][lane] = g[row + x];
      tc[r][lane] = c[row + x];
    }
  }
  __syncthreads();
  const int e = y0 + lane;
  float4* dst = (float4*)ps.cpg;
  for (int xi = w; xi < TL; xi += 4) {
    const int l = x0 + xi;
    if (l >= ps.ncl || e >= ps.rows) continue;
    float4 rec;
    rec.x = tg[lane][xi];
    rec.y = tg[lane][xi + 1];
    rec.z = tg[lane][xi + 2];
    rec.w = __builtin_bit_cast(float, (uint32_t)tc[lane][xi] | ((uint32_t)tc[lane][xi + 1] << 8) |
                                          ((uint32_t)tc[lane][xi + 2] << 16));
    dst[((size_t)z * ps.ncl + l) * ps.pitch_t + e] = rec;
  }
}

// The planes derived from k_prep's output in ONE launch instead of nine (each of these passes is a few microseconds of
// work behind ~10 us of launch latency, and a Match starts with all of them in a row): the four transposes (u8 image,
// f32 gradient, u8 gradient, u16 packed), the row triples, the reference quads and the column triples -- all read
// k_prep's planes only.  One linear grid; a block finds its section and its (bx, by, bz) there from the section sizes.
struct SetupGrid {
  unsigned tx, ty, tz;  // transpose sections: (ceil(cols/64), ceil(rows/64), B * 4) each
  unsigned lx, ly, lz;  // row triples / quads: (ceil(cols/256), ceil(nrl / kLinesPerSetupThread), B * 2) each
  unsigned cx, cy, cz;  // column triples: (ceil(ncl/32), ceil(rows/64), B * 2)
  int with_lines;       // 0: transposes only (PM_SEM_GPU, plane mode, anchor engines)
  int view;             // -1: both views (tz = B * 4, lz = cz = B * 2); 0 / 1: that view's planes only (tz = B * 2, lz = cz = B)
};
__device__ __forceinline__ void setup_block(const PlaneSet& ps, const SetupGrid& sg, unsigned b, float* lds) {
  // plane / slot indices of a one-view launch: planes 2 * view, 2 * view + 1 of every pair; slot = pair * 2 + view
  auto plane_of = [&](int bz) { return sg.view < 0 ? bz : (bz >> 1) * 4 + 2 * sg.view + (bz & 1); };
  auto slot_of = [&](int z) { return sg.view < 0 ? z : z * 2 + sg.view; };
  const unsigned nt = sg.tx * sg.ty * sg.tz;
  if (b < 4 * nt) {
    const unsigned kind = b / nt;
    b -= kind * nt;
    const int bx = (int)(b % sg.tx), by = (int)((b / sg.tx) % sg.ty), bz = plane_of((int)(b / (sg.tx * sg.ty)));
    if (kind == 0)
      transpose_block<uint8_t>(ps.img8, ps.timg8, ps.rows, ps.cols, ps.pitch, ps.pitch_t, ps.plane, ps.plane_t, bx, by, bz, lds);
    else if (kind == 1)
      transpose_block<float>(ps.g32, ps.tg32, ps.rows, ps.cols, ps.pitch, ps.pitch_t, ps.plane, ps.plane_t, bx, by, bz, lds);
    else if (kind == 2)
      transpose_block<uint8_t>(ps.g8, ps.tg8, ps.rows, ps.cols, ps.pitch, ps.pitch_t, ps.plane, ps.plane_t, bx, by, bz, lds);
    else
      transpose_block<uint16_t>(ps.pk16, ps.tpk16, ps.rows, ps.cols, ps.pitch, ps.pitch_t, ps.plane, ps.plane_t, bx, by, bz, lds);
    return;
  }
  b -= 4 * nt;
  const unsigned nl = sg.lx * sg.ly * sg.lz;
  if (b < 2 * nl) {
    const int bx = (int)(b % sg.lx), l = (int)((b / sg.lx) % sg.ly), z = slot_of((int)((b % nl) / (sg.lx * sg.ly)));
    if (b < nl) triples_block(ps, 1, bx, l, z);
    else quads_block(ps, bx, l, z);
    return;
  }
  b -= 2 * nl;
  col_triples_tile(ps, (int)(b % sg.cx), (int)((b / sg.cx) % sg.cy), slot_of((int)(b / (sg.cx * sg.cy))), lds);
}
// blocks of a SetupGrid
inline unsigned setup_blocks(const SetupGrid& sg) {
  return 4 * sg.tx * sg.ty * sg.tz + (sg.with_lines ? 2 * sg.lx * sg.ly * sg.lz + sg.cx * sg.cy * sg.cz : 0);
}
__global__ void __launch_bounds__(256) k_setup(PlaneSet ps, SetupGrid sg) {
  __shared__ float lds[64 * 65];
  setup_block(ps, sg, blockIdx.x, lds);
}

// seed maps -> disparity planes; the right-view seed is mirrored like the images
// (patchmatch_gpu.cu:362-366).  A null seed pointer means "all background".
__global__ void __launch_bounds__(256) k_seed(PlaneSet ps, const float* __restrict__ seed_l,
                                              const float* __restrict__ seed_r, size_t seed_stride, int view_sel) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z;
  if (x >= ps.cols) return;
  const size_t sp = (size_t)ps.rows * seed_stride;
  const size_t o = state_at(x, y, ps.pitch);
  if (view_sel != 1)
    ps.disp[((size_t)b * 2 + 0) * ps.splane + o] = seed_l ? seed_l[(size_t)b * sp + (size_t)y * seed_stride + x] : 0.f;
  if (ps.n_views > 1 && view_sel != 0)
    ps.disp[((size_t)b * 2 + 1) * ps.splane + o] =
        seed_r ? seed_r[(size_t)b * sp + (size_t)y * seed_stride + (ps.cols - 1 - x)] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// noise + clamp + cost of the current disparity, one lane per pixel.
//   AddForegroundNoise (patchmatch_gpu.cu:298-304): d = max((d + s*U) * [d > 0], 0) -- four
//   elementwise OpenCV-CUDA passes in the reference, fused here;
//   == Patchmatch::AddNoise(disp, amount, disp > 0) (patchmatch.cpp:143-155, called at
//   patchmatch_test.cpp:173-179): both add (int32)rng * amount * 2^-31 where d > 0 (amount is a
//   power of two on every reference call site, so scaling the unit noise is exact).
// Then, for the pixels the sweeps visit, PM_SEM_CPU applies the clamp of patchmatch.cpp:175 and
// both semantics store cost(d) so that no sweep ever re-evaluates the cost of the current value.
// amount < 0 skips the noise (used by the single-stage entry points).
// grid = (ceil(cols/256), rows, slots).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_noise_cost(PlaneSet ps, CostParams cp, Interior in, float amount) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, slot = blockIdx.z;
  if (x >= ps.cols) return;
  const View v = make_view(ps, slot);
  const size_t o = state_at(x, y, ps.pitch);
  float d = v.disp[o];
  if (amount >= 0.f) {
    if (d > 0.f) {
      const float m = ps.noise[(size_t)y * ps.pitch + x] * amount;
      const float s = m + d;
      d = s > 0.f ? s : 0.f;
    } else {
      d = 0.f;
    }
  }
  if (x >= in.x_lo && x <= in.x_hi && y >= in.y_lo && y <= in.y_hi) {
    float c;
    if (cp.semantics == 0) {
      const float hi = (float)x - (float)(cp.pw / 2);
      d = d > 0.f ? d : 0.f;
      d = d < hi ? d : hi;
      c = cpu_cost_lane(v, ps.pitch, ps.cols, x, y, d, cp);
    } else {
      c = gpu_cost_lane(v, ps.pitch, x, y, fmaxf((float)x - d, 1.f), cp);
    }
    v.cost[o] = c;
  }
  v.disp[o] = d;
}

// ---------------------------------------------------------------------------------------------
// The same stage for PM_SEM_CPU with the windows staged in LDS.  Every pixel evaluates its OWN
// disparity, so nothing is shared between lanes except the image data: a 32x8 tile of pixels reads
// (32+pw-1) x (8+ph-1) reference pixels and, on the target side, the same rows over the column range
// [min ipx, max ipx + pw] of the tile's lanes (ipx = first target column of a lane's window).  The
// tile is filled with coalesced row reads once (about 12 target pixels per output pixel instead of
// 2*pw*ph scattered loads) and the 3*pw*ph reads per evaluation go to LDS.  If the disparities of a
// tile span more than kTileDispRange columns the block falls back to the global-memory path of
// k_noise_cost -- same arithmetic, same result.
// grid = (ceil(cols/32), ceil(rows/8), slots), block = 256.
// ---------------------------------------------------------------------------------------------
constexpr int kTileW = 32, kTileH = 8;
constexpr int kTileRW = 224;  // columns of the target tile held in LDS

// keep_zero != 0: the cost plane already holds cost(d) for this window (it does after any sweep:
// adoption updates it), so a pixel whose disparity is 0 before the noise -- the noise leaves it 0
// (patchmatch_gpu.cu:300-303) and the clamp too -- needs no new evaluation.  Exact; saves whole
// tiles in background regions.
template <int TPW, int TPH>
__global__ void __launch_bounds__(256) k_noise_cost_tiled(PlaneSet ps, CostParams cp, Interior in, float amount,
                                                          int keep_zero) {
  constexpr int PW = TPW, PH = TPH;
  constexpr int LW = kTileW + PW - 1, TR = kTileH + PH - 1;
  // The reference tile starts at an EVEN image column (x0 - PW / 2 - OFF): its rows are whole pixel pairs of the packed
  // plane (one dword load = two pixels); the row stride is a multiple of four bytes, so a lane's byte shift is the
  // same in every row.
  constexpr int OFF = (PW / 2) & 1;
  constexpr int NPAIR = (OFF + LW + 1) / 2;
  constexpr int LWP = (2 * NPAIR + 3) & ~3;
  constexpr int NREF = TR * NPAIR, NREFK = (NREF + 255) / 256;  // pixel pairs of the reference tile, per thread
  constexpr int LPR = 256 / TR;                                 // threads per row of the target tile
  constexpr int KMAX = (kTileRW / 4 + LPR - 1) / LPR;           // four-column groups per thread
  // byte tiles as dword arrays: window bytes are fetched as ALIGNED dwords and shifted into place
  // (v_alignbyte_b32).  Unaligned ds_read_b96/b128 made this kernel LDS-bound: SQ_LDS_UNALIGNED_STALL was
  // 70 % of SQ_LDS_IDX_ACTIVE, which itself equalled the kernel's duration (profiles/r01f_pmc_lds.txt).
  __shared__ unsigned s_l8w[TR * LWP / 4 + 4];
  __shared__ unsigned s_lgw[TR * LWP / 4 + 4];
  __shared__ unsigned s_r8w[TR * kTileRW / 4 + 4];
  __shared__ __attribute__((aligned(16))) float s_rg[TR * kTileRW];
  uint8_t* const s_l8 = (uint8_t*)s_l8w;
  uint8_t* const s_lg = (uint8_t*)s_lgw;
  uint8_t* const s_r8 = (uint8_t*)s_r8w;
  __shared__ int s_red[8];

#ifdef PM_TUNING
  const int dbg = keep_zero >> 8;  // timing experiments of the tuning build (bit 0: no window rows, bit 1: no fill)
  keep_zero &= 0xff;
#endif
  const int tid = threadIdx.x, tx = tid & (kTileW - 1), ty = tid / kTileW;
  const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
  const int x = x0 + tx, y = y0 + ty, slot = blockIdx.z;
  const View v = make_view(ps, slot);
  const int cols = ps.cols, rows = ps.rows, pitch = ps.pitch;
  const bool inimg = x < cols && y < rows;
  const size_t o = state_at(x, y, pitch);  // state planes

  float d = 0.f;
  bool was_zero = false;
  if (inimg) {
    // (the unit noise is loaded beside the disparity, not behind it: one memory latency instead of two)
    const float unit = amount >= 0.f ? ps.noise[(size_t)y * pitch + x] : 0.f;
    d = v.disp[o];
    was_zero = !(d > 0.f);
    if (amount >= 0.f) {
      if (d > 0.f) {
        const float m = unit * amount;
        const float s = m + d;
        d = s > 0.f ? s : 0.f;
      } else {
        d = 0.f;
      }
    }
  }
  const bool inside = inimg && x >= in.x_lo && x <= in.x_hi && y >= in.y_lo && y <= in.y_hi;
  if (keep_zero && inside && was_zero) {  // d == 0 stays 0 and its cost is already stored
    v.disp[o] = 0.f;
  }
  const bool interior = inside && !(keep_zero && was_zero);
  CpuLerp l;
  l.ipx = 0;
  if (interior) {
    const float hi = (float)x - (float)(PW / 2);
    d = d > 0.f ? d : 0.f;
    d = d < hi ? d : hi;
    l = cpu_lerp(x, d, PW);
  }

  // ---- reference tile: the loads go out now (they do not depend on the disparities) and land in LDS after the
  // column range is known.  Fast form: whole pixel pairs, when the tile's columns lie inside the image ---------------
  const int ry0 = y0 - PH / 2, ls = x0 - PW / 2 - OFF;
  const bool ref_fast = ls >= 0 && ls + 2 * NPAIR <= cols && (reinterpret_cast<uintptr_t>(v.refpk) & 3u) == 0;  // uniform
  unsigned refv[NREFK];
  if (ref_fast) {
#pragma unroll
    for (int kk = 0; kk < NREFK; ++kk) {
      const int e = tid + 256 * kk;
      refv[kk] = 0u;
      if (e < NREF) {
        const int rr = e / NPAIR, pp = e - rr * NPAIR;
        const int gy = min(max(ry0 + rr, 0), rows - 1);
        refv[kk] = *reinterpret_cast<const unsigned*>(v.refpk + (size_t)gy * pitch + (ls + 2 * pp));
      }
    }
  }

  // ---- column range of the target tile: min / max of ipx over the block's interior lanes --------------
  int lo = interior ? l.ipx : 0x7fffffff, hi_ = interior ? l.ipx : -0x7fffffff;
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) {
    lo = min(lo, __shfl_xor(lo, ofs, 64));
    hi_ = max(hi_, __shfl_xor(hi_, ofs, 64));
  }
  if ((tid & 63) == 0) {
    s_red[(tid >> 6) * 2] = lo;
    s_red[(tid >> 6) * 2 + 1] = hi_;
  }
  __syncthreads();
  lo = min(min(s_red[0], s_red[2]), min(s_red[4], s_red[6]));
  hi_ = max(max(s_red[1], s_red[3]), max(s_red[5], s_red[7]));
  if (hi_ < lo) {  // no pixel of this tile needs an evaluation (uniform)
    if (inimg) v.disp[o] = d;
    return;
  }
  // columns needed: [lo, hi_ + PW]; the tile starts at the multiple of four below lo (16-byte loads of the gradients)
  const int lo4 = lo & ~3;
  const int rw = hi_ + PW + 1 - lo4;
  const bool tiled = rw <= kTileRW;  // uniform

  float c = 0.f;
  if (tiled) {
    // ---- fill: rows y0-PH/2 .. y0+kTileH-1+PH/2, clamped (clamped rows/columns are only ever read by
    // pixels that are not interior, or carry weight 0) -----------------------------------------------
#ifdef PM_TUNING
    if (!(dbg & 2)) {
#else
    {
#endif
      if (ref_fast) {
#pragma unroll
        for (int kk = 0; kk < NREFK; ++kk) {
          const int e = tid + 256 * kk;
          if (e < NREF) {
            const int rr = e / NPAIR, pp = e - rr * NPAIR;
            // dword = colour, gradient, colour, gradient of two pixels -> two colour bytes, two gradient bytes
            *reinterpret_cast<uint16_t*>(s_l8 + rr * LWP + 2 * pp) = (uint16_t)__builtin_amdgcn_perm(0u, refv[kk], 0x0c0c0200u);
            *reinterpret_cast<uint16_t*>(s_lg + rr * LWP + 2 * pp) = (uint16_t)__builtin_amdgcn_perm(0u, refv[kk], 0x0c0c0301u);
          }
        }
      } else {  // a tile at the image border: element by element, clamped columns
        const int fw = tid >> 6, fl = tid & 63, lx0 = x0 - PW / 2;
        for (int rr = fw; rr < TR; rr += 4) {
          const int gy = min(max(ry0 + rr, 0), rows - 1);
          const uint16_t* prow = v.refpk + (size_t)gy * pitch;
          for (int cc = fl; cc < LW; cc += 64) {
            const unsigned pk = prow[min(max(lx0 + cc, 0), cols - 1)];
            s_l8[rr * LWP + cc + OFF] = (uint8_t)(pk & 0xffu);
            s_lg[rr * LWP + cc + OFF] = (uint8_t)(pk >> 8);
          }
        }
      }
      const int n4 = (rw + 3) >> 2;
      const bool tgt_fast = lo4 >= 0 && lo4 + 4 * n4 <= cols && (reinterpret_cast<uintptr_t>(v.tgt8) & 3u) == 0 &&
                            (reinterpret_cast<uintptr_t>(v.tgtg) & 15u) == 0;  // uniform
      if (tgt_fast) {
        // LPR threads per tile row, four columns each: one 16-byte load of gradients and one dword of colour bytes
        const int rr = tid / LPR, q0 = tid - rr * LPR;
        if (rr < TR) {
          const int gy = min(max(ry0 + rr, 0), rows - 1);
          typedef float f32x4 __attribute__((ext_vector_type(4)));
          const f32x4* g4 = reinterpret_cast<const f32x4*>(v.tgtg + (size_t)gy * pitch + lo4);
          const unsigned* t4 = reinterpret_cast<const unsigned*>(v.tgt8 + (size_t)gy * pitch + lo4);
          f32x4* const sg4 = reinterpret_cast<f32x4*>(s_rg + rr * kTileRW);
          unsigned* const st4 = s_r8w + rr * (kTileRW / 4);
          f32x4 gq0 = {0.f, 0.f, 0.f, 0.f}, gq1 = gq0, gq2 = gq0, gq3 = gq0;
          unsigned tq0 = 0u, tq1 = 0u, tq2 = 0u, tq3 = 0u;
          static_assert(KMAX <= 4, "four-column groups per thread");
          const int q1 = q0 + LPR, q2 = q0 + 2 * LPR, q3 = q0 + 3 * LPR;
          if (q0 < n4) { gq0 = g4[q0]; tq0 = t4[q0]; }
          if (KMAX > 1 && q1 < n4) { gq1 = g4[q1]; tq1 = t4[q1]; }
          if (KMAX > 2 && q2 < n4) { gq2 = g4[q2]; tq2 = t4[q2]; }
          if (KMAX > 3 && q3 < n4) { gq3 = g4[q3]; tq3 = t4[q3]; }
          if (q0 < n4) { sg4[q0] = gq0; st4[q0] = tq0; }
          if (KMAX > 1 && q1 < n4) { sg4[q1] = gq1; st4[q1] = tq1; }
          if (KMAX > 2 && q2 < n4) { sg4[q2] = gq2; st4[q2] = tq2; }
          if (KMAX > 3 && q3 < n4) { sg4[q3] = gq3; st4[q3] = tq3; }
        }
      } else {
        const int fw = tid >> 6, fl = tid & 63;
        for (int rr = fw; rr < TR; rr += 4) {
          const int gy = min(max(ry0 + rr, 0), rows - 1);
          const uint8_t* trow = v.tgt8 + (size_t)gy * pitch;
          const float* grow = v.tgtg + (size_t)gy * pitch;
          for (int cc = fl; cc < rw; cc += 64) {
            const int gx = min(max(lo4 + cc, 0), cols - 1);
            s_r8[rr * kTileRW + cc] = trow[gx];
            s_rg[rr * kTileRW + cc] = grow[gx];
          }
        }
      }
    }
    __syncthreads();
#ifdef PM_TUNING
    if (interior && !(dbg & 1)) {
#else
    if (interior) {
#endif
      // Four taps per v_sad_u8: the window row's reference bytes are consecutive in LDS and are used as
      // loaded; the four colour samples (byte 2 of the 16.16 fixed-point value) are gathered with
      // v_perm_b32, the four gradient samples are rounded and packed by v_cvt_pk_u8_f32.
      unsigned sc = 0, sg = 0;
      const int rc = l.ipx - lo4;
      constexpr int NG = (PW + 3) / 4;  // groups of four taps; the last one may be partial
      constexpr int NR = (PW + 4) / 4;  // dwords covering r[0 .. PW]
      // 4 * NG reference bytes from flat byte offset fl0 (+ a row), 4 * NR target bytes from fr0 (+ a row): the row strides
      // are multiples of four bytes, so dword index and byte shift of row i are those of row 0 plus a constant
      const int fl0 = ty * LWP + tx + OFF, fr0 = ty * kTileRW + rc;
      // LDS pointers the compiler cannot see through (an absolute LDS address does not fit a ds_read2 offset field: every
      // pair of dwords would get its own address add); re-based where the row offset outgrows the 8-bit dword offset
      typedef __attribute__((address_space(3))) const unsigned* LdsU32;
      typedef __attribute__((address_space(3))) const float* LdsF32;
      LdsU32 pl = (LdsU32)(s_l8w + (fl0 >> 2));
      LdsU32 pg = (LdsU32)(s_lgw + (fl0 >> 2));
      LdsU32 pr = (LdsU32)(s_r8w + (fr0 >> 2));
      LdsF32 rg = (LdsF32)(s_rg + fr0);
      asm volatile("" : "+v"(pl), "+v"(pg), "+v"(pr), "+v"(rg));
      const unsigned shl = (unsigned)fl0 & 3u, shr = (unsigned)fr0 & 3u;
      static_assert(LWP % 4 == 0 && kTileRW % 4 == 0, "row strides in whole dwords");
      static_assert((PH - 1) * (LWP / 4) + NG < 256, "reference rows within one ds_read2 offset range");
      constexpr int kRowsPerBase = 4;  // target byte rows per base: 4 * kTileRW / 4 + NR dwords < 256
      static_assert((kRowsPerBase - 1) * (kTileRW / 4) + NR < 256, "target rows within one ds_read2 offset range");
#pragma unroll
      for (int i = 0; i < PH; ++i) {  // unrolled: every LDS address is a base register plus a constant
        if (i > 0) {
          rg += kTileRW;
          asm volatile("" : "+v"(rg));
          if (i % kRowsPerBase == 0) {
            pr += kRowsPerBase * (kTileRW / 4);
            asm volatile("" : "+v"(pr));
          }
        }
        unsigned lw[NG], lgw[NG], rw[NR];
        {
          const LdsU32 pl_i = pl + i * (LWP / 4), pg_i = pg + i * (LWP / 4), pr_i = pr + (i % kRowsPerBase) * (kTileRW / 4);
          unsigned a0 = pl_i[0], b0 = pg_i[0], c0 = pr_i[0];
#pragma unroll
          for (int q = 0; q < NR; ++q) {
            if (q < NG) {
              const unsigned a1 = pl_i[q + 1], b1 = pg_i[q + 1];
              lw[q] = __builtin_amdgcn_alignbyte(a1, a0, shl);
              lgw[q] = __builtin_amdgcn_alignbyte(b1, b0, shl);
              a0 = a1;
              b0 = b1;
            }
            const unsigned c1 = pr_i[q + 1];
            rw[q] = __builtin_amdgcn_alignbyte(c1, c0, shr);
            c0 = c1;
          }
        }
        // target bytes j and j + 1 as the two halfwords of a dword (they may straddle two dwords of rw[]): the
        // operand of the one-instruction colour lerp (cpu_color_sum_pk)
        auto rpair = [&](int j) -> unsigned {
          const unsigned sel = 0x0c000c00u | (unsigned)(j % 4) | ((unsigned)(j % 4 + 1) << 16);
          return __builtin_amdgcn_perm(rw[min(j / 4 + 1, NR - 1)], rw[j / 4], sel);
        };
        const unsigned cw = cpu_color_weights(l);
        // gradient lerp g[j] * (1 - a) + g[j + 1] * a: both products of every sample with packed-f32
        // multiplies (two samples per instruction), one add per tap; each product and each sum is a single
        // IEEE operation exactly as in the scalar form
        constexpr int NP = (PW + 2) / 2;  // pairs covering g[0 .. PW]
        f32x2 ga[NP], gb[NP];
        const f32x2 ia2 = {l.ia, l.ia}, a2 = {l.a, l.a};
#pragma unroll
        for (int k2 = 0; k2 < NP; ++k2) {
          const f32x2 gg = {rg[2 * k2], rg[min(2 * k2 + 1, PW)]};
          ga[k2] = gg * ia2;
          gb[k2] = gg * a2;
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
          // bytes of the last group that lie beyond the window take the REFERENCE's byte on both sides of the SAD (their
          // difference is 0): no masks
          const int rem = PW - 4 * q;  // taps in this group
          unsigned t[4] = {0, 0, 0, 0};
          unsigned pg = lgw[q];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int j = 4 * q + k;
            if (j < PW) {
              t[k] = cpu_color_sum_pk(rpair(j), cw);
              const float sgr = ga[j / 2][j % 2] + gb[(j + 1) / 2][(j + 1) % 2];
              pg = __builtin_amdgcn_cvt_pk_u8_f32(sgr, k, pg);  // saturate_cast<uchar> into byte k
            }
          }
          // v_perm_b32(S0, S1, sel): selector 0-3 = bytes of S1, 4-7 = bytes of S0, 0x0c = 0x00
          unsigned pc;
          if (rem >= 4) {
            pc = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0602u) | __builtin_amdgcn_perm(t[3], t[2], 0x06020c0cu);
          } else if (rem == 3) {
            pc = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0602u) | __builtin_amdgcn_perm(lw[q], t[2], 0x07020c0cu);
          } else if (rem == 2) {
            pc = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0602u) | (lw[q] & 0xffff0000u);
          } else {
            pc = __builtin_amdgcn_perm(lw[q], t[0], 0x07060502u);
          }
          sc = __builtin_amdgcn_sad_u8(lw[q], pc, sc);
          sg = __builtin_amdgcn_sad_u8(lgw[q], pg, sg);
        }
      }
      c = cpu_cost_from_sums((int)sc, (int)sg, cp);
    }
  } else if (interior) {
    c = cpu_cost_lane(v, pitch, cols, x, y, d, cp);
  }
  if (interior) v.cost[o] = c;
  if (inimg) v.disp[o] = d;
}

// ---------------------------------------------------------------------------------------------
// Background mask, one lane per pixel.
//   PM_SEM_CPU  RemoveBackground (patchmatch.cpp:314-360): d0 = clamp(d); zero d if
//               cost(d0) > cost(0) / win_by_factor.  cost(d0) comes from the cost plane when the
//               window is the one the last sweeps used and d is already inside the clamp.
//   PM_SEM_GPU  MaskBackground (patchmatch_gpu.cu:233-270): zero d unless cost(d) < f * cost(0).
// `cached` != 0: ps.cost holds cost(d) for this window.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_background(PlaneSet ps, CostParams cp, Interior in, float factor, int cached) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, slot = blockIdx.z;
  if (x < in.x_lo || x > in.x_hi || y < in.y_lo || y > in.y_hi) return;
  const View v = make_view(ps, slot);
  const size_t o = state_at(x, y, ps.pitch);
  const float d = v.disp[o];
  if (cp.semantics == 0) {
    const float hi = (float)x - (float)(cp.pw / 2);
    float d0 = d > 0.f ? d : 0.f;
    d0 = d0 < hi ? d0 : hi;
    const float c = (cached && d0 == d) ? v.cost[o] : cpu_cost_lane(v, ps.pitch, ps.cols, x, y, d0, cp);
    const float c_bg = cpu_cost_lane(v, ps.pitch, ps.cols, x, y, 0.f, cp);
    const float thr = c_bg / factor;
    if (c > thr) v.disp[o] = 0.f;
  } else {
    const float cost0 = gpu_cost_lane(v, ps.pitch, x, y, (float)x, cp);
    const float cost1 = cached ? v.cost[o] : gpu_cost_lane(v, ps.pitch, x, y, fmaxf((float)x - d, 1.f), cp);
    const float thr = factor * cost0;
    if (!(cost1 < thr)) v.disp[o] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// RemoveBackground for PM_SEM_CPU square windows, tiled.  cost(0) compares the two windows at the SAME
// position: the bilinear fraction is 0, so the colour sample is the target byte itself
// ((r0 * 65536 + 32768) >> 16) and the gradient sample is saturate_cast<uchar>(g0 * 1 + g1 * 0) = the
// target's g8 byte -- cost(0) is a plain byte SAD of two 4-plane windows.  A 32x8 tile stages the
// (32 + PW - 1) x (8 + PH - 1) bytes of the four u8 planes in LDS (as dwords: aligned reads +
// v_alignbyte_b32) and every pixel does 2 * ceil(PW / 4) v_sad_u8 per window row.
// grid = (ceil(cols/32), ceil(rows/8), slots), block = 256.
// ---------------------------------------------------------------------------------------------
template <int TPW, int TPH>
__global__ void __launch_bounds__(256) k_background_tiled(PlaneSet ps, CostParams cp, Interior in, float factor,
                                                          int cached) {
  constexpr int PW = TPW, PH = TPH;
  constexpr int LW = kTileW + PW - 1, TR = kTileH + PH - 1;
  constexpr int NW = (TR * LW + 3) / 4 + 4;
  __shared__ unsigned s_w[4][NW];  // l8, lg8, r8, rg8

  const int tid = threadIdx.x, tx = tid & (kTileW - 1), ty = tid / kTileW;
  const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
  const int x = x0 + tx, y = y0 + ty, slot = blockIdx.z;
  const View v = make_view(ps, slot);
  const int cols = ps.cols, rows = ps.rows, pitch = ps.pitch;
  const int ry0 = y0 - PH / 2, lx0 = x0 - PW / 2;
  {
    // colour and gradient byte of a pixel in one u16 load (the packed planes), every load of a thread before its stores
    const uint16_t* tgtpk = v.refpk + (v.tgt8 - v.ref8);  // the packed plane of the target image (same plane order)
    constexpr int NE = (TR * LW + 255) / 256;
    unsigned pr[NE], pt[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {  // (an element beyond the tile re-reads its last one: loaded, never stored)
      const int e = min(tid + 256 * u, TR * LW - 1);
      const int rr = e / LW, cc = e - rr * LW;
      const int gy = min(max(ry0 + rr, 0), rows - 1), gx = min(max(lx0 + cc, 0), cols - 1);
      const size_t go = (size_t)gy * pitch + gx;
      pr[u] = v.refpk[go];
      pt[u] = tgtpk[go];
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + 256 * u;
      if (e < TR * LW) {
        ((uint8_t*)s_w[0])[e] = (uint8_t)(pr[u] & 0xffu);
        ((uint8_t*)s_w[1])[e] = (uint8_t)(pr[u] >> 8);
        ((uint8_t*)s_w[2])[e] = (uint8_t)(pt[u] & 0xffu);
        ((uint8_t*)s_w[3])[e] = (uint8_t)(pt[u] >> 8);
      }
    }
  }
  __syncthreads();
  if (x < in.x_lo || x > in.x_hi || y < in.y_lo || y > in.y_hi) return;

  constexpr int NG = (PW + 3) / 4;
  unsigned sc = 0, sg = 0;
#pragma unroll 1
  for (int i = 0; i < PH; ++i) {
    const int fl = (ty + i) * LW + tx;
    const unsigned sh = (unsigned)fl & 3u;
    const int w0 = fl >> 2;
    unsigned a0 = s_w[0][w0], b0 = s_w[1][w0], c0 = s_w[2][w0], d0 = s_w[3][w0];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const unsigned a1 = s_w[0][w0 + q + 1], b1 = s_w[1][w0 + q + 1], c1 = s_w[2][w0 + q + 1], d1 = s_w[3][w0 + q + 1];
      const int rem = PW - 4 * q;
      const unsigned mask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
      sc = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(a1, a0, sh) & mask,
                                   __builtin_amdgcn_alignbyte(c1, c0, sh) & mask, sc);
      sg = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(b1, b0, sh) & mask,
                                   __builtin_amdgcn_alignbyte(d1, d0, sh) & mask, sg);
      a0 = a1;
      b0 = b1;
      c0 = c1;
      d0 = d1;
    }
  }
  const float c_bg = cpu_cost_from_sums((int)sc, (int)sg, cp);
  const size_t o = state_at(x, y, pitch);
  const float d = v.disp[o];
  const float hi = (float)x - (float)(PW / 2);
  float dd = d > 0.f ? d : 0.f;
  dd = dd < hi ? dd : hi;
  const float c = (cached && dd == d) ? v.cost[o] : cpu_cost_lane(v, pitch, cols, x, y, dd, cp);
  const float thr = c_bg / factor;
  if (c > thr) v.disp[o] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// finalize: un-mirror the right view (cu::flip, patchmatch_gpu.cu:368) and MaskOcclusions
// (:273-295) fused with the copy into the caller's tightly packed outputs.
//   dr = dispr(y, (int)max(x - dl, 0));  zero dl if dr > 1.4*dl || dr < 0.7*dl  (double compare)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_finalize(PlaneSet ps, float* __restrict__ out_l, float* __restrict__ out_r,
                                                  size_t out_stride) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z;
  if (x >= ps.cols) return;
  const size_t op = (size_t)ps.rows * out_stride;
  const float* dl_plane = ps.disp + ((size_t)b * 2 + 0) * ps.splane;
  float dl = dl_plane[state_at(x, y, ps.pitch)];
  if (ps.n_views > 1) {
    const float* dr_plane = ps.disp + ((size_t)b * 2 + 1) * ps.splane;
    const int xr = (int)fmaxf((float)x - dl, 0.f);
    const float dr = dr_plane[state_at(ps.cols - 1 - xr, y, ps.pitch)];
    if ((double)dr > 1.4 * (double)dl || (double)dr < 0.7 * (double)dl) dl = 0.f;
    if (out_r) out_r[(size_t)b * op + (size_t)y * out_stride + x] = dr_plane[state_at(ps.cols - 1 - x, y, ps.pitch)];
  }
  out_l[(size_t)b * op + (size_t)y * out_stride + x] = dl;
}

// Stand-alone MaskOcclusions on caller planes (single-stage entry point).
__global__ void __launch_bounds__(256) k_mask_occlusions(float* __restrict__ displ, const float* __restrict__ dispr,
                                                         int rows, int cols) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= cols || y >= rows) return;
  const float dl = displ[(size_t)y * cols + x];
  const int xr = (int)fmaxf((float)x - dl, 0.f);
  const float dr = dispr[(size_t)y * cols + xr];
  if ((double)dr > 1.4 * (double)dl || (double)dr < 0.7 * (double)dl) displ[(size_t)y * cols + x] = 0.f;
}

// Row-tiled mode: columns flagged in mask[n_views][cols] get their disparity / cost back from the snapshot taken
// before the vertical sweep (the re-sweep of those columns then starts from the pre-sweep state).
__global__ void __launch_bounds__(256) k_restore_cols(PlaneSet ps, const float* __restrict__ snap_disp,
                                                      const float* __restrict__ snap_cost, const int* __restrict__ mask) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, v = blockIdx.z;
  if (x >= ps.cols || !mask[v * ps.cols + x]) return;
  const size_t o = (size_t)v * ps.splane + state_at(x, y, ps.pitch);
  ps.disp[o] = snap_disp[o];
  ps.cost[o] = snap_cost[o];
}

// Row-tiled mode, one exchange round of a vertical sweep in ONE launch (round 4: a row copy, a row compare, a column
// restore over the whole band and a row store were four launches and a copy).  `incoming` is the neighbour band's
// boundary row as it stands NOW -- read where the neighbour published it, also across devices when peer access is
// enabled -- `used` the one the last sweep saw.  A column whose value changed is flagged in `mask`, its rows
// y_lo .. y_hi (the rows the sweeps of this band write) go back to the snapshot, and row `pred_r` of the band (the
// neighbour's row, outside those) takes the incoming value; `used_next` (if not the incoming row itself) receives a copy
// for the next round's comparison.  grid = (cols / 256, chunks of kTileRoundRows rows, views): a thread restores one
// chunk of one column, all loads before the stores; blocks without a changed column end after two loads.
constexpr int kTileRoundRows = 16;
__global__ void __launch_bounds__(256) k_tile_round(PlaneSet ps, const float* __restrict__ snap_disp,
                                                    const float* __restrict__ snap_cost, const float* incoming,
                                                    const float* __restrict__ used, float* used_next,
                                                    int* __restrict__ mask, int pred_r, int y_lo, int y_hi) {
  // (incoming and used_next may be the same buffer: no __restrict__ on them)
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.z;
  if (x >= ps.cols) return;
  const size_t e = (size_t)v * ps.cols + x;
  const float in = incoming[e];
  const bool changed = in != used[e];
  const size_t base = (size_t)v * ps.splane;
  if (blockIdx.y == 0) {
    mask[e] = changed ? 1 : 0;
    if (used_next != incoming) used_next[e] = in;
    if (changed) ps.disp[base + state_at(x, pred_r, ps.pitch)] = in;
  }
  if (!changed) return;
  const int y0 = y_lo + (int)blockIdx.y * kTileRoundRows;
  float d[kTileRoundRows], c[kTileRoundRows];
#pragma unroll
  for (int i = 0; i < kTileRoundRows; ++i) {
    const int y = y0 + i;
    const size_t o = base + state_at(x, y <= y_hi ? y : y_hi, ps.pitch);
    d[i] = snap_disp[o];
    c[i] = snap_cost[o];
  }
#pragma unroll
  for (int i = 0; i < kTileRoundRows; ++i) {
    const int y = y0 + i;
    if (y <= y_hi) {
      const size_t o = base + state_at(x, y, ps.pitch);
      ps.disp[o] = d[i];
      ps.cost[o] = c[i];
    }
  }
}
// Row-tiled mode, in front of a vertical sweep, ONE launch (round 4: a row store and two runtime copies): the snapshot of
// both state planes of every view (whole 16-byte pieces: four interleaved rows of a column), and -- if `row` is given --
// the neighbour band's boundary row into row `pred_r` of the disparity plane (in the snapshot as well: a restore never
// touches that row, but the snapshot is "the planes as the sweep found them").  grid = (ceil(pitch / 256), rows4 / 4... see launch).
__global__ void __launch_bounds__(256) k_tile_presweep(PlaneSet ps, float* __restrict__ snap_disp, float* __restrict__ snap_cost,
                                                       const float* __restrict__ row, int pred_r) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int x = blockIdx.x * blockDim.x + threadIdx.x;  // column (one 16-byte piece = rows 4 q .. 4 q + 3 of column x)
  const int q = blockIdx.y, v = blockIdx.z;
  if (x >= ps.pitch) return;
  const size_t o = (size_t)v * ps.splane + (((size_t)q * (size_t)ps.pitch + (size_t)x) << 2);
  f32x4 d = *reinterpret_cast<const f32x4*>(ps.disp + o);
  const f32x4 c = *reinterpret_cast<const f32x4*>(ps.cost + o);
  if (row && (pred_r >> 2) == q && x < ps.cols) {
    const float in = row[(size_t)v * ps.cols + x];
    d[pred_r & 3] = in;
    ps.disp[o + (pred_r & 3)] = in;
  }
  *reinterpret_cast<f32x4*>(snap_disp + o) = d;
  *reinterpret_cast<f32x4*>(snap_cost + o) = c;
}

// ... and the question at the end of a vertical sweep in one launch (round 4: row copy + compare): does image row r of
// the planes differ from `ref` ([n_views][cols], the row last sent to the successor)?  Then *flag becomes 1.
__global__ void __launch_bounds__(256) k_state_row_moved(PlaneSet ps, int r, const float* __restrict__ ref,
                                                         int* __restrict__ flag) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (x >= ps.cols) return;
  const bool moved = ps.disp[(size_t)v * ps.splane + state_at(x, r, ps.pitch)] != ref[(size_t)v * ps.cols + x];
  if (__builtin_amdgcn_ballot_w64(moved) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// Row-tiled mode: image row r of every view between the disparity planes and a tight [n_views][cols] buffer.
__global__ void __launch_bounds__(256) k_state_row(PlaneSet ps, int r, float* __restrict__ buf, int to_buf) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (x >= ps.cols) return;
  float* p = ps.disp + (size_t)v * ps.splane + state_at(x, r, ps.pitch);
  if (to_buf) buf[(size_t)v * ps.cols + x] = *p;
  else *p = buf[(size_t)v * ps.cols + x];
}

// Plain copies between caller planes and the pitched disparity plane of view 0.
__global__ void __launch_bounds__(256) k_copy_in(PlaneSet ps, const float* __restrict__ src) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= ps.cols) return;
  ps.disp[state_at(x, y, ps.pitch)] = src[(size_t)y * ps.cols + x];
}
// caller plane with its own row stride (elements); to_caller != 0 copies the other way
__global__ void __launch_bounds__(256) k_copy_disp_strided(PlaneSet ps, float* __restrict__ buf, size_t stride,
                                                           int to_caller) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= ps.cols) return;
  if (to_caller) buf[(size_t)y * stride + x] = ps.disp[state_at(x, y, ps.pitch)];
  else ps.disp[state_at(x, y, ps.pitch)] = buf[(size_t)y * stride + x];
}
// Disparity maps out of device memory into page-locked HOST memory, by a few wavefronts.  The runtime performs a
// device-to-host hipMemcpyAsync on a busy stream with a blit kernel of its own, one per map, whose waves wait on the
// bus for 71 us while they hold CU slots the sweeps of the next frames want (rocprofv3 kernel trace of a frame sequence:
// 2 x __amd_rocclr_copyBuffer per frame; the sequence ran 6 % below the same frames without downloads).  PCIe takes
// ~55 GB/s whatever the number of waves behind it, so this copy runs on kDownloadBlocks workgroups: a grid-stride loop,
// 16 bytes per lane and step.  rows x cols floats, source tight, destination row stride dst_step floats.
constexpr int kDownloadBlocks = 16;
__global__ void __launch_bounds__(256) k_download(float* __restrict__ dst, size_t dst_step, const float* __restrict__ src,
                                                  int rows, int cols) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (size_t)gridDim.x * blockDim.x;
  if (dst_step == (size_t)cols && (((size_t)dst | (size_t)src) & 15) == 0) {
    const size_t total = (size_t)rows * cols, n4 = total / 4;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f* s4 = (const v4f*)src;
    v4f* d4 = (v4f*)dst;
    for (size_t i = tid; i < n4; i += nthreads) __builtin_nontemporal_store(__builtin_nontemporal_load(s4 + i), d4 + i);
    for (size_t i = n4 * 4 + tid; i < total; i += nthreads) dst[i] = src[i];
    return;
  }
  const size_t total = (size_t)rows * cols;
  for (size_t i = tid; i < total; i += nthreads) {
    const size_t y = i / (size_t)cols, x = i - y * (size_t)cols;
    dst[y * dst_step + x] = src[i];
  }
}
__global__ void __launch_bounds__(256) k_copy_out(PlaneSet ps, float* __restrict__ dst, int which) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= ps.cols) return;
  const float* src = which == 0 ? ps.disp : (which == 1 ? ps.g32 : ps.noise);
  dst[(size_t)y * ps.cols + x] = src[which == 0 ? state_at(x, y, ps.pitch) : (size_t)y * ps.pitch + x];
}

}  // namespace pm
