// pm_seed.hpp -- sparse seeding on the device: the replacement of PatchmatchGpu::SparseInit
// (src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:414-442), which the reference runs on the CPU twice
// per Match() through cv::GFTTDetector (feature_tracking/feature_detector.cpp:44-57,89-122) and
// cv::matchTemplate (feature_tracking/stereo_matcher.cpp:22-116).
//
// Pipeline (all on the handle's stream, no host synchronisation):
//   k_seed_sobel   u8 image -> Sobel dx, dy (int16)
//   k_seed_eig     block^2 box sums of dx^2, dxdy, dy^2 (exact integers) -> min-eigenvalue response,
//                  global maximum by atomicMax on the float bit pattern (responses >= 0)
//   k_seed_nms     quality threshold + 3x3 non-maximum suppression -> candidate keys
//                  (response bits << 32 | y << 16 | x: raster order)
//   hipcub radix sort, descending: strongest first, ties by larger index (cv::goodFeaturesToTrack's
//                  greaterThanPtr order)
//   k_seed_select  greedy minimum-distance selection, one wavefront (the accepted list lives in LDS)
//   k_seed_match   one workgroup per corner: normalised squared difference of the templ_cols x templ_rows
//                  template against every position of the max_disp x (templ_rows+2) stripe, exact
//                  integer sums, first minimum
//   k_seed_splat   the (2k+1)^2 max-dilation of the <= 1024 matched corners written straight into the zeroed seed
//                  map: one workgroup per corner raises its rectangle with atomicMax on the float bits (what
//                  cv::dilate of the scattered map gives), optionally scaled and nearest-resized as
//                  Patchmatch::Initialize does (patchmatch.cpp:75-81)
// The arithmetic is this build's definition of the seeder (see oracle/pm_oracle.h): OpenCV's float
// pipelines are not reproducible without OpenCV; parity is against oracle/pm_seed_oracle.c.
#pragma once

#include <hipcub/hipcub.hpp>

#include "pm_device.hpp"
#include "pm_tune.hpp"
#include "pm_seed_api.hpp"

namespace pm {

// General reflect-101 (the box window may reach further out than one pixel).
__device__ __forceinline__ int reflect101n(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// cv::cornerMinEigenVal (OpenCV 3.4 imgproc/corner.cpp) in one kernel: 3x3 Sobel derivatives (scale 1, image
// border reflect-101), the block x block box sums of dx^2, dx dy, dy^2 (border reflect-101 ON THE DERIVATIVE
// PLANES: outside sample p = D(reflect(p))), and the smaller eigenvalue.  A workgroup produces a 64 x 32 tile:
// the derivatives of the tile plus its halo are computed once into LDS (dx | dy << 16 per pixel), each thread then
// sums its windows from LDS.  All sums are exact integers (|d| <= 1020, block <= 15: < 2^28).  Also folds the
// block maximum of the positive responses into counters[0] with ONE atomic per block, and only if it can raise
// the maximum (14 400 per-wave atomics on one address serialised: 172 us for a 720p image).
// This replaced a Sobel kernel + a 25-tap gather from global memory (6 + 40 us at 720p).
// BLOCK = 3, 5, 7: compile-time window, every thread owns 8 consecutive rows of its column and forms the
// horizontal sums of the 8 + BLOCK - 1 rows it touches once (7.5 taps per output instead of 25 for BLOCK = 5);
// BLOCK = 0: any odd `block` <= 15, plain double loop.
constexpr int kEigTileW = 64, kEigTileH = 32, kEigRowsPerThread = kEigTileH / 4;
template <int BLOCK>
__global__ void __launch_bounds__(256) k_seed_response(const uint8_t* __restrict__ im, int rows, int cols, int pitch,
                                                       int block_rt, int use_harris, double harris_k,
                                                       float* __restrict__ eig, unsigned* __restrict__ counters) {
  extern __shared__ unsigned s_d[];  // [kEigTileH + 2h][kEigTileW + 2h]
  const int block = BLOCK ? BLOCK : block_rt;
  const int h = block / 2;
  const int W = kEigTileW + 2 * h, H = kEigTileH + 2 * h;
  const int x0 = blockIdx.x * kEigTileW, y0 = blockIdx.y * kEigTileH;
  for (int e = threadIdx.x; e < W * H; e += 256) {
    const int py = e / W, px = e - py * W;
    const int X = reflect101n(x0 - h + px, cols), Y = reflect101n(y0 - h + py, rows);
    const uint8_t* r0 = im + (size_t)reflect101(Y - 1, rows) * pitch;
    const uint8_t* r1 = im + (size_t)Y * pitch;
    const uint8_t* r2 = im + (size_t)reflect101(Y + 1, rows) * pitch;
    const int xm = reflect101(X - 1, cols), xp = reflect101(X + 1, cols);
    const int a = r0[xm], b = r0[X], c = r0[xp], d = r1[xm], f = r1[xp], g = r2[xm], hh = r2[X], k = r2[xp];
    const int gx = (c - a) + 2 * (f - d) + (k - g);
    const int gy = (g - a) + 2 * (hh - b) + (k - c);
    s_d[e] = (unsigned)(gx & 0xffff) | ((unsigned)gy << 16);
  }
  __syncthreads();
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int x = x0 + tx;
  float m = 0.f;
  auto finish = [&](int y, int sxx, int sxy, int syy) {
    if (x < cols && y < rows) {
      float e;
      if (use_harris) {
        // calcHarris (OpenCV 3.4 imgproc/corner.cpp): (float)(a*c - b*b - k*(a + c)*(a + c)) with float a, b, c and a
        // double k -- the products and the difference round to binary32, the trace term is formed in binary64
        const float a = (float)sxx, b = (float)sxy, c = (float)syy;
        const float ac = a * c, bb = b * b;
        const float det = ac - bb;
        const float tr = a + c;
        const double kt = harris_k * (double)tr;
        const double ktt = kt * (double)tr;
        e = (float)((double)det - ktt);
      } else {
        const float a = (float)sxx * 0.5f, b = (float)sxy, c = (float)syy * 0.5f;
        const float t = a - c;
        const float tt = t * t, bb = b * b;
        const float s = a + c;
        e = s - sqrtf(tt + bb);
      }
      eig[(size_t)y * pitch + x] = e;
      m = fmaxf(m, e);  // m starts at 0: only positive responses count
    }
  };
  if constexpr (BLOCK != 0) {
    constexpr int NR = kEigRowsPerThread + BLOCK - 1;
    int hxx[NR], hxy[NR], hyy[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const unsigned* row = s_d + (ty * kEigRowsPerThread + j) * W + tx;
      int sxx = 0, sxy = 0, syy = 0;
#pragma unroll
      for (int i = 0; i < BLOCK; ++i) {
        const unsigned v = row[i];
        const int gx = (int)(short)(v & 0xffffu), gy = (int)v >> 16;
        sxx += gx * gx;
        sxy += gx * gy;
        syy += gy * gy;
      }
      hxx[j] = sxx;
      hxy[j] = sxy;
      hyy[j] = syy;
    }
#pragma unroll
    for (int r = 0; r < kEigRowsPerThread; ++r) {
      int sxx = 0, sxy = 0, syy = 0;
#pragma unroll
      for (int j = 0; j < BLOCK; ++j) {
        sxx += hxx[r + j];
        sxy += hxy[r + j];
        syy += hyy[r + j];
      }
      finish(y0 + ty * kEigRowsPerThread + r, sxx, sxy, syy);
    }
  } else {
    for (int r = 0; r < kEigRowsPerThread; ++r) {
      const int yy = ty * kEigRowsPerThread + r;
      int sxx = 0, sxy = 0, syy = 0;
      for (int j = 0; j < block; ++j) {
        const unsigned* row = s_d + (yy + j) * W + tx;
        for (int i = 0; i < block; ++i) {
          const unsigned v = row[i];
          const int gx = (int)(short)(v & 0xffffu), gy = (int)v >> 16;
          sxx += gx * gx;
          sxy += gx * gy;
          syy += gy * gy;
        }
      }
      finish(y0 + yy, sxx, sxy, syy);
    }
  }
  __shared__ float s_m[4];
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) m = fmaxf(m, __shfl_xor(m, ofs, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (m > __builtin_bit_cast(float, *(volatile unsigned*)&counters[0]))
      atomicMax(&counters[0], __builtin_bit_cast(unsigned, m));
  }
}

// ---- constants shared by k_seed_nms and k_seed_select_fused (see the latter)
constexpr int kSelChunk = 2048;
constexpr int kSelBins = 2048, kSelDigit = 11, kSelUnroll = 16;
constexpr int kSelHist = kSelBins + kSelBins / 32;  // bin b lives at b + (b >> 5): 32-bin runs fall on distinct banks
__host__ __device__ inline size_t seed_select_lds_bytes(int gx, int gy) {
  return sizeof(unsigned long long) * 2 * kSelChunk + sizeof(unsigned) * (kSelHist + 8) +
         sizeof(int) * 4 * (size_t)(gx + 2) * (gy + 2);
}
// shift of the first digit: the kSelDigit bits ending at the highest bit in which max and threshold differ
__device__ inline int seed_first_shift(unsigned maxbits, unsigned thrbits) {
  const unsigned diff = maxbits ^ thrbits;
  const int ptop = diff != 0u ? 31 - __clz((int)diff) : 0;
  return max(32 + ptop - (kSelDigit - 1), 0);
}

// A block covers 256 columns x kNmsRows rows, collects its candidates in LDS and appends them with ONE
// global atomic (same-address atomics cost ~11 ns each on this chip: per-candidate or even per-wavefront
// appends made this kernel take 159 us at 720p).  The order of the keys is irrelevant: they are sorted next.
constexpr int kNmsRows = 8;
// `zero` (may be null): n4 float4 the launch clears on the side -- the plane the corners are splatted into later in the
// sequence (one memset launch less per map).
__global__ void __launch_bounds__(256) k_seed_nms(const float* __restrict__ eig, int rows, int cols, int pitch,
                                                  double quality, unsigned long long* __restrict__ keys,
                                                  unsigned* __restrict__ counters, int cap, float4* __restrict__ zero,
                                                  unsigned n4) {
  __shared__ unsigned long long s_keys[256 * kNmsRows];
  __shared__ unsigned s_count, s_base;
  if (zero) {
    const unsigned nthreads = gridDim.x * gridDim.y * 256u;
    for (unsigned e = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.x; e < n4; e += nthreads)
      zero[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const float maxv = __builtin_bit_cast(float, counters[0]);
  const float thr = (float)((double)maxv * quality);
  for (int r = 0; r < kNmsRows; ++r) {
    const int y = blockIdx.y * kNmsRows + r;
    if (x < 1 || x >= cols - 1 || y < 1 || y >= rows - 1) continue;
    const float v = eig[(size_t)y * pitch + x];
    if (!(v > thr)) continue;
    bool is_max = true;
#pragma unroll
    for (int j = -1; j <= 1; ++j)
#pragma unroll
      for (int i = -1; i <= 1; ++i) {
        const float u = eig[(size_t)(y + j) * pitch + (x + i)];
        is_max = is_max && !(u > v);
      }
    if (!is_max) continue;
    const unsigned slot = atomicAdd(&s_count, 1u);
    // low word = y << 16 | x: the same order as the raster index y * cols + x (x < cols <= 65535), no division to decode
    s_keys[slot] = ((unsigned long long)__builtin_bit_cast(unsigned, v) << 32) | ((unsigned)y << 16) | (unsigned)x;
  }
  __syncthreads();
  const unsigned n = s_count;
  if (n == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(&counters[1], n);
  __syncthreads();
  const unsigned base = s_base;
  for (unsigned i = threadIdx.x; i < n; i += 256)
    if ((int)(base + i) < cap) keys[base + i] = s_keys[i];
}

// One wavefront.  Keys are sorted descending (0 = unused slot).  Accepts a corner when no accepted corner
// lies closer than min_distance; stops at max_features (cv::goodFeaturesToTrack's greedy loop).
__global__ void __launch_bounds__(64) k_seed_select(const unsigned long long* __restrict__ keys, int cap, int cols,
                                                    int min_distance, int max_features, int* __restrict__ kp_xy,
                                                    unsigned* __restrict__ counters) {
  __shared__ int s_x[kSeedMaxFeatures], s_y[kSeedMaxFeatures];
  const int lane = threadIdx.x;
  const int ncand = min((int)counters[1], cap);
  const long long md2 = (long long)min_distance * min_distance;
  int count = 0;
  for (int base = 0; base < ncand && count < max_features; base += 64) {
    const unsigned long long mine = base + lane < ncand ? keys[base + lane] : 0ull;
    const int nk = min(64, ncand - base);
    for (int k = 0; k < nk && count < max_features; ++k) {
      const unsigned idx = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine & 0xffffffffull), k);
      const int y = (int)(idx >> 16), x = (int)(idx & 0xffffu);
      bool bad = false;
      if (min_distance >= 1)
        for (int j = lane; j < count; j += 64) {
          const long long ddx = x - s_x[j], ddy = y - s_y[j];
          bad = bad || (ddx * ddx + ddy * ddy < md2);
        }
      if (!__any(bad)) {
        if (lane == 0) {
          s_x[count] = x;
          s_y[count] = y;
          kp_xy[2 * count] = x;
          kp_xy[2 * count + 1] = y;
        }
        ++count;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the list entry is visible to the next check
      }
    }
  }
  if (lane == 0) {
    counters[2] = (unsigned)count;
    counters[0] = counters[1] = counters[3] = 0u;  // ready for the next map (SeedScratch::counters_clean)
  }
}

// The same greedy selection with the accepted corners kept in a uniform GRID in LDS (cells of min_distance pixels,
// cv::goodFeaturesToTrack's own data structure): a candidate only meets the corners of its 3x3 cell neighbourhood,
// and 64 candidates are examined per step -- every lane checks its own candidate against the grid, then the
// survivors of the batch are taken in sorted order, each one knocking out the later lanes within min_distance.
// Same result as the sequential loop: a candidate is accepted iff no earlier accepted one is closer than
// min_distance.  Two corners at least min_distance apart cannot share more than a cell diagonal: <= 2 per cell;
// four slots are kept and counters[3] flags an overflow (never observed; the caller may assert on it).
// grid = 1 workgroup of 64, dynamic LDS = gx * gy * 4 ints.  Requires min_distance >= 1.
__global__ void __launch_bounds__(64) k_seed_select_grid(const unsigned long long* __restrict__ keys, int cap, int cols,
                                                         int min_distance, int max_features, int gx, int gy,
                                                         int* __restrict__ kp_xy, unsigned* __restrict__ counters) {
  extern __shared__ int s_cell[];  // [gy][gx][4] packed x | y << 16, -1 = free
  const int lane = threadIdx.x;
  for (int e = lane; e < gx * gy * 4; e += 64) s_cell[e] = -1;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  const int ncand = min((int)counters[1], cap);
  if (lane == 0) counters[3] = 0u;  // this map's overflow flag (set below, by this lane)
  const int md2 = min_distance * min_distance;
  // cell of a coordinate: floor(v / min_distance) up to float rounding -- any monotone map whose cells are at
  // least min_distance - 1 wide keeps two points closer than min_distance in adjacent cells, which is all the
  // 3x3 lookup needs (insertion and lookup use the same map)
  const float inv_md = 1.0f / (float)min_distance;
  int count = 0;
  unsigned long long next_key = lane < ncand ? keys[lane] : 0ull;  // one batch ahead: hides the load latency
  for (int base = 0; base < ncand && count < max_features; base += 64) {
    const bool valid = base + lane < ncand;
    const unsigned idx = (unsigned)(next_key & 0xffffffffull);
    next_key = base + 64 + lane < ncand ? keys[base + 64 + lane] : 0ull;
    const int y = (int)(idx >> 16), x = (int)(idx & 0xffffu);
    const int cx = min((int)((float)x * inv_md), gx - 1), cy = min((int)((float)y * inv_md), gy - 1);
    bool bad = !valid;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int ccx = cx + dx, ccy = cy + dy;
        if (ccx < 0 || ccy < 0 || ccx >= gx || ccy >= gy) continue;
        const int* c = s_cell + (ccy * gx + ccx) * 4;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
          const int e = c[sl];
          if (e >= 0) {
            const int ddx = x - (e & 0xffff), ddy = y - (e >> 16);
            bad = bad || (ddx * ddx + ddy * ddy < md2);
          }
        }
      }
    unsigned long long alive = __ballot(!bad);
    while (alive != 0ull && count < max_features) {
      const int w = __ffsll((long long)alive) - 1;
      const int wx = __builtin_amdgcn_readlane(x, w), wy = __builtin_amdgcn_readlane(y, w);
      if (lane == 0) {
        kp_xy[2 * count] = wx;
        kp_xy[2 * count + 1] = wy;
        const int wcx = min((int)((float)wx * inv_md), gx - 1), wcy = min((int)((float)wy * inv_md), gy - 1);
        int* c = s_cell + (wcy * gx + wcx) * 4;
        int sl = 0;
        while (sl < 4 && c[sl] >= 0) ++sl;
        if (sl < 4) c[sl] = wx | (wy << 16);
        else counters[3] = 1u;
      }
      ++count;
      const int ddx = x - wx, ddy = y - wy;
      if (ddx * ddx + ddy * ddy < md2) bad = true;  // includes lane w itself
      alive = __ballot(!bad) & ~((2ull << w) - 1ull);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the grid entries are visible to the next batch
  }
  if (lane == 0) {
    counters[2] = (unsigned)count;
    counters[0] = counters[1] = 0u;  // ready for the next map (SeedScratch::counters_clean)
  }
}

// Sort + selection in ONE workgroup, without sorting what the greedy loop never looks at.  The loop stops after
// max_features corners (200), which it finds among the strongest few hundred candidates of tens of thousands;
// sorting all `cap` slots (hipcub radix sort, ~100 us at 720p) was the seeder's largest item.  Here the candidates
// are consumed in CHUNKS of at most kSelChunk keys in descending order:
//   1. a radix descent (11-bit digits) finds a threshold tau such that between want / 2 and want of the keys not
//      yet consumed are >= tau (keys are unique: value bits | y << 16 | x; want ~ 4 max_features), histograms in
//      LDS (counting the first digit in k_seed_nms with global atomics instead cost that kernel 40 us);
//   2. those keys are gathered into LDS and sorted: counting sort on the 11 bits below the highest bit in which
//      the chunk's bounds differ (descending bin offsets), then every key is ranked inside its
//      bin (bins hold a handful of keys);
//   3. wavefront 0 runs the batch-greedy loop of k_seed_select_grid over the sorted chunk (grid of accepted
//      corners in LDS, kept across chunks) -- the accept loop works on registers only, the accepted lanes then write
//      their corners and grid entries in parallel;
// until max_features corners are accepted or no candidate is left.  The chunks partition the keys by value and are
// visited from the top, so the sequence of candidates the greedy loop sees is exactly the fully sorted order.
// All candidate values lie in (thr, max]: the first digit starts at the highest bit in which the two differ.
// grid = 1 workgroup of 1024; dynamic LDS = seed_select_lds_bytes(gx, gy).
// Descending exclusive offsets of a histogram, by one wavefront: on return hist[b] = number of keys in bins > b.
// Optionally finds the bin in which the running count from the top reaches `want` (-> *bin, *above).
__device__ inline void seed_hist_scan_desc(unsigned* hist, int lane, bool write_offsets, unsigned want, unsigned* bin,
                                           unsigned* above) {
  const int btop = kSelBins - 1 - 32 * lane;  // lane l owns bins btop ... btop - 31 (descending)
  unsigned mine = 0u;
  for (int q = 0; q < 32; ++q) {
    const int b = btop - q;
    mine += hist[b + (b >> 5)];
  }
  unsigned incl = mine;
#pragma unroll
  for (int ofs = 1; ofs < 64; ofs <<= 1) {
    const unsigned o = __shfl_up(incl, ofs, 64);
    if (lane >= ofs) incl += o;
  }
  if (bin) {
    const unsigned long long cross = __ballot(incl >= want);
    const int cl = cross != 0ull ? __ffsll((long long)cross) - 1 : 63;
    if (lane == cl) {
      unsigned acc = incl - mine;
      int b = btop;
      for (int q = 0; q < 32; ++q, --b) {
        const unsigned hb = hist[b + (b >> 5)];
        if (acc + hb >= want || q == 31) break;
        acc += hb;
      }
      *bin = (unsigned)b;
      *above = acc;
    }
  }
  if (write_offsets) {
    unsigned acc = incl - mine;
    for (int q = 0; q < 32; ++q) {
      const int b = btop - q;
      const unsigned hb = hist[b + (b >> 5)];
      hist[b + (b >> 5)] = acc;
      acc += hb;
    }
  }
}
__global__ void __launch_bounds__(1024) k_seed_select_fused(const unsigned long long* __restrict__ keys, int cap,
                                                            int min_distance, int max_features, double quality,
                                                            int gx, int gy, int* __restrict__ kp_xy,
                                                            unsigned* __restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_sel[];
  unsigned long long* s_keys = s_sel;                 // [kSelChunk] the chunk grouped by bin
  unsigned long long* s_tmp = s_sel + kSelChunk;      // [kSelChunk] the chunk as gathered, then sorted
  unsigned* s_hist = (unsigned*)(s_sel + 2 * kSelChunk);  // [kSelHist]
  unsigned* s_misc = s_hist + kSelHist;               // 0 fill, 1 bin, 2 above, 3 count
  int* s_cell = (int*)(s_misc + 8);                   // [gy + 2][gx + 2][4] packed x | y << 16, -1 = free
  const int tid = threadIdx.x, lane = tid & 63;
  const int pgx = gx + 2;  // padded grid: one ring of cells that stay empty
  for (int e = tid; e < pgx * (gy + 2) * 4; e += 1024) s_cell[e] = -1;
  const int ncand = min((int)counters[1], cap);
  const int md2 = min_distance * min_distance;
  const float inv_md = 1.0f / (float)min_distance;
  const unsigned maxbits = counters[0];
  if (tid == 0) counters[3] = 0u;  // this map's overflow flag (set below by lanes of the same wavefront, later in program order)
  const unsigned thrbits = __builtin_bit_cast(unsigned, (float)((double)__builtin_bit_cast(float, maxbits) * quality));
  const int sh_top = seed_first_shift(maxbits, thrbits);
  const int want0 = min(kSelChunk, max(512, 4 * max_features));
  int count = 0, remaining = ncand;
  unsigned long long upper = ~0ull;  // keys >= upper are consumed
  __syncthreads();
#ifdef PM_SEL_TRACE
  unsigned long long tr_t[5] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0};
  int tr_chunks = 0;
#define TR(k) { const unsigned long long t_ = wall_clock64(); tr_t[k] += t_ - tr_t[0]; tr_t[0] = t_; }
#else
#define TR(k)
#endif
  while (remaining > 0 && count < max_features) {
    TR(4);
    // ---- 1. threshold
    unsigned long long tau = 0ull;
    if (remaining > want0) {
      unsigned long long prefix = ((unsigned long long)maxbits << 32) >> (sh_top + kSelDigit) << (sh_top + kSelDigit);
      int want = want0, taken = 0;
      int sh = sh_top, width = kSelDigit;
      for (;;) {
        {
          for (int e = tid; e < kSelHist; e += 1024) s_hist[e] = 0u;
          __syncthreads();
          const unsigned dmask = (1u << width) - 1u;
          const int shp = sh + width;  // <= 63
          for (int i0 = tid; i0 < ncand; i0 += 1024 * kSelUnroll) {
            unsigned long long kk[kSelUnroll];  // independent loads first: one memory latency per kSelUnroll keys
#pragma unroll
            for (int u = 0; u < kSelUnroll; ++u) kk[u] = keys[min(i0 + 1024 * u, ncand - 1)];  // unconditional loads
#pragma unroll
            for (int u = 0; u < kSelUnroll; ++u) kk[u] = i0 + 1024 * u < ncand ? kk[u] : ~0ull;
#pragma unroll
            for (int u = 0; u < kSelUnroll; ++u) {
              const unsigned long long k = kk[u];
              if (k < upper && (k >> shp) == (prefix >> shp)) {
                const unsigned b = (unsigned)(k >> sh) & dmask;
                atomicAdd(&s_hist[b + (b >> 5)], 1u);
              }
            }
          }
        }
        __syncthreads();
        if (tid < 64) seed_hist_scan_desc(s_hist, lane, false, (unsigned)want, &s_misc[1], &s_misc[2]);
        __syncthreads();
        const unsigned b = s_misc[1], acc = s_misc[2];
        __syncthreads();
        if (sh == 0) {  // single keys: the crossing key is the want-th itself
          tau = prefix | b;
          break;
        }
        if (taken + (int)acc >= want0 / 2) {  // enough above the crossing bin: leave that bin to the next chunk
          tau = prefix + ((unsigned long long)(b + 1u) << sh);
          break;
        }
        prefix |= (unsigned long long)b << sh;
        taken += (int)acc;
        want -= (int)acc;
        const int nsh = max(sh - kSelDigit, 0);
        width = sh - nsh;
        sh = nsh;
      }
    }
    TR(1);
    // ---- 2. gather [tau, upper) ...
    if (tid == 0) s_misc[0] = 0u;
    for (int e = tid; e < kSelHist; e += 1024) s_hist[e] = 0u;
    __syncthreads();
    for (int i0 = tid; i0 < ncand + tid; i0 += 1024 * kSelUnroll) {  // whole wavefronts iterate together (ballot)
      unsigned long long kk[kSelUnroll];
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u) kk[u] = keys[min(i0 + 1024 * u, ncand - 1)];  // unconditional loads
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u) kk[u] = i0 + 1024 * u < ncand ? kk[u] : ~0ull;
#pragma unroll
      for (int u = 0; u < kSelUnroll; ++u) {
        const unsigned long long k = kk[u];
        const bool in = k >= tau && k < upper;  // the filler ~0 is never below `upper`
        const unsigned long long m = __ballot(in);
        if (m != 0ull) {
          const int leader = __ffsll((long long)m) - 1;
          unsigned base = 0u;
          if (lane == leader) base = atomicAdd(&s_misc[0], (unsigned)__popcll(m));
          base = (unsigned)__shfl((int)base, leader, 64);
          if (in) {
            const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (pos < (unsigned)kSelChunk) s_tmp[pos] = k;
          }
        }
      }
    }
    __syncthreads();
    const int n = min((int)s_misc[0], kSelChunk);
    if (n == 0) break;  // cannot happen (every threshold keeps at least one key); uniform
    // ... and sort it: bins of the 11 bits below the highest bit in which the bounds of the chunk differ (every key
    // lies in [max(tau, threshold), min(upper - 1, maximum)], so the bits above that one are common to all)
    const unsigned long long klo = max(tau, (unsigned long long)thrbits << 32);
    const unsigned long long khi = min(upper - 1ull, ((unsigned long long)maxbits << 32) | 0xffffffffull);
    const unsigned long long kdiff = klo ^ khi;
    const int csh = kdiff != 0ull ? max(63 - __clzll((long long)kdiff) - (kSelDigit - 1), 0) : 0;
    unsigned long long mykey[2];
    unsigned mybin[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + 1024 * u;
      mykey[u] = e < n ? s_tmp[e] : 0ull;
      mybin[u] = (unsigned)(mykey[u] >> csh) & (unsigned)(kSelBins - 1);
      if (e < n) atomicAdd(&s_hist[mybin[u] + (mybin[u] >> 5)], 1u);
    }
    __syncthreads();
    if (tid < 64) seed_hist_scan_desc(s_hist, lane, true, 0u, nullptr, nullptr);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)  // slot inside the bin: any order; s_hist[b] ends as the END of bin b
      if (tid + 1024 * u < n) s_keys[atomicAdd(&s_hist[mybin[u] + (mybin[u] >> 5)], 1u)] = mykey[u];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {  // rank inside the bin (the bins above it end where it starts)
      const int e = tid + 1024 * u;
      if (e < n) {
        const unsigned long long k = s_keys[e];
        const unsigned bn = (unsigned)(k >> csh) & (unsigned)(kSelBins - 1);
        const unsigned end = s_hist[bn + (bn >> 5)];
        const unsigned start = bn == (unsigned)(kSelBins - 1) ? 0u : s_hist[bn + 1 + ((bn + 1) >> 5)];
        unsigned r = start;
        for (unsigned q = start; q < end; ++q) r += s_keys[q] > k ? 1u : 0u;
        mykey[u] = k;
        mybin[u] = r;
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (tid + 1024 * u < n) s_tmp[mybin[u]] = mykey[u];
    __syncthreads();
    TR(2);
    // ---- 3. greedy selection over the sorted chunk (wavefront 0; see k_seed_select_grid).  The grid carries a
    // border of cells that stay empty, so the 3x3 lookup needs no bounds tests: nine 16-byte reads, no branches.
    if (tid < 64) {
      const unsigned long long below = (1ull << lane) - 1ull;
      for (int base = 0; base < n && count < max_features; base += 64) {
        const bool valid = base + lane < n;
        const unsigned idx = valid ? (unsigned)(s_tmp[base + lane] & 0xffffffffull) : 0u;
        const int y = (int)(idx >> 16), x = (int)(idx & 0xffffu);
        const int cx = min((int)((float)x * inv_md), gx - 1), cy = min((int)((float)y * inv_md), gy - 1);
        const int4* c3 = (const int4*)s_cell + (cy * pgx + cx);  // cell (cx - 1, cy - 1) of the padded grid
        int4 cell[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) cell[q] = c3[(q / 3) * pgx + (q % 3)];
        bool bad = !valid;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int e4[4] = {cell[q].x, cell[q].y, cell[q].z, cell[q].w};
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) {
            const int ddx = x - (e4[sl] & 0xffff), ddy = y - (e4[sl] >> 16);
            bad = bad | ((e4[sl] >= 0) & (__mul24(ddx, ddx) + __mul24(ddy, ddy) < md2));
          }
        }
        // the survivors in sorted order, each one knocking out the later lanes within min_distance: wavefront
        // masks only (the lanes below the winner have already left `alive`, the winner is within distance 0)
        unsigned long long alive = __ballot(!bad), accepted = 0ull;
        int room = max_features - count;
        while (alive != 0ull && room > 0) {
          const int w = __ffsll((long long)alive) - 1;
          const int wx = __builtin_amdgcn_readlane(x, w), wy = __builtin_amdgcn_readlane(y, w);
          accepted |= 1ull << w;
          --room;
          const int ddx = x - wx, ddy = y - wy;
          alive &= ~__ballot(__mul24(ddx, ddx) + __mul24(ddy, ddy) < md2);
        }
        if ((accepted >> lane) & 1ull) {
          const int slot = count + __popcll(accepted & below);
          kp_xy[2 * slot] = x;
          kp_xy[2 * slot + 1] = y;
          int* c = s_cell + ((cy + 1) * pgx + cx + 1) * 4;
          int sl = 0;
          while (sl < 4 && atomicCAS(&c[sl], -1, x | (y << 16)) != -1) ++sl;
          if (sl == 4) counters[3] = 1u;
        }
        count += __popcll(accepted);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the grid entries are visible to the next batch
      }
      if (tid == 0) s_misc[3] = (unsigned)count;
    }
    __syncthreads();
    count = (int)s_misc[3];
    remaining -= n;
    upper = tau;
    __syncthreads();
    TR(3);
#ifdef PM_SEL_TRACE
    ++tr_chunks;
#endif
  }
#ifdef PM_SEL_TRACE
  if (tid == 0) printf("select: ncand %d accepted %d chunks %d; ticks (10 ns) threshold %llu gather+sort %llu greedy %llu\n", ncand, count, tr_chunks, tr_t[1], tr_t[2], tr_t[3]);
#endif
  __syncthreads();  // every thread has read counters[0] and [1]
  if (tid == 0) {
    counters[2] = (unsigned)count;
    counters[0] = counters[1] = 0u;  // ready for the next map (SeedScratch::counters_clean)
  }
#undef TR
}

// ---- cv::cornerSubPix on the device (oracle: pm_seed_oracle.c::pmo_corner_subpix, which cites the OpenCV source it
// restates).  One LANE per point, strictly sequential: the 2 x 2 normal equations are accumulated in double in raster
// order, so that the result is the oracle's bit for bit.  `buf` = the lane's neighbourhood, element e at buf[e * bs]
// (corner-major interleave: the lanes of a wavefront read and write neighbouring words).
__device__ inline int sp_floor(float v) {
  const int i = (int)v;
  return i - (v < (float)i ? 1 : 0);
}
// getRectSubPix 8u -> 32f (samplers.cpp: getRectSubPix_8u32f inside the image, getRectSubPix_Cn_ + adjustRect at the border)
__device__ inline void sp_get_rect_8u32f(const uint8_t* src, int rows, int cols, int pitch, int ww, int wh, float cx,
                                         float cy, float* dst, int bs) {
  cx -= (float)(ww - 1) * 0.5f;
  cy -= (float)(wh - 1) * 0.5f;
  const int ipx = sp_floor(cx), ipy = sp_floor(cy);
  if (0 <= ipx && ipx + ww < cols && 0 <= ipy && ipy + wh < rows && ww > 0 && wh > 0) {
    float a = cx - (float)ipx;
    const float b = cy - (float)ipy;
    a = a > 0.0001f ? a : 0.0001f;
    const float a12 = a * (1.f - b), a22 = a * b, b1 = 1.f - b, b2 = b;
    const double s = (1. - (double)a) / (double)a;
    const uint8_t* r = src + (size_t)ipy * pitch + ipx;
    for (int i = 0; i < wh; ++i, r += pitch) {
      const float t0 = b1 * (float)r[0], t1 = b2 * (float)r[pitch];
      float prev = (1.f - a) * (t0 + t1);
      float* d = dst + (size_t)i * ww * bs;
      for (int j = 0; j < ww; ++j) {
        const float u0 = a12 * (float)r[j + 1], u1 = a22 * (float)r[j + 1 + pitch];
        const float t = u0 + u1;
        d[(size_t)j * bs] = prev + t;
        prev = (float)((double)t * s);
      }
    }
    return;
  }
  const float a = cx - (float)ipx, b = cy - (float)ipy;
  const float ia = 1.f - a, ib = 1.f - b;
  const float a11 = ia * ib, a12 = a * ib, a21 = ia * b, a22 = a * b, b1 = ib, b2 = b;
  int rx, ry, rw, rh;
  long off = 0;
  if (ipx >= 0) { off += ipx; rx = 0; } else { rx = -ipx; if (rx > ww) rx = ww; }
  if (ipx < cols - ww) rw = ww; else { rw = cols - ipx - 1; if (rw < 0) { off += rw; rw = 0; } }
  if (ipy >= 0) { off += (long)ipy * pitch; ry = 0; } else ry = -ipy;
  if (ipy < rows - wh) rh = wh; else { rh = rows - ipy - 1; if (rh < 0) { off += (long)rh * pitch; rh = 0; } }
  const uint8_t* r = src + (off - rx);
  for (int i = 0; i < wh; ++i) {
    const uint8_t* r2 = r + pitch;
    if (i < ry || i >= rh) r2 -= pitch;
    float* d = dst + (size_t)i * ww * bs;
    float s0 = (float)r[rx] * b1 + (float)r2[rx] * b2;
    for (int j = 0; j < rx; ++j) d[(size_t)j * bs] = s0;
    s0 = (float)r[rw] * b1 + (float)r2[rw] * b2;
    for (int j = rw; j < ww; ++j) d[(size_t)j * bs] = s0;
    for (int j = rx; j < rw; ++j) {
      float v = (float)r[j] * a11;
      v = v + (float)r[j + 1] * a12;
      v = v + (float)r2[j] * a21;
      v = v + (float)r2[j + 1] * a22;
      d[(size_t)j * bs] = v;
    }
    if (i < rh) r = r2;
  }
}
__device__ inline void sp_corner_subpix(const uint8_t* img, int rows, int cols, int pitch, float& x, float& y, int win,
                                        const float* __restrict__ mask, int max_iters, double eps, float* buf, int bs) {
  const int ww = 2 * win + 1, bw = ww + 2;
  max_iters = max_iters < 1 ? 1 : (max_iters > 100 ? 100 : max_iters);
  eps = eps > 0. ? eps : 0.;
  eps *= eps;
  const float tx = x, ty = y;
  float ix = tx, iy = ty;
  int iter = 0;
  double err = 0.;
  do {
    double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
    sp_get_rect_8u32f(img, rows, cols, pitch, bw, bw, ix, iy, buf, bs);
    for (int i = 0, k = 0; i < ww; ++i) {
      const float* sp = buf + (size_t)((i + 1) * bw + 1) * bs;
      const double py = i - win;
      for (int j = 0; j < ww; ++j, ++k) {
        const double m = mask[k];
        const float fgx = sp[(size_t)(j + 1) * bs] - sp[(long)(j - 1) * bs];
        const float fgy = sp[(size_t)(j + bw) * bs] - sp[(long)(j - bw) * bs];
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
  x = ix;
  y = iy;
}
// FeatureDetector::Detect's refinement (feature_detector.cpp:110-120): corner i of the selection -> its sub-pixel
// position (kp_f) and the rounded position the matcher and the scatter use (std::round, back into kp_xy)
__global__ void __launch_bounds__(64) k_seed_subpix_corners(const uint8_t* __restrict__ img, int rows, int cols, int pitch,
                                                            int* __restrict__ kp_xy, float* __restrict__ kp_f,
                                                            const unsigned* __restrict__ counters, SeedParams sp,
                                                            const float* __restrict__ mask, float* __restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int)counters[2]) return;
  float x = (float)kp_xy[2 * i], y = (float)kp_xy[2 * i + 1];
  sp_corner_subpix(img, rows, cols, pitch, x, y, sp.subpix_winsize, mask, sp.subpix_maxiters, (double)sp.subpix_epsilon,
                   buf + i, kSeedMaxFeatures);
  kp_f[2 * i] = x;
  kp_f[2 * i + 1] = y;
  kp_xy[2 * i] = (int)roundf(x);
  kp_xy[2 * i + 1] = (int)roundf(y);
}

// cv::cornerSubPix as a stage of its own (pm_corner_subpix): n points, xs / ys in and out
__global__ void __launch_bounds__(64) k_seed_subpix_points(const uint8_t* __restrict__ img, int rows, int cols, int pitch,
                                                           float* __restrict__ xs, float* __restrict__ ys, int n, int win,
                                                           int max_iters, double eps, const float* __restrict__ mask,
                                                           float* __restrict__ buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = xs[i], y = ys[i];
  sp_corner_subpix(img, rows, cols, pitch, x, y, win, mask, max_iters, eps, buf + i, kSeedMaxFeatures);
  xs[i] = x;
  ys[i] = y;
}

// One workgroup per accepted corner (StereoMatcher::MatchRectified).  Writes the corner's disparity (>= 0) or -1.
__global__ void __launch_bounds__(256) k_seed_match(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                    int rows, int cols, int pitch, const int* __restrict__ kp_xy,
                                                    const float* __restrict__ kp_f,
                                                    const unsigned* __restrict__ counters, SeedParams sp,
                                                    float* __restrict__ kp_d, const float* __restrict__ sp_mask,
                                                    float* __restrict__ sp_buf) {
  const int kp = blockIdx.x;
  if (kp >= (int)counters[2]) return;
  if (threadIdx.x == 0) kp_d[kp] = -1.f;  // every early exit below means "no match"; thread 0 also writes the result
  // the rounded corner (an integer corner is its own rounding; sub-pixel corners were rounded by k_seed_subpix_corners)
  const int rx = kp_xy[2 * kp], ry = kp_xy[2 * kp + 1];
  const float kx = kp_f ? kp_f[2 * kp] : (float)rx;  // left_keypoint.x
  const int tc = sp.templ_cols, tr = sp.templ_rows, md = sp.max_disp;
  const int stripe_rows = tr + 2;
  int ty = ry - (tr - 1) / 2;
  if (ty < 0 || ty + tr >= rows) return;
  int offset_x = 0;
  int tx = rx - (tc - 1) / 2;
  if (tx < 0) {
    offset_x = tx;
    tx = 0;
  }
  if (tx + tc >= cols) {
    if (offset_x != 0) return;
    offset_x = (tx + tc) - (cols - 1);
    tx -= offset_x;
  }
  const int sy = ry - (stripe_rows - 1) / 2;
  if (sy < 0 || sy + stripe_rows >= rows) return;
  int sx = rx + (tc - 1) / 2 - md;
  if (sx + md > cols - 1) sx -= (sx + md) - (cols - 1);
  if (sx < 0) sx = 0;
  if (sx + md > cols || tx < 0) return;
  const int rw = md - tc + 1, rh = stripe_rows - tr + 1;

  __shared__ unsigned long long s_best[4];
  // Template and search stripe staged in LDS once, rows padded to whole dwords (template padding = 0).
  // num = sum (t - q)^2 = sum t^2 - 2 sum t*q + sum q^2, every sum an exact integer: four taps per v_dot4_u32_u8.
  extern __shared__ unsigned s_seed_w[];
  const int tcw = (tc + 3) >> 2;            // template row, dwords
  const int mdw = ((md + 3) >> 2) + 1;      // stripe row, dwords (+1: the last window may read one dword past)
  unsigned* s_t = s_seed_w;                 // [tr][tcw]
  unsigned* s_i = s_seed_w + tr * tcw;      // [stripe_rows][mdw]
  for (int e = threadIdx.x; e < tr * tcw + stripe_rows * mdw; e += blockDim.x) s_seed_w[e] = 0u;
  __syncthreads();
  for (int e = threadIdx.x; e < tr * tc; e += blockDim.x) {
    const int j = e / tc, i = e - j * tc;
    ((uint8_t*)(s_t + j * tcw))[i] = left[(size_t)(ty + j) * pitch + tx + i];
  }
  for (int e = threadIdx.x; e < stripe_rows * md; e += blockDim.x) {
    const int j = e / md, i = e - j * md;
    ((uint8_t*)(s_i + j * mdw))[i] = right[(size_t)(sy + j) * pitch + sx + i];
  }
  __syncthreads();
  unsigned t2 = 0;  // the template's own sum of squares does not depend on the position
  for (int e = 0; e < tr * tcw; ++e) t2 = __builtin_amdgcn_udot4(s_t[e], s_t[e], t2, false);
  const int rem = tc - 4 * (tcw - 1);  // taps in the last dword of a row (1..4)
  const unsigned last_mask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
  unsigned long long best = ~0ull;
  for (int pos = threadIdx.x; pos < rw * rh; pos += blockDim.x) {
    const int v = pos / rw, u = pos - v * rw;
    unsigned tq = 0, i2 = 0;
    const unsigned sh = (unsigned)u & 3u;
    for (int j = 0; j < tr; ++j) {
      const unsigned* T = s_t + j * tcw;
      const unsigned* I = s_i + (v + j) * mdw + (u >> 2);
      unsigned w0 = I[0];
      for (int g = 0; g < tcw; ++g) {
        const unsigned w1 = I[g + 1];
        unsigned q = __builtin_amdgcn_alignbyte(w1, w0, sh);
        if (g == tcw - 1) q &= last_mask;  // stripe bytes beyond the template's width do not belong to the window
        tq = __builtin_amdgcn_udot4(T[g], q, tq, false);
        i2 = __builtin_amdgcn_udot4(q, q, i2, false);
        w0 = w1;
      }
    }
    const unsigned num = t2 + i2 - 2u * tq;  // = sum (t - q)^2 >= 0
    const double den = sqrt((double)t2 * (double)i2);
    const float r = den > 0.0 ? (float)((double)num / den) : 1.f;
    // first minimum in row-major order = minimum of (value bits, position) as one 64-bit key (r >= 0)
    const unsigned long long key = ((unsigned long long)__builtin_bit_cast(unsigned, r) << 32) | (unsigned)pos;
    best = key < best ? key : best;
  }
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) {
    const unsigned long long o = __shfl_xor(best, ofs, 64);
    best = o < best ? o : best;
  }
  if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) best = s_best[w] < best ? s_best[w] : best;
    const float minv = __builtin_bit_cast(float, (unsigned)(best >> 32));
    const int pos = (int)(best & 0xffffffffull);
    const int bx = pos % rw, by = pos / rw;
    const int mx = bx + sx + (tc - 1) / 2 + offset_x;
    float mpx = (float)mx;
    if (sp.subpixel_refinement) {  // stereo_matcher.cpp:94-103: cornerSubPix on the right image, 10 x 10, 40 steps, 0.001
      float mpy = (float)(by + sy + (tr - 1) / 2);
      sp_corner_subpix(right, rows, cols, pitch, mpx, mpy, kSubpixMatchWin, sp_mask + kSubpixMaskStride, 40, 0.001,
                       sp_buf + kp, kSeedMaxFeatures);
    }
    if ((double)minv < sp.max_matching_cost && kx >= mpx) kp_d[kp] = kx - mpx;
  }
}

// Seed map from the matched corners = cv::dilate (MORPH_RECT (2k+1)^2, anchor (k, k), samples outside the image
// ignored) of the map that holds d at round(kp) and 0 elsewhere (patchmatch.cpp:61-78, patchmatch_gpu.cu:422-439):
// every pixel takes the largest disparity among the corners whose rectangle covers it.  One workgroup per corner
// writes its rectangle with atomicMax on the float bit patterns (disparities are >= 0, so unsigned order = float
// order; a maximum does not depend on the order of the updates: deterministic) into a zeroed map.  `inv_scale`
// (1, or 2^-f for Initialize, patchmatch.cpp:81) is applied to the corner value first: scaling by a power of two
// is exact and monotone, so it commutes with the maximum.
// The output map is row-major with `out_pitch` elements per row, or -- out_pitch < 0 -- one of the engine's state
// planes (four rows interleaved, pm_device.hpp::state_at, pitch -out_pitch).
__host__ __device__ __forceinline__ size_t seed_out_at(int x, int y, int out_pitch) {
  return out_pitch < 0 ? state_at(x, y, -out_pitch) : (size_t)y * (size_t)out_pitch + (size_t)x;
}
// `dedup` (sub-pixel corners only): two corners may have been refined onto the same pixel; the scatter
// `disps.at(round(kp.y), round(kp.x)) = d` (patchmatch_gpu.cu:426-432) is sequential, so the LAST corner with a valid
// match (d >= 0, even 0) owns the pixel -- a corner with such a successor does not count.
__global__ void __launch_bounds__(256) k_seed_splat(const int* __restrict__ kp_xy, const float* __restrict__ kp_d,
                                                    const unsigned* __restrict__ counters, int rows, int cols, int k,
                                                    float inv_scale, float* __restrict__ out, int out_pitch, int dedup) {
  const int kp = blockIdx.x;
  const int count = (int)counters[2];
  if (kp >= count) return;
  const float d = kp_d[kp];
  if (!(d > 0.f)) return;  // no match (-1) or disparity 0: nothing to raise above the zero background
  const unsigned bits = __builtin_bit_cast(unsigned, d * inv_scale);
  const int kx = kp_xy[2 * kp], ky = kp_xy[2 * kp + 1];
  if (kx < 0 || kx >= cols || ky < 0 || ky >= rows) return;  // a sub-pixel corner rounded out of the image: no pixel to set
  if (dedup) {
    int later = 0;
    for (int j = kp + 1 + (int)threadIdx.x; j < count; j += (int)blockDim.x)
      later |= (kp_xy[2 * j] == kx && kp_xy[2 * j + 1] == ky && kp_d[j] >= 0.f) ? 1 : 0;
    if (__syncthreads_or(later)) return;
  }
  const int x0 = max(kx - k, 0), x1 = min(kx + k, cols - 1), y0 = max(ky - k, 0), y1 = min(ky + k, rows - 1);
  const int w = x1 - x0 + 1, n = w * (y1 - y0 + 1);
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const int yy = e / w, xx = e - yy * w;
    atomicMax((unsigned*)out + seed_out_at(x0 + xx, y0 + yy, out_pitch), bits);
  }
}
// Initialize's down-sampled map: cv::resize(INTER_NEAREST) of the dilated full-size map (patchmatch.cpp:79):
// dst(y, x) = src(min(floor(y * rows / out_rows), rows - 1), min(floor(x * cols / out_cols), cols - 1)).
__global__ void __launch_bounds__(256) k_seed_resize_nearest(const float* __restrict__ src, int rows, int cols,
                                                             int src_pitch, float* __restrict__ dst, int out_rows,
                                                             int out_cols, int dst_pitch) {
  const int xo = blockIdx.x * blockDim.x + threadIdx.x, yo = blockIdx.y;
  if (xo >= out_cols) return;
  const double ifx = 1.0 / ((double)out_cols / (double)cols), ify = 1.0 / ((double)out_rows / (double)rows);
  const int sx = min((int)floor((double)xo * ifx), cols - 1);
  const int sy = min((int)floor((double)yo * ify), rows - 1);
  dst[seed_out_at(xo, yo, dst_pitch)] = src[(size_t)sy * src_pitch + sx];
}

// PM_SEED_FUSED=0 keeps the radix sort + separate selection (A/B and the fallback's own test); read once.
inline bool seed_fused_enabled() {
  static const bool on = [] {
    const char* v = pm::tune_env("PM_SEED_FUSED");
    return !(v && v[0] == '0');
  }();
  return on;
}

// SparseInit / Initialize for one pair of pitched u8 planes: corners of `left` matched into `right`, dilated with
// half-width k, written as an out_rows x out_cols map (row pitch out_pitch elements) scaled by inv_scale.
//   PatchmatchGpu::SparseInit(iml, imr, f)   k = 2^f + 1,     out = image size, inv_scale = 1   (patchmatch_gpu.cu:436)
//   Patchmatch::Initialize(iml, imr, f)      k = 2^(f-1) + 1, out = size / f,   inv_scale = 2^-f (patchmatch.cpp:75-81)
// Enqueue-only.  `stages`: which of the kSeedStages parts of the sequence to enqueue (bit i = part i; the parts of one
// map must be enqueued in order on one stream) -- a caller with two views on two streams enqueues part i of both before
// part i + 1 of either, so neither stream waits for the host to get through the other's whole sequence.
static hipError_t seed_map(const SeedScratch& sc, const SeedParams& sp, const uint8_t* left, const uint8_t* right,
                           int rows, int cols, int pitch, int k, int out_rows, int out_cols, float inv_scale, float* out,
                           int out_pitch, hipStream_t stream, unsigned stages) {
  const dim3 grid((unsigned)((cols + 255) / 256), (unsigned)rows), block(256);
  hipError_t e;
  if (stages & 1u) {
    if (!sc.counters_clean && (e = hipMemsetAsync(sc.counters, 0, kSeedCounters * sizeof(unsigned), stream)) != hipSuccess)
      return e;
    sc.counters_clean = false;  // until this map's selection kernel is behind them
  }
  // the plane the corners are splatted into: the output itself, or the (idle by then) response plane when a resize
  // follows.  An engine state plane (16-byte multiples) is cleared by the suppression kernel on the side.
  const bool resized = out_rows != rows || out_cols != cols;
  float* full = resized ? sc.eig : out;
  const int full_pitch = resized ? pitch : out_pitch;
  const size_t full_elems = full_pitch < 0 ? (size_t)((rows + 3) & ~3) * (size_t)(-full_pitch) : (size_t)rows * full_pitch;
  const bool zero_in_nms = !resized && full_pitch < 0 && (full_elems % 4) == 0 && ((uintptr_t)full % 16) == 0 &&
                           full_elems / 4 < 0xffffffffull;
  const int maxf = sp.max_features < kSeedMaxFeatures ? sp.max_features : kSeedMaxFeatures;
  const int md = sp.min_distance;
  const int gx = md >= 1 ? (cols + md - 1) / md : 0, gy = md >= 1 ? (rows + md - 1) / md : 0;
  // x | y << 16 packing, and squared distances (any two candidates of a batch) that fit an int
  const bool packed_ok = md >= 1 && cols <= 32768 && rows <= 32768;
  const bool fused = packed_ok && seed_select_lds_bytes(gx, gy) <= 150 * 1024 && seed_fused_enabled();
  // the sort treats 0 as "unused slot"; the fused selection only reads the first counters[1] keys
  if ((stages & 1u) && !fused &&
      (e = hipMemsetAsync(sc.keys, 0, sizeof(unsigned long long) * sc.cap, stream)) != hipSuccess)
    return e;
  if (stages & 1u) {
    const int hb = sp.block_size / 2;
    const size_t lds = sizeof(unsigned) * (size_t)(kEigTileW + 2 * hb) * (kEigTileH + 2 * hb);
    const dim3 tiles((unsigned)((cols + kEigTileW - 1) / kEigTileW), (unsigned)((rows + kEigTileH - 1) / kEigTileH));
    auto kern = sp.block_size == 3   ? k_seed_response<3>
                : sp.block_size == 5 ? k_seed_response<5>
                : sp.block_size == 7 ? k_seed_response<7>
                                     : k_seed_response<0>;
    hipLaunchKernelGGL(kern, tiles, block, lds, stream, left, rows, cols, pitch, sp.block_size, sp.use_harris,
                       sp.harris_k, sc.eig, sc.counters);
  }
  if (stages & 2u)
    hipLaunchKernelGGL(k_seed_nms, dim3(grid.x, (unsigned)((rows + kNmsRows - 1) / kNmsRows)), block, 0, stream, sc.eig,
                       rows, cols, pitch, sp.quality_level, sc.keys, sc.counters, sc.cap,
                       zero_in_nms ? (float4*)full : (float4*)nullptr, (unsigned)(full_elems / 4));
  if ((stages & 4u) && fused) {
    const size_t lds = seed_select_lds_bytes(gx, gy);
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k_seed_select_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_seed_select_fused, dim3(1), dim3(1024), lds, stream, (const unsigned long long*)sc.keys,
                       sc.cap, md, maxf, sp.quality_level, gx, gy, sc.kp_xy, sc.counters);
  } else if (stages & 4u) {
    size_t tmp_bytes = sc.sort_tmp_bytes;
    if ((e = hipcub::DeviceRadixSort::SortKeysDescending(sc.sort_tmp, tmp_bytes, sc.keys, sc.keys_sorted, sc.cap, 0, 64,
                                                         stream)) != hipSuccess)
      return e;
    const size_t grid_bytes = (size_t)gx * gy * 4 * sizeof(int);
    if (packed_ok && grid_bytes <= 150 * 1024) {  // LDS capacity
      if (grid_bytes > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_seed_select_grid, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)grid_bytes);
      hipLaunchKernelGGL(k_seed_select_grid, dim3(1), dim3(64), grid_bytes, stream, sc.keys_sorted, sc.cap, cols, md,
                         maxf, gx, gy, sc.kp_xy, sc.counters);
    } else {
      hipLaunchKernelGGL(k_seed_select, dim3(1), dim3(64), 0, stream, sc.keys_sorted, sc.cap, cols, sp.min_distance,
                         maxf, sc.kp_xy, sc.counters);
    }
  }
  if (stages & 4u) sc.counters_clean = hipPeekAtLastError() == hipSuccess;
  if (stages & 8u) {
    const size_t px_bytes = 4 * ((size_t)sp.templ_rows * ((sp.templ_cols + 3) / 4) +
                                 (size_t)(sp.templ_rows + 2) * ((sp.max_disp + 3) / 4 + 1));
    if (sp.subpixel_corners)
      hipLaunchKernelGGL(k_seed_subpix_corners, dim3((unsigned)((maxf + 63) / 64 > 0 ? (maxf + 63) / 64 : 1)), dim3(64), 0,
                         stream, left, rows, cols, pitch, sc.kp_xy, sc.kp_f, (const unsigned*)sc.counters, sp,
                         (const float*)sc.sp_mask, sc.sp_buf);
    hipLaunchKernelGGL(k_seed_match, dim3((unsigned)(maxf > 0 ? maxf : 1)), dim3(256), px_bytes, stream, left, right,
                       rows, cols, pitch, sc.kp_xy, sp.subpixel_corners ? (const float*)sc.kp_f : (const float*)nullptr,
                       sc.counters, sp, sc.kp_d, (const float*)sc.sp_mask, sc.sp_buf);
  }
  if (!(stages & 16u)) return hipGetLastError();
  if (!zero_in_nms && (e = hipMemsetAsync(full, 0, sizeof(float) * full_elems, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(k_seed_splat, dim3((unsigned)(maxf > 0 ? maxf : 1)), block, 0, stream, (const int*)sc.kp_xy,
                     (const float*)sc.kp_d, (const unsigned*)sc.counters, rows, cols, k, inv_scale, full, full_pitch,
                     sp.subpixel_corners);
  if (resized)
    hipLaunchKernelGGL(k_seed_resize_nearest, dim3((unsigned)((out_cols + 255) / 256), (unsigned)out_rows), block, 0,
                       stream, (const float*)full, rows, cols, full_pitch, out, out_rows, out_cols, out_pitch);
  return hipGetLastError();
}
// pm_corner_subpix: the detector's window of `sp` (sc.sp_mask) on n <= kSeedMaxFeatures device points
hipError_t seed_corner_subpix(const SeedScratch& sc, const SeedParams& sp, const uint8_t* img, int rows, int cols, int pitch,
                              float* d_xs, float* d_ys, int n, hipStream_t stream) {
  if (n < 1) return hipSuccess;
  hipLaunchKernelGGL(k_seed_subpix_points, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, img, rows, cols, pitch, d_xs,
                     d_ys, n, sp.subpix_winsize, sp.subpix_maxiters, (double)sp.subpix_epsilon, (const float*)sc.sp_mask,
                     sc.sp_buf);
  return hipGetLastError();
}
hipError_t seed_sparse_init(const SeedScratch& sc, const SeedParams& sp, const uint8_t* left,
                                   const uint8_t* right, int rows, int cols, int pitch, int dilate_factor, float* out,
                                   int out_pitch, hipStream_t stream, unsigned stages) {
  return seed_map(sc, sp, left, right, rows, cols, pitch, (1 << dilate_factor) + 1, rows, cols, 1.0f, out, out_pitch,
                  stream, stages);
}
// downsample_factor >= 1
hipError_t seed_initialize(const SeedScratch& sc, const SeedParams& sp, const uint8_t* left,
                                  const uint8_t* right, int rows, int cols, int pitch, int downsample_factor, float* out,
                                  int out_pitch, hipStream_t stream, unsigned stages) {
  const float inv = 1.0f / (float)(1 << downsample_factor);
  return seed_map(sc, sp, left, right, rows, cols, pitch, (1 << (downsample_factor - 1)) + 1, rows / downsample_factor,
                  cols / downsample_factor, inv, out, out_pitch, stream, stages);
}

}  // namespace pm
