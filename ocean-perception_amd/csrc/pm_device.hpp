// pm_device.hpp -- device-side arithmetic of the PatchMatch stereo engine (gfx950).
//
// Everything here is compiled with -ffp-contract=off: each float operation is one IEEE-754
// binary32 rounding, in the same order as the reference code it reproduces, so disparities
// are bit-identical to the CPU path.  `file:line` citations are relative to the reference tree.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pm {

// BORDER_REFLECT_101 index (one reflection: |overshoot| < len)
__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  if (p < 0) p = -p;
  if (p >= len) p = 2 * (len - 1) - p;
  return p;
}

// The transposed planes carry kTransPad extra rows (= image columns cols .. cols + kTransPad - 1) that
// replicate the last column: a window that leaves the image on the right reads them instead of clamping
// its column index (cv::getRectSubPix replicates the border), which keeps the column sweep's target row
// offsets affine in the window column (pm_run2.hpp).
constexpr int kTransPad = 16;

struct Interior {
  int x_lo, x_hi, y_lo, y_hi;  // inclusive bounds of the pixels visited by the sweeps
};

// Device memory of one call: B pairs, each with four u8 images (L, R, mirrored L, mirrored R),
// their Sobel magnitudes as f32 and as saturated u8, and per view a disparity plane and the
// cost of that disparity.  All planes share one pitch (elements per row, multiple of 64) so that
// row y of every plane starts on a 256-byte boundary for f32 and a 64-byte one for u8.
struct PlaneSet {
  uint8_t* img8;       // [B][4][rows][pitch]   0=L 1=R 2=Lm 3=Rm
  float* g32;          // [B][4][rows][pitch]
  uint8_t* g8;         // [B][4][rows][pitch]   saturate_cast<uchar>(g32)
  // Transposed copies ([cols][pitch_t], element (x, y) at x * pitch_t + y): a column sweep gives one
  // lane to each image ROW, so with these the 64 lanes of a load read 64 consecutive bytes/words.
  uint8_t* timg8;      // [B][4][cols][pitch_t]
  float* tg32;         // [B][4][cols][pitch_t]
  uint8_t* tg8;        // [B][4][cols][pitch_t]
  int pitch_t;         // align_up(rows, 64)
  size_t plane_t;      // cols * pitch_t
  uint16_t* pk16;      // [B][4][rows][pitch]   img8 | g8 << 8: what a window reads of its own image
  uint16_t* tpk16;     // [B][4][cols][pitch_t] transposed copy of pk16
  // Line-TRIPLE planes of the run engine (pm_run3.hpp): element (l, e) holds position e of the three consecutive
  // lines l, l + 1, l + 2 as one aligned 16-byte record, so a step fetches three window lines per load instruction
  // and a window starting at any line finds its lines as whole triples l, l + 3, l + 6 ...  (The sweeps are bound by
  // the vector memory pipeline, which works per 4 lanes and cache line: fewer, wider loads are what counts.)
  // "Lines" are image rows for the row sweeps (rpg) and image columns, on the transposed planes, for the column
  // sweeps (cpg).  Per pair b and view v (target image of the view for the records, reference image for the quads):
  float* rpg;          // [B][2][nrl][pitch][4]    target records {gradient of lines l, l + 1, l + 2, u32 colour bytes}
  // reference QUADS (row sweeps): element (l, e) = image rows l .. l + 3 at column e as {four colour bytes, four
  // gradient bytes}: the form the sweeps' v_sad_u8 takes
  uint32_t* rqk;       // [B][2][nrl][pitch][2]
  float* cpg;          // [B][2][ncl][pitch_t][4]  the same records on the transposed target planes
  int nrl, ncl;        // lines per view: rows + 2, cols + kTransPad + 2 (lines beyond the last repeat it)
  // STATE planes: four image rows interleaved -- element (x, y) at state_at(x, y, pitch) = ((y >> 2) * pitch + x) * 4 +
  // (y & 3), plane stride `splane` -- so that a column chain finds four of its positions in one 16-byte piece (180
  // cache lines per plane for 720 rows instead of 720) while a row chain still reads every fourth word of a contiguous
  // stretch (round 4: DESIGN.md 6, "what a launch costs besides its steps").  Every access goes through state_at() / chain_at().
  float* disp;         // [B][2][rows4][pitch][4]   view 0 = left, view 1 = right (mirrored coordinates)
  float* cost;         // [B][2][rows4][pitch][4]   cost of disp under the current window
  const float* noise;  // [rows][pitch]         cv::RNG(seed) uniform [-1,1), shared by all slots
  unsigned long long* counters;  // [8] work counters (see pm_debug_counters), one atomic per wavefront
  // Row-tiled mode: chains a sweep launch works on, [n_views][cols] (chain index = plane column), nullptr = all.
  // A re-sweep after a boundary exchange only touches the columns whose incoming value changed.
  const int* chain_mask;
  int rows, cols, pitch;
  int n_views;         // 1 or 2
  int view_fixed;      // -1: slot = pair * n_views + view; 0 / 1: slot = pair, this view only (per-view streams)
  size_t plane;        // rows * pitch
  size_t splane;       // state planes: align_up(rows, 4) * pitch
};

// index of pixel (x, y) in a state plane (PlaneSet::disp / cost)
__host__ __device__ __forceinline__ size_t state_at(int x, int y, int pitch) {
  return (((size_t)(y >> 2) * (size_t)pitch + (size_t)x) << 2) + (size_t)(y & 3);
}
// position s of chain `chain`: a row chain (axis 0) runs along x, a column chain along y
__host__ __device__ __forceinline__ size_t chain_at(int axis, int chain, int s, int pitch) {
  return axis == 0 ? state_at(s, chain, pitch) : state_at(chain, s, pitch);
}

// The planes one view works on.  View 1 is "the same algorithm on the horizontally mirrored
// (R, L) pair" (src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:357-368): the mirrored copies are
// written once by the prep kernel, so no flip pass exists.
struct View {
  const uint8_t* ref8;   // reference image ("iml")
  const uint8_t* tgt8;   // target image ("imr")
  const float* refg;     // gradient magnitude of ref ("Gl")
  const float* tgtg;     // gradient magnitude of tgt ("Gr")
  const uint8_t* refg8;  // saturated u8 of refg
  const uint8_t* tref8;  // transposed copies of ref8 / tgt8 / tgtg / refg8
  const uint8_t* ttgt8;
  const float* ttgtg;
  const uint8_t* trefg8;
  const float* trefg;      // transposed refg (PM_SEM_GPU column sweeps)
  const uint16_t* refpk;   // ref8 | refg8 << 8
  const uint16_t* trefpk;  // transposed
  // column sweeps: the reference window lines of the chain staged in LDS, [image row][kLref4Stride] dwords, four
  // window columns per dword (pm_run3.hpp, LREF)
  const unsigned* lds_ref4;
  // line-triple / quad planes of this view (first line)
  const float* rpg;
  const uint32_t* rqk;
  const float* cpg;
  float* disp;
  float* cost;
};

// view index of a launch slot (the same rule make_view applies)
__device__ __forceinline__ int slot_view(const PlaneSet& ps, int slot) {
  return ps.view_fixed >= 0 ? ps.view_fixed : slot - (slot / ps.n_views) * ps.n_views;
}
// false: this chain is masked off in this launch (uniform per workgroup for the chain engines)
__device__ __forceinline__ bool chain_active(const PlaneSet& ps, int slot, int chain) {
  return !ps.chain_mask || ps.chain_mask[slot_view(ps, slot) * ps.cols + chain] != 0;
}

__device__ __forceinline__ View make_view(const PlaneSet& ps, int slot) {
  const int b = ps.view_fixed >= 0 ? slot : slot / ps.n_views;
  const int v = ps.view_fixed >= 0 ? ps.view_fixed : slot - b * ps.n_views;
  const int iref = v == 0 ? 0 : 3, itgt = v == 0 ? 1 : 2;
  const size_t base4 = (size_t)b * 4;
  View w;
  w.ref8 = ps.img8 + (base4 + iref) * ps.plane;
  w.tgt8 = ps.img8 + (base4 + itgt) * ps.plane;
  w.refg = ps.g32 + (base4 + iref) * ps.plane;
  w.tgtg = ps.g32 + (base4 + itgt) * ps.plane;
  w.refg8 = ps.g8 + (base4 + iref) * ps.plane;
  w.tref8 = ps.timg8 + (base4 + iref) * ps.plane_t;
  w.ttgt8 = ps.timg8 + (base4 + itgt) * ps.plane_t;
  w.ttgtg = ps.tg32 + (base4 + itgt) * ps.plane_t;
  w.trefg8 = ps.tg8 + (base4 + iref) * ps.plane_t;
  w.trefg = ps.tg32 + (base4 + iref) * ps.plane_t;
  w.refpk = ps.pk16 + (base4 + iref) * ps.plane;
  w.trefpk = ps.tpk16 + (base4 + iref) * ps.plane_t;
  w.lds_ref4 = nullptr;
  const size_t pv = (size_t)b * 2 + v;
  w.rpg = ps.rpg + pv * (size_t)ps.nrl * ps.pitch * 4;
  w.rqk = ps.rqk + pv * (size_t)ps.nrl * ps.pitch * 2;
  w.cpg = ps.cpg + pv * (size_t)ps.ncl * ps.pitch_t * 4;
  const size_t dofs = ((size_t)b * 2 + v) * ps.splane;
  w.disp = ps.disp + dofs;
  w.cost = ps.cost + dofs;
  return w;
}

// Constants of one cost evaluation.
struct CostParams {
  int semantics;  // PM_SEM_CPU / PM_SEM_GPU
  int pw, ph;     // window (PM_SEM_CPU); PM_SEM_GPU uses the 5-tap 3x3 and radius 1
  // PM_SEM_CPU: L1GradientCostFunction (test/stereo_matching/patchmatch_test.cpp:30-45)
  float alpha, one_minus_alpha, tau_color, tau_grad;
  double inv_n;  // 1./(pw*ph): cv::mean multiplies the sum by the reciprocal in double
  float inv_n_hi, inv_n_lo;  // inv_n split into two floats (mean_from_sum)
  // PM_SEM_GPU: L1GradientCost3x3 (patchmatch_gpu.cu:72-114)
  float g_alpha, g_one_minus_alpha;
};

// saturate_cast<uchar>(float) = clamp(cvRound(v), 0, 255); cvRound rounds half to even.
__device__ __forceinline__ int sat_u8(float v) {
  const int iv = __float2int_rn(v);
  return min(max(iv, 0), 255);
}

// ---------------------------------------------------------------------------------------------
// PM_SEM_CPU tap arithmetic.  One window sample of cv::getRectSubPix on the path (integer y, so
// the vertical weight is 0 and the sample is the horizontal two-tap lerp):
//   8u:  (r0*a11 + r1*a12 + 2^15) >> 16,  a11 = cvRound((1-a)*2^16), a12 = cvRound(a*2^16)
//   32f: r0*(1-a) + r1*a
// (OpenCV 3.4 modules/imgproc/src/samplers.cpp getRectSubPix_Cn_; called from GetPatchSubpix,
// src/vehicle/stereo_matching/patchmatch.cpp:98-111.)
// ---------------------------------------------------------------------------------------------
struct CpuLerp {
  int ipx;  // column of the first tap of the window row
  int a11, a12;
  float a, ia;
};

__device__ __forceinline__ CpuLerp cpu_lerp(int x, float d, int pw) {
  float cx = (float)x - d;
  cx = cx - (float)(pw - 1) * 0.5f;
  CpuLerp l;
  const float fl = floorf(cx);
  l.ipx = (int)fl;
  l.a = cx - fl;
  l.ia = 1.f - l.a;
  l.a11 = __float2int_rn(l.ia * 65536.f);
  l.a12 = __float2int_rn(l.a * 65536.f);
  return l;
}

// r0, r1 <= 255 and a11, a12 <= 65536 fit 24 bits: v_mad_u32_u24 (full rate) instead of v_mul_lo_u32.
__device__ __forceinline__ int cpu_tap_color(int left, int r0, int r1, const CpuLerp& l) {
  unsigned t = __umul24((unsigned)r1, (unsigned)l.a12) + (1u << 15);
  t = __umul24((unsigned)r0, (unsigned)l.a11) + t;
  const int v = (int)(t >> 16);
  return abs(left - v);
}
__device__ __forceinline__ int cpu_tap_grad(int left_g8, float g0, float g1, const CpuLerp& l) {
  float s = g0 * l.ia;
  s = s + g1 * l.a;
  return abs(left_g8 - sat_u8(s));
}
// Accumulate forms for the VALU-bound kernels: acc + |left - sample| as ONE v_sad_u8 each.
// v_sad_u8 adds the absolute differences of all four bytes of its operands; `left` and the colour
// sample are < 256, so only byte 0 contributes.
// The colour lerp sum as ONE v_dot2_u32_u16: r01 = r0 | r1 << 16, w = cpu_color_weights().  Both weights are
// clamped to 65535 so that they fit 16 bits: a weight of 65536 happens only next to a = 0 or a = 1, where the other
// weight is 0 or 1, and then r * 65535 + r' * k + 2^15 = r * 65536 + (2^15 + r' * k - r) has the same byte 2 (= r)
// as r * 65536 + r' * k + 2^15 (the low part stays within [0, 2^16)): checked for every pair that occurs by
// tests/test_oracle_primitives.py::test_lerp_weights_fit_16_bits_when_clamped.  Only byte 2 of the result -- the
// sample -- may be used.
typedef unsigned short pm_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cpu_color_weights(const CpuLerp& l) {
  return (unsigned)min(l.a11, 65535) | ((unsigned)min(l.a12, 65535) << 16);
}
__device__ __forceinline__ unsigned cpu_color_sum_pk(unsigned r01, unsigned w) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(pm_u16x2, r01), __builtin_bit_cast(pm_u16x2, w), 1u << 15, false);
}
// The colour lerp sum alone: r0 * a11 + r1 * a12 + 2^15 < 2^24, the sample is its byte 2.
__device__ __forceinline__ unsigned cpu_color_sum(int r0, int r1, const CpuLerp& l) {
  const unsigned t = __umul24((unsigned)r1, (unsigned)l.a12) + (1u << 15);
  return __umul24((unsigned)r0, (unsigned)l.a11) + t;
}
__device__ __forceinline__ unsigned cpu_acc_color(unsigned acc, int left, int r0, int r1, const CpuLerp& l) {
  unsigned t = __umul24((unsigned)r1, (unsigned)l.a12) + (1u << 15);
  t = __umul24((unsigned)r0, (unsigned)l.a11) + t;
  return __builtin_amdgcn_sad_u8((unsigned)left, t >> 16, acc);
}
// saturate_cast<uchar>(s) = clamp(rint(s), 0, 255), ties to even: exactly what v_cvt_pk_u8_f32 computes
// (checked on the device for every multiple of 1/16 in [-0.5, 260), the floats next to every .5 tie and
// +-1e30: tools/probe/cvt_pk_u8.hip), and it drops the byte into a chosen lane of a dword -- one
// instruction instead of clamp + magic add (+ v_perm when four taps are packed).
__device__ __forceinline__ unsigned cpu_acc_grad(unsigned acc, int left_g8, float g0, float g1, const CpuLerp& l) {
  float s = g0 * l.ia;
  s = s + g1 * l.a;
  return __builtin_amdgcn_sad_u8((unsigned)left_g8, __builtin_amdgcn_cvt_pk_u8_f32(s, 0, 0u), acc);
}
// The same with the lerp sum already formed (packed-f32 products, pm_run2.hpp / k_noise_cost_tiled).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cpu_acc_grad_sum(unsigned acc, int left_g8, float s) {
  return __builtin_amdgcn_sad_u8((unsigned)left_g8, __builtin_amdgcn_cvt_pk_u8_f32(s, 0, 0u), acc);
}
// Loads addressed as (wave-uniform plane base) + (32-bit unsigned byte offset): the form that maps
// to global_load ... v_off, s[base:base+1] -- no 64-bit per-lane address arithmetic, no VGPR pairs.
// Offsets are relative to one view's plane, i.e. < 4 GiB for any image this engine accepts.
__device__ __forceinline__ int ld_u8(const uint8_t* base, unsigned off) { return base[(size_t)off]; }
__device__ __forceinline__ int ld_u16(const uint16_t* base, unsigned byte_off) {
  return *(const uint16_t*)((const char*)base + (size_t)byte_off);
}
__device__ __forceinline__ float ld_f32(const float* base, unsigned byte_off) {
  return *(const float*)((const char*)base + (size_t)byte_off);
}
// Two adjacent elements with one load (the pair may start at any byte / dword).
__device__ __forceinline__ void load_pair_u8(const uint8_t* base, unsigned off, int& a, int& b) {
  unsigned short v;
  __builtin_memcpy(&v, base + (size_t)off, 2);
  a = v & 0xff;
  b = v >> 8;
}
__device__ __forceinline__ void load_pair_f32(const float* base, unsigned byte_off, float& a, float& b) {
  struct P { float x, y; } v;
  __builtin_memcpy(&v, (const char*)base + (size_t)byte_off, 8);
  a = v.x;
  b = v.y;
}

// (float)((double)sum * inv_n) without f64 conversions: sum * (hi + lo) with the rounding error of the leading
// product recovered by one FMA.  Equal to the double form for EVERY window 3..15 x 3..15 and every sum
// 0 .. 255 * pw * ph (exhaustive check: tests/test_oracle_primitives.py::test_mean_from_sum_is_exact).
__device__ __forceinline__ float mean_from_sum(int sum, const CostParams& cp) {
  const float sf = (float)sum;
  const float p = sf * cp.inv_n_hi;
  const float e1 = __builtin_fmaf(sf, cp.inv_n_hi, -p);
  const float e2 = sf * cp.inv_n_lo;
  return p + (e1 + e2);
}

// mean = (float)(sum * (1./N)); cost = alpha*min(mean_c, tau_c) + (1-alpha)*min(mean_g, tau_g).
__device__ __forceinline__ float cpu_cost_from_sums(int sc, int sg, const CostParams& cp) {
  const float mc = mean_from_sum(sc, cp);
  const float mg = mean_from_sum(sg, cp);
  const float ec = fminf(mc, cp.tau_color);
  const float eg = fminf(mg, cp.tau_grad);
  const float t0 = cp.alpha * ec;
  const float t1 = cp.one_minus_alpha * eg;
  return t0 + t1;
}

// Whole-window cost by one lane.  Valid for pw/2 <= x <= cols-pw/2-1, ph/2 <= y <= rows-ph/2-1 and
// 0 <= d <= x - pw/2 (then 0 <= ipx and ipx + pw <= cols, the second tap of the last column has
// weight 0 when it would fall outside and is clamped).
__device__ __forceinline__ float cpu_cost_lane(const View& v, int pitch, int cols, int x, int y, float d,
                                const CostParams& cp) {
  const CpuLerp l = cpu_lerp(x, d, cp.pw);
  const int x0 = x - cp.pw / 2, y0 = y - cp.ph / 2;
  int sc = 0, sg = 0;
  for (int i = 0; i < cp.ph; ++i) {
    const size_t row = (size_t)(y0 + i) * pitch;
    const uint8_t* lp = v.ref8 + row + x0;
    const uint8_t* lg = v.refg8 + row + x0;
    const uint8_t* rp = v.tgt8 + row;
    const float* rg = v.tgtg + row;
    int c0 = l.ipx;
    int r0 = rp[c0];
    float g0 = rg[c0];
    for (int j = 0; j < cp.pw; ++j) {
      const int c1 = min(c0 + 1, cols - 1);
      const int r1 = rp[c1];
      const float g1 = rg[c1];
      sc += cpu_tap_color(lp[j], r0, r1, l);
      sg += cpu_tap_grad(lg[j], g0, g1, l);
      r0 = r1;
      g0 = g1;
      c0 = c1;
    }
  }
  return cpu_cost_from_sums(sc, sg, cp);
}

// ---------------------------------------------------------------------------------------------
// PM_SEM_GPU: GetSubpixel (patchmatch_gpu.cu:18-42) at an integer row, and L1GradientCost3x3
// (:72-114): taps (-1,-1) (-1,+1) (0,0) (+1,-1) (+1,+1), accumulated in that order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float gpu_tap(float il, float gl, float r0, float r1, float g0, float g1, float tcol,
                         const CostParams& cp) {
  const float ic = 1.0f - tcol;
  float rs = ic * r0;
  rs = rs + tcol * r1;
  float gs = ic * g0;
  gs = gs + tcol * g1;
  const float e0 = fabsf(il - rs);
  const float e1 = fabsf(gl - gs);
  const float t0 = cp.g_alpha * e0;
  const float t1 = cp.g_one_minus_alpha * e1;
  return t0 + t1;
}

__device__ __forceinline__ float gpu_cost_lane(const View& v, int pitch, int x, int y, float xr, const CostParams& cp) {
  const int dy[5] = {-1, -1, 0, 1, 1};
  const int dx[5] = {-1, 1, 0, -1, 1};
  float cost = 0.f;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const float xs = xr + (float)dx[t];
    const float f0 = floorf(xs);
    const int col0 = (int)f0, col1 = (int)ceilf(xs);
    const float tcol = xs - f0;
    const size_t lrow = (size_t)(y + dy[t]) * pitch;
    const float il = (float)v.ref8[lrow + x + dx[t]];
    const float gl = v.refg[lrow + x + dx[t]];
    const float r0 = (float)v.tgt8[lrow + col0], r1 = (float)v.tgt8[lrow + col1];
    const float g0 = v.tgtg[lrow + col0], g1 = v.tgtg[lrow + col1];
    const float s = gpu_tap(il, gl, r0, r1, g0, g1, tcol, cp);
    cost = cost + s;
  }
  return cost;
}

// ---------------------------------------------------------------------------------------------
// Candidate rule of one sweep step: what a pixel holding (d0, c0) does when its predecessor in
// the sweep holds `cand`.  Returns true and fills (nd, nc) when the pixel changes.
//   PM_SEM_CPU  PropagateNeighbors (patchmatch.cpp:158-196): candidate considered only if
//               x - cand >= pw/2, adopted on a strictly smaller cost; d0 is already clamped to
//               [0, x - pw/2] (the noise kernel applies the clamp of :175 to every interior pixel,
//               which every pass-A visit would otherwise do before any neighbour reads it).
//   PM_SEM_GPU  PropagateRow/Col (patchmatch_gpu.cu:156-171): xr = max(x - cand, r), adopted on a
//               strictly smaller cost, stored as min(cand, x - r).
// A candidate equal to d0 (or mapping to the same xr) has an equal cost and is never adopted, so
// its evaluation is skipped; this is exact, not an approximation.
// ---------------------------------------------------------------------------------------------
template <typename CostFn>
__device__ __forceinline__ bool sweep_step(int semantics, int x, int half_w, float d0, float c0, float cand,
                                           float& nd, float& nc, CostFn&& cost_at) {
  if (semantics == 0) {
    if (cand == d0) return false;
    if (!(((float)x - cand) >= (float)half_w)) return false;
    const float c1 = cost_at(cand);
    if (c1 < c0) {
      nd = cand;
      nc = c1;
      return true;
    }
    return false;
  } else {
    const float r = (float)half_w;
    const float xr0 = fmaxf((float)x - d0, r), xr1 = fmaxf((float)x - cand, r);
    if (xr0 == xr1) return false;
    const float c1 = cost_at(xr1);
    if (c1 < c0) {
      nd = fminf(cand, (float)x - r);
      nc = c1;
      return true;
    }
    return false;
  }
}

}  // namespace pm
