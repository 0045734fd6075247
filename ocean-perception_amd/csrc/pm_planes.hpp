// pm_planes.hpp -- PM_MODE_PLANES: slanted-plane PatchMatch stereo kernels for gfx950.
//
// BASELINE.json's north_star names random slanted-plane initialisation, red-black spatial propagation, view
// propagation, random plane refinement and a per-pixel windowed match cost.  The reference has no such code
// (its device entry point, src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:379-411, keeps one scalar disparity
// per pixel); the algorithm is DEFINED by oracle/pm_planes_oracle.{h,c} and these kernels reproduce that
// definition bit for bit.  What is shared with the reference path: the images / Sobel gradient planes of
// k_prep, the cost functor's form (test/stereo_matching/patchmatch_test.cpp:30-45), cv::RNG's
// multiply-with-carry step, "right view = same algorithm on the mirrored (R, L) pair" (patchmatch_gpu.cu:357-368).
//
// State per pixel and view, SoA in HBM ([rows][pitch] each, f32 or f16): plane (a, b, z) in pixel-local form
// (z = disparity AT the pixel, d(x+dx, y+dy) = z + a*dx + b*dy) and the cost of that plane.
//
// One kernel template, four stages.  A block of 512 threads owns a tile of 8 rows x 64 pixels (x 128 columns for the
// red-black stage, whose lanes hold only the pixels of the active colour), stages in LDS
//   * the reference window bytes of the tile (colour and saturated gradient, (TW+P-1) x (8+P-1)),
//   * the target rows over the whole column range any admissible plane can reach
//     (TW + P-1 + max_disp + slope margin; 8 bytes per column: the pixel AND its right neighbour, each as
//     colour | gradient << 16, so that one aligned ds_read_b64 fetches both bilinear taps),
//   * for the red-black stage the planes of the tile with a 1-pixel ring (neighbour candidates),
// and every lane then evaluates its candidates out of LDS.  The window cost is integer arithmetic:
// 16.16 fixed-point column per tap, 8-bit lerp weights applied to both channels with ONE multiply-add pair
// (the channels sit 16 bits apart in the dword and cannot carry into each other), four taps per v_sad_u8.
#pragma once

#include "pm_device.hpp"
#include "pm_sweep_defs.hpp"

namespace pm {

enum { PL_INIT = 0, PL_SPATIAL = 1, PL_VIEW = 2, PL_REFINE = 3, PL_VIEW_REFINE = 4 };  // 4 = 2 then 3, one tile fill
enum { PL_RAND_INIT = 0, PL_RAND_REFINE = 1 };  // `stage` of the random key (oracle: ST_INIT / ST_REFINE)

// Tile height (rows = wavefronts of a block) per stage.  Results do not depend on it; what it trades is the halo (P - 1
// extra rows of reference and target tile per block), the wavefronts a CU holds, and how evenly a launch's blocks fill
// the chip's block slots (a 1280x720 view is 1800 blocks of 8 rows: 2.3 rounds on the 768 slots three blocks per CU give).
// -DPL_TILE_H_SPATIAL / -DPL_TILE_H_OTHER: A/B builds (profiles/r06_planes_tile_height.txt).
#ifndef PL_TILE_H_SPATIAL
#define PL_TILE_H_SPATIAL 8
#endif
#ifndef PL_TILE_H_OTHER
#define PL_TILE_H_OTHER 8
#endif
#ifndef PL_SPATIAL_REF_REGS
#define PL_SPATIAL_REF_REGS 1
#endif
__host__ __device__ constexpr int pl_tile_h(int stage) { return stage == 1 /* PL_SPATIAL */ ? PL_TILE_H_SPATIAL : PL_TILE_H_OTHER; }

struct PlanesParams {
  int patch, max_disp, refine_steps, margin;
  float slope_max;  // effective bound (f16-representable in f16 mode)
  float slope_init, slope_per_disp;
  float alpha, one_minus_alpha, tau_color, tau_grad, inv_n;
  float lr_tol;
  unsigned long long seed;
  int n_views;
  int window;  // PM_PL_WINDOW_FULL / PM_PL_WINDOW_CHECKER (pm/patchmatch.h): which taps of the P x P window count
  int neighbours;  // PM_PL_NEIGH_FOUR / PM_PL_NEIGH_TWO: the spatial stage's candidates
};

// State of all slots: slot (pair, view) holds 4 arrays of `plane` elements: a, b, z, cost.
// An array is split by COLOUR of the red-black stage: the pixels with x + y even first ([rows][pitch / 2], element
// x >> 1 of row y), then those with x + y odd.  A red / black launch then reads and writes contiguous half-arrays --
// interleaved, the colour it does not touch shares every 32-byte sector with the one it writes, and the stage wrote all
// four arrays whole (round 2: 28.8 MiB written per launch for 14.7 MiB of changed state).
template <typename ST>
struct PlaneState {
  ST* base;
  size_t plane;  // rows * pitch
  int half_pitch;  // pitch / 2
  __device__ __forceinline__ ST* arr(int pair, int view, int k) const {
    return base + (((size_t)pair * 2 + view) * 4 + k) * plane;
  }
  __device__ __forceinline__ size_t idx(int x, int y) const {
    return (size_t)((x + y) & 1) * (plane >> 1) + (size_t)y * half_pitch + (size_t)(x >> 1);
  }
};

template <typename ST>
__device__ __forceinline__ float pl_quant(float v) {
  if constexpr (sizeof(ST) == 2) return (float)(_Float16)v;
  else return v;
}
template <typename ST>
__device__ __forceinline__ float pl_load(const ST* p, size_t o) { return (float)p[o]; }
template <typename ST>
__device__ __forceinline__ void pl_store(ST* p, size_t o, float v) { p[o] = (ST)v; }

// oracle: pmo_planes_rand
__device__ __forceinline__ unsigned pl_rand(unsigned long long seed, int stage, int it, int k, int view, int draw,
                                            int x, int y) {
  const unsigned long long tag = (unsigned long long)stage | ((unsigned long long)it << 4) |
                                 ((unsigned long long)k << 12) | ((unsigned long long)view << 20) |
                                 ((unsigned long long)draw << 24);
  unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (tag + 1);
  s ^= ((unsigned long long)(unsigned)y << 32) | (unsigned long long)(unsigned)x;
  s ^= s >> 30;
  s *= 0xBF58476D1CE4E5B9ull;
  s ^= s >> 27;
  s *= 0x94D049BB133111EBull;
  s ^= s >> 31;
  s = (unsigned long long)(unsigned)s * 4164903690u + (s >> 32);
  return (unsigned)s;
}
__device__ __forceinline__ float pl_pm1(unsigned r) { return (float)(int)r * 4.656612873077392578125e-10f; }
__device__ __forceinline__ float pl_u01(unsigned r) { return (float)(r >> 8) * 5.9604644775390625e-08f; }

__device__ __forceinline__ float pl_clamp_slope(float v, float smax) { return fminf(fmaxf(v, -smax), smax); }

// LDS images of one tile.
typedef unsigned pl_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) pl_u2 pl_lds_u2;
struct PlTile {
  int tgt_entry0;       // ABSOLUTE LDS address / 8 of the target tile: [TR][rw] 8-byte entries pl_entry(P(c), P(c + 1)),
                        // P = colour | gradient << 16, c = xs_lo + index
  // Reference window bytes: FOUR copies of the tile per channel, copy s shifted left by s bytes
  // (copy_s[k] = tile[k + s]), rows padded to whole dwords: a lane whose window starts at byte offset f of a row reads
  // ALIGNED dwords (f >> 2) of copy (f & 3) and gets its window bytes in place -- no v_alignbyte per dword.
  // Checkerboard window: the tile's columns are split by parity first (the taps of a window row are every other column),
  // each half then kept as four shifted copies: [2 parities][4][TR][2 LWW].
  // A row holds its LWW colour dwords and then its LWW gradient dwords (round 6; two planes before): the gradient sits
  // at a compile-time offset from the colour, so one address register and the offset fields of ds_read2_b32 serve both.
  const unsigned* rc;   // [4][TR][2 LWW] dwords: colour, gradient = + LWW
  int rw, copy_w;       // target row entries; dwords per copy (TR * 2 LWW)
  int tr;               // rows of the staged tiles: tile height + P - 1
};

// A target entry and the lerp of a tap (round 5).  The sample of a tap is (p0 (256 - w) + p1 w + 128) >> 8 per channel
// (oracle: pmo_planes_cost); with the channels 16 bits apart that is ONE packed 16-bit multiply-add if the entry holds
// {A, D}:  A = (P(c) << 8) + 0x00800080 (both channels times 256, plus the rounding term),  D = P(c + 1) - P(c) per
// 16-bit channel (two's complement), and  sample = byte 1 / byte 3 of  A + D * w  in arithmetic modulo 2^16 per channel
// -- the true value p0 * 256 + (p1 - p0) * w + 128 lies in [128, 65408], so the wrap-around of a negative D * w is
// exact.  v_pk_mad_u16 with the weight's low half feeding both channels (op_sel_hi) replaces a subtract, two 24-bit
// multiplies and a three-way add: 6 -> 3.5 vector instructions per tap besides the byte gathers (column add, address
// shift, half a weight gather, the multiply-add).
typedef short pl_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pl_u2 pl_entry(unsigned p0, unsigned p1) {  // p = colour | gradient << 16
  pl_u2 v;
  v.x = (p0 << 8) + 0x00800080u;
  v.y = __builtin_bit_cast(unsigned, (pl_s16x2)(__builtin_bit_cast(pl_s16x2, p1) - __builtin_bit_cast(pl_s16x2, p0)));
  return v;
}
// The weights of two taps in one register (one v_perm_b32 per PAIR of taps instead of one bit-field extract per tap):
// byte 1 of the first tap's 16.16 column in the low half, of the second tap's in the high half.
__device__ __forceinline__ unsigned pl_weights2(int x_first, int x_second) {
  return __builtin_amdgcn_perm((unsigned)x_second, (unsigned)x_first, 0x0c050c01u);
}
// HALF = 0 / 1: the low / high half of `w2` is the tap's weight, fed to BOTH channels (op_sel / op_sel_hi of source 1)
template <int HALF>
__device__ __forceinline__ unsigned pl_lerp(pl_u2 e, unsigned w2) {
  unsigned r;
  if constexpr (HALF == 0) asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(e.y), "v"(w2), "v"(e.x));
  else asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(e.y), "v"(w2), "v"(e.x));
  return r;
}

// Window cost of plane (a, b, z) for the pixel at tile position (lx, ty); xrel = its column - h - xs_lo.
// oracle: pmo_planes_cost.
// The 16.16 column of a tap carries the LDS entry index of its row in the integer part (t.tgt_entry0 + row * rw is
// folded into it), so the byte address of the tap's 8-byte entry is ((X >> 16) << 3) -- one SDWA shift -- and the
// low half is untouched: bits 8..15 are the lerp weight.
// Measured at 720p: 225 pairs/s with the early termination below, 228 without -- a wavefront leaves the window only
// when all 64 candidates are hopeless, and one competitive lane (a neighbour's near-identical plane, a sub-pixel
// refinement) is almost always there.  Off by default; -DPL_EARLY_EXIT=1 for experiments (bit-identical results).
#ifndef PL_EARLY_EXIT
#define PL_EARLY_EXIT 0
#endif
template <int P, int LWW>
__device__ __forceinline__ float pl_cost(const PlTile& t, int lx, int ty, int xrel, float a, float b, float z,
                                         const PlanesParams& pp, float bound = __builtin_inff()) {
  constexpr int h = P / 2;
  constexpr int NG = (P + 3) / 4;
  const int Z = __float2int_rn(z * 65536.0f), A = __float2int_rn(a * 65536.0f), B = __float2int_rn(b * 65536.0f);
  int xrow = ((xrel + t.tgt_entry0 + ty * t.rw) << 16) - Z + A * h + B * h;  // tap (0, 0)
  int stepj = 65536 - A;
  // opaque to the optimiser: otherwise it splits j * stepj into j * 65536 - j * A and spends three adds per tap
  asm volatile("" : "+v"(stepj));
  // column offsets j * stepj by running additions, each made opaque: written as products they compile to one
  // 64-bit v_mad_u64_u32 (quarter rate) per odd multiple and candidate
  int xoff[P];
  xoff[0] = 0;
#pragma unroll
  for (int j = 1; j < P; ++j) {
    int t = xoff[j - 1] + stepj;
    asm volatile("" : "+v"(t));
    xoff[j] = t;
  }
  const int rowstep = (t.rw << 16) - B;
  unsigned sc = 0, sg = 0;
  // reference side: the lane's copy and dword column are fixed for the whole window (rows are whole dwords)
  const int ref0 = (lx & 3) * t.copy_w + ty * (2 * LWW) + (lx >> 2);
  const unsigned* pl = t.rc + ref0;
#pragma unroll 1
  for (int i = 0; i < P; ++i) {
    // all LDS reads of the window row first (target pairs, then reference bytes), arithmetic afterwards: the
    // reads of a row are in flight together instead of one wait per tap
    pl_u2 tp[P];
    unsigned wq[(P + 1) / 2];  // the weights of taps 2 m and 2 m + 1
    int xs[P + 1];
#pragma unroll
    for (int j = 0; j < P; ++j) {
      xs[j] = xrow + xoff[j];
      tp[j] = *(const pl_lds_u2*)(uintptr_t)(((unsigned)xs[j] >> 16) << 3);  // absolute LDS byte address
    }
    xs[P] = 0;
#pragma unroll
    for (int m = 0; m < (P + 1) / 2; ++m) wq[m] = pl_weights2(xs[2 * m], xs[2 * m + 1]);
    unsigned lw[NG], lgw[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      lw[q] = pl[q];
      lgw[q] = pl[LWW + q];
    }
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      unsigned s[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = 4 * q + k;
        if (j < P) s[k] = (j & 1) ? pl_lerp<1>(tp[j], wq[j / 2]) : pl_lerp<0>(tp[j], wq[j / 2]);
      }
      // bytes 1 / 3 of every s[k] are the colour / gradient samples: gather four of each (a tap the window
      // does not have contributes s = 0, i.e. sample bytes 0: only the reference side needs the mask)
      const unsigned u01 = __builtin_amdgcn_perm(s[1], s[0], 0x07030501u);
      const unsigned u23 = __builtin_amdgcn_perm(s[3], s[2], 0x07030501u);
      const unsigned pc = __builtin_amdgcn_perm(u23, u01, 0x05040100u);
      const unsigned pg = __builtin_amdgcn_perm(u23, u01, 0x07060302u);
      const int rem = P - 4 * q;
      const unsigned mask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
      sc = __builtin_amdgcn_sad_u8(lw[q] & mask, pc, sc);
      sg = __builtin_amdgcn_sad_u8(lgw[q] & mask, pg, sg);
    }
    xrow += rowstep;
    pl += 2 * LWW;
    // Early termination (exact): the sums only grow and the cost is monotone in them (one rounding per monotone
    // operation), so a partial cost that has reached `bound` -- the cost the candidate has to beat -- can only end
    // in a reject.  When that holds for every lane of the wavefront the remaining rows are skipped and the partial
    // cost (>= bound) is returned.  bound < 0: the lane has no candidate; +inf (the default): never terminate.
    if (PL_EARLY_EXIT && (i & 1) == 0 && i >= 2 && i < P - 2) {
      const float mc = (float)(int)sc * pp.inv_n, mg = (float)(int)sg * pp.inv_n;
      const float part = pp.alpha * fminf(mc, pp.tau_color) + pp.one_minus_alpha * fminf(mg, pp.tau_grad);
      if (!__any(part < bound)) return part;
    }
  }
  const float mc = (float)(int)sc * pp.inv_n, mg = (float)(int)sg * pp.inv_n;
  const float t0 = pp.alpha * fminf(mc, pp.tau_color);
  const float t1 = pp.one_minus_alpha * fminf(mg, pp.tau_grad);
  return t0 + t1;
}

// The same cost on the CHECKERBOARD window (oracle: PMO_PL_WINDOW_CHECKER, the mode's default since round 4): tap (i, j)
// counts iff i + j is even -- window rows 0, 2, 4 ... take columns 0, 2, 4 ..., rows 1, 3, ... take columns 1, 3, ...
// (61 of the 121 taps of an 11 x 11 window; the mean divides by the taps that count).  The stage is bound by vector issue
// at ~8 instructions per tap, so half the taps is what buys time; on the 1280x720 benchmark pairs the disparities within
// 1 px of the truth stay at 99.86 % (DESIGN.md 5b has the table).  The reference bytes of a row's taps are every other
// column of the tile: the tile is kept split by column parity, so that they are consecutive bytes again and the aligned
// dword reads of the shifted copies work as for the full window.
// The colour bytes (byte 1) and the gradient bytes (byte 3) of up to four lerp results, four of each per dword, zeros where
// the group is short.  n = 4: four v_perm_b32; 3 and 2: three; 1: two.
__device__ __forceinline__ void pl_gather(const unsigned (&s)[4], int n, unsigned& pc, unsigned& pg) {
  if (n >= 4) {
    const unsigned u01 = __builtin_amdgcn_perm(s[1], s[0], 0x07030501u);
    const unsigned u23 = __builtin_amdgcn_perm(s[3], s[2], 0x07030501u);
    pc = __builtin_amdgcn_perm(u23, u01, 0x05040100u);
    pg = __builtin_amdgcn_perm(u23, u01, 0x07060302u);
  } else if (n == 3) {
    const unsigned u01 = __builtin_amdgcn_perm(s[1], s[0], 0x07030501u);  // c0 c1 g0 g1
    pc = __builtin_amdgcn_perm(s[2], u01, 0x0c050100u);
    pg = __builtin_amdgcn_perm(s[2], u01, 0x0c070302u);
  } else if (n == 2) {
    const unsigned u01 = __builtin_amdgcn_perm(s[1], s[0], 0x07030501u);
    pc = __builtin_amdgcn_perm(0u, u01, 0x0c0c0100u);
    pg = __builtin_amdgcn_perm(0u, u01, 0x0c0c0302u);
  } else {
    pc = __builtin_amdgcn_perm(0u, s[0], 0x0c0c0c01u);
    pg = __builtin_amdgcn_perm(0u, s[0], 0x0c0c0c03u);
  }
}
template <int P, int NT, int J0, int LWW>
__device__ __forceinline__ void pl_row_taps(int xrow, const int (&xoff)[P], const unsigned* pl, unsigned& sc, unsigned& sg) {
  constexpr int NG = (NT + 3) / 4;
  pl_u2 tp[NT];
  unsigned wq[(NT + 1) / 2];  // the weights of taps 2 m and 2 m + 1 of this row
  int xs[NT + 1];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    xs[k] = xrow + xoff[J0 + 2 * k];
    tp[k] = *(const pl_lds_u2*)(uintptr_t)(((unsigned)xs[k] >> 16) << 3);  // absolute LDS byte address
  }
  xs[NT] = 0;
#pragma unroll
  for (int m = 0; m < (NT + 1) / 2; ++m) wq[m] = pl_weights2(xs[2 * m], xs[2 * m + 1]);
  unsigned lw[NG], lgw[NG];
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    lw[q] = pl[q];
    lgw[q] = pl[LWW + q];
  }
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    unsigned s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = 4 * q + k;
      if (j < NT) s[k] = (j & 1) ? pl_lerp<1>(tp[j], wq[j / 2]) : pl_lerp<0>(tp[j], wq[j / 2]);
    }
    unsigned pc, pgs;
    pl_gather(s, NT - 4 * q, pc, pgs);
    const int rem = NT - 4 * q;
    const unsigned mask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
    sc = __builtin_amdgcn_sad_u8(lw[q] & mask, pc, sc);
    sg = __builtin_amdgcn_sad_u8(lgw[q] & mask, pgs, sg);
  }
}
// Two window rows of the checkerboard at once (round 6): an even row has NE = (P + 1) / 2 taps, the odd row behind it
// NO = P / 2 -- 6 and 5 for P = 11 -- so both end in a partial group of four (2 taps and 1 tap), and a partial group costs
// what a full one does (byte gathers, two v_sad_u8, masks).  Where the two leftovers fit ONE group they share it: the
// samples are taken in the order [even full groups | odd full groups | even leftover, odd leftover], the reference bytes of
// the shared group come from one v_perm_b32 of the two rows' last dwords (zero where the group is short: no masks).
// P = 11: 3 groups per row pair instead of 4, 20 vector instructions for gathers and SADs instead of 26.
template <int P>
struct PlPairOrder {
  static constexpr int NE = (P + 1) / 2, NO = P / 2, QE = NE / 4, QO = NO / 4, RE = NE % 4, RO = NO % 4, NT = NE + NO;
  static constexpr bool shared = RE > 0 && RO > 0 && RE + RO <= 4;
  int odd[NT], idx[NT];
  constexpr PlPairOrder() : odd{}, idx{} {
    int k = 0;
    for (int j = 0; j < 4 * QE; ++j, ++k) { odd[k] = 0; idx[k] = j; }
    for (int j = 0; j < 4 * QO; ++j, ++k) { odd[k] = 1; idx[k] = j; }
    for (int j = 4 * QE; j < NE; ++j, ++k) { odd[k] = 0; idx[k] = j; }
    for (int j = 4 * QO; j < NO; ++j, ++k) { odd[k] = 1; idx[k] = j; }
  }
  // v_perm_b32 selector of the shared group's reference bytes: RE bytes of the even row's last dword (source 1: bytes
  // 0..3), then RO bytes of the odd row's (source 0: bytes 4..7), then zeros (0x0c)
  static constexpr unsigned sel() {
    unsigned v = 0;
    for (int b = 0; b < 4; ++b) {
      const unsigned s = b < RE ? (unsigned)b : (b < RE + RO ? (unsigned)(4 + b - RE) : 0x0cu);
      v |= s << (8 * b);
    }
    return v;
  }
};
template <int P, int LWW>
__device__ __forceinline__ void pl_rowpair_taps(int xrow_e, int xrow_o, const int (&xoff)[P], const unsigned* ple,
                                                const unsigned* plo, unsigned& sc, unsigned& sg) {
  using O = PlPairOrder<P>;
  constexpr O ord{};
  constexpr int NT = O::NT, NQ = O::QE + O::QO + 1;
  pl_u2 tp[NT];
  unsigned wq[(NT + 1) / 2];
  int xs[NT + 1];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    xs[k] = (ord.odd[k] ? xrow_o : xrow_e) + xoff[2 * ord.idx[k] + ord.odd[k]];
    tp[k] = *(const pl_lds_u2*)(uintptr_t)(((unsigned)xs[k] >> 16) << 3);  // absolute LDS byte address
  }
  xs[NT] = 0;
#pragma unroll
  for (int m = 0; m < (NT + 1) / 2; ++m) wq[m] = pl_weights2(xs[2 * m], xs[2 * m + 1]);
  unsigned ec[O::QE + 1], eg[O::QE + 1], oc[O::QO + 1], og[O::QO + 1];
#pragma unroll
  for (int q = 0; q <= O::QE; ++q) {
    ec[q] = ple[q];
    eg[q] = ple[LWW + q];
  }
#pragma unroll
  for (int q = 0; q <= O::QO; ++q) {
    oc[q] = plo[q];
    og[q] = plo[LWW + q];
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    unsigned s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = 4 * q + k;
      if (j < NT) s[k] = (j & 1) ? pl_lerp<1>(tp[j], wq[j / 2]) : pl_lerp<0>(tp[j], wq[j / 2]);
    }
    unsigned pc, pgs;
    pl_gather(s, NT - 4 * q, pc, pgs);
    unsigned rc, rg;
    if (q < O::QE) {
      rc = ec[q];
      rg = eg[q];
    } else if (q < O::QE + O::QO) {
      rc = oc[q - O::QE];
      rg = og[q - O::QE];
    } else {
      rc = __builtin_amdgcn_perm(oc[O::QO], ec[O::QE], O::sel());
      rg = __builtin_amdgcn_perm(og[O::QO], eg[O::QE], O::sel());
    }
    sc = __builtin_amdgcn_sad_u8(rc, pc, sc);
    sg = __builtin_amdgcn_sad_u8(rg, pgs, sg);
  }
}
template <int P, int LWW>
__device__ __forceinline__ float pl_cost_checker(const PlTile& t, int lx, int ty, int xrel, float a, float b, float z,
                                                 const PlanesParams& pp, float bound = __builtin_inff()) {
  constexpr int h = P / 2;
  const int TR = t.tr;
  constexpr int NE = (P + 1) / 2, NO = P / 2;  // taps of an even / an odd window row
  const int Z = __float2int_rn(z * 65536.0f), A = __float2int_rn(a * 65536.0f), B = __float2int_rn(b * 65536.0f);
  int xrow = ((xrel + t.tgt_entry0 + ty * t.rw) << 16) - Z + A * h + B * h;  // tap (0, 0)
  int stepj = 65536 - A;
  asm volatile("" : "+v"(stepj));
  int xoff[P];
  xoff[0] = 0;
#pragma unroll
  for (int j = 1; j < P; ++j) {
    int tt = xoff[j - 1] + stepj;
    asm volatile("" : "+v"(tt));
    xoff[j] = tt;
  }
  const int rowstep = (t.rw << 16) - B;
  unsigned sc = 0, sg = 0;
  // even window rows start at tile column lx, odd ones at lx + 1: parity plane, shifted copy and dword of both
  const int ie = lx >> 1, io = (lx + 1) >> 1;
  const int refe = (((lx & 1) * 4 + (ie & 3)) * TR + ty) * (2 * LWW) + (ie >> 2);
  const int refo = ((((lx + 1) & 1) * 4 + (io & 3)) * TR + ty) * (2 * LWW) + (io >> 2);
  const unsigned* ple = t.rc + refe;
  const unsigned* plo = t.rc + refo + 2 * LWW;  // window row 1
  constexpr int lww2 = 4 * LWW;                 // two rows further
  if constexpr (PlPairOrder<P>::shared) {
    // row pairs (0, 1), (2, 3), ... share their leftover group; the last (even) row stands alone
#pragma unroll 1
    for (int i = 0; i < P / 2; ++i) {
      // (opaque: otherwise the loop optimiser keeps one running column per tap and spends an add on each per iteration)
      asm volatile("" : "+v"(xrow));
      int xrow_o = xrow + rowstep;
      asm volatile("" : "+v"(xrow_o));
      pl_rowpair_taps<P, LWW>(xrow, xrow_o, xoff, ple, plo, sc, sg);
      xrow = xrow_o + rowstep;
      ple += lww2;
      plo += lww2;
    }
    pl_row_taps<P, NE, 0, LWW>(xrow, xoff, ple, sc, sg);
  } else {
#pragma unroll 1
    for (int i = 0; i < P; i += 2) {
      pl_row_taps<P, NE, 0, LWW>(xrow, xoff, ple, sc, sg);
      xrow += rowstep;
      ple += lww2;
      if (i + 1 < P) {
        pl_row_taps<P, NO, 1, LWW>(xrow, xoff, plo, sc, sg);
        xrow += rowstep;
        plo += lww2;
      }
      // early termination (exact, see pl_cost; -DPL_EARLY_EXIT=1 builds only): after window rows 0..3, 0..5, 0..7
      if (PL_EARLY_EXIT && i >= 2 && i + 4 < P) {
        const float mc = (float)(int)sc * pp.inv_n, mg = (float)(int)sg * pp.inv_n;
        const float part = pp.alpha * fminf(mc, pp.tau_color) + pp.one_minus_alpha * fminf(mg, pp.tau_grad);
        if (!__any(part < bound)) return part;
      }
    }
  }
  const float mc = (float)(int)sc * pp.inv_n, mg = (float)(int)sg * pp.inv_n;
  const float t0 = pp.alpha * fminf(mc, pp.tau_color);
  const float t1 = pp.one_minus_alpha * fminf(mg, pp.tau_grad);
  return t0 + t1;
}
// The red-black stage keeps a pixel's reference window in REGISTERS (round 6): the window is the same for all four
// neighbour candidates, the stage runs four wavefronts per SIMD (two 78 KB blocks per CU) and has the registers, and a
// timing build said what the reads cost it (reference dwords out of a register: 74 -> 67 us per launch, while fewer
// vector instructions and conflict-free target reads gave it nothing: profiles/r06_planes_experiments.txt).  Per row
// pair the three group dwords of each channel in the order pl_rowpair_taps consumes them, then the last row's.
template <int P>
struct PlRefRegs {
  using O = PlPairOrder<P>;
  static constexpr int NP = P / 2, NQ = O::QE + O::QO + 1, NL = (O::NE + 3) / 4;
  unsigned c[NP][NQ], g[NP][NQ], lc[NL], lg[NL];
};
template <int P, int LWW>
__device__ __forceinline__ void pl_ref_load(const PlTile& t, int lx, int ty, PlRefRegs<P>& R) {
  using O = PlPairOrder<P>;
  const int TR = t.tr;
  const int ie = lx >> 1, io = (lx + 1) >> 1;
  const unsigned* ple = t.rc + (((lx & 1) * 4 + (ie & 3)) * TR + ty) * (2 * LWW) + (ie >> 2);
  const unsigned* plo = t.rc + ((((lx + 1) & 1) * 4 + (io & 3)) * TR + ty) * (2 * LWW) + (io >> 2) + 2 * LWW;
#pragma unroll
  for (int i = 0; i < PlRefRegs<P>::NP; ++i) {
    const unsigned* e = ple + 4 * LWW * i;
    const unsigned* o = plo + 4 * LWW * i;
#pragma unroll
    for (int q = 0; q < O::QE; ++q) {
      R.c[i][q] = e[q];
      R.g[i][q] = e[LWW + q];
    }
#pragma unroll
    for (int q = 0; q < O::QO; ++q) {
      R.c[i][O::QE + q] = o[q];
      R.g[i][O::QE + q] = o[LWW + q];
    }
    R.c[i][O::QE + O::QO] = __builtin_amdgcn_perm(o[O::QO], e[O::QE], O::sel());
    R.g[i][O::QE + O::QO] = __builtin_amdgcn_perm(o[LWW + O::QO], e[LWW + O::QE], O::sel());
  }
  const unsigned* e = ple + 4 * LWW * PlRefRegs<P>::NP;
#pragma unroll
  for (int q = 0; q < PlRefRegs<P>::NL; ++q) {
    const int rem = O::NE - 4 * q;
    const unsigned mask = rem >= 4 ? 0xffffffffu : ((1u << (8 * rem)) - 1u);
    R.lc[q] = e[q] & mask;
    R.lg[q] = e[LWW + q] & mask;
  }
}
template <int P>
__device__ __forceinline__ float pl_cost_checker_regs(const PlTile& t, int ty, int xrel, float a, float b, float z,
                                                      const PlanesParams& pp, const PlRefRegs<P>& R) {
  using O = PlPairOrder<P>;
  constexpr O ord{};
  constexpr int h = P / 2, NT = O::NT, NQ = PlRefRegs<P>::NQ, NP = PlRefRegs<P>::NP, NE = O::NE;
  const int Z = __float2int_rn(z * 65536.0f), A = __float2int_rn(a * 65536.0f), B = __float2int_rn(b * 65536.0f);
  int xrow = ((xrel + t.tgt_entry0 + ty * t.rw) << 16) - Z + A * h + B * h;  // tap (0, 0)
  int stepj = 65536 - A;
  asm volatile("" : "+v"(stepj));
  int xoff[P];
  xoff[0] = 0;
#pragma unroll
  for (int j = 1; j < P; ++j) {
    int tt = xoff[j - 1] + stepj;
    asm volatile("" : "+v"(tt));
    xoff[j] = tt;
  }
  const int rowstep = (t.rw << 16) - B;
  unsigned sc = 0, sg = 0;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    asm volatile("" : "+v"(xrow));  // (one row base per pair: see pl_cost_checker)
    int xrow_o = xrow + rowstep;
    asm volatile("" : "+v"(xrow_o));
    pl_u2 tp[NT];
    unsigned wq[(NT + 1) / 2];
    int xs[NT + 1];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      xs[k] = (ord.odd[k] ? xrow_o : xrow) + xoff[2 * ord.idx[k] + ord.odd[k]];
      tp[k] = *(const pl_lds_u2*)(uintptr_t)(((unsigned)xs[k] >> 16) << 3);
    }
    xs[NT] = 0;
#pragma unroll
    for (int m = 0; m < (NT + 1) / 2; ++m) wq[m] = pl_weights2(xs[2 * m], xs[2 * m + 1]);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      unsigned s[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = 4 * q + k;
        if (j < NT) s[k] = (j & 1) ? pl_lerp<1>(tp[j], wq[j / 2]) : pl_lerp<0>(tp[j], wq[j / 2]);
      }
      unsigned pc, pgs;
      pl_gather(s, NT - 4 * q, pc, pgs);
      sc = __builtin_amdgcn_sad_u8(R.c[i][q], pc, sc);
      sg = __builtin_amdgcn_sad_u8(R.g[i][q], pgs, sg);
    }
    xrow = xrow_o + rowstep;
    // (no scheduling barrier between the unrolled pairs: with one the kernel spills 12 registers at the 128 allowed)
  }
  {  // the last (even) row
    pl_u2 tp[NE];
    unsigned wq[(NE + 1) / 2];
    int xs[NE + 1];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      xs[k] = xrow + xoff[2 * k];
      tp[k] = *(const pl_lds_u2*)(uintptr_t)(((unsigned)xs[k] >> 16) << 3);
    }
    xs[NE] = 0;
#pragma unroll
    for (int m = 0; m < (NE + 1) / 2; ++m) wq[m] = pl_weights2(xs[2 * m], xs[2 * m + 1]);
#pragma unroll
    for (int q = 0; q < PlRefRegs<P>::NL; ++q) {
      unsigned s[4] = {0, 0, 0, 0};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = 4 * q + k;
        if (j < NE) s[k] = (j & 1) ? pl_lerp<1>(tp[j], wq[j / 2]) : pl_lerp<0>(tp[j], wq[j / 2]);
      }
      unsigned pc, pgs;
      pl_gather(s, NE - 4 * q, pc, pgs);
      sc = __builtin_amdgcn_sad_u8(R.lc[q], pc, sc);
      sg = __builtin_amdgcn_sad_u8(R.lg[q], pgs, sg);
    }
  }
  const float mc = (float)(int)sc * pp.inv_n, mg = (float)(int)sg * pp.inv_n;
  const float t0 = pp.alpha * fminf(mc, pp.tau_color);
  const float t1 = pp.one_minus_alpha * fminf(mg, pp.tau_grad);
  return t0 + t1;
}
template <int P, int WIN, int LWW>
__device__ __forceinline__ float pl_cost_w(const PlTile& t, int lx, int ty, int xrel, float a, float b, float z,
                                           const PlanesParams& pp, float bound = __builtin_inff()) {
  if constexpr (WIN == 1) return pl_cost_checker<P, LWW>(t, lx, ty, xrel, a, b, z, pp, bound);
  else return pl_cost<P, LWW>(t, lx, ty, xrel, a, b, z, pp, bound);
}

// Pixel state in registers + the candidate rule (oracle: offer()).
struct PlPix {
  float a, b, z, c;
};
template <int P, typename ST, int WIN, int LWW>
__device__ __forceinline__ void pl_offer(const PlTile& t, int lx, int ty, int xrel, int x, bool on, float ca, float cb,
                                         float cz, PlPix& px, const PlanesParams& pp) {
  ca = pl_quant<ST>(ca);
  cb = pl_quant<ST>(cb);
  cz = pl_quant<ST>(cz);
  const float zmax = fminf((float)pp.max_disp, (float)x);
  const bool need = on && cz >= 0.0f && cz <= zmax && !(ca == px.a && cb == px.b && cz == px.z);
  if (!__any(need)) return;  // wave-uniform skip
  // lanes without a candidate evaluate a harmless plane (their result is discarded)
  const float c = pl_quant<ST>(pl_cost_w<P, WIN, LWW>(t, lx, ty, xrel, need ? ca : 0.f, need ? cb : 0.f, need ? cz : 0.f, pp,
                                                 need ? px.c : -1.0f));
  if (need && c < px.c) {
    px.a = ca;
    px.b = cb;
    px.z = cz;
    px.c = c;
  }
}

// pl_offer with the reference window in registers (red-black stage, checkerboard windows whose row pairs share a group)
template <int P, typename ST>
__device__ __forceinline__ void pl_offer_regs(const PlTile& t, int ty, int xrel, int x, bool on, float ca, float cb, float cz,
                                              PlPix& px, const PlanesParams& pp, const PlRefRegs<P>& R) {
  ca = pl_quant<ST>(ca);
  cb = pl_quant<ST>(cb);
  cz = pl_quant<ST>(cz);
  const float zmax = fminf((float)pp.max_disp, (float)x);
  const bool need = on && cz >= 0.0f && cz <= zmax && !(ca == px.a && cb == px.b && cz == px.z);
  if (!__any(need)) return;  // wave-uniform skip
  const float c = pl_quant<ST>(pl_cost_checker_regs<P>(t, ty, xrel, need ? ca : 0.f, need ? cb : 0.f, need ? cz : 0.f, pp, R));
  if (need && c < px.c) {
    px.a = ca;
    px.b = cb;
    px.z = cz;
    px.c = c;
  }
}

struct PlArgs {
  int stage;
  int arg;         // SPATIAL: colour + 2 * iteration; VIEW: the view; REFINE / VIEW_REFINE: iteration
  int view_fixed;  // >= 0: blockIdx.z = pair, this view; -1: blockIdx.z = pair * n_views + view
  float refine_amp;
  const float* seed_l;  // INIT: tightly packed [n][rows][cols] seed maps in left / right image coordinates, or null
  const float* seed_r;
  // INIT: bit v set = view v's seeds are in the scalar engine's disparity plane of (pair, v) (pitched, view 1
  // already mirrored): where the device SparseInit leaves them
  int seed_in_disp;
  int dbg;  // tuning build only (PM_PLANES_DBG): bit 0 no window evaluation, bit 1 no tile fill -- timing experiments
};

// grid = (ceil(cols / TW), ceil(rows / tile height), slots), block = 64 x tile height, dynamic LDS = pl_lds_bytes().
// Tile width.  The spatial stage updates one colour: 64 active lanes per row of a 128-wide tile.  The other stages update
// every pixel: 64-wide tiles, one pixel per lane.  -DPL_WIDE_TW=128 gives them 128-wide tiles whose lanes take two pixels
// of a row one after the other (half as many blocks per launch, the 164 extra target columns of a tile row amortised
// over 128 pixels); tried for the last, partly filled round of blocks of a launch (900 blocks on 512 slots instead of
// 1800 on 768) and dropped: 241 -> 231 pairs/s -- two blocks per CU instead of three leave 4 wavefronts per SIMD
// instead of 6, which costs more than the tail gains (bit-identical either way).
#ifndef PL_WIDE_TW
#define PL_WIDE_TW 64
#endif
__host__ __device__ constexpr int pl_tile_w(int stage) { return stage == PL_SPATIAL ? 128 : PL_WIDE_TW; }
__host__ __device__ constexpr int pl_ref_row_dwords(int LW, int win) {
  // dwords per reference row (+1: a shifted copy reads 3 bytes further); checkerboard: per column-parity half
  return win == 1 ? ((LW + 1) / 2 + 3) / 4 + 1 : (LW + 3) / 4 + 1;
}
template <int P, int STAGE, typename ST, int WIN>
// (second launch bound = wavefronts per SIMD the register budget must allow: two of the red-black stage's 78 KB blocks fit
// a CU's LDS -- four wavefronts per SIMD, 128 VGPRs -- and its register-resident reference window uses them)
__global__ void __launch_bounds__(64 * pl_tile_h(STAGE), STAGE == PL_SPATIAL ? 4 : 1) k_planes(PlaneSet ps, PlaneState<ST> st, PlanesParams pp, PlArgs ar) {
  constexpr int h = P / 2;
  constexpr int kPlTileH = pl_tile_h(STAGE), kPlThreads = 64 * kPlTileH;
  constexpr int TW = pl_tile_w(STAGE);  // 64 lanes per tile row either way
  constexpr int SUBS = STAGE == PL_SPATIAL ? 1 : TW / 64;  // pixels of a row per lane
  constexpr int TR = kPlTileH + P - 1;
  constexpr int LW = TW + P - 1;
  constexpr int LWW = pl_ref_row_dwords(LW, WIN);
  constexpr int COPYW = TR * 2 * LWW;            // dwords per shifted copy: rows of LWW colour + LWW gradient dwords
  constexpr int NREF = (WIN == 1 ? 8 : 4) * (COPYW / 2);  // per channel (COPYW / 2 = TR * LWW); 2 NREF is even
  extern __shared__ __attribute__((aligned(16))) unsigned pl_lds[];
  const int rw = TW + 2 * h + pp.max_disp + 2 * pp.margin + 2;
  unsigned* s_rc = pl_lds;
  pl_u2* s_tgt = (pl_u2*)(pl_lds + 2 * NREF);  // 2 NREF is even: 8-byte aligned
  float* s_pl = (float*)(s_tgt + TR * rw);     // SPATIAL only: [3][kPlTileH + 2][TW + 2]

  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * kPlTileH;
  const int pair = ar.view_fixed >= 0 ? (int)blockIdx.z : (int)blockIdx.z / pp.n_views;
  const int view = ar.view_fixed >= 0 ? ar.view_fixed : (int)blockIdx.z - pair * pp.n_views;
  const int rows = ps.rows, cols = ps.cols, pitch = ps.pitch;
  const int y = y0 + ty;

  // images of this view: reference / target packed planes (colour | gradient << 8)
  const int iref = view == 0 ? 0 : 3, itgt = view == 0 ? 1 : 2;
  const uint16_t* refpk = ps.pk16 + ((size_t)pair * 4 + iref) * ps.plane;
  const uint16_t* tgtpk = ps.pk16 + ((size_t)pair * 4 + itgt) * ps.plane;

  // ---- fill: wavefront w takes tile rows w, w + 8, ...; lanes walk along the row (coalesced u16 reads).  Every load of
  // a thread is issued before its first store: the fill is a chain of memory latencies, not of bytes (timed with the
  // window evaluation switched off, it was 34 of a spatial launch's 96 us when each element waited for its own load) ---
  {
    uint8_t* rc8 = (uint8_t*)s_rc;
    uint8_t* rg8 = rc8 + 4 * LWW;  // the gradient half of every row
    const int ry0 = y0 - h, lx0 = x0 - h;
    const int xs_lo = x0 - h - pp.max_disp - pp.margin;
    constexpr int NW = kPlThreads / 64;          // wavefronts
    constexpr int NRR = (TR + NW - 1) / NW;      // tile rows per wavefront
    constexpr int NCM = (LW + 63) / 64;          // reference columns per lane
    const uint16_t* rrow[NRR];
    const uint16_t* trw[NRR];
#pragma unroll
    for (int k = 0; k < NRR; ++k) {  // (a row beyond the tile reads the last image row: loaded, never stored)
      const size_t gy = (size_t)min(max(ry0 + ty + NW * k, 0), rows - 1);
      rrow[k] = refpk + gy * pitch;
      trw[k] = tgtpk + gy * pitch;
    }
#ifdef PM_TUNING
    if (!(ar.dbg & 2))
#endif
    {
      unsigned rpk[NRR][NCM];
#pragma unroll
      for (int k = 0; k < NRR; ++k)
#pragma unroll
        for (int m = 0; m < NCM; ++m) rpk[k][m] = rrow[k][min(max(lx0 + tx + 64 * m, 0), cols - 1)];
      // the first target columns are on their way while the reference bytes are stored
      auto store_ref = [&]() {
#ifdef PM_TUNING
        if (ar.dbg & 4) return;  // timing: no reference stores
#endif
#pragma unroll
        for (int k = 0; k < NRR; ++k) {
          const int rr = ty + NW * k;
#pragma unroll
          for (int m = 0; m < NCM; ++m) {
            const int rcc = tx + 64 * m;
            if (rr < TR && rcc < LW) {
              const unsigned pk = rpk[k][m];
              // byte cc of the row goes to byte cc - s of copy s (s = 0..3); checkerboard: byte cc >> 1 of its parity's half
              const int par = WIN == 1 ? (rcc & 1) : 0, ci = WIN == 1 ? (rcc >> 1) : rcc;
#pragma unroll
              for (int sft = 0; sft < 4; ++sft)
                if (ci >= sft) {
                  const int o = 4 * ((par * 4 + sft) * COPYW + rr * 2 * LWW) + ci - sft;
                  rc8[o] = (uint8_t)(pk & 0xffu);
                  rg8[o] = (uint8_t)(pk >> 8);
                }
            }
          }
        }
      };
      // target entry of column x = {pixel x, pixel x + 1}, each as colour | gradient << 16
      const bool tfast = (cols & 1) == 0 && (reinterpret_cast<uintptr_t>(tgtpk) & 3u) == 0;  // uniform
      constexpr int MAXT = 4;  // 128-column strips of the target tile whose loads are all in flight together
      if (tfast && rw + 1 <= 128 * MAXT) {
        // one dword (two pixels) per lane, row and strip: the entries of both its columns; the pixel behind them comes
        // from the next lane (lane 63: one more u16).  Columns outside the image repeat the border pixel, as the clamp did.
        const int xs_e = xs_lo & ~1, eoff = xs_lo - xs_e;  // even start column; entry index of column c = c - xs_lo
        unsigned dw[MAXT][NRR], nx[MAXT][NRR];
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          const int colA = xs_e + 128 * t + 2 * tx;
          const int ce = min(max(colA, 0), cols - 2);
#pragma unroll
          for (int k = 0; k < NRR; ++k) {
            dw[t][k] = 0u;
            nx[t][k] = 0u;
            if (128 * t < rw + eoff) {  // uniform
              dw[t][k] = *reinterpret_cast<const unsigned*>(trw[k] + ce);
              if (tx == 63) nx[t][k] = trw[k][min(max(colA + 2, 0), cols - 1)];
            }
          }
        }
        store_ref();
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
          if (128 * t >= rw + eoff) break;  // uniform
          const int colA = xs_e + 128 * t + 2 * tx;
#pragma unroll
          for (int k = 0; k < NRR; ++k) {
            unsigned d = dw[t][k];
            if (colA < 0) d = __builtin_amdgcn_perm(0u, d, 0x01000100u);            // pixel 0 twice
            else if (colA > cols - 2) d = __builtin_amdgcn_perm(0u, d, 0x03020302u);  // pixel cols - 1 twice
            const unsigned fromnext = (unsigned)__builtin_amdgcn_update_dpp(0, (int)d, 0x130, 0xF, 0xF, true);  // lane + 1
            const unsigned n = tx == 63 ? nx[t][k] : fromnext;
            const unsigned ea = __builtin_amdgcn_perm(0u, d, 0x0c010c00u);  // pixel A: colour | gradient << 16
            const unsigned eb = __builtin_amdgcn_perm(0u, d, 0x0c030c02u);  // pixel B
            const unsigned en = __builtin_amdgcn_perm(0u, n, 0x0c010c00u);  // the pixel behind B
            const int rr = ty + NW * k, ia = 128 * t + 2 * tx - eoff;
#ifdef PM_TUNING
            if (ar.dbg & 8) continue;  // timing: no target stores
#endif
            if (rr < TR) {
              pl_u2* trow = s_tgt + rr * rw;
              if (ia >= 0 && ia < rw) trow[ia] = pl_entry(ea, eb);
              if (ia + 1 < rw) trow[ia + 1] = pl_entry(eb, en);
            }
          }
        }
      } else {
        for (int c0 = 0; c0 < rw; c0 += 64) {
          const int cc = c0 + tx;
          unsigned p0[NRR], p1[NRR];
#pragma unroll
          for (int k = 0; k < NRR; ++k) {
            p0[k] = trw[k][min(max(xs_lo + cc, 0), cols - 1)];
            p1[k] = trw[k][min(max(xs_lo + cc + 1, 0), cols - 1)];
          }
          if (c0 == 0) store_ref();
#pragma unroll
          for (int k = 0; k < NRR; ++k) {
            const int rr = ty + NW * k;
            if (rr < TR && cc < rw)
              s_tgt[rr * rw + cc] = pl_entry((p0[k] & 0xffu) | ((p0[k] & 0xff00u) << 8), (p1[k] & 0xffu) | ((p1[k] & 0xff00u) << 8));
          }
        }
      }
    }
  }
  ST* const pa = st.arr(pair, view, 0);
  ST* const pb = st.arr(pair, view, 1);
  ST* const pz = st.arr(pair, view, 2);
  ST* const pc = st.arr(pair, view, 3);
  if constexpr (STAGE == PL_SPATIAL) {
    constexpr int PW2 = TW + 2, PH2 = kPlTileH + 2;
    constexpr int NE = (PW2 * PH2 + kPlThreads - 1) / kPlThreads;
    float va[NE], vb[NE], vz[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {  // (an element beyond the halo tile re-reads its last one: loaded, never stored)
      const int e = min(tid + kPlThreads * u, PW2 * PH2 - 1);
      const int rr = e / PW2, cc = e - rr * PW2;
      const int gy = min(max(y0 - 1 + rr, 0), rows - 1), gx = min(max(x0 - 1 + cc, 0), cols - 1);
      const size_t o = st.idx(gx, gy);
      va[u] = pl_load(pa, o);
      vb[u] = pl_load(pb, o);
      vz[u] = pl_load(pz, o);
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + kPlThreads * u;
      if (e < PW2 * PH2) {
        s_pl[e] = va[u];
        s_pl[PW2 * PH2 + e] = vb[u];
        s_pl[2 * PW2 * PH2 + e] = vz[u];
      }
    }
  }
  __syncthreads();
#ifdef PM_TUNING
  if (ar.dbg & 1) return;
#endif

  PlTile t;
  // absolute LDS address of the target tile in 8-byte units (the dynamic LDS block is 16-byte aligned)
  t.tgt_entry0 = (int)((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)pl_lds >> 3) + NREF;
  t.rc = s_rc;
  t.rw = rw;
  t.copy_w = COPYW;
  t.tr = TR;
  const float smax = pp.slope_max;
#pragma unroll 1
  for (int sub = 0; sub < SUBS; ++sub) {
  const int lx = STAGE == PL_SPATIAL ? 2 * tx + ((y + (ar.arg & 1)) & 1) : tx + 64 * sub;
  const int x = x0 + lx;
  const bool on = x < cols && y < rows;
  const int xrel = lx + pp.max_disp + pp.margin;
  const size_t o = st.idx(min(x, cols - 1), min(y, rows - 1));  // (lanes beyond the image never use it)

  PlPix px = {0.f, 0.f, 0.f, 0.f};
  if constexpr (STAGE == PL_INIT) {
    const float zmax = fminf((float)pp.max_disp, (float)x);
    const float* seed = view == 0 ? ar.seed_l : ar.seed_r;
    float s = 0.f;
    if (((ar.seed_in_disp >> view) & 1) && on) {
      s = ps.disp[((size_t)pair * 2 + view) * ps.splane + state_at(x, y, pitch)];
    } else if (seed && on) {
      // the right view's seed map is given in right-image coordinates: view 1 works on the mirrored pair
      const int sx = view == 0 ? x : cols - 1 - x;
      s = seed[(size_t)pair * rows * cols + (size_t)y * cols + sx];
    }
    const float u = pl_u01(pl_rand(pp.seed, PL_RAND_INIT, 0, 0, view, 0, x, y));
    float z = s > 0.0f ? fminf(s, zmax) : u * zmax;
    float a = pp.slope_init * pl_pm1(pl_rand(pp.seed, PL_RAND_INIT, 0, 0, view, 1, x, y));
    float b = pp.slope_init * pl_pm1(pl_rand(pp.seed, PL_RAND_INIT, 0, 0, view, 2, x, y));
    a = pl_quant<ST>(pl_clamp_slope(a, smax));
    b = pl_quant<ST>(pl_clamp_slope(b, smax));
    z = pl_quant<ST>(z);
    if (!(z <= zmax)) z = 0.0f;
    px.a = a;
    px.b = b;
    px.z = z;
    px.c = pl_quant<ST>(pl_cost_w<P, WIN, LWW>(t, lx, ty, xrel, on ? a : 0.f, on ? b : 0.f, on ? z : 0.f, pp));
  } else {
    if (on) {
      px.a = pl_load(pa, o);
      px.b = pl_load(pb, o);
      px.z = pl_load(pz, o);
      px.c = pl_load(pc, o);
    }
    if constexpr (STAGE == PL_SPATIAL) {
      constexpr int PW2 = TW + 2, PH2 = kPlTileH + 2;
      const float* sa = s_pl;
      const float* sb = s_pl + PW2 * PH2;
      const float* sz = s_pl + 2 * PW2 * PH2;
      const int c = (ty + 1) * PW2 + lx + 1;
      // left, right, up, down: the neighbour's plane evaluated at this pixel.  PM_PL_NEIGH_TWO: left + up in the passes of
      // an even iteration, right + down in those of an odd one (uniform for the launch)
      const bool two = pp.neighbours == 1, odd_it = ((ar.arg >> 1) & 1) != 0;
      const bool lu = !two || !odd_it, rd = !two || odd_it;
      if constexpr (WIN == 1 && PlPairOrder<P>::shared && PL_SPATIAL_REF_REGS) {
        PlRefRegs<P> R;
        pl_ref_load<P, LWW>(t, lx, ty, R);
        if (lu) {
          const float na = sa[c - 1], nb = sb[c - 1], nz = sz[c - 1];
          pl_offer_regs<P, ST>(t, ty, xrel, x, on && x > 0, na, nb, nz + na, px, pp, R);
        }
        if (rd) {
          const float na = sa[c + 1], nb = sb[c + 1], nz = sz[c + 1];
          pl_offer_regs<P, ST>(t, ty, xrel, x, on && x < cols - 1, na, nb, nz - na, px, pp, R);
        }
        if (lu) {
          const float na = sa[c - PW2], nb = sb[c - PW2], nz = sz[c - PW2];
          pl_offer_regs<P, ST>(t, ty, xrel, x, on && y > 0, na, nb, nz + nb, px, pp, R);
        }
        if (rd) {
          const float na = sa[c + PW2], nb = sb[c + PW2], nz = sz[c + PW2];
          pl_offer_regs<P, ST>(t, ty, xrel, x, on && y < rows - 1, na, nb, nz - nb, px, pp, R);
        }
      } else {
      if (lu) {
        const float na = sa[c - 1], nb = sb[c - 1], nz = sz[c - 1];
        pl_offer<P, ST, WIN, LWW>(t, lx, ty, xrel, x, on && x > 0, na, nb, nz + na, px, pp);
      }
      if (rd) {
        const float na = sa[c + 1], nb = sb[c + 1], nz = sz[c + 1];
        pl_offer<P, ST, WIN, LWW>(t, lx, ty, xrel, x, on && x < cols - 1, na, nb, nz - na, px, pp);
      }
      if (lu) {
        const float na = sa[c - PW2], nb = sb[c - PW2], nz = sz[c - PW2];
        pl_offer<P, ST, WIN, LWW>(t, lx, ty, xrel, x, on && y > 0, na, nb, nz + nb, px, pp);
      }
      if (rd) {
        const float na = sa[c + PW2], nb = sb[c + PW2], nz = sz[c + PW2];
        pl_offer<P, ST, WIN, LWW>(t, lx, ty, xrel, x, on && y < rows - 1, na, nb, nz - nb, px, pp);
      }
      }
    }
    if constexpr (STAGE == PL_VIEW || STAGE == PL_VIEW_REFINE) {
      const ST* oa = st.arr(pair, 1 - view, 0);
      const ST* ob = st.arr(pair, 1 - view, 1);
      const ST* oz = st.arr(pair, 1 - view, 2);
      const float xo = (float)(cols - 1 - x) + px.z;
      const int xoi = min(max(__float2int_rn(xo), 0), cols - 1);
      float ao = 0.f, bo = 0.f, zo = 0.f;
      if (on) {
        const size_t oo = st.idx(xoi, y);
        ao = pl_load(oa, oo);
        bo = pl_load(ob, oo);
        zo = pl_load(oz, oo);
      }
      const float den = 1.0f - ao;
      const bool ok = on && den >= 0.25f;
      const float dsafe = ok ? den : 1.0f;
      const float na = (-ao) / dsafe, nb = bo / dsafe;
      const float xc = (float)(cols - 1 - xoi) + zo;
      const float dx = (float)x - xc;
      const float tt = na * dx;
      const float nz = zo + tt;
      pl_offer<P, ST, WIN, LWW>(t, lx, ty, xrel, x, ok, pl_clamp_slope(na, smax), pl_clamp_slope(nb, smax), nz, px, pp);
    }
    if constexpr (STAGE == PL_REFINE || STAGE == PL_VIEW_REFINE) {
      float dz = ar.refine_amp;
      for (int k = 0; k < pp.refine_steps; ++k) {
        const float ds = dz * pp.slope_per_disp;
        const float u0 = pl_pm1(pl_rand(pp.seed, PL_RAND_REFINE, ar.arg, k, view, 0, x, y));
        const float u1 = pl_pm1(pl_rand(pp.seed, PL_RAND_REFINE, ar.arg, k, view, 1, x, y));
        const float u2 = pl_pm1(pl_rand(pp.seed, PL_RAND_REFINE, ar.arg, k, view, 2, x, y));
        const float t0 = dz * u0, t1 = ds * u1, t2 = ds * u2;
        const float nz = px.z + t0, na = px.a + t1, nb = px.b + t2;
        pl_offer<P, ST, WIN, LWW>(t, lx, ty, xrel, x, on, pl_clamp_slope(na, smax), pl_clamp_slope(nb, smax), nz, px, pp);
        dz = dz * 0.5f;
      }
    }
  }
  if (on) {
    pl_store(pa, o, px.a);
    pl_store(pb, o, px.b);
    pl_store(pz, o, px.z);
    pl_store(pc, o, px.c);
  }
  }  // sub
}

template <int STAGE>
inline size_t pl_lds_bytes(int P, const PlanesParams& pp) {
  const int h = P / 2;
  const int TW = pl_tile_w(STAGE), kPlTileH = pl_tile_h(STAGE);
  const int TR = kPlTileH + P - 1, LW = TW + P - 1;
  const int nref = ((pp.window == 1 ? 8 : 4) * TR * pl_ref_row_dwords(LW, pp.window) + 1) & ~1;
  const int rw = TW + 2 * h + pp.max_disp + 2 * pp.margin + 2;
  size_t words = 2 * (size_t)nref + 2 * (size_t)TR * rw + 4;
  if (STAGE == PL_SPATIAL) words += 3 * (size_t)(kPlTileH + 2) * (TW + 2);
  return words * 4;
}

template <int P, int STAGE, typename ST, int WIN>
inline hipError_t pl_launch_w(const PlaneSet& ps, void* state, const PlanesParams& pp, const PlArgs& ar, int slots,
                              hipStream_t stream) {
  const int TW = pl_tile_w(STAGE), kPlTileH = pl_tile_h(STAGE);
  const size_t lds = pl_lds_bytes<STAGE>(P, pp);
  if (lds > kChainLdsMax) return hipErrorInvalidValue;
  allow_big_lds(k_planes<P, STAGE, ST, WIN>, lds);
  PlaneState<ST> st;
  st.base = (ST*)state;
  st.plane = ps.plane;
  st.half_pitch = ps.pitch / 2;
  const dim3 grid((unsigned)((ps.cols + TW - 1) / TW), (unsigned)((ps.rows + kPlTileH - 1) / kPlTileH), (unsigned)slots);
  hipLaunchKernelGGL((k_planes<P, STAGE, ST, WIN>), grid, dim3(64 * kPlTileH), lds, stream, ps, st, pp, ar);
  return hipGetLastError();
}
template <int P, int STAGE, typename ST>
inline hipError_t pl_launch_t(const PlaneSet& ps, void* state, const PlanesParams& pp, const PlArgs& ar, int slots,
                              hipStream_t stream) {
  return pp.window == 1 ? pl_launch_w<P, STAGE, ST, 1>(ps, state, pp, ar, slots, stream)
                        : pl_launch_w<P, STAGE, ST, 0>(ps, state, pp, ar, slots, stream);
}

template <int P, int STAGE>
inline hipError_t pl_launch_p(const PlaneSet& ps, void* state, bool f16, const PlanesParams& pp, const PlArgs& ar,
                              int slots, hipStream_t stream) {
  return f16 ? pl_launch_t<P, STAGE, _Float16>(ps, state, pp, ar, slots, stream)
             : pl_launch_t<P, STAGE, float>(ps, state, pp, ar, slots, stream);
}

template <int STAGE>
inline hipError_t pl_launch(const PlaneSet& ps, void* state, bool f16, const PlanesParams& pp, const PlArgs& ar,
                            int slots, hipStream_t stream) {
  switch (pp.patch) {
    case 3: return pl_launch_p<3, STAGE>(ps, state, f16, pp, ar, slots, stream);
    case 5: return pl_launch_p<5, STAGE>(ps, state, f16, pp, ar, slots, stream);
    case 7: return pl_launch_p<7, STAGE>(ps, state, f16, pp, ar, slots, stream);
    case 9: return pl_launch_p<9, STAGE>(ps, state, f16, pp, ar, slots, stream);
    case 11: return pl_launch_p<11, STAGE>(ps, state, f16, pp, ar, slots, stream);
    case 13: return pl_launch_p<13, STAGE>(ps, state, f16, pp, ar, slots, stream);
    case 15: return pl_launch_p<15, STAGE>(ps, state, f16, pp, ar, slots, stream);
    default: return hipErrorInvalidValue;
  }
}

// Disparity maps out (view 1 un-mirrored) + left/right consistency mask on the left map
// (the role of MaskOcclusions, patchmatch_gpu.cu:273-295; rule: |dl - dr| > lr_tol).  grid = pixel grid x pairs.
template <typename ST>
__global__ void __launch_bounds__(256) k_planes_finish(PlaneSet ps, PlaneState<ST> st, PlanesParams pp,
                                                       float* __restrict__ out_l, float* __restrict__ out_r,
                                                       size_t out_stride) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, b = blockIdx.z;
  if (x >= ps.cols) return;
  const size_t op = (size_t)ps.rows * out_stride;
  const ST* z0 = st.arr(b, 0, 2);
  float dl = (float)z0[st.idx(x, y)];
  if (pp.n_views > 1) {
    const ST* z1 = st.arr(b, 1, 2);
    const float fx = (float)x - dl;
    const int xt = min(max(__float2int_rn(fx), 0), ps.cols - 1);
    const float dr = (float)z1[st.idx(ps.cols - 1 - xt, y)];
    const float df = dl - dr;
    if (fabsf(df) > pp.lr_tol) dl = 0.0f;
    if (out_r) out_r[(size_t)b * op + (size_t)y * out_stride + x] = (float)z1[st.idx(ps.cols - 1 - x, y)];
  }
  out_l[(size_t)b * op + (size_t)y * out_stride + x] = dl;
}

// state <-> tightly packed f32 host-side planes (pm_planes_read / pm_planes_write)
template <typename ST>
__global__ void __launch_bounds__(256) k_planes_copy(PlaneSet ps, PlaneState<ST> st, int pair, int view, float* buf,
                                                     int to_state) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, k = blockIdx.z;
  if (x >= ps.cols) return;
  ST* p = st.arr(pair, view, k) + st.idx(x, y);
  float* q = buf + ((size_t)k * ps.rows + y) * ps.cols + x;
  if (to_state) *p = (ST)*q;
  else *q = (float)*p;
}

}  // namespace pm
