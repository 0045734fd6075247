// pm_wave.hpp -- PM_ENGINE_WAVE: directional sweeps with one wavefront (64 lanes) per chain.
//
// A sweep is a first-order recurrence along each row (column): the value a pixel ends up with
// depends on what its predecessor ended up with (patchmatch.cpp:158-196 read in pass order), so a
// chain is inherently sequential while different chains are independent.  What can be spread over
// the lanes of a wavefront is the window itself: the pw*ph taps of one cost evaluation go to the
// 64 lanes (ceil(pw*ph/64) taps each) and both integer sums of the cost functor (colour SAD and
// saturated-gradient SAD, patchmatch_test.cpp:30-45) are reduced across the wave with DPP
// row/bank operations -- no LDS, no barriers.  Because the sums are integers the reduction order
// is irrelevant and the result is bit-identical to the sequential CPU loop.
//
// The disparity / cost values of the next 64 positions of the chain are fetched with one coalesced
// (row sweep) or strided (column sweep) load, handed to the deciding code with v_readlane, and the
// changed ones written back with one store per 64 steps.
#pragma once

#include "pm_kernels.hpp"

namespace pm {

// Sum of `v` over the 64 lanes of the wavefront, returned wave-uniform.
// quad_perm / row_half_mirror / row_mirror give every lane its 16-lane row sum, row_bcast15 and
// row_bcast31 (gfx9 DPP) fold the four rows into lane 63.
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast31 -> rows 2,3
  return __builtin_amdgcn_readlane(v, 63);
}

// Taps owned by one lane: K = ceil(pw*ph / 64) (i, j) window coordinates, flattened row-major.
template <int K>
struct LaneTaps {
  int left_off[K];  // (i * pitch + j), added to the window origin of the reference image
  int row_off[K];   // i * pitch
  int j[K];
  bool on[K];
};

template <int K>
__device__ __forceinline__ LaneTaps<K> lane_taps(int lane, int pw, int ph, int pitch) {
  LaneTaps<K> t;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int idx = lane + k * kWave;
    const int i = idx / pw, j = idx - i * pw;
    t.on[k] = idx < pw * ph;
    t.row_off[k] = t.on[k] ? i * pitch : 0;
    t.j[k] = t.on[k] ? j : 0;
    t.left_off[k] = t.row_off[k] + t.j[k];
  }
  return t;
}

// PM_SEM_CPU window cost of disparity d at (x, y), evaluated by the whole wavefront; wave-uniform
// result.  Same validity conditions as cpu_cost_lane.
template <int K>
__device__ __forceinline__ float cpu_cost_wave(const View& v, int pitch, int cols, int x, int y, float d,
                                               const CostParams& cp, const LaneTaps<K>& t) {
  const CpuLerp l = cpu_lerp(x, d, cp.pw);
  const size_t org = (size_t)(y - cp.ph / 2) * pitch;
  const uint8_t* lp = v.ref8 + org + (x - cp.pw / 2);
  const uint8_t* lg = v.refg8 + org + (x - cp.pw / 2);
  const uint8_t* rp = v.tgt8 + org;
  const float* rg = v.tgtg + org;
  int packed = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int c0 = l.ipx + t.j[k];
    const int c1 = min(c0 + 1, cols - 1);
    const int left = lp[t.left_off[k]];
    const int leftg = lg[t.left_off[k]];
    const int r0 = rp[t.row_off[k] + c0], r1 = rp[t.row_off[k] + c1];
    const float g0 = rg[t.row_off[k] + c0], g1 = rg[t.row_off[k] + c1];
    const int sc = cpu_tap_color(left, r0, r1, l);
    const int sg = cpu_tap_grad(leftg, g0, g1, l);
    // both sums stay below 2^16 (<= 225 taps * 255), so they travel through one reduction
    packed += t.on[k] ? (sc | (sg << 16)) : 0;
  }
  const int tot = wave_sum_i32(packed);
  return cpu_cost_from_sums(tot & 0xffff, (int)((unsigned)tot >> 16), cp);
}

// grid = (chains, 1, slots), block = 64 (one wavefront).
template <int K>
__global__ void __launch_bounds__(64) k_sweep_wave_cpu(PlaneSet ps, CostParams cp, SweepGeom g) {
  const int chain = g.c_lo + blockIdx.x;
  const int slot = blockIdx.z;
  const int lane = threadIdx.x;
  const View v = make_view(ps, slot);
  const int half_w = cp.pw / 2;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const LaneTaps<K> taps = lane_taps<K>(lane, cp.pw, cp.ph, ps.pitch);

  // predecessor of the first visited position: a pixel this sweep never writes
  float prev;
  {
    const int px = g.axis == 0 ? g.s_first - g.dir : chain;
    const int py = g.axis == 0 ? chain : g.s_first - g.dir;
    prev = v.disp[(size_t)py * ps.pitch + px];
  }
  const int stride = g.axis == 0 ? g.dir : g.dir * ps.pitch;  // element step along the chain
  const size_t first = g.axis == 0 ? (size_t)chain * ps.pitch + g.s_first : (size_t)g.s_first * ps.pitch + chain;

  for (int base = 0; base < n; base += kWave) {
    const int cnt = min(kWave, n - base);
    const bool mine = lane < cnt;
    const ptrdiff_t o = (ptrdiff_t)first + (ptrdiff_t)(base + lane) * stride;
    const float dreg = mine ? v.disp[o] : 0.f;
    const float creg = mine ? v.cost[o] : 0.f;
    float dnew = dreg, cnew = creg;
    bool changed = false;
    for (int k = 0; k < cnt; ++k) {
      const int s = g.s_first + (base + k) * g.dir;
      const int x = g.axis == 0 ? s : chain;
      const int y = g.axis == 0 ? chain : s;
      const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dreg), k));
      const float c0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, creg), k));
      float nd = d0, nc = c0;
      const bool adopt = sweep_step(0, x, half_w, d0, c0, prev, nd, nc, [&](float cand) {
        return cpu_cost_wave<K>(v, ps.pitch, ps.cols, x, y, cand, cp, taps);
      });
      if (adopt && lane == k) {
        dnew = nd;
        cnew = nc;
        changed = true;
      }
      prev = nd;
    }
    if (changed) {
      v.disp[o] = dnew;
      v.cost[o] = cnew;
    }
  }
}

inline void launch_sweep_wave(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots,
                              hipStream_t stream) {
  const int chains = g.c_hi - g.c_lo + 1;
  if (cp.semantics != 0) {
    // PM_SEM_GPU: the 5-tap cost is too small to spread over a wavefront; one lane per chain.
    hipLaunchKernelGGL(k_sweep_serial, dim3((unsigned)((chains + 63) / 64), 1, (unsigned)slots), dim3(64), 0, stream,
                       ps, cp, g);
    return;
  }
  const dim3 grid((unsigned)chains, 1, (unsigned)slots), block(kWave);
  const int k = (cp.pw * cp.ph + kWave - 1) / kWave;
  switch (k) {
    case 1: hipLaunchKernelGGL(k_sweep_wave_cpu<1>, grid, block, 0, stream, ps, cp, g); break;
    case 2: hipLaunchKernelGGL(k_sweep_wave_cpu<2>, grid, block, 0, stream, ps, cp, g); break;
    case 3: hipLaunchKernelGGL(k_sweep_wave_cpu<3>, grid, block, 0, stream, ps, cp, g); break;
    default: hipLaunchKernelGGL(k_sweep_wave_cpu<4>, grid, block, 0, stream, ps, cp, g); break;
  }
}

}  // namespace pm
