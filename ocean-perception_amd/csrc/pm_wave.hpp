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

#include "pm_sweep_defs.hpp"

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
  if (!chain_active(ps, slot, chain)) return;
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
    prev = v.disp[state_at(px, py, ps.pitch)];
  }

  for (int base = 0; base < n; base += kWave) {
    const int cnt = min(kWave, n - base);
    const bool mine = lane < cnt;
    const size_t o = chain_at(g.axis, chain, g.s_first + (base + (mine ? lane : 0)) * g.dir, ps.pitch);
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

// ---------------------------------------------------------------------------------------------
// PM_SEM_GPU sweeps: one wavefront per chain, one LANE per segment of the chain.
// The reference's rule (PropagateRow/Col, patchmatch_gpu.cu:156-171, :214-229) evaluates a 5-tap cost
// per step -- too little to spread over lanes, so the lanes of a wavefront take 64 consecutive
// segments of the chain instead.  Round 1: every lane sweeps its segment from the OLD value of the
// pixel before it (exact for segment 0).  Later rounds: a lane whose predecessor segment ended on a
// different value re-runs its segment from the start with that value until the value it produces
// equals what it had stored (its trajectory merged) or the segment ends (then its own last value may
// change and trigger the next lane).  Segment k is final after round k+1; the fixpoint is the unique
// solution of the recurrence, i.e. exactly the sequential sweep (same argument as pm_run.hpp).  The
// chain's disparity / cost values live in LDS; a single wavefront needs no barriers.
// grid = (chains, 1, slots), block = 64, dynamic LDS = 4 * (n + 1) + 65 floats.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_sweep_gpu_lanes(PlaneSet ps, CostParams cp, SweepGeom g) {
  extern __shared__ float lds[];
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int n1 = (n + 1 + 3) & ~3;
  float* din = lds;
  float* cin = lds + n1;
  float* dout = lds + 2 * n1;
  float* cout = lds + 3 * n1;
  float* s_last = lds + 4 * n1;  // [65]

  const int chain = g.c_lo + blockIdx.x;
  if (!chain_active(ps, blockIdx.z, chain)) return;
  const View v = make_view(ps, blockIdx.z);
  const int lane = threadIdx.x;
  for (int k = lane; k <= n; k += kWave) {
    const size_t o = chain_at(g.axis, chain, g.s_first + (k - 1) * g.dir, ps.pitch);
    const float d = v.disp[o];
    const float cc = k > 0 ? v.cost[o] : 0.f;
    din[k] = d;
    cin[k] = cc;
    dout[k] = d;
    cout[k] = cc;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS image is complete (one wavefront: no barrier)

  const int len = (n + kWave - 1) / kWave;
  const int i0 = lane * len, i1 = min(n, i0 + len);
  const bool active = i0 < n;

  auto step = [&](int i, float prev, float& nd, float& nc) {
    const int s = g.s_first + i * g.dir;
    const int x = g.axis == 0 ? s : chain;
    const int y = g.axis == 0 ? chain : s;
    const float d0 = din[i + 1], c0 = cin[i + 1];
    nd = d0;
    nc = c0;
    sweep_step(1, x, 1, d0, c0, prev, nd, nc, [&](float xr) { return gpu_cost_lane(v, ps.pitch, x, y, xr, cp); });
  };

  // ---- round 1 ------------------------------------------------------------------------------------
  float in_used = active ? din[i0] : 0.f;
  float prev = in_used;
  for (int k = 0; k < len; ++k) {
    const int i = i0 + k;
    if (active && i < i1) {
      float nd, nc;
      step(i, prev, nd, nc);
      dout[i + 1] = nd;
      cout[i + 1] = nc;
      prev = nd;
    }
  }
  float lastv = prev;
  if (active) s_last[lane + 1] = lastv;
  if (lane == 0) s_last[0] = din[0];

  // ---- fix-up rounds ---------------------------------------------------------------------------------
  for (int round = 1; round < kWave; ++round) {
    const float in = (active && lane > 0) ? s_last[lane] : in_used;
    const bool redo = active && lane > 0 && in != in_used;
    if (!__any(redo)) break;
    bool merged = false;
    if (redo) {
      in_used = in;
      prev = in;
    }
    for (int k = 0; k < len; ++k) {
      const int i = i0 + k;
      if (redo && !merged && i < i1) {
        float nd, nc;
        step(i, prev, nd, nc);
        if (nd == dout[i + 1]) {
          merged = true;  // same state as the stored trajectory from here on
        } else {
          dout[i + 1] = nd;
          cout[i + 1] = nc;
          prev = nd;
        }
      }
    }
    if (redo && !merged && prev != lastv) {
      lastv = prev;
      s_last[lane + 1] = lastv;
    }
  }

  for (int k = lane + 1; k <= n; k += kWave) {
    const float d = dout[k];
    if (d != din[k]) {
      const size_t o = chain_at(g.axis, chain, g.s_first + (k - 1) * g.dir, ps.pitch);
      v.disp[o] = d;
      v.cost[o] = cout[k];
    }
  }
}

inline void launch_sweep_wave(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots,
                              hipStream_t stream) {
  const int chains = g.c_hi - g.c_lo + 1;
  if (cp.semantics != 0) {
    // PM_SEM_GPU: the 5-tap cost is too small to spread over a wavefront; lanes take chain segments.
    const int n = (g.s_last - g.s_first) * g.dir + 1;
    const int n1 = (n + 1 + 3) & ~3;
    allow_big_lds(k_sweep_gpu_lanes, sizeof(float) * (4 * (size_t)n1 + kWave + 1));
    hipLaunchKernelGGL(k_sweep_gpu_lanes, dim3((unsigned)chains, 1, (unsigned)slots), dim3(kWave),
                       sizeof(float) * (4 * (size_t)n1 + kWave + 1), stream, ps, cp, g);
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
