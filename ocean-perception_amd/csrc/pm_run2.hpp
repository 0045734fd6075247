// pm_run2.hpp -- PM_ENGINE_RUNBLK2: directional sweeps that advance a whole adoption run per step.
// This header holds the idea, the exactness notes, the shared helpers and the PM_SEM_GPU form of the engine
// (k_runblk2 + pm_run_gpu.hpp); the PM_SEM_CPU form -- the benchmarked one -- is pm_run3.hpp.
//
// Measured on the benchmark workload (8 iterations, 11x11): in the first pass of an iteration 85-97 % of the pixels
// adopt their predecessor's value and the adopted value travels in runs of 7-50 pixels; in the reverse passes
// 60-90 % of the steps offer a candidate equal to the pixel's own value.  A sweep is therefore mostly "one value
// walking along the chain until some pixel rejects it" (patchmatch.cpp:158-196 applied pixel after pixel, :264-310).
//
// The run step.  A group of GS lanes (32 or 16: two or four groups per wavefront, each on its own segment of the
// chain) tests ONE candidate value v at up to GS - win consecutive positions per step:
//   * lane l computes the window LINE sum of one image column (row sweep) or image row (column sweep, on the
//     transposed planes): the sum over the window's other dimension of the colour and saturated-gradient absolute
//     differences.  Lanes hold consecutive coordinates, so every load is a coalesced row read, and for a row sweep
//     the second bilinear tap of a lane is the first tap of its neighbour (DPP, no second load);
//   * the window sums of a position are `win` adjacent lines: a sliding sum across lanes.  Both sums are integers
//     (< 2^16 each, packed into one register), so regrouping them is exact;
//   * every lane turns its window sums into the cost functor's value and compares with the stored cost of its pixel;
//     the first position (in sweep order) that does not continue the run ends the step.  Positions before it adopt v
//     (or already hold it), that position keeps its own value, which becomes the next candidate.
// The disparity / cost values of the chain live in LDS for the whole kernel (loaded and stored once), so a step touches
// global memory only for the image lines.  The result is bit-identical to the sequential loop.
//
// Exactness notes.  (1) cv::getRectSubPix derives the bilinear weight from fl(fl(x - d) - (pw-1)/2); for a fixed d
// that fraction is the same for all x with x - d in one binade and may change when x - d crosses a power of two, so
// a step only decides positions whose parameters equal the first position's; the rest wait for the next step.
// (2) a candidate with x - d < pw/2 is not considered (patchmatch.cpp:186): such a position ends the run without an
// evaluation.
//
// Segments + fix-up.  One WORKGROUP per chain, the chain cut into segments, fix-up iterated to a fixpoint.  Round 1
// sweeps every segment speculatively, starting from the OLD value of the pixel before it (exact for the first
// segment, whose predecessor is never swept).  In each later round a segment whose predecessor ended on a different
// value than the one it started from re-runs from its start with that value until its state merges with the
// trajectory it had stored (or the segment ends, which may change ITS last value and trigger its successor in the
// next round).  Segment k is final after round k + 1, so at most S rounds happen and the fixpoint is the unique
// solution of the recurrence = the sequential sweep; in practice a value crosses one or two boundaries and 2-3
// rounds suffice.
#pragma once

#include "pm_sweep_defs.hpp"

namespace pm {

constexpr int kMaxSegWaves = 16;
constexpr int kLref4Stride = 7;  // dwords per image row of the column sweeps' staged reference bytes (odd; 6 measures the same)

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an XCD and its 4 MiB L2).
// Adjacent chains read almost the same image rows, so chain k of the sweep goes to the block whose
// XCD owns the band around k: block b -> chain (b % 8) * band + b / 8 (bijective for any count).
// Speed only: any placement gives the same result.
__device__ __forceinline__ int xcd_band_index(int b, int nb) {
  const int xcd = b & 7, j = b >> 3;
  const int q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// lane l <- lane l+1 across the whole wavefront (DPP wave_shl:1, gfx9 incl. gfx950); lane 63 has no
// source and receives 0.
__device__ __forceinline__ int wave_shl1(int v) {
  // bound_ctrl: the lane without a source reads 0 and the old value is dead -- no zero-initialising v_mov per use
  return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true);
}
__device__ __forceinline__ float wave_shl1f(float v) {
  return __builtin_bit_cast(float, wave_shl1(__builtin_bit_cast(int, v)));
}

// ---- PM_SEM_GPU --------------------------------------------------------------------------------------------------
struct RunStep2 {
  // group-uniform
  int advance;   // positions resolved (0 if the group is idle)
  int rej_pos;   // -1: none
  float rej_d0;  // value the position rej_pos holds after the step = candidate of the next step
  // per lane
  int mpos;
  bool adopt;
  float d0, c0, cost;
  float val;  // what the position holds after the step (read where 0 <= mpos < advance)
};

// Ballot of this lane's group (GS = 32, 16 or 8 lanes), in the low GS bits.
template <int GS>
__device__ __forceinline__ unsigned gballot(bool p, int gbase) {
  const unsigned long long b = __builtin_amdgcn_ballot_w64(p);
  if constexpr (GS == 32) return gbase ? (unsigned)(b >> 32) : (unsigned)b;
  else return (unsigned)(b >> gbase) & ((1u << GS) - 1u);
}

}  // namespace pm
#include "pm_run_gpu.hpp"
namespace pm {

// One workgroup per chain; wavefront w carries segments (64 / GS) * w ...  Rounds and fix-up as described above, per
// group.  grid = (chains, 1, slots), block = 64 * nw, dynamic LDS = 5 * (n + 1) floats + segments + 3 words.
template <int GS, int AXIS>
__global__ void __launch_bounds__(64 * kMaxSegWaves) k_runblk2(PlaneSet ps, CostParams cp, SweepGeom g, int seg_len) {
  extern __shared__ float lds[];
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int n1 = (n + 1 + 3) & ~3;
  float* din = lds;
  float* cin = lds + n1;
  float* dout = lds + 2 * n1;
  float* cout = lds + 3 * n1;
  float* offer = lds + 4 * n1;   // what a position does with its predecessor's OLD value (pm_run_gpu.hpp::run2_gpu_offer)
  float* s_last = lds + 5 * n1;  // [nseg + 1] last values + [2] change flags
  int* s_changed = (int*)(lds + 5 * n1 + (kWave / GS) * (blockDim.x >> 6) + 1);

  const int chain = g.c_lo + xcd_band_index(blockIdx.x, gridDim.x);
  if (!chain_active(ps, blockIdx.z, chain)) return;  // uniform for the workgroup, before any barrier
#ifdef PM_RUN2_PHASES  // analysis builds (tools/refshape_steps.py --phases): device wall clock (10 ns) per phase of the chain
  const unsigned long long t_start = wall_clock64();
#endif
  const View v = make_view(ps, blockIdx.z);
  const int lane = threadIdx.x & 63;
  constexpr int kPerWave = kWave / GS;
  const int gl = lane & (GS - 1);
  const int gbase = lane & ~(GS - 1);
  const int w = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int nseg = kPerWave * nw;
  const int sidx = kPerWave * w + lane / GS;
  constexpr int nd = GS - 2;
  {
    // the chain's state into LDS, four positions per thread in flight (one memory latency, not four in a row)
    constexpr int U = 4;
    const int bd = blockDim.x;
    for (int k0 = threadIdx.x; k0 <= n; k0 += U * bd) {
      float dd[U], cc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u * bd;
        dd[u] = cc[u] = 0.f;
        if (k <= n) {
          const size_t o = chain_at(AXIS, chain, g.s_first + (k - 1) * g.dir, ps.pitch);
          dd[u] = v.disp[o];
          if (k > 0) cc[u] = v.cost[o];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u * bd;
        if (k <= n) {
          din[k] = dd[u];
          cin[k] = cc[u];
          dout[k] = dd[u];
          cout[k] = cc[u];
        }
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x + 1; k <= n; k += blockDim.x) offer[k] = run2_gpu_offer<AXIS>(v, ps, cp, g, chain, k, din, cin);
  __syncthreads();
#ifdef PM_RUN2_PHASES
  const unsigned long long t_loaded = wall_clock64();
#endif

  const int i0 = sidx * seg_len;
  const int i1 = min(n, i0 + seg_len);
  const bool active = i0 < n;
  unsigned n_steps = 0, n_fix = 0, n_rounds = 0;

  // ---- round 1 ------------------------------------------------------------------------------------
  float in_used = active ? din[i0] : 0.f;
  float cand = in_used;
  {
    int i = i0;
    while (__builtin_amdgcn_ballot_w64(active && i < i1) != 0ull) {
      const bool act = active && i < i1;
      const RunStep2 st = run_step2_gpu<GS, AXIS>(v, ps, cp, g, chain, act, i, i1, cand, din, cin, offer);
      ++n_steps;
      if (st.mpos >= 0 && st.mpos < st.advance) {
        dout[i + st.mpos + 1] = st.val;
        cout[i + st.mpos + 1] = st.adopt ? st.cost : st.c0;
      }
      if (st.rej_pos >= 0) cand = st.rej_d0;
      i += st.advance;
    }
  }
#ifdef PM_RUN2_PHASES
  const unsigned long long t_round1 = wall_clock64();
#endif
  float lastv = cand;
  if (active && gl == 0) s_last[sidx + 1] = lastv;
  if (threadIdx.x == 0) s_last[0] = in_used;

  // ---- fix-up rounds ---------------------------------------------------------------------------------
  for (int round = 1; round < nseg; ++round) {
    if (threadIdx.x == 0) s_changed[round & 1] = 0;
    __syncthreads();
    const float in = (active && sidx > 0) ? s_last[sidx] : in_used;
    bool redo = active && sidx > 0 && (in != in_used);
    bool new_last = false;
    if (__any(redo)) {
      if (redo) in_used = in;
      float c2 = in;
      int i = i0;
      bool merged = false;
      while (__builtin_amdgcn_ballot_w64(redo && !merged && i < i1) != 0ull) {
        const bool act = redo && !merged && i < i1;
        const RunStep2 st = run_step2_gpu<GS, AXIS>(v, ps, cp, g, chain, act, i, i1, c2, din, cin, offer);
        ++n_fix;
        const bool mine = st.mpos >= 0 && st.mpos < st.advance;
        const float val = st.val;
        const float spec = mine ? dout[i + st.mpos + 1] : 0.f;
        const unsigned eq = gballot<GS>(mine && val == spec, gbase);
        int ms = -1;
        if (eq) {  // first merged position in sweep order (lane <-> position mapping of the step function)
          const int lo_lane = __ffs((int)eq) - 1, hi_lane = 31 - __clz((int)eq);
          ms = g.dir > 0 ? lo_lane - 1 : nd - hi_lane;
        }
        const int wlim = ms >= 0 ? ms : st.advance;
        if (st.mpos >= 0 && st.mpos < wlim) {
          dout[i + st.mpos + 1] = val;
          cout[i + st.mpos + 1] = st.adopt ? st.cost : st.c0;
        }
        if (act && ms >= 0) merged = true;
        if (act && st.rej_pos >= 0) c2 = st.rej_d0;
        i += st.advance;
      }
      if (redo && !merged && c2 != lastv) {
        lastv = c2;
        new_last = true;
      }
    }
    __syncthreads();
    if (new_last && gl == 0) {
      s_last[sidx + 1] = lastv;
      s_changed[round & 1] = 1;
    }
    __syncthreads();
    ++n_rounds;
    if (!s_changed[round & 1]) break;
  }
  __syncthreads();
  if (ps.counters && lane == 0) {
    const int base = AXIS * 4;
#ifndef PM_RUN2_PHASES
    atomicAdd(&ps.counters[base + 0], (unsigned long long)n_steps);
    atomicAdd(&ps.counters[base + 1], (unsigned long long)n_fix);
    if (w == 0) atomicAdd(&ps.counters[base + 2], (unsigned long long)n_rounds);
    if (w == 0) atomicAdd(&ps.counters[base + 3], (unsigned long long)n);
#endif
  }

#ifdef PM_RUN2_PHASES
  const unsigned long long t_fixed = wall_clock64();
#endif
  for (int k = threadIdx.x + 1; k <= n; k += blockDim.x) {
    const float d = dout[k];
    if (d != din[k]) {
      const size_t o = chain_at(AXIS, chain, g.s_first + (k - 1) * g.dir, ps.pitch);
      v.disp[o] = d;
      v.cost[o] = cout[k];
    }
  }
#ifdef PM_RUN2_PHASES
  if (ps.counters && threadIdx.x == 0) {
    const int base = AXIS * 4;
    const unsigned long long ph[4] = {t_loaded - t_start, t_round1 - t_loaded, t_fixed - t_round1, wall_clock64() - t_fixed};
    for (int e = 0; e < 4; ++e) {
      if (PM_RUN2_PHASES == 2) atomicMax(&ps.counters[base + e], ph[e]);
      else atomicAdd(&ps.counters[base + e], ph[e]);
    }
    if (PM_RUN2_PHASES == 3 && ph[0] + ph[1] + ph[2] + ph[3] > 1800)  // chains slower than 18 us, one line each
      printf("slow chain: axis %d dir %d chain %d view %d: load %llu round1 %llu fixup %llu wb %llu ticks; wave 0: %u + %u steps, %u rounds\n",
             AXIS, g.dir, chain, (int)blockIdx.z, ph[0], ph[1], ph[2], ph[3], n_steps, n_fix, n_rounds);
  }
#endif
}

template <int GS, int AXIS>
inline void launch_run2_k(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                          hipStream_t stream) {
  const int chains = g.c_hi - g.c_lo + 1;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int nwv = waves < 1 ? 1 : (waves > kMaxSegWaves ? kMaxSegWaves : waves);
  const int per_block = (kWave / GS) * nwv;
  int len = (n + per_block - 1) / per_block;
  if (len < 8) len = 8;
  const int n1 = (n + 1 + 3) & ~3;
  const size_t lds_bytes = sizeof(float) * (5 * (size_t)n1 + per_block + 1 + 2);
  allow_big_lds(k_runblk2<GS, AXIS>, lds_bytes);
  hipLaunchKernelGGL((k_runblk2<GS, AXIS>), dim3((unsigned)chains, 1, (unsigned)slots), dim3(kWave * nwv), lds_bytes,
                     stream, ps, cp, g, len);
}

// PM_SEM_GPU, in place.  group = lanes per chain segment: 32, 16 or 8 (its window is 3 lanes).
inline void launch_sweep_run2(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                              int group, hipStream_t stream) {
  if (g.axis == 0) {
    if (group <= 8) launch_run2_k<8, 0>(ps, cp, g, slots, waves, stream);
    else if (group <= 16) launch_run2_k<16, 0>(ps, cp, g, slots, waves, stream);
    else launch_run2_k<32, 0>(ps, cp, g, slots, waves, stream);
  } else {
    if (group <= 8) launch_run2_k<8, 1>(ps, cp, g, slots, waves, stream);
    else if (group <= 16) launch_run2_k<16, 1>(ps, cp, g, slots, waves, stream);
    else launch_run2_k<32, 1>(ps, cp, g, slots, waves, stream);
  }
}

}  // namespace pm
