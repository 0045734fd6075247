// pm_run.hpp -- PM_ENGINE_RUN: directional sweeps that advance a whole adoption run per step.
//
// Measured on the benchmark workload (8 iterations, 11x11): in the first pass of an iteration
// 85-97 % of the pixels adopt their predecessor's value and the adopted value travels in runs of
// 7-50 pixels; in the reverse passes 60-90 % of the steps offer a candidate equal to the pixel's own
// value.  A sweep is therefore mostly "one value walking along the chain until some pixel rejects
// it" (patchmatch.cpp:158-196 applied pixel after pixel, :264-310).
//
// One wavefront owns one chain and, per step, tests ONE candidate value v at up to 64-pw+1
// consecutive positions at once:
//   * lane l computes the window LINE sum of image column (row sweep) / image row (column sweep)
//     number l of the strip: sum over the window's other dimension of the colour and
//     saturated-gradient absolute differences -- for a row sweep these are fully coalesced row reads;
//   * the window sums of position m are lines m .. m+pw-1: a sliding sum across lanes; both sums are
//     integers (< 2^16 each, packed into one register), so regrouping them is exact;
//   * every lane turns its window sums into the cost functor's value and compares with the stored
//     cost of its pixel; a ballot finds the first position that does not continue the run.
// Positions before it adopt v (or already hold it), that position keeps its own value, which becomes
// the next candidate.  The result is bit-identical to the sequential loop.
//
// Exactness notes.  (1) cv::getRectSubPix derives the bilinear weight from fl(fl(x - d) - (pw-1)/2);
// for a fixed d that fraction is the same for all x with x - d in one binade and changes when x - d
// crosses a power of two, so a step only decides positions whose (fraction, integer offset) equal
// the reference position's; the rest wait for the next step.  (2) a candidate with x - d < pw/2 is not
// considered (patchmatch.cpp:186): such a position ends the run without an evaluation.
#pragma once

#include "pm_kernels.hpp"

// Window rows whose loads are in flight together in one run step.  All 11 (full unroll) costs ~76
// VGPRs and halves the resident workgroups; 4 keeps the kernels at <= 64 VGPRs (8 waves per SIMD).
#ifndef PM_RUN_UNROLL
#define PM_RUN_UNROLL 11
#endif
#ifndef PM_PAIR_LOADS
#define PM_PAIR_LOADS 0
#endif

namespace pm {

// lane l <- lane l+1 across the whole wavefront (DPP wave_shl:1, available on gfx9 incl. gfx950);
// lane 63 has no source and receives 0.
__device__ __forceinline__ int wave_shl1(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, false);
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// What one step resolved: positions [i, i + advance) of the chain.  Lane m < advance holds position
// i + m; its new value is `d0` if m == rej_lane (the pixel that refused the candidate and keeps its own
// value) and the candidate otherwise (adopted, or already equal to it); its new cost is `cost` where
// `adopt` is set and `c0` elsewhere.
struct RunStep {
  int advance;    // >= 1, wave-uniform
  int rej_lane;   // -1 if the step ended for another reason (reach of the strip, end, other binade)
  float rej_d0;   // d0 of rej_lane: the next candidate (valid when rej_lane >= 0)
  bool evaluated; // line sums were computed
  // per lane
  bool adopt;
  float d0, c0, cost;
  ptrdiff_t o;    // element offset of this lane's position in the disparity / cost planes
};

// AXIS 0: chain = image row, positions are columns, lines are image columns (row-major planes).
// AXIS 1: chain = image column, positions are rows, lines are image rows (TRANSPOSED planes, so that
//         lane l reading row Y+l is again a coalesced access).
// TPW / TPH > 0 fix the window at compile time (fully unrolled loads: all of a step's reads are in
// flight together); 0 = take it from CostParams.
// One step of candidate `cand` from position index i (sweep order) of `chain`; positions >= n_end are
// out of reach.  din / cin: disparity and cost planes to read.
template <int AXIS, int TPW, int TPH>
__device__ __forceinline__ RunStep run_step(const View& v, const PlaneSet& ps, const CostParams& cp,
                                            const SweepGeom& g, int chain, int i, int n_end, float cand,
                                            const float* __restrict__ din, const float* __restrict__ cin) {
  const int lane = threadIdx.x & (kWave - 1);  // blocks may hold several wavefronts (k_runblk)
  const int pitch = ps.pitch, cols = ps.cols, rows = ps.rows;
  const int pw = TPW > 0 ? TPW : cp.pw, ph = TPH > 0 ? TPH : cp.ph;
  const int half_w = pw / 2, half_h = ph / 2;
  const int win = AXIS == 0 ? pw : ph;  // window extent along the chain
  const int half = win / 2;
  const int nd = kWave - win + 1;       // positions one step can decide
  const int dir = g.dir;
  const float shift = (float)(pw - 1) * 0.5f;
  const int stride = AXIS == 0 ? dir : dir * pitch;
  const ptrdiff_t first = AXIS == 0 ? (ptrdiff_t)chain * pitch + g.s_first : (ptrdiff_t)g.s_first * pitch + chain;

  RunStep st;
  st.rej_lane = -1;
  st.rej_d0 = 0.f;
  st.evaluated = false;
  st.adopt = false;
  st.cost = 0.f;
  const bool inr = (lane < nd) && (i + lane < n_end);
  st.o = first + (ptrdiff_t)(i + lane) * stride;
  st.d0 = inr ? din[st.o] : 0.f;
  st.c0 = inr ? cin[st.o] : 0.f;
  const bool neutral = inr && (st.d0 == cand);
  const unsigned long long need = __ballot(inr && !neutral);
  if (need == 0ull) {  // every position in reach already holds the candidate: it simply walks on
    st.advance = min(nd, n_end - i);
    return st;
  }
  const int r = __ffsll((long long)need) - 1;  // first position that has to decide

  // bilinear parameters of candidate `cand` seen from this lane's pixel
  const int pos = g.s_first + (i + lane) * dir;
  const int px = AXIS == 0 ? pos : chain;
  float cx = (float)px - cand;
  const bool valid = cx >= (float)half_w;
  cx = cx - shift;
  const float fl = floorf(cx);
  const int ipx = (int)fl;
  const float a = cx - fl;
  const int delta = (px - half_w) - ipx;  // left column X pairs with right column X - delta

  const unsigned long long valid_b = __ballot(valid);
  if (!((valid_b >> r) & 1ull)) {  // candidate not admissible at r: r keeps its own value
    st.rej_lane = r;
    st.rej_d0 = readlane_f(st.d0, r);
    st.advance = r + 1;
    return st;
  }
  const float a_r = readlane_f(a, r);
  const int delta_r = __builtin_amdgcn_readlane(delta, r);
  const bool same = valid && (a == a_r) && (delta == delta_r);
  const float ia_r = 1.f - a_r;
  CpuLerp l;
  l.a = a_r;
  l.ia = ia_r;
  l.a11 = __float2int_rn(ia_r * 65536.f);
  l.a12 = __float2int_rn(a_r * 65536.f);
  l.ipx = 0;

  st.evaluated = true;
  // ---- line sums --------------------------------------------------------------------------
  // Row bases are wave-uniform (scalar registers), the lane contributes a 32-bit offset.  The two
  // bilinear taps are adjacent elements and come with one load; when the first tap is the last
  // column its neighbour is read from the (zero-initialised) padding and carries weight 0.
  unsigned sc = 0, sg = 0;
  if (AXIS == 0) {
    // lane l <-> image column X = x_i + dir*(l - half); window of position m = lanes m..m+pw-1
    const int x_i = g.s_first + i * dir;
    const unsigned X = (unsigned)min(max(x_i + dir * (lane - half), 0), cols - 1);
    const unsigned R0 = (unsigned)min(max((int)X - delta_r, 0), cols - 1);
    const unsigned org = (unsigned)((chain - half_h) * pitch);
    const unsigned ol = org + X, orr = org + R0;
#pragma unroll PM_RUN_UNROLL
    for (int t = 0; t < ph; ++t) {
      const unsigned st = (unsigned)(t * pitch);  // uniform
      int r0, r1;
      float g0, g1;
#if PM_PAIR_LOADS
      load_pair_u8(v.tgt8, orr + st, r0, r1);
      load_pair_f32(v.tgtg, (orr + st) * 4u, g0, g1);
#else
      r0 = ld_u8(v.tgt8, orr + st);
      r1 = ld_u8(v.tgt8, orr + st + 1u);
      g0 = ld_f32(v.tgtg, (orr + st) * 4u);
      g1 = ld_f32(v.tgtg, (orr + st) * 4u + 4u);
#endif
      sc = cpu_acc_color(sc, ld_u8(v.ref8, ol + st), r0, r1, l);
      sg = cpu_acc_grad(sg, ld_u8(v.refg8, ol + st), g0, g1, l);
    }
  } else {
    // lane l <-> image row Y = y_i + dir*(l - half); window of position m = lanes m..m+ph-1.
    // Transposed planes: element (x, Y) at x * pitch_t + Y.
    const int pt = ps.pitch_t;
    const int y_i = g.s_first + i * dir;
    const unsigned Y = (unsigned)min(max(y_i + dir * (lane - half), 0), rows - 1);
    const int ipx_r = (chain - half_w) - delta_r;
    int r0 = ld_u8(v.ttgt8, (unsigned)(ipx_r * pt) + Y);
    float g0 = ld_f32(v.ttgtg, ((unsigned)(ipx_r * pt) + Y) * 4u);
#pragma unroll PM_RUN_UNROLL
    for (int t = 0; t < pw; ++t) {
      const unsigned lrow = (unsigned)((chain - half_w + t) * pt);          // uniform
      const unsigned rrow = (unsigned)(min(ipx_r + t + 1, cols - 1) * pt);  // uniform
      const int r1 = ld_u8(v.ttgt8, rrow + Y);
      const float g1 = ld_f32(v.ttgtg, (rrow + Y) * 4u);
      sc = cpu_acc_color(sc, ld_u8(v.tref8, lrow + Y), r0, r1, l);
      sg = cpu_acc_grad(sg, ld_u8(v.trefg8, lrow + Y), g0, g1, l);
      r0 = r1;
      g0 = g1;
    }
  }

  // ---- window sums: W[m] = line[m] + ... + line[m + win - 1]  (both sums < 2^16, packed) ------
  // Horner form: acc <- line + shl1(acc), win-1 times.
  const int line = (int)(sc | (sg << 16));
  int wsum = line;
#pragma unroll
  for (int t = 1; t < win; ++t) wsum = line + wave_shl1(wsum);
  st.cost = cpu_cost_from_sums(wsum & 0xffff, (int)((unsigned)wsum >> 16), cp);

  // ---- decide ----------------------------------------------------------------------------------
  const bool adopt = inr && !neutral && same && (st.cost < st.c0);
  const bool cont = (lane < r) || neutral || adopt;
  const unsigned long long stop = __ballot(!cont);
  const int q = stop ? __ffsll((long long)stop) - 1 : kWave;
  st.adopt = adopt && lane < q;
  st.advance = q;
  if (q < kWave) {
    const unsigned long long bit = 1ull << q;
    const bool q_inr = (__ballot(inr) & bit) != 0ull;
    const bool q_same = (__ballot(same) & bit) != 0ull;
    const bool q_valid = (valid_b & bit) != 0ull;
    if (q_inr && (q_same || !q_valid)) {  // a real rejection: q keeps its value, which walks on
      st.rej_lane = q;
      st.rej_d0 = readlane_f(st.d0, q);
      st.advance = q + 1;
    }
    // else: out of reach (window strip, end, or another binade): same candidate again from q
  }
  return st;
}

__device__ __forceinline__ void run_count(const PlaneSet& ps, int axis, unsigned steps, unsigned evals,
                                          unsigned adopted, unsigned positions) {
  if (threadIdx.x == 0 && ps.counters) {
    const int base = axis * 4;
    atomicAdd(&ps.counters[base + 0], (unsigned long long)steps);
    atomicAdd(&ps.counters[base + 1], (unsigned long long)evals);
    atomicAdd(&ps.counters[base + 2], (unsigned long long)adopted);
    atomicAdd(&ps.counters[base + 3], (unsigned long long)positions);
  }
}

// PM_ENGINE_RUN: one wavefront per chain, in place on buffer `cur`.  grid = (chains, 1, slots), block = 64.
template <int AXIS, int TPW, int TPH>
__global__ void __launch_bounds__(64) k_sweep_run_cpu(PlaneSet ps, CostParams cp, SweepGeom g) {
  const int chain = g.c_lo + blockIdx.x;
  const View v = make_view(ps, blockIdx.z);
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int stride = AXIS == 0 ? g.dir : g.dir * ps.pitch;
  const ptrdiff_t first =
      AXIS == 0 ? (ptrdiff_t)chain * ps.pitch + g.s_first : (ptrdiff_t)g.s_first * ps.pitch + chain;
  float cand = v.disp[first - stride];  // predecessor of the first position: never written by this sweep
  unsigned n_steps = 0, n_evals = 0, n_adopt = 0;
  int i = 0;
  while (i < n) {
    const RunStep st = run_step<AXIS, TPW, TPH>(v, ps, cp, g, chain, i, n, cand, v.disp, v.cost);
    if (st.adopt) {
      v.disp[st.o] = cand;
      v.cost[st.o] = st.cost;
    }
    ++n_steps;
    n_evals += st.evaluated ? 1u : 0u;
    n_adopt += (unsigned)__popcll(__ballot(st.adopt));
    if (st.rej_lane >= 0) cand = st.rej_d0;
    i += st.advance;
  }
  run_count(ps, AXIS, n_steps, n_evals, n_adopt, (unsigned)n);
}

// ---------------------------------------------------------------------------------------------
// PM_ENGINE_RUNSEG: run steps + speculative segments.  The chain is cut into segments; pass 1 sweeps
// all segments at once (S times more wavefronts in flight), each starting from the OLD value of the
// pixel before it (exact for the first segment), reading buffer `cur` and writing every position of
// buffer `cur ^ 1`.  Pass 2 (one wavefront per chain) walks the boundaries in sweep order: where the
// final value of the pixel before a segment differs from the guess, it re-runs run steps from the
// segment start with the true value, overwriting pass 1's output, until a position gets the value
// pass 1 had stored there -- the two trajectories have merged and pass 1's remainder is the truth.
// Sequential walk => every boundary is checked against final data => exactly the sequential sweep,
// for any run length; a run crossing a boundary costs one or two extra steps, not its length.
// ---------------------------------------------------------------------------------------------
template <int AXIS, int TPW, int TPH>
__global__ void __launch_bounds__(64) k_runseg_pass1(PlaneSet ps, CostParams cp, SweepGeom g, int seg_len) {
  const int chain = g.c_lo + blockIdx.x;
  const View v = make_view(ps, blockIdx.z);
  const int lane = threadIdx.x;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int i0 = blockIdx.y * seg_len;
  const int i1 = min(n, i0 + seg_len);
  if (i0 >= n) return;
  const int stride = AXIS == 0 ? g.dir : g.dir * ps.pitch;
  const ptrdiff_t first =
      AXIS == 0 ? (ptrdiff_t)chain * ps.pitch + g.s_first : (ptrdiff_t)g.s_first * ps.pitch + chain;
  float cand = v.disp[first + (ptrdiff_t)(i0 - 1) * stride];
  unsigned n_steps = 0, n_evals = 0, n_adopt = 0;
  int i = i0;
  while (i < i1) {
    const RunStep st = run_step<AXIS, TPW, TPH>(v, ps, cp, g, chain, i, i1, cand, v.disp, v.cost);
    if (lane < st.advance) {
      v.disp_out[st.o] = lane == st.rej_lane ? st.d0 : cand;
      v.cost_out[st.o] = st.adopt ? st.cost : st.c0;
    }
    ++n_steps;
    n_evals += st.evaluated ? 1u : 0u;
    n_adopt += (unsigned)__popcll(__ballot(st.adopt));
    if (st.rej_lane >= 0) cand = st.rej_d0;
    i += st.advance;
  }
  run_count(ps, AXIS, n_steps, n_evals, n_adopt, (unsigned)(i1 - i0));
}

template <int AXIS, int TPW, int TPH>
__global__ void __launch_bounds__(64) k_runseg_pass2(PlaneSet ps, CostParams cp, SweepGeom g, int seg_len) {
  const int chain = g.c_lo + blockIdx.x;
  const View v = make_view(ps, blockIdx.z);
  const int lane = threadIdx.x;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int stride = AXIS == 0 ? g.dir : g.dir * ps.pitch;
  const ptrdiff_t first =
      AXIS == 0 ? (ptrdiff_t)chain * ps.pitch + g.s_first : (ptrdiff_t)g.s_first * ps.pitch + chain;
  unsigned n_steps = 0, n_evals = 0;
  int idx = seg_len;
  while (idx < n) {
    const ptrdiff_t pb = first + (ptrdiff_t)(idx - 1) * stride;
    const float true_in = v.disp_out[pb];  // final: everything before idx is settled
    const float guessed = v.disp[pb];
    if (true_in == guessed) {
      idx += seg_len;
      continue;
    }
    float cand = true_in;
    int i = idx;
    int merged_at = -1;
    while (i < n) {
      const RunStep st = run_step<AXIS, TPW, TPH>(v, ps, cp, g, chain, i, n, cand, v.disp, v.cost);
      ++n_steps;
      n_evals += st.evaluated ? 1u : 0u;
      const bool mine = lane < st.advance;
      const float val = lane == st.rej_lane ? st.d0 : cand;
      const float spec = mine ? v.disp_out[st.o] : 0.f;
      const unsigned long long eq = __ballot(mine && val == spec);
      const int ms = eq ? __ffsll((long long)eq) - 1 : -1;
      const int wlim = ms >= 0 ? ms : st.advance;
      if (lane < wlim) {
        v.disp_out[st.o] = val;
        v.cost_out[st.o] = st.adopt ? st.cost : st.c0;
      }
      if (ms >= 0) {
        merged_at = i + ms;
        break;
      }
      if (st.rej_lane >= 0) cand = st.rej_d0;
      i += st.advance;
    }
    idx = merged_at >= 0 ? (merged_at / seg_len + 1) * seg_len : n;
  }
  run_count(ps, AXIS, n_steps, n_evals, 0u, 0u);
}

// ---------------------------------------------------------------------------------------------
// PM_ENGINE_RUNBLK: one WORKGROUP per chain, one wavefront per segment, fix-up iterated to a fixpoint
// inside the kernel.  Round 1 is the speculative sweep of every segment (as k_runseg_pass1).  In each
// later round a wavefront whose predecessor segment ended on a different value than the one it
// started from re-runs its segment from the start with that value until its state merges with the
// trajectory it had stored (or the segment ends, which may change ITS last value and trigger its
// successor in the next round).  Segment k is final after round k+1, so at most S rounds happen and
// the fixpoint is the unique solution of the recurrence = the sequential sweep; in practice a value
// crosses one or two boundaries and 2-3 rounds suffice.  Last values travel through LDS.
// grid = (chains, 1, slots), block = 64 * S.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxSegWaves = 16;

template <int AXIS, int TPW, int TPH>
__global__ void __launch_bounds__(64 * kMaxSegWaves) k_runblk(PlaneSet ps, CostParams cp, SweepGeom g, int seg_len) {
  __shared__ float s_last[kMaxSegWaves + 1];
  __shared__ int s_changed[2];  // alternating per round: a fast wave resetting the next round's flag cannot race a slow reader
  const int chain = g.c_lo + blockIdx.x;
  const View v = make_view(ps, blockIdx.z);
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int i0 = w * seg_len;
  const int i1 = min(n, i0 + seg_len);
  const bool active = i0 < n;
  const int stride = AXIS == 0 ? g.dir : g.dir * ps.pitch;
  const ptrdiff_t first =
      AXIS == 0 ? (ptrdiff_t)chain * ps.pitch + g.s_first : (ptrdiff_t)g.s_first * ps.pitch + chain;

  // ---- round 1: speculative sweep from the old value of the pixel before the segment --------------
  float in_used = 0.f, lastv = 0.f;
  if (active) {
    in_used = v.disp[first + (ptrdiff_t)(i0 - 1) * stride];
    float cand = in_used;
    int i = i0;
    while (i < i1) {
      const RunStep st = run_step<AXIS, TPW, TPH>(v, ps, cp, g, chain, i, i1, cand, v.disp, v.cost);
      if (lane < st.advance) {
        v.disp_out[st.o] = lane == st.rej_lane ? st.d0 : cand;
        v.cost_out[st.o] = st.adopt ? st.cost : st.c0;
      }
      if (st.rej_lane >= 0) cand = st.rej_d0;  // also the value of the last resolved position
      i += st.advance;
    }
    lastv = cand;  // value of position i1-1: the candidate after the last step (rejected -> its d0, else cand)
    if (lane == 0) s_last[w + 1] = lastv;
  }
  if (threadIdx.x == 0) s_last[0] = in_used;  // wave 0: the true predecessor

  // ---- fix-up rounds ---------------------------------------------------------------------------------
  for (int round = 1; round < nw; ++round) {
    if (threadIdx.x == 0) s_changed[round & 1] = 0;
    __syncthreads();  // s_last of the previous round and all global stores of this workgroup are visible
    bool new_last = false;
    if (active && w > 0) {
      const float in = s_last[w];
      if (in != in_used) {
        in_used = in;
        float cand = in;
        int i = i0;
        bool merged = false;
        while (i < i1) {
          const RunStep st = run_step<AXIS, TPW, TPH>(v, ps, cp, g, chain, i, i1, cand, v.disp, v.cost);
          const bool mine = lane < st.advance;
          const float val = lane == st.rej_lane ? st.d0 : cand;
          const float spec = mine ? v.disp_out[st.o] : 0.f;
          const unsigned long long eq = __ballot(mine && val == spec);
          const int ms = eq ? __ffsll((long long)eq) - 1 : -1;
          const int wlim = ms >= 0 ? ms : st.advance;
          if (lane < wlim) {
            v.disp_out[st.o] = val;
            v.cost_out[st.o] = st.adopt ? st.cost : st.c0;
          }
          if (ms >= 0) {
            merged = true;
            break;
          }
          if (st.rej_lane >= 0) cand = st.rej_d0;
          i += st.advance;
        }
        if (!merged && cand != lastv) {
          lastv = cand;
          new_last = true;
        }
      }
    }
    __syncthreads();  // every wave has read its s_last entry
    if (new_last && lane == 0) {
      s_last[w + 1] = lastv;
      s_changed[round & 1] = 1;
    }
    __syncthreads();
    if (!s_changed[round & 1]) break;
  }
}

template <int AXIS, int TPW, int TPH>
inline void launch_run_k(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int seg_len,
                         int blk_waves, hipStream_t stream) {
  const int chains = g.c_hi - g.c_lo + 1;
  const dim3 block(kWave);
  if (blk_waves > 0) {
    const int n = (g.s_last - g.s_first) * g.dir + 1;
    int nwv = blk_waves > kMaxSegWaves ? kMaxSegWaves : blk_waves;
    int len = (n + nwv - 1) / nwv;
    if (len < 16) len = 16;  // tiny chains: fewer waves do work
    hipLaunchKernelGGL((k_runblk<AXIS, TPW, TPH>), dim3((unsigned)chains, 1, (unsigned)slots), dim3(kWave * nwv), 0,
                       stream, ps, cp, g, len);
    return;
  }
  if (seg_len <= 0) {
    hipLaunchKernelGGL((k_sweep_run_cpu<AXIS, TPW, TPH>), dim3((unsigned)chains, 1, (unsigned)slots), block, 0,
                       stream, ps, cp, g);
    return;
  }
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int nseg = (n + seg_len - 1) / seg_len;
  hipLaunchKernelGGL((k_runseg_pass1<AXIS, TPW, TPH>), dim3((unsigned)chains, (unsigned)nseg, (unsigned)slots),
                     block, 0, stream, ps, cp, g, seg_len);
  if (nseg > 1)
    hipLaunchKernelGGL((k_runseg_pass2<AXIS, TPW, TPH>), dim3((unsigned)chains, 1, (unsigned)slots), block, 0,
                       stream, ps, cp, g, seg_len);
}

template <int AXIS>
inline void launch_run_axis(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int seg_len,
                            int blk_waves, hipStream_t stream) {
  const int sq = cp.pw == cp.ph ? cp.pw : 0;
  switch (sq) {
    case 3: launch_run_k<AXIS, 3, 3>(ps, cp, g, slots, seg_len, blk_waves, stream); break;
    case 5: launch_run_k<AXIS, 5, 5>(ps, cp, g, slots, seg_len, blk_waves, stream); break;
    case 7: launch_run_k<AXIS, 7, 7>(ps, cp, g, slots, seg_len, blk_waves, stream); break;
    case 9: launch_run_k<AXIS, 9, 9>(ps, cp, g, slots, seg_len, blk_waves, stream); break;
    case 11: launch_run_k<AXIS, 11, 11>(ps, cp, g, slots, seg_len, blk_waves, stream); break;
    default: launch_run_k<AXIS, 0, 0>(ps, cp, g, slots, seg_len, blk_waves, stream); break;
  }
}

// blk_waves > 0: PM_ENGINE_RUNBLK (reads buffer ps.cur, result in ps.cur ^ 1).  Otherwise seg_len <= 0:
// in place on buffer ps.cur (PM_ENGINE_RUN); seg_len > 0: PM_ENGINE_RUNSEG (result in ps.cur ^ 1).
inline void launch_sweep_run(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int seg_len,
                             int blk_waves, hipStream_t stream) {
  if (g.axis == 0)
    launch_run_axis<0>(ps, cp, g, slots, seg_len, blk_waves, stream);
  else
    launch_run_axis<1>(ps, cp, g, slots, seg_len, blk_waves, stream);
}

}  // namespace pm
