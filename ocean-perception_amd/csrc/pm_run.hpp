// pm_run.hpp -- PM_ENGINE_RUN / PM_ENGINE_RUNBLK: directional sweeps that advance a whole adoption
// run per step.
//
// Measured on the benchmark workload (8 iterations, 11x11): in the first pass of an iteration
// 85-97 % of the pixels adopt their predecessor's value and the adopted value travels in runs of
// 7-50 pixels; in the reverse passes 60-90 % of the steps offer a candidate equal to the pixel's own
// value.  A sweep is therefore mostly "one value walking along the chain until some pixel rejects
// it" (patchmatch.cpp:158-196 applied pixel after pixel, :264-310).
//
// A wavefront tests ONE candidate value v at up to 64-win consecutive positions per step:
//   * lane l computes the window LINE sum of image column X0+l (row sweep) or image row Y0+l (column
//     sweep, on the transposed planes): the sum over the window's other dimension of the colour and
//     saturated-gradient absolute differences.  Lanes are always in increasing coordinate order, so
//     every load is a coalesced row read, and for a row sweep the second bilinear tap of a lane is
//     the first tap of its right neighbour (DPP wave_shl:1, no second load);
//   * the window sums of a position are `win` adjacent lines: a sliding sum across lanes.  Both sums
//     are integers (< 2^16 each, packed into one register), so regrouping them is exact;
//   * every lane turns its window sums into the cost functor's value and compares with the stored
//     cost of its pixel; a ballot finds the first position (in sweep order) that does not continue
//     the run.  Positions before it adopt v (or already hold it), that position keeps its own value,
//     which becomes the next candidate.
// The disparity / cost values of the chain live in LDS for the whole kernel (loaded and stored once),
// so a step touches global memory only for the image lines.  The result is bit-identical to the
// sequential loop.
//
// Exactness notes.  (1) cv::getRectSubPix derives the bilinear weight from fl(fl(x - d) - (pw-1)/2);
// for a fixed d that fraction is the same for all x with x - d in one binade and changes when x - d
// crosses a power of two, so a step only decides positions whose (fraction, integer offset) equal
// the reference position's; the rest wait for the next step.  (2) a candidate with x - d < pw/2 is not
// considered (patchmatch.cpp:186): such a position ends the run without an evaluation.
#pragma once

#include "pm_kernels.hpp"

namespace pm {

constexpr int kMaxSegWaves = 16;


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
  return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, false);
}
__device__ __forceinline__ float wave_shl1f(float v) {
  return __builtin_bit_cast(float, wave_shl1(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Chain values in LDS: index k <-> position k-1 in sweep order, index 0 = the predecessor of the chain
// (a pixel the sweep never writes).  `in` = state before the sweep, `out` = state after.
struct ChainLds {
  float* din;
  float* cin;
  float* dout;
  float* cout;
};

// What one step resolved: positions [i, i + advance).  A lane with 0 <= mpos < advance holds position
// i + mpos; its new value is `d0` if mpos == rej_pos (the pixel that refused the candidate and keeps
// its own value) and the candidate otherwise; its new cost is `cost` where `adopt` is set, else `c0`.
struct RunStep {
  int advance;    // >= 1, wave-uniform
  int rej_pos;    // -1 if the step ended for another reason (reach of the strip, end, other binade)
  float rej_d0;   // d0 of the rejecting position: the next candidate (valid when rej_pos >= 0)
  // per lane
  int mpos;       // position index relative to i held by this lane (< 0 or >= nd: none)
  bool adopt;
  float d0, c0, cost;
};

template <int AXIS, int TPW, int TPH>
__device__ __forceinline__ int run_nd(const CostParams& cp) {
  // A row sweep takes the second bilinear tap from the next lane, so its last lane has no valid line.
  return AXIS == 0 ? kWave - (TPW > 0 ? TPW : cp.pw) : kWave - (TPH > 0 ? TPH : cp.ph) + 1;
}

// One step of candidate `cand` from position index i (sweep order); positions >= n_end are out of reach.
// AXIS 0: chain = image row, lines = image columns.  AXIS 1: chain = image column, lines = image rows.
// TPW / TPH > 0 fix the window at compile time (unrolled loads); 0 = take it from CostParams.
template <int AXIS, int TPW, int TPH>
__device__ __forceinline__ RunStep run_step(const View& v, const PlaneSet& ps, const CostParams& cp,
                                            const SweepGeom& g, int chain, int i, int n_end, float cand,
                                            const float* din, const float* cin) {
  const int lane = threadIdx.x & (kWave - 1);
  const int pitch = ps.pitch, cols = ps.cols, rows = ps.rows;
  const int pw = TPW > 0 ? TPW : cp.pw, ph = TPH > 0 ? TPH : cp.ph;
  const int half_w = pw / 2, half_h = ph / 2;
  const int win = AXIS == 0 ? pw : ph;  // window extent along the chain
  const int half = win / 2;
  const int nd = run_nd<AXIS, TPW, TPH>(cp);  // positions one step can decide
  const int dir = g.dir;
  const float shift = (float)(pw - 1) * 0.5f;

  // lane <-> position: lanes are in increasing image coordinate, positions in sweep order
  RunStep st;
  st.rej_pos = -1;
  st.rej_d0 = 0.f;
  st.adopt = false;
  st.cost = 0.f;
  st.mpos = dir > 0 ? lane : nd - 1 - lane;
  const bool inr = (lane < nd) && (i + st.mpos < n_end);
  st.d0 = inr ? din[i + st.mpos + 1] : 0.f;
  st.c0 = inr ? cin[i + st.mpos + 1] : 0.f;
  const bool neutral = inr && (st.d0 == cand);
  // "first in sweep order": lowest lane for dir > 0, highest lane for dir < 0
  auto first_pos = [&](unsigned long long m) -> int {
    return dir > 0 ? __ffsll((long long)m) - 1 : nd - 1 - (63 - __clzll((long long)m));
  };
  auto lane_of = [&](int m) -> int { return dir > 0 ? m : nd - 1 - m; };

  const unsigned long long need = __ballot(inr && !neutral);
  if (need == 0ull) {  // every position in reach already holds the candidate: it simply walks on
    st.advance = min(nd, n_end - i);
    return st;
  }
  const int r = first_pos(need);  // first position that has to decide
  const int r_lane = lane_of(r);

  // bilinear parameters of candidate `cand` seen from this lane's position
  const int pos = g.s_first + (i + st.mpos) * dir;
  const int px = AXIS == 0 ? pos : chain;
  float cx = (float)px - cand;
  const bool valid = cx >= (float)half_w;
  cx = cx - shift;
  const float fl = floorf(cx);
  const int ipx = (int)fl;
  const float a = cx - fl;
  const int delta = (px - half_w) - ipx;  // left column X pairs with right column X - delta

  const unsigned long long valid_b = __ballot(valid);
  if (!((valid_b >> r_lane) & 1ull)) {  // candidate not admissible at r: r keeps its own value
    st.rej_pos = r;
    st.rej_d0 = readlane_f(st.d0, r_lane);
    st.advance = r + 1;
    return st;
  }
  const float a_r = readlane_f(a, r_lane);
  const int delta_r = __builtin_amdgcn_readlane(delta, r_lane);
  const bool same = valid && (a == a_r) && (delta == delta_r);
  const float ia_r = 1.f - a_r;
  CpuLerp l;
  l.a = a_r;
  l.ia = ia_r;
  l.a11 = __float2int_rn(ia_r * 65536.f);
  l.a12 = __float2int_rn(a_r * 65536.f);
  l.ipx = 0;

  // ---- line sums --------------------------------------------------------------------------
  // first coordinate of the strip: position 0 of the step is at c_i; for dir < 0 the strip is mirrored
  const int c_i = g.s_first + i * dir;
  const int c_base = dir > 0 ? c_i - half : c_i - half - (nd - 1);
  unsigned sc = 0, sg = 0;
  if (AXIS == 0) {
    // lane l <-> image column X = c_base + l.  Right column R0 = X - delta; its neighbour R0 + 1 is
    // lane l+1's R0 (for the last image column the neighbour is that column again, weight 0).
    // R0 is clamped on its own (not derived from the clamped X): past the right image edge it keeps
    // growing up to cols-1, so the last image column still finds its true right neighbour one lane up.
    const unsigned X = (unsigned)min(max(c_base + lane, 0), cols - 1);
    const unsigned R0 = (unsigned)min(max(c_base + lane - delta_r, 0), cols - 1);
    const unsigned org = (unsigned)((chain - half_h) * pitch);
    const unsigned ol = org + X, orr = org + R0;
#pragma unroll
    for (int t = 0; t < ph; ++t) {
      const unsigned ro = (unsigned)(t * pitch);  // uniform
      const int l8 = ld_u8(v.ref8, ol + ro);
      const int lg = ld_u8(v.refg8, ol + ro);
      const int r0 = ld_u8(v.tgt8, orr + ro);
      const float g0 = ld_f32(v.tgtg, (orr + ro) * 4u);
      const int r1 = wave_shl1(r0);
      const float g1 = wave_shl1f(g0);
      sc = cpu_acc_color(sc, l8, r0, r1, l);
      sg = cpu_acc_grad(sg, lg, g0, g1, l);
    }
  } else {
    // lane l <-> image row Y = c_base + l, on the transposed planes: element (x, Y) at x * pitch_t + Y.
    const int pt = ps.pitch_t;
    const unsigned Y = (unsigned)min(max(c_base + lane, 0), rows - 1);
    const int ipx_r = (chain - half_w) - delta_r;
    int r0 = ld_u8(v.ttgt8, (unsigned)(ipx_r * pt) + Y);
    float g0 = ld_f32(v.ttgtg, ((unsigned)(ipx_r * pt) + Y) * 4u);
#pragma unroll
    for (int t = 0; t < pw; ++t) {
      const unsigned lrow = (unsigned)((chain - half_w + t) * pt);          // uniform
      const unsigned rrow = (unsigned)(min(ipx_r + t + 1, cols - 1) * pt);  // uniform
      const int l8 = ld_u8(v.tref8, lrow + Y);
      const int lg = ld_u8(v.trefg8, lrow + Y);
      const int r1 = ld_u8(v.ttgt8, rrow + Y);
      const float g1 = ld_f32(v.ttgtg, (rrow + Y) * 4u);
      sc = cpu_acc_color(sc, l8, r0, r1, l);
      sg = cpu_acc_grad(sg, lg, g0, g1, l);
      r0 = r1;
      g0 = g1;
    }
  }

  // ---- window sums: W[l] = line[l] + ... + line[l + win - 1]  (both sums < 2^16, packed) ----------
  // Horner form: acc <- line + shl1(acc), win-1 times.  Lane l then holds the window whose first
  // line is l, i.e. position mpos(l).
  const int line = (int)(sc | (sg << 16));
  int wsum = line;
#pragma unroll
  for (int t = 1; t < win; ++t) wsum = line + wave_shl1(wsum);
  st.cost = cpu_cost_from_sums(wsum & 0xffff, (int)((unsigned)wsum >> 16), cp);

  // ---- decide ----------------------------------------------------------------------------------
  const bool adopt = inr && !neutral && same && (st.cost < st.c0);
  const bool cont = (inr && st.mpos < r) || neutral || adopt;
  const unsigned long long lanes_nd = (1ull << nd) - 1ull;       // nd <= 62
  const unsigned long long stop = __ballot(!cont) & lanes_nd;    // lanes >= nd hold no position
  const int q = stop ? first_pos(stop) : nd;
  st.adopt = adopt && st.mpos < q;
  st.advance = q;
  if (q < nd) {
    const unsigned long long bit = 1ull << lane_of(q);
    const bool q_inr = (__ballot(inr) & bit) != 0ull;
    const bool q_same = (__ballot(same) & bit) != 0ull;
    const bool q_valid = (valid_b & bit) != 0ull;
    if (q_inr && (q_same || !q_valid)) {  // a real rejection: q keeps its value, which walks on
      st.rej_pos = q;
      st.rej_d0 = readlane_f(st.d0, lane_of(q));
      st.advance = q + 1;
    }
    // else: out of reach (chain end or another binade): same candidate again from q
  }
  return st;
}

// ---------------------------------------------------------------------------------------------
// One WORKGROUP per chain, one wavefront per segment of the chain, fix-up iterated to a fixpoint.
// Round 1 sweeps every segment speculatively, starting from the OLD value of the pixel before it
// (exact for the first segment, whose predecessor is never swept).  In each later round a wavefront
// whose predecessor segment ended on a different value than the one it started from re-runs its
// segment from the start with that value until its state merges with the trajectory it had stored
// (or the segment ends, which may change ITS last value and trigger its successor in the next
// round).  Segment k is final after round k+1, so at most S rounds happen and the fixpoint is the
// unique solution of the recurrence = the sequential sweep; in practice a value crosses one or two
// boundaries and 2-3 rounds suffice.  With one wavefront (PM_ENGINE_RUN) this is the plain run sweep.
// grid = (chains, 1, slots), block = 64 * S, dynamic LDS = 4 * (n + 1) floats + S + 3 words.
// ---------------------------------------------------------------------------------------------
template <int AXIS, int TPW, int TPH>
__global__ void __launch_bounds__(64 * kMaxSegWaves) k_runblk(PlaneSet ps, CostParams cp, SweepGeom g, int seg_len) {
  extern __shared__ float lds[];
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int n1 = (n + 1 + 3) & ~3;  // padded array length
  ChainLds c;
  c.din = lds;
  c.cin = lds + n1;
  c.dout = lds + 2 * n1;
  c.cout = lds + 3 * n1;
  float* s_last = lds + 4 * n1;                              // [kMaxSegWaves + 1]
  int* s_changed = (int*)(lds + 4 * n1 + kMaxSegWaves + 1);  // [2], alternating per round

  const int chain = g.c_lo + xcd_band_index(blockIdx.x, gridDim.x);
  const View v = make_view(ps, blockIdx.z);
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int nd = run_nd<AXIS, TPW, TPH>(cp);
  const int stride = AXIS == 0 ? g.dir : g.dir * ps.pitch;
  const ptrdiff_t first =
      AXIS == 0 ? (ptrdiff_t)chain * ps.pitch + g.s_first : (ptrdiff_t)g.s_first * ps.pitch + chain;

  // ---- chain values -> LDS (index k <-> position k-1) -------------------------------------------
  for (int k = threadIdx.x; k <= n; k += blockDim.x) {
    const ptrdiff_t o = first + (ptrdiff_t)(k - 1) * stride;
    const float d = v.disp[o];
    const float cc = k > 0 ? v.cost[o] : 0.f;
    c.din[k] = d;
    c.cin[k] = cc;
    c.dout[k] = d;
    c.cout[k] = cc;
  }
  __syncthreads();

  const int i0 = w * seg_len;
  const int i1 = min(n, i0 + seg_len);
  const bool active = i0 < n;

  // ---- round 1: speculative sweep from the old value of the pixel before the segment --------------
  unsigned n_steps = 0, n_fix = 0, n_rounds = 0;  // opt-in work counters (pm_debug_counters)
  float in_used = 0.f, lastv = 0.f;
  if (active) {
    in_used = c.din[i0];
    float cand = in_used;
    int i = i0;
    while (i < i1) {
      const RunStep st = run_step<AXIS, TPW, TPH>(v, ps, cp, g, chain, i, i1, cand, c.din, c.cin);
      ++n_steps;
      if (st.mpos >= 0 && st.mpos < st.advance) {
        c.dout[i + st.mpos + 1] = st.mpos == st.rej_pos ? st.d0 : cand;
        c.cout[i + st.mpos + 1] = st.adopt ? st.cost : st.c0;
      }
      if (st.rej_pos >= 0) cand = st.rej_d0;  // also the value of the last resolved position
      i += st.advance;
    }
    lastv = cand;  // value of position i1-1
    if (lane == 0) s_last[w + 1] = lastv;
  }
  if (threadIdx.x == 0) s_last[0] = in_used;  // wave 0 started from the true predecessor

  // ---- fix-up rounds -------------------------------------------------------------------------------
  for (int round = 1; round < nw; ++round) {
    if (threadIdx.x == 0) s_changed[round & 1] = 0;
    __syncthreads();  // s_last and the LDS chain values of the previous round are visible
    bool new_last = false;
    if (active && w > 0) {
      const float in = s_last[w];
      if (in != in_used) {
        in_used = in;
        float cand = in;
        int i = i0;
        bool merged = false;
        while (i < i1) {
          const RunStep st = run_step<AXIS, TPW, TPH>(v, ps, cp, g, chain, i, i1, cand, c.din, c.cin);
          ++n_fix;
          const bool mine = st.mpos >= 0 && st.mpos < st.advance;
          const float val = st.mpos == st.rej_pos ? st.d0 : cand;
          const float spec = mine ? c.dout[i + st.mpos + 1] : 0.f;
          // first position (sweep order) whose value equals what is stored: the trajectories merged
          const unsigned long long eq = __ballot(mine && val == spec);
          int ms = -1;
          if (eq) ms = g.dir > 0 ? __ffsll((long long)eq) - 1 : nd - 1 - (63 - __clzll((long long)eq));
          const int wlim = ms >= 0 ? ms : st.advance;
          if (st.mpos >= 0 && st.mpos < wlim) {
            c.dout[i + st.mpos + 1] = val;
            c.cout[i + st.mpos + 1] = st.adopt ? st.cost : st.c0;
          }
          if (ms >= 0) {
            merged = true;
            break;
          }
          if (st.rej_pos >= 0) cand = st.rej_d0;
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
    ++n_rounds;
    if (!s_changed[round & 1]) break;
  }
  __syncthreads();
  if (ps.counters && lane == 0) {
    const int base = AXIS * 4;
    atomicAdd(&ps.counters[base + 0], (unsigned long long)n_steps);
    atomicAdd(&ps.counters[base + 1], (unsigned long long)n_fix);
    if (w == 0) atomicAdd(&ps.counters[base + 2], (unsigned long long)n_rounds);
    if (w == 0) atomicAdd(&ps.counters[base + 3], (unsigned long long)n);
  }

  // ---- LDS -> chain values (only what changed) --------------------------------------------------------
  for (int k = threadIdx.x + 1; k <= n; k += blockDim.x) {
    const float d = c.dout[k];
    if (d != c.din[k]) {
      const ptrdiff_t o = first + (ptrdiff_t)(k - 1) * stride;
      v.disp[o] = d;
      v.cost[o] = c.cout[k];
    }
  }
}

template <int AXIS, int TPW, int TPH>
inline void launch_run_k(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                         hipStream_t stream) {
  const int chains = g.c_hi - g.c_lo + 1;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  int nwv = waves < 1 ? 1 : (waves > kMaxSegWaves ? kMaxSegWaves : waves);
  int len = (n + nwv - 1) / nwv;
  if (len < 16) len = 16;  // tiny chains: fewer waves do work
  const int n1 = (n + 1 + 3) & ~3;
  const size_t lds_bytes = sizeof(float) * (4 * (size_t)n1 + kMaxSegWaves + 1 + 2);
  allow_big_lds(k_runblk<AXIS, TPW, TPH>, lds_bytes);
  hipLaunchKernelGGL((k_runblk<AXIS, TPW, TPH>), dim3((unsigned)chains, 1, (unsigned)slots), dim3(kWave * nwv),
                     lds_bytes, stream, ps, cp, g, len);
}

template <int AXIS>
inline void launch_run_axis(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                            hipStream_t stream) {
  const int sq = cp.pw == cp.ph ? cp.pw : 0;
  switch (sq) {
    case 3: launch_run_k<AXIS, 3, 3>(ps, cp, g, slots, waves, stream); break;
    case 5: launch_run_k<AXIS, 5, 5>(ps, cp, g, slots, waves, stream); break;
    case 7: launch_run_k<AXIS, 7, 7>(ps, cp, g, slots, waves, stream); break;
    case 9: launch_run_k<AXIS, 9, 9>(ps, cp, g, slots, waves, stream); break;
    case 11: launch_run_k<AXIS, 11, 11>(ps, cp, g, slots, waves, stream); break;
    default: launch_run_k<AXIS, 0, 0>(ps, cp, g, slots, waves, stream); break;
  }
}

// PM_SEM_CPU only, in place.  waves = 1: PM_ENGINE_RUN; waves > 1: PM_ENGINE_RUNBLK.
inline void launch_sweep_run(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                             hipStream_t stream) {
  if (g.axis == 0)
    launch_run_axis<0>(ps, cp, g, slots, waves, stream);
  else
    launch_run_axis<1>(ps, cp, g, slots, waves, stream);
}

}  // namespace pm
