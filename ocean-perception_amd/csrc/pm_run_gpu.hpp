// pm_run_gpu.hpp -- the run step of PM_ENGINE_RUNBLK2 for PM_SEM_GPU (PropagateRow / PropagateCol of
// patchmatch_gpu.cu:156-171, :214-229 with the 5-tap cost L1GradientCost3x3, :72-114).
//
// The 5-tap cost is a float sum in a fixed order,
//     cost(x, y) = (((t(y-1, x-1) + t(y-1, x+1)) + t(y, x)) + t(y+1, x-1)) + t(y+1, x+1),
// so window sums may not be regrouped as in PM_SEM_CPU -- but the TAP VALUES are shareable: the tap of
// reference pixel (row, col) depends only on that pixel and on the position in the other image it is
// compared with.  For one candidate disparity v every unclamped position x of a row samples
// xr = x - v, its left taps sample fl(xr - 1) and its right taps fl(xr + 1); the tap of column X is
// therefore the same for every position that uses it iff those sample positions agree BITWISE with
// S[X] = fl(X - v).  They do unless x - v and x -+ 1 - v fall into different binades, which the step
// checks per position (the samples are compared as floats, no tolerance).
//
// A step therefore works like pm_run2.hpp::run_step2: the lanes of a group hold consecutive columns (row sweep) or rows
// (column sweep, on the transposed planes); every lane computes the three tap values of its column/row the five-tap
// pattern can ask for; a position gathers its five taps from its own lane and its two neighbours (DPP wave_shl /
// wave_shr) and adds them in the reference's order: GS - 2 positions per group and step.  Positions whose candidate is
// clamped (x - v < 1) or whose neighbour samples disagree compute all five taps themselves (gpu_cost_own), in the same
// step.
//
// Rule per position holding (d0, c0), predecessor value v (pm_device.hpp::sweep_step, semantics 1):
//   xr0 = max(x - d0, 1), xr1 = max(x - v, 1);  xr0 == xr1 -> unchanged;
//   else c1 = cost(xr1); c1 < c0 -> (min(v, x - 1), c1); else unchanged.
// The run of v continues through a position iff the position ends up holding v.
//
// Two candidates per step (round 6).  With a 3 x 3 window most positions of a converged map REJECT what their
// predecessor offers: the run of v stops after one or two positions and a step was spent per stop -- 13 steps for a
// segment of 23 positions on the reference's own 376 x 240 call.  But behind a stop that kept its own value the next
// candidate is that position's OLD value, and so on for every further position that keeps its value: what position p
// does when offered d0[p - 1] depends on the state before the sweep only.  k_runblk2 evaluates that for every position
// of the chain in one parallel pass before the first step (`offer`: the cost if p adopts, -1 if it declines -- one
// five-tap evaluation per position, all lanes busy), and a step then runs through the stop AND through all following
// positions that decline, up to and including the first one that adopts its predecessor's old value (whose new value is
// the next step's candidate).  Steps per segment on that call: 12.9 -> 4.9 (rows), 8.7 -> 3.0 (columns); results
// identical by construction -- every decision is the rule above with the value the predecessor really ends up holding.
#pragma once

namespace pm {

// lane l <- lane l-1 across the whole wavefront (DPP wave_shr:1); lane 0 receives 0.
__device__ __forceinline__ float wave_shr1f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}

struct GpuTaps {
  float a, b, c;
};

// Tap values of one reference pixel column/row against sample position s (>= 0) in the other image.
// `lo` = byte offset of the reference pixel, `t0`/`t1` = offsets of the two target pixels.
__device__ __forceinline__ float gpu_tap_at(const uint8_t* ref8, const float* refg, const uint8_t* tgt8,
                                            const float* tgtg, unsigned lo, unsigned t0, unsigned t1, float tcol,
                                            const CostParams& cp) {
  const float il = (float)ld_u8(ref8, lo);
  const float gl = ld_f32(refg, lo * 4u);
  const float r0 = (float)ld_u8(tgt8, t0), r1 = (float)ld_u8(tgt8, t1);
  const float g0 = ld_f32(tgtg, t0 * 4u), g1 = ld_f32(tgtg, t1 * 4u);
  return gpu_tap(il, gl, r0, r1, g0, g1, tcol, cp);
}

// The five-tap cost of ONE position against sample column xr (>= 1), all taps by this lane -- what a thread of the
// reference's kernels computes (L1GradientCost3x3, patchmatch_gpu.cu:72-114; pm_device.hpp::gpu_cost_lane is the same
// sum with 64-bit addressing).  Row sweeps read the image planes, column sweeps the transposed ones, where the lanes'
// consecutive rows are consecutive bytes.
template <int AXIS>
__device__ __forceinline__ float gpu_cost_own(const View& v, const PlaneSet& ps, int px, int py, float xr,
                                              const CostParams& cp) {
  const int dys[5] = {-1, -1, 0, 1, 1};
  const int dxs[5] = {-1, 1, 0, -1, 1};
  float cost = 0.f;
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const float xs = xr + (float)dxs[t];
    const float f0 = floorf(xs);
    const unsigned col0 = (unsigned)(int)f0, col1 = (unsigned)(int)ceilf(xs);
    const float tcol = xs - f0;
    float s;
    if (AXIS == 0) {
      const unsigned row = __umul24((unsigned)(py + dys[t]), (unsigned)ps.pitch);
      s = gpu_tap_at(v.ref8, v.refg, v.tgt8, v.tgtg, row + (unsigned)(px + dxs[t]), row + col0, row + col1, tcol, cp);
    } else {
      const unsigned Y = (unsigned)(py + dys[t]), pt = (unsigned)ps.pitch_t;
      s = gpu_tap_at(v.tref8, v.trefg, v.ttgt8, v.ttgtg, __umul24((unsigned)(px + dxs[t]), pt) + Y,
                     __umul24(col0, pt) + Y, __umul24(col1, pt) + Y, tcol, cp);
    }
    cost = cost + s;
  }
  return cost;
}

// One step of a group: the run of `cand` from position i on, then -- behind a stop that kept its own value -- the
// positions that decline their predecessor's old value, up to and including the first one that adopts it.
template <int GS, int AXIS>
__device__ __forceinline__ RunStep2 run_step2_gpu(const View& v, const PlaneSet& ps, const CostParams& cp,
                                                  const SweepGeom& g, int chain, bool act, int i, int n_end,
                                                  float cand, const float* din, const float* cin,
                                                  const float* offer) {
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & (GS - 1);
  const int gbase = lane & ~(GS - 1);
  const int pitch = ps.pitch, cols = ps.cols, rows = ps.rows;
  constexpr int nd = GS - 2;  // positions <-> lanes 1..GS-2
  const int dir = g.dir;

  RunStep2 st;
  st.mpos = dir > 0 ? gl - 1 : nd - gl;
  const bool has_pos = gl >= 1 && gl <= nd;
  const bool inr = act && has_pos && (i + st.mpos < n_end);
  st.d0 = inr ? din[i + st.mpos + 1] : 0.f;
  st.c0 = inr ? cin[i + st.mpos + 1] : 0.f;
  const float d_pred = inr ? din[i + st.mpos] : 0.f;     // the predecessor's old value and what this position does
  const float cost_b = inr ? offer[i + st.mpos + 1] : -1.f;  // with it (k_runblk2: evaluated once per chain)
  auto first_pos = [&](unsigned m) -> int {  // m != 0, bits of lanes 1..GS-2
    return dir > 0 ? __ffs((int)m) - 2 : nd - (31 - __clz((int)m));
  };
  auto glane_of = [&](int m) -> int { return dir > 0 ? m + 1 : nd - m; };

  const int c_i = g.s_first + i * dir;
  const int c_base = dir > 0 ? c_i - 1 : c_i - nd;  // column / row held by lane 0
  const int pos = c_base + gl;
  const int px = AXIS == 0 ? pos : chain;
  const int py = AXIS == 0 ? chain : pos;
  const float fx = (float)px;
  const float xu = fx - cand;
  const float xr0 = fmaxf(fx - st.d0, 1.f), xr1 = fmaxf(xu, 1.f);
  const bool neutral = inr && (st.d0 == cand);
  const bool same_xr = inr && !neutral && (xr0 == xr1);
  const bool need_eval = inr && !neutral && !same_xr;
  const float newval = fminf(cand, fx - 1.f);

  // ---- sample positions and the bitwise consistency of the neighbours ---------------------------
  // AXIS 0: lane column X samples S = X - v (kept inside the row for addressing; a position only uses a
  // neighbour's tap if the sample it needs equals the neighbour's S).  AXIS 1: x is the chain, the three
  // sample positions are group-uniform.
  bool wide_ok;
  float sL, sC, sR;  // AXIS 1
  float S;           // AXIS 0
  if (AXIS == 0) {
    S = fminf(fmaxf(xu, 0.f), (float)(cols - 1));
    const float s_left = wave_shr1f(S), s_right = wave_shl1f(S);
    wide_ok = (xu >= 1.f) && (S == xu) && (s_left == xu - 1.f) && (s_right == xu + 1.f);
    sL = sC = sR = 0.f;
  } else {
    S = 0.f;
    sL = xu - 1.f;
    sC = xu;
    sR = xu + 1.f;
    wide_ok = (xu >= 1.f) && (sR <= (float)(cols - 1));
  }
  const bool eval_wide = need_eval && wide_ok, eval_own = need_eval && !wide_ok;

  // ---- wide evaluation: tap values per lane, gathered by the positions ---------------------------
  float cost_w = 0.f;
  if (__any(eval_wide)) {
    if (AXIS == 0) {
      const unsigned X = (unsigned)min(max(pos, 0), cols - 1);
      const float f0 = floorf(S);
      const unsigned col0 = (unsigned)(int)f0, col1 = (unsigned)(int)ceilf(S);
      const float tcol = S - f0;
      const unsigned rowm = (unsigned)((chain - 1) * pitch);
      float t[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const unsigned ro = rowm + (unsigned)(k * pitch);
        t[k] = gpu_tap_at(v.ref8, v.refg, v.tgt8, v.tgtg, ro + X, ro + col0, ro + col1, tcol, cp);
      }
      float c = wave_shr1f(t[0]) + wave_shl1f(t[0]);
      c = c + t[1];
      c = c + wave_shr1f(t[2]);
      c = c + wave_shl1f(t[2]);
      cost_w = c;
    } else {
      const int pt = ps.pitch_t;
      const unsigned Y = (unsigned)min(max(pos, 0), rows - 1);
      // a group that does not evaluate may carry meaningless samples: keep its addresses in range
      const float lim = (float)(cols - 1);
      const float s3[3] = {fminf(fmaxf(sL, 0.f), lim), fminf(fmaxf(sC, 0.f), lim), fminf(fmaxf(sR, 0.f), lim)};
      float t[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float f0 = floorf(s3[k]);
        const unsigned col0 = (unsigned)(int)f0, col1 = (unsigned)(int)ceilf(s3[k]);
        const float tcol = s3[k] - f0;
        const unsigned lcol = (unsigned)min(max(chain - 1 + k, 0), cols - 1);
        // column * pitch: both factors below 2^24, the 24-bit multiply is exact and full rate (v_mul_lo_u32 is not)
        t[k] = gpu_tap_at(v.tref8, v.trefg, v.ttgt8, v.ttgtg, __umul24(lcol, (unsigned)pt) + Y,
                          __umul24(col0, (unsigned)pt) + Y, __umul24(col1, (unsigned)pt) + Y, tcol, cp);
      }
      // rows y-1 / y+1 are the neighbour lanes; left / centre / right columns are t[0] / t[1] / t[2]
      float c = wave_shr1f(t[0]) + wave_shr1f(t[2]);
      c = c + t[1];
      c = c + wave_shl1f(t[0]);
      c = c + wave_shl1f(t[2]);
      cost_w = c;
    }
  }

  // ---- positions whose samples cannot be shared (clamped candidate, binade crossing): all five taps by the lane ----
  // (until round 6 such a position was evaluated ALONE, by five lanes, and ended the step: whole chains near the left
  // image border, where every candidate is clamped, went one position per step)
  float cost_o = 0.f;
  if (__any(eval_own)) {
    if (eval_own) cost_o = gpu_cost_own<AXIS>(v, ps, px, py, xr1, cp);
  }

  // ---- decide -----------------------------------------------------------------------------------
  st.cost = wide_ok ? cost_w : cost_o;
  const bool adopt = need_eval && (st.cost < st.c0);
  // the run of `cand` passes a position iff the position ends up holding `cand`; lanes outside the segment stop it
  const bool pass = neutral || (adopt && newval == cand);
  const unsigned lanes_pos = ((1u << (GS - 1)) - 1u) & ~1u;  // lanes 1..GS-2
  const unsigned stop = gballot<GS>(!pass, gbase) & lanes_pos;
  const int q = stop ? first_pos(stop) : nd;
  const int q_gl = glane_of(min(q, nd - 1));
  const unsigned inr_m = gballot<GS>(inr, gbase), adopt_m = gballot<GS>(adopt, gbase);
  const bool q_in = (q < nd) && ((inr_m >> q_gl) & 1u);  // the stop is a position of the segment: decided in this step
  const int n_in = min(nd, n_end - i);                   // positions of this step inside the segment
  // second phase: behind a stop that kept its own value (PropagateRow's rule leaves d0 in place on a reject), every
  // position is offered its predecessor's old value until one adopts
  const bool phase_b = q_in && !((adopt_m >> q_gl) & 1u) && (q + 1 < n_in);
  const unsigned takes_m = gballot<GS>(cost_b >= 0.f && st.mpos > q, gbase) & lanes_pos;

  int advance, rej_pos;
  bool ends_adopting = false;
  if (!q_in) {
    advance = q;  // = n_in unless the group is idle
    rej_pos = -1;
  } else if (phase_b) {
    ends_adopting = takes_m != 0u;
    rej_pos = ends_adopting ? first_pos(takes_m) : n_in - 1;
    advance = rej_pos + 1;
  } else {
    advance = q + 1;
    rej_pos = q;
  }
  const bool in_b = phase_b && st.mpos > q;  // (only read where mpos < advance)
  const bool takes = in_b && ends_adopting && st.mpos == rej_pos;
  // what the position holds after the step
  const float stop_val = takes ? fminf(d_pred, fx - 1.f) : ((!in_b && adopt) ? newval : st.d0);
  const int src_gl = glane_of(max(rej_pos, 0));
  st.rej_d0 = __shfl(stop_val, gbase + src_gl, kWave);
  st.rej_pos = act ? rej_pos : -1;
  st.advance = act ? advance : 0;
  st.adopt = (adopt && !in_b) || takes;
  if (takes) st.cost = cost_b;
  st.val = st.mpos >= q ? stop_val : cand;
  if (!has_pos) st.mpos = -1;
  return st;
}

// offer[k] for chain position k (1-based as din / cin; din[k - 1] = the OLD value of its predecessor): the cost at
// which the position adopts that value, or -1 if it keeps its own (equal value, equal sample column, or no smaller
// cost) -- PropagateRow / PropagateCol's rule for one position (patchmatch_gpu.cu:156-171), evaluated by the lane alone.
template <int AXIS>
__device__ __forceinline__ float run2_gpu_offer(const View& v, const PlaneSet& ps, const CostParams& cp,
                                                const SweepGeom& g, int chain, int k, const float* din,
                                                const float* cin) {
  const int pos = g.s_first + (k - 1) * g.dir;
  const int px = AXIS == 0 ? pos : chain;
  const int py = AXIS == 0 ? chain : pos;
  const float fx = (float)px;
  const float d0 = din[k], bval = din[k - 1];
  const float xr0 = fmaxf(fx - d0, 1.f), xr1 = fmaxf(fx - bval, 1.f);
  if (d0 == bval || xr0 == xr1) return -1.f;
  const float c = gpu_cost_own<AXIS>(v, ps, px, py, xr1, cp);
  return c < cin[k] ? c : -1.f;
}

}  // namespace pm
