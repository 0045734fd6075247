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
// A step therefore works like pm_run2.hpp::run_step2: the 32 lanes of a group hold 32 consecutive
// columns (row sweep) or rows (column sweep, on the transposed planes); every lane computes the three
// tap values of its column/row the five-tap pattern can ask for; a position gathers its five taps from
// its own lane and its two neighbours (DPP wave_shl / wave_shr) and adds them in the reference's order.
// 30 positions per group and step.  Positions whose candidate is clamped (x - v < 1) or whose neighbour
// samples disagree are evaluated alone, exactly like k_sweep_gpu_lanes does (slow path: lanes 0-4 of the
// group take one tap each).
//
// Rule per position holding (d0, c0), predecessor value v (pm_device.hpp::sweep_step, semantics 1):
//   xr0 = max(x - d0, 1), xr1 = max(x - v, 1);  xr0 == xr1 -> unchanged;
//   else c1 = cost(xr1); c1 < c0 -> (min(v, x - 1), c1); else unchanged.
// The run of v continues through a position iff the position ends up holding v.
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

template <int GS, int AXIS>
__device__ __forceinline__ RunStep2 run_step2_gpu(const View& v, const PlaneSet& ps, const CostParams& cp,
                                                  const SweepGeom& g, int chain, bool act, int i, int n_end,
                                                  float cand, const float* din, const float* cin) {
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

  const unsigned nonneutral = gballot<GS>(inr && !neutral, gbase);
  const bool has_need = nonneutral != 0u;
  const int r = has_need ? first_pos(nonneutral) : 0;
  const int r_gl = glane_of(r);
  const unsigned need_m = gballot<GS>(need_eval, gbase);
  const bool r_eval = has_need && ((need_m >> r_gl) & 1u);

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
  const unsigned wide_m = gballot<GS>(wide_ok, gbase);
  const bool r_wide = r_eval && ((wide_m >> r_gl) & 1u);
  const bool r_slow = r_eval && !r_wide;

  // ---- wide evaluation: tap values per lane, gathered by the positions ---------------------------
  float cost_w = 0.f;
  if (__any(r_wide)) {
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

  // ---- slow evaluation of position r alone: lanes 0..4 of the group take one tap each ------------
  float cost_s = 0.f;
  if (__any(r_slow)) {
    const int src = gbase + r_gl;
    const int x_r = __shfl(px, src, kWave), y_r = __shfl(py, src, kWave);
    const float xr_r = __shfl(xr1, src, kWave);
    const int tt = min(gl, 4);
    const int dy = tt < 2 ? -1 : (tt == 2 ? 0 : 1);
    const int dx = tt == 2 ? 0 : ((tt == 0 || tt == 3) ? -1 : 1);
    float s = 0.f;
    if (r_slow) {
      const float xs = xr_r + (float)dx;
      const float f0 = floorf(xs);
      const int col0 = (int)f0, col1 = (int)ceilf(xs);
      const float tcol = xs - f0;
      const unsigned lrow = __umul24((unsigned)(y_r + dy), (unsigned)pitch);  // 32-bit offsets: no 64-bit multiply-adds
      s = gpu_tap_at(v.ref8, v.refg, v.tgt8, v.tgtg, lrow + (unsigned)(x_r + dx), lrow + (unsigned)col0,
                     lrow + (unsigned)col1, tcol, cp);
    }
    float c = 0.f + __shfl(s, gbase + 0, kWave);
    c = c + __shfl(s, gbase + 1, kWave);
    c = c + __shfl(s, gbase + 2, kWave);
    c = c + __shfl(s, gbase + 3, kWave);
    c = c + __shfl(s, gbase + 4, kWave);
    cost_s = c;
  }

  // ---- decide -----------------------------------------------------------------------------------
  const bool is_r = has_pos && st.mpos == r;
  const bool evald = need_eval && (r_wide ? wide_ok : (r_slow && is_r));
  st.cost = r_wide ? cost_w : cost_s;
  const bool adopt = evald && (st.cost < st.c0);
  // the run of `cand` passes a position iff the position ends up holding `cand`
  const bool pass = neutral || (adopt && newval == cand);
  const bool before_r = inr && st.mpos < r;
  // slow path: the step ends at r whatever the outcome (the next step continues from r + 1)
  const bool cont = before_r || (pass && !(r_slow && is_r));
  const unsigned lanes_pos = ((1u << (GS - 1)) - 1u) & ~1u;  // lanes 1..GS-2
  const unsigned stop = gballot<GS>(!cont, gbase) & lanes_pos;
  const int q = stop ? first_pos(stop) : nd;
  const int q_gl = glane_of(min(q, nd - 1));
  const unsigned inr_m = gballot<GS>(inr, gbase), evald_m = gballot<GS>(evald, gbase), same_m = gballot<GS>(same_xr, gbase);
  // q is decided in this step if it was evaluated or needs no evaluation; otherwise the next step starts there
  const bool q_in = (q < nd) && ((inr_m >> q_gl) & 1u);
  const bool q_real = q_in && (((evald_m >> q_gl) & 1u) || ((same_m >> q_gl) & 1u));

  int advance, rej_pos;
  if (!has_need) {
    advance = min(nd, n_end - i);
    rej_pos = -1;
  } else {
    advance = q_real ? q + 1 : q;
    rej_pos = q_real ? q : -1;
  }
  const int src_gl = glane_of(max(rej_pos, 0));
  const float stop_val = adopt ? newval : st.d0;  // what the position holds afterwards
  st.rej_d0 = __shfl(stop_val, gbase + src_gl, kWave);
  st.rej_pos = act ? rej_pos : -1;
  st.advance = act ? advance : 0;
  st.adopt = adopt && has_need && (st.mpos < q || (q_real && st.mpos == q));
  if (!has_pos) st.mpos = -1;
  return st;
}

}  // namespace pm
