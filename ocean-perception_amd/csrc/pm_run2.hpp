// pm_run2.hpp -- PM_ENGINE_RUNBLK2: the run engine with TWO chain segments per wavefront.
//
// PMC profiles show the run-step kernels are VALU-issue bound (SQ_ACTIVE_INST_VALU ~ 86 % of SIMD
// time) while a 64-lane step consumes on average only ~8 positions, i.e. ~18 of the 64 window lines
// it computes.  Here a wavefront carries two independent segments of the same chain, one per 32-lane
// half ("group"): each instruction of a step now serves two steps, and a 32-line strip wastes far less
// (win = 11: up to 21 positions per step and group).  Everything that was wave-uniform in
// pm_run.hpp::run_step (position index, candidate, reference lane, bilinear parameters, outcome) is
// group-uniform here and lives in vector registers; cross-lane traffic stays inside a group:
// ballots are split into their 32-bit halves, broadcasts use ds_bpermute.  The DPP neighbour tap and
// the sliding window sum may cross from lane 31 into lane 32, which only touches lines no position of
// group 0 uses (its last lane is the spare one, exactly as lane 63 was).
// Semantics, exactness arguments and the fix-up scheme are those of pm_run.hpp; results are
// bit-identical and checked against the other engines and the oracle.
#pragma once

#include "pm_run.hpp"

namespace pm {

constexpr int kGroup = 32;

// Occupancy experiment knob: -DPM_RUNBLK2_MIN_WAVES=8 caps the kernel at 64 VGPRs (8 waves per SIMD).
#ifdef PM_RUNBLK2_MIN_WAVES
#define PM_RUNBLK2_BOUNDS __launch_bounds__(64 * kMaxSegWaves, PM_RUNBLK2_MIN_WAVES)
#else
#define PM_RUNBLK2_BOUNDS __launch_bounds__(64 * kMaxSegWaves)
#endif

struct RunStep2 {
  // group-uniform
  int advance;   // positions resolved (0 if the group is idle)
  int rej_pos;   // -1: none
  float rej_d0;  // value the position rej_pos holds after the step = candidate of the next step
  // per lane
  int mpos;
  bool adopt;
  float d0, c0, cost;
};

template <int AXIS, int TPW, int TPH>
__device__ __forceinline__ int run2_nd(const CostParams& cp) {
  return AXIS == 0 ? kGroup - (TPW > 0 ? TPW : cp.pw) : kGroup - (TPH > 0 ? TPH : cp.ph) + 1;
}

// 32-bit ballot of this lane's group.
__device__ __forceinline__ unsigned gballot(bool p, bool upper) {
  const unsigned long long b = __ballot(p);
  return upper ? (unsigned)(b >> 32) : (unsigned)b;
}

template <int AXIS, int TPW, int TPH>
__device__ __forceinline__ RunStep2 run_step2(const View& v, const PlaneSet& ps, const CostParams& cp,
                                              const SweepGeom& g, int chain, bool act, int i, int n_end, float cand,
                                              const float* din, const float* cin) {
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & (kGroup - 1);
  const bool upper = (lane & kGroup) != 0;
  const int gbase = lane & kGroup;
  const int pitch = ps.pitch, cols = ps.cols, rows = ps.rows;
  const int pw = TPW > 0 ? TPW : cp.pw, ph = TPH > 0 ? TPH : cp.ph;
  const int half_w = pw / 2, half_h = ph / 2;
  const int win = AXIS == 0 ? pw : ph;
  const int half = win / 2;
  const int nd = run2_nd<AXIS, TPW, TPH>(cp);
  const int dir = g.dir;
  const float shift = (float)(pw - 1) * 0.5f;
  const unsigned lanes_nd = (1u << nd) - 1u;

  RunStep2 st;
  st.mpos = dir > 0 ? gl : nd - 1 - gl;
  const bool inr = act && (gl < nd) && (i + st.mpos < n_end);
  st.d0 = inr ? din[i + st.mpos + 1] : 0.f;
  st.c0 = inr ? cin[i + st.mpos + 1] : 0.f;
  const bool neutral = inr && (st.d0 == cand);
  auto first_pos = [&](unsigned m) -> int {  // m != 0
    return dir > 0 ? __ffs((int)m) - 1 : nd - 1 - (31 - __clz((int)m));
  };
  auto glane_of = [&](int m) -> int { return dir > 0 ? m : nd - 1 - m; };

  const unsigned need = gballot(inr && !neutral, upper);
  const bool has_need = need != 0u;
  const int r = has_need ? first_pos(need) : 0;
  const int r_gl = glane_of(r);

  const int pos = g.s_first + (i + st.mpos) * dir;
  const int px = AXIS == 0 ? pos : chain;
  float cx = (float)px - cand;
  const bool valid = cx >= (float)half_w;
  cx = cx - shift;
  const float fl = floorf(cx);
  const int ipx = (int)fl;
  const float a = cx - fl;
  const int delta = (px - half_w) - ipx;

  const unsigned valid_m = gballot(valid, upper);
  const bool valid_r = has_need && ((valid_m >> r_gl) & 1u);
  const float a_r = __shfl(a, gbase + r_gl, kWave);
  const int delta_r = __shfl(delta, gbase + r_gl, kWave);
  const bool same = valid && (a == a_r) && (delta == delta_r);

  st.cost = 0.f;
  if (__any(valid_r)) {  // at least one group evaluates; the other computes along and ignores the result
    const float ia_r = 1.f - a_r;
    CpuLerp l;
    l.a = a_r;
    l.ia = ia_r;
    l.a11 = __float2int_rn(ia_r * 65536.f);
    l.a12 = __float2int_rn(a_r * 65536.f);
    l.ipx = 0;
    const int c_i = g.s_first + i * dir;
    const int c_base = dir > 0 ? c_i - half : c_i - half - (nd - 1);
    unsigned sc = 0, sg = 0;
    if (AXIS == 0) {
      const unsigned X = (unsigned)min(max(c_base + gl, 0), cols - 1);
      const unsigned R0 = (unsigned)min(max(c_base + gl - delta_r, 0), cols - 1);
      const unsigned org = (unsigned)((chain - half_h) * pitch);
      const unsigned ol = org + X, orr = org + R0;
#pragma unroll
      for (int t = 0; t < ph; ++t) {
        const unsigned ro = (unsigned)(t * pitch);
        const int l8 = ld_u8(v.ref8, ol + ro);
        const int lg = ld_u8(v.refg8, ol + ro);
        const int r0 = ld_u8(v.tgt8, orr + ro);
        const float g0 = ld_f32(v.tgtg, (orr + ro) * 4u);
        const int r1 = wave_shl1(r0);
        const float g1 = wave_shl1f(g0);
        sc = cpu_acc_color(sc, l8, r0, r1, l);
        sg = cpu_acc_grad(sg, lg, g0, g1, l);
      }
      sg -= cpu_grad_bias(ph);
    } else {
      const int pt = ps.pitch_t;
      const unsigned Y = (unsigned)min(max(c_base + gl, 0), rows - 1);
      // a group that does not evaluate may carry a meaningless delta_r: keep its addresses in range
      const int ipx_r = min(max((chain - half_w) - delta_r, 0), cols - 1);
      int r0 = ld_u8(v.ttgt8, (unsigned)(ipx_r * pt) + Y);
      float g0 = ld_f32(v.ttgtg, ((unsigned)(ipx_r * pt) + Y) * 4u);
#pragma unroll
      for (int t = 0; t < pw; ++t) {
        const unsigned lrow = (unsigned)((chain - half_w + t) * pt);  // wave-uniform
        const unsigned rrow = (unsigned)(min(ipx_r + t + 1, cols - 1) * pt);  // group-uniform
        const int l8 = ld_u8(v.tref8, lrow + Y);
        const int lg = ld_u8(v.trefg8, lrow + Y);
        const int r1 = ld_u8(v.ttgt8, rrow + Y);
        const float g1 = ld_f32(v.ttgtg, (rrow + Y) * 4u);
        sc = cpu_acc_color(sc, l8, r0, r1, l);
        sg = cpu_acc_grad(sg, lg, g0, g1, l);
        r0 = r1;
        g0 = g1;
      }
      sg -= cpu_grad_bias(pw);
    }
    const int line = (int)(sc | (sg << 16));
    int wsum = line;
#pragma unroll
    for (int t = 1; t < win; ++t) wsum = line + wave_shl1(wsum);
    st.cost = cpu_cost_from_sums(wsum & 0xffff, (int)((unsigned)wsum >> 16), cp);
  }

  const bool adopt = valid_r && inr && !neutral && same && (st.cost < st.c0);
  const bool cont = (inr && st.mpos < r) || neutral || adopt;
  const unsigned stop = gballot(!cont, upper) & lanes_nd;
  const int q = stop ? first_pos(stop) : nd;
  const int q_gl = glane_of(min(q, nd - 1));
  const unsigned inr_m = gballot(inr, upper), same_m = gballot(same, upper);
  const bool q_real = (q < nd) && ((inr_m >> q_gl) & 1u) && (((same_m >> q_gl) & 1u) || !((valid_m >> q_gl) & 1u));

  // outcome (group-uniform selects; see pm_run.hpp::run_step for the case analysis)
  int advance, rej_pos;
  if (!has_need) {
    advance = min(nd, n_end - i);
    rej_pos = -1;
  } else if (!valid_r) {
    advance = r + 1;
    rej_pos = r;
  } else {
    advance = q_real ? q + 1 : q;
    rej_pos = q_real ? q : -1;
  }
  const int src_gl = glane_of(max(rej_pos, 0));
  st.rej_d0 = __shfl(st.d0, gbase + src_gl, kWave);
  st.rej_pos = act ? rej_pos : -1;
  st.advance = act ? advance : 0;
  st.adopt = adopt && has_need && valid_r && st.mpos < q;
  return st;
}

// One workgroup per chain; wavefront w carries segments 2w (lanes 0-31) and 2w+1 (lanes 32-63).
// Rounds and fix-up exactly as pm_run.hpp::k_runblk, per group.
// grid = (chains, 1, slots), block = 64 * nw, dynamic LDS = 4 * (n + 1) floats + 2 * kMaxSegWaves + 3 words.
// SEM = 0: PM_SEM_CPU (run_step2 above); SEM = 1: PM_SEM_GPU (run_step2_gpu, pm_run_gpu.hpp).
template <int SEM, int AXIS, int TPW, int TPH>
__device__ __forceinline__ RunStep2 run_step2_any(const View& v, const PlaneSet& ps, const CostParams& cp,
                                                  const SweepGeom& g, int chain, bool act, int i, int n_end,
                                                  float cand, const float* din, const float* cin);

template <int SEM, int AXIS, int TPW, int TPH>
__global__ void PM_RUNBLK2_BOUNDS k_runblk2(PlaneSet ps, CostParams cp, SweepGeom g, int seg_len) {
  extern __shared__ float lds[];
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  const int n1 = (n + 1 + 3) & ~3;
  float* din = lds;
  float* cin = lds + n1;
  float* dout = lds + 2 * n1;
  float* cout = lds + 3 * n1;
  float* s_last = lds + 4 * n1;                                   // [2 * kMaxSegWaves + 1]
  int* s_changed = (int*)(lds + 4 * n1 + 2 * kMaxSegWaves + 1);   // [2]

  const int chain = g.c_lo + xcd_band_index(blockIdx.x, gridDim.x);
  const View v = make_view(ps, blockIdx.z);
  const int lane = threadIdx.x & 63;
  const int gl = lane & (kGroup - 1);
  const bool upper = (lane & kGroup) != 0;
  const int w = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int nseg = 2 * nw;
  const int sidx = 2 * w + (upper ? 1 : 0);
  const int nd = SEM == 0 ? run2_nd<AXIS, TPW, TPH>(cp) : kGroup - 2;
  const int stride = AXIS == 0 ? g.dir : g.dir * ps.pitch;
  const ptrdiff_t first =
      AXIS == 0 ? (ptrdiff_t)chain * ps.pitch + g.s_first : (ptrdiff_t)g.s_first * ps.pitch + chain;

  for (int k = threadIdx.x; k <= n; k += blockDim.x) {
    const ptrdiff_t o = first + (ptrdiff_t)(k - 1) * stride;
    const float d = v.disp[o];
    const float cc = k > 0 ? v.cost[o] : 0.f;
    din[k] = d;
    cin[k] = cc;
    dout[k] = d;
    cout[k] = cc;
  }
  __syncthreads();

  const int i0 = sidx * seg_len;
  const int i1 = min(n, i0 + seg_len);
  const bool active = i0 < n;
  unsigned n_steps = 0, n_fix = 0, n_rounds = 0;

  // ---- round 1 ------------------------------------------------------------------------------------
  float in_used = active ? din[i0] : 0.f;
  float cand = in_used;
  {
    int i = i0;
    while (__any(active && i < i1)) {
      const bool act = active && i < i1;
      const RunStep2 st = run_step2_any<SEM, AXIS, TPW, TPH>(v, ps, cp, g, chain, act, i, i1, cand, din, cin);
      ++n_steps;
      if (st.mpos >= 0 && st.mpos < st.advance) {
        dout[i + st.mpos + 1] = st.mpos == st.rej_pos ? st.rej_d0 : cand;
        cout[i + st.mpos + 1] = st.adopt ? st.cost : st.c0;
      }
      if (st.rej_pos >= 0) cand = st.rej_d0;
      i += st.advance;
    }
  }
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
      while (__any(redo && !merged && i < i1)) {
        const bool act = redo && !merged && i < i1;
        const RunStep2 st = run_step2_any<SEM, AXIS, TPW, TPH>(v, ps, cp, g, chain, act, i, i1, c2, din, cin);
        ++n_fix;
        const bool mine = st.mpos >= 0 && st.mpos < st.advance;
        const float val = st.mpos == st.rej_pos ? st.rej_d0 : c2;
        const float spec = mine ? dout[i + st.mpos + 1] : 0.f;
        const unsigned eq = gballot(mine && val == spec, upper);
        int ms = -1;
        if (eq) {  // first merged position in sweep order (lane <-> position mapping of the step function)
          const int lo_lane = __ffs((int)eq) - 1, hi_lane = 31 - __clz((int)eq);
          if (SEM == 0)
            ms = g.dir > 0 ? lo_lane : nd - 1 - hi_lane;
          else
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
    atomicAdd(&ps.counters[base + 0], (unsigned long long)n_steps);
    atomicAdd(&ps.counters[base + 1], (unsigned long long)n_fix);
    if (w == 0) atomicAdd(&ps.counters[base + 2], (unsigned long long)n_rounds);
    if (w == 0) atomicAdd(&ps.counters[base + 3], (unsigned long long)n);
  }

  for (int k = threadIdx.x + 1; k <= n; k += blockDim.x) {
    const float d = dout[k];
    if (d != din[k]) {
      const ptrdiff_t o = first + (ptrdiff_t)(k - 1) * stride;
      v.disp[o] = d;
      v.cost[o] = cout[k];
    }
  }
}

template <int SEM, int AXIS, int TPW, int TPH>
inline void launch_run2_k(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                          hipStream_t stream) {
  const int chains = g.c_hi - g.c_lo + 1;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  int nwv = waves < 1 ? 1 : (waves > kMaxSegWaves ? kMaxSegWaves : waves);
  int len = (n + 2 * nwv - 1) / (2 * nwv);
  if (len < 8) len = 8;
  const int n1 = (n + 1 + 3) & ~3;
  const size_t lds_bytes = sizeof(float) * (4 * (size_t)n1 + 2 * kMaxSegWaves + 1 + 2);
  hipLaunchKernelGGL((k_runblk2<SEM, AXIS, TPW, TPH>), dim3((unsigned)chains, 1, (unsigned)slots), dim3(kWave * nwv),
                     lds_bytes, stream, ps, cp, g, len);
}

template <int AXIS>
inline void launch_run2_axis(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                             hipStream_t stream) {
  if (cp.semantics != 0) {
    launch_run2_k<1, AXIS, 3, 3>(ps, cp, g, slots, waves, stream);
    return;
  }
  const int sq = cp.pw == cp.ph ? cp.pw : 0;
  switch (sq) {
    case 3: launch_run2_k<0, AXIS, 3, 3>(ps, cp, g, slots, waves, stream); break;
    case 5: launch_run2_k<0, AXIS, 5, 5>(ps, cp, g, slots, waves, stream); break;
    case 7: launch_run2_k<0, AXIS, 7, 7>(ps, cp, g, slots, waves, stream); break;
    case 9: launch_run2_k<0, AXIS, 9, 9>(ps, cp, g, slots, waves, stream); break;
    case 11: launch_run2_k<0, AXIS, 11, 11>(ps, cp, g, slots, waves, stream); break;
    default: launch_run2_k<0, AXIS, 0, 0>(ps, cp, g, slots, waves, stream); break;
  }
}

}  // namespace pm
#include "pm_run_gpu.hpp"
namespace pm {

template <int SEM, int AXIS, int TPW, int TPH>
__device__ __forceinline__ RunStep2 run_step2_any(const View& v, const PlaneSet& ps, const CostParams& cp,
                                                  const SweepGeom& g, int chain, bool act, int i, int n_end,
                                                  float cand, const float* din, const float* cin) {
  if constexpr (SEM == 0)
    return run_step2<AXIS, TPW, TPH>(v, ps, cp, g, chain, act, i, n_end, cand, din, cin);
  else
    return run_step2_gpu<AXIS>(v, ps, cp, g, chain, act, i, n_end, cand, din, cin);
}

// In place.
inline void launch_sweep_run2(const PlaneSet& ps, const CostParams& cp, const SweepGeom& g, int slots, int waves,
                              hipStream_t stream) {
  if (g.axis == 0)
    launch_run2_axis<0>(ps, cp, g, slots, waves, stream);
  else
    launch_run2_axis<1>(ps, cp, g, slots, waves, stream);
}

}  // namespace pm
