// pm_serial.hpp -- PM_ENGINE_SERIAL: the sweep whose equivalence with the reference loops is evident (pm_sweeps.hip only).
#pragma once

#include "pm_sweep_defs.hpp"

namespace pm {

// ---------------------------------------------------------------------------------------------
// Directional sweep, PM_ENGINE_SERIAL: one lane per chain (a row for axis 0, a column for axis 1),
// strictly sequential along the chain -- the form whose equivalence with the reference loops
// (patchmatch.cpp:264-310; patchmatch_gpu.cu:156-171, :214-229) is evident.  Used as the
// on-device anchor for the faster engines, never for the benchmark.
//   axis 0: chain index = row y in [c_lo, c_hi], steps x from s_first to s_last (step dir)
//   axis 1: chain index = column x,              steps y
// The predecessor value is read once before the first step (a border or not-yet-visited pixel,
// which this sweep never writes) and then carried in a register.
// grid = (ceil(chains/64), 1, slots), block = 64.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_sweep_serial(PlaneSet ps, CostParams cp, SweepGeom g) {
  const int chain = g.c_lo + blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = blockIdx.z;
  if (chain > g.c_hi || !chain_active(ps, slot, chain)) return;
  const View v = make_view(ps, slot);
  const int half_w = cp.semantics == 0 ? cp.pw / 2 : 1;
  const int n = (g.s_last - g.s_first) * g.dir + 1;
  if (n <= 0) return;
  int x = g.axis == 0 ? g.s_first - g.dir : chain;
  int y = g.axis == 0 ? chain : g.s_first - g.dir;
  float prev = v.disp[state_at(x, y, ps.pitch)];
  for (int s = 0; s < n; ++s) {
    if (g.axis == 0) x += g.dir; else y += g.dir;
    const size_t o = state_at(x, y, ps.pitch);
    const float d0 = v.disp[o];
    const float c0 = v.cost[o];
    float nd = d0, nc = c0;
    const bool changed = sweep_step(cp.semantics, x, half_w, d0, c0, prev, nd, nc, [&](float arg) {
      return cp.semantics == 0 ? cpu_cost_lane(v, ps.pitch, ps.cols, x, y, arg, cp)
                               : gpu_cost_lane(v, ps.pitch, x, y, arg, cp);
    });
    if (changed) {
      v.disp[o] = nd;
      v.cost[o] = nc;
    }
    prev = nd;
  }
}

}  // namespace pm
